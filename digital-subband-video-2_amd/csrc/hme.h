// hme.h -- interfaces of the analysis / motion-estimation kernels (intra.hip, hme.hip).
#pragma once

#include "bmc.h"
#include "dev.h"

namespace dsv2 {

struct AnalysisParams {
    int width, height; // luma picture size
    int blk_w, blk_h, nbh, nbv, hshift, vshift;
    int do_psy;
    int scale; // 2 * spatial_psy_factor(-1), hme.c:1851
};

// dsv_intra_analysis (hme.c:1836): flags-only DSV_MV field for an I frame
void intra_analysis(hipStream_t s, const DFrame &src, const AnalysisParams &p, DSV_MV *d_out);
struct IntraJob {
    Planes3 src;
    DSV_MV *out;
};
void intra_analysis_batch(hipStream_t s, const IntraJob *d_jobs, int n, const AnalysisParams &p);

struct HmeParams {
    AnalysisParams a;
    int effort, lossless;
    int quant; // quantiser of the previous frame (dsv_encoder.c:665)
    int skip_block_thresh;
    int pyr_levels;
};

// device operands of one motion search: luma of every pyramid level (0 = full size) for the
// source, the reconstructed reference and the original reference; chroma of level 0
struct HmeFrames {
    DPlane src[6], ref[6], ogr[6];
    DPlane srcc[2], refc[2];
    DSV_MV *mvf[6];        // out: one field per level, nblocks entries each
    const DSV_MV *ref_mvf; // previous frame's transmitted field or null
    DSV_MV *host_mvs = nullptr;   // batched driver only: pinned host mirror of mvf[0], filled by the search itself
    int *host_counters = nullptr; // batched driver only: pinned host copy of counters[0..7]; word 12 (kHmeHostTailWord) of the FIRST
                                  // stream's block is set when the level-0 launch has handed out its last block row
    void *src_stats = nullptr; // hme_src_stats_bytes() of device memory for the source pre-pass, or null (hme_run: allocated per call)
    void *l0_pre = nullptr;    // hme_l0_pre_bytes() of device memory for level 0's pre-pass records, or null (hme_run: allocated per call)
    int *counters;         // hme_counter_words(nbv) ints. out: [0] nintra [1] ndiff [2] eligible [3] total_err
                           // ([4],[5] global motion, [7] row-pipeline timeout flag, [16..] row progress)
};

// dsv_hme (hme.c:2001): all levels coarse to fine, asynchronous on `s`
inline size_t hme_counter_words(int nbv) { return 16 + (size_t) nbv; }
// source statistics of every block of levels 0 and 1, 16 bytes each (k_hme_src_stats_b)
inline size_t hme_src_stats_bytes(int nbh, int nbv) { return ((size_t) nbh * nbv + (size_t) ((nbh + 1) / 2) * ((nbv + 1) / 2)) * 16; }
// level 0's pre-pass records (k_hme_l0_pre_b), 256 bytes a block
inline size_t hme_l0_pre_bytes(int nbh, int nbv) { return (size_t) nbh * nbv * 256; }
constexpr int kHmeHostTailWord = 12;
int hme_run(hipStream_t s, const HmeFrames &f, const HmeParams &hp); // returns the number of front launches

// lockstep variant for n streams of identical geometry; h_table (pinned) / d_table hold hme_table_bytes(n)
size_t hme_table_bytes(int n);
struct StageProf;
// prof (optional): HIP events around the level-0 launch alone (stage ST_HME_L0), for the roofline of the dominant kernel
// level_hi .. level_lo: the pyramid levels this call runs (default: all, coarse to fine); a search may be split over calls --
// the one that starts at the coarsest level also ships the job table and clears the hand-off words
// phases: HME_PREPARE = what needs neither the reference's search slots nor any order -- the job table, the clears and the
// source pre-pass (k_hme_src_stats_b) --, HME_LEVELS = the levels' launches; a caller that serialises searches (the
// encoder's search token) prepares before it queues for its turn
enum { HME_PREPARE = 1, HME_LEVELS = 2 };
int hme_run_batch(hipStream_t s, const HmeFrames *f, const HmeParams *hp, int n, void *h_table, void *d_table, StageProf *prof = nullptr,
                  int level_hi = -1, int level_lo = 0, int phases = HME_PREPARE | HME_LEVELS);

// ---- per-frame block statistics of the finished level-0 field (host controller inputs) -------------------------
// What the reference's controller sums over the motion field of a P frame before it decides anything
// (dsv_encoder.c:129-250 avg_motion / scene_complexity, :545-650 the running intra map of scene_change_detection,
// :992-1037 gather_stats): every term is a function of a block's own vector record and its left / top / top-left
// neighbours', so the sums are taken by one thread per block right behind the search, and the host's H1 phase starts
// from ten integers instead of four passes over the field.  Sums are plain 32-bit adds (the reference's `int`s).
enum {
    BS_AX = 0, BS_AY, BS_CHAOS, BS_STAT,    // avg_motion: sum of x / y of the non-skipped vectors, chaotic / static block counts
    BS_COMPLEXITY,                           // scene_complexity numerator (rc_mode decides the terms)
    BS_NINTRA, BS_SKIPN,                     // scene_change_detection's second test, over map_in | this frame's intra flags
    BS_MODE, BS_EPRM, BS_STABLE,             // gather_stats majorities (+1 / -1 votes)
    // the block counts of DSV_STATS (dsv_encoder.c:1505-1550): flags and sub-pel phases of the field (an inter block's
    // vector and every flag are the same before and after the field is finalised)
    BS_ST_EPRM, BS_ST_SKIP, BS_ST_MBI, BS_ST_MBDC, BS_ST_MBSUB, BS_ST_SUB0, BS_ST_SUB1, BS_ST_SUB2, BS_ST_SUB3, BS_ST_MBP,
    BS_ST_QPX, BS_ST_HPX, BS_ST_FPX, BS_ST_QPY, BS_ST_HPY, BS_ST_FPY,
    BS_USED,
    BS_WORDS = 32
};
struct BlockStatsJob {
    const DSV_MV *mvs;      // level-0 field as the search left it
    const int *counters;    // the search's counters ([3] = total_err: avg_err = total_err / nblocks)
    const uint8_t *map_in;  // running intra map of the GOP so far, or null: all zero (first P frame after an I frame)
    uint8_t *map_out;       // map_in | this frame's intra flags (committed by the host if the frame stays a P frame)
    DSV_MV *host_mvs;       // pinned host mirror of the field: written here in whole 16-byte records, coalesced, instead of by
                            // the search itself block by block (two lone PCIe stores inside every block's drain-and-publish)
    int *out;               // BS_WORDS ints, zeroed before the launch
    int b2sr;               // mv_cost's bits-to-SSE ratio for the previous frame's quantiser (dsv.c:357)
    int rc_mode;            // DSV_RATE_CONTROL_*: which scene_complexity variant
};
void block_stats_batch(hipStream_t s, const BlockStatsJob *d_jobs, int n, int nbh, int nbv);

struct CodecDev;
struct PicSet;
int hme_estimate(hipStream_t s, CodecDev &dv, const PicSet &cur, const PicSet &ref, const HmeParams &hp);

} // namespace dsv2
