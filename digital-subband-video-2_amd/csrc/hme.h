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
    int *host_counters = nullptr; // batched driver only: pinned host copy of counters[0..7]
    int *counters;         // hme_counter_words(nbv) ints. out: [0] nintra [1] ndiff [2] eligible [3] total_err
                           // ([4],[5] global motion, [7] row-pipeline timeout flag, [16..] row progress)
};

// dsv_hme (hme.c:2001): all levels coarse to fine, asynchronous on `s`
inline size_t hme_counter_words(int nbv) { return 16 + (size_t) nbv; }
int hme_run(hipStream_t s, const HmeFrames &f, const HmeParams &hp); // returns the number of front launches

// lockstep variant for n streams of identical geometry; h_table (pinned) / d_table hold hme_table_bytes(n)
size_t hme_table_bytes(int n);
struct StageProf;
// prof (optional): HIP events around the level-0 launch alone (stage ST_HME_L0), for the roofline of the dominant kernel
int hme_run_batch(hipStream_t s, const HmeFrames *f, const HmeParams *hp, int n, void *h_table, void *d_table, StageProf *prof = nullptr);

struct CodecDev;
struct PicSet;
int hme_estimate(hipStream_t s, CodecDev &dv, const PicSet &cur, const PicSet &ref, const HmeParams &hp);

} // namespace dsv2
