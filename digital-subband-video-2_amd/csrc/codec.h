// codec.h -- device-resident state of one encoder / decoder instance and the stage drivers
// shared by encoder.cpp and decoder.cpp.
#pragma once

#include <vector>

#include "bmc.h"
#include "dev.h"
#include "hme.h"
#include "quant.h"
#include "entropy_gpu.h"

namespace dsv2 {

// one picture and everything derived from it that a later frame may need as its reference
struct PicSet {
    DFrame src;                            // padded source picture
    DFrame src_pyr[DSV_MAX_PYRAMID_LEVELS]; // luma decimation pyramid of the source
    DFrame recon;                          // residual -> reconstruction (the "residual" frame of the reference)
    DFrame recon_pyr[DSV_MAX_PYRAMID_LEVELS];
    bool recon_pyr_valid = false;
    DSV_MV *d_final_mvs = nullptr;         // motion field as finally transmitted (device)
    bool has_final_mvs = false;
};

struct PlaneSyms { // host view of one plane's entropy input
    int32_t LL;
    const uint32_t *pos;
    const int32_t *val;
    int n;
};

// ---- optional stage timing with HIP events on the codec's own stream (bench.py roofline) ----
enum Stage { ST_INGEST = 0, ST_HME, ST_PREDICT, ST_FWD_SBT, ST_QUANT, ST_INV_SBT, ST_RECON_FILTER, ST_EXTEND, ST_HME_L0 /* the level-0 search launch alone (inside ST_HME) */, ST_COUNT };

struct StageProf {
    bool created = false;
    hipEvent_t ev[ST_COUNT][2];
    bool used[ST_COUNT];
    long long launches[ST_COUNT], units[ST_COUNT], mark = 0;
    void destroy();
    void begin(hipStream_t s, int st);
    // nunits = stream-frames the stage processed; nlaunch >= 0 overrides the counted kernel launches
    void end(hipStream_t s, int st, int nunits, int nlaunch = -1);
    void collect(); // after a stream synchronise: fold the step's event pairs into the global totals
};
bool prof_enabled();

// geometry + buffers that persist for the life of a codec instance
struct CodecDev {
    // Created on first request: a lockstep batch runs on the stream of its FIRST instance only, so a GPU with hundreds of
    // encoder instances holds a handful of streams, not one per instance (the runtime spreads the streams that exist
    // over a few hardware queues).
    hipStream_t stream = nullptr;
    hipStream_t ensure_stream();
    bool alive = false;
    int format = 0, w = 0, h = 0;
    int blk_w = 0, blk_h = 0, nbh = 0, nbv = 0, pyr_levels = 0;
    int cw[3], ch[3];
    PicSet pics[2];
    DFrame pred;
    int32_t *coefs[3] = {nullptr, nullptr, nullptr};
    int32_t *qv = nullptr; // dense quantised values of the 3 planes, concatenated (encoder: inside `work`, see init)
    int32_t *work = nullptr; // encoder: ONE block for the transform scratch of all planes and the dense quantised values
    DevArena arena;          // encoder: the one device allocation everything below lives in (dev.h)
    size_t qv_off[4] = {0, 0, 0, 0};
    ScanGeom scan[3];
    SbtScratch scratch;        // luma (and, one plane at a time, any plane of the single-stream calls)
    SbtScratch scratch_uv[2];  // chroma planes of the table-driven encoder path, where U and V share a launch
    Compactor comp;
    EntBuffers ent;            // plane sections of the packet assembled on the GPU (encoder, batch engine)
    uint8_t *d_blockdata = nullptr;
    DSV_MV *d_mvs_stage = nullptr; // analysis output / upload staging
    DSV_MV *d_mvf[DSV_MAX_PYRAMID_LEVELS + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int *d_counters = nullptr;
    void *d_src_stats = nullptr; // hme_src_stats_bytes(): the search's source pre-pass (encoder)
    void *d_l0_pre = nullptr;    // hme_l0_pre_bytes(): level 0's pre-pass records (encoder)
    uint8_t *d_intra_map[2] = {nullptr, nullptr}; // encoder: running intra map of the GOP (committed / being written), hme.h BlockStatsJob
    int32_t *d_ll = nullptr;
    // decoder-side symbol upload
    uint32_t *d_sym_pos = nullptr;
    int32_t *d_sym_val = nullptr;
    size_t sym_cap = 0;
    StageProf prof;
    // pinned host staging
    uint8_t *h_frame = nullptr; // one packed planar picture
    size_t h_frame_bytes = 0;
    DSV_MV *h_mvs = nullptr;
    DSV_MV *h_intra = nullptr; // intra-analysis flags read-back
    int *h_counters = nullptr;
    int32_t *h_ll = nullptr;
    uint8_t *h_small = nullptr; // coarsest pyramid level readback
    uint32_t *h_pos = nullptr;
    int32_t *h_val = nullptr;
    size_t h_sym_cap = 0;

    // list_symbols: symbols the compaction lists (and the entropy coder's per-symbol buffers) hold at first; 0 = every coefficient
    // (the worst case: lossless).  A picture that has more is worked out again into enlarged lists (encoder.cpp: redo_overflow).
    void init(int format, int w, int h, int blk_w, int blk_h, int pyr_levels, bool encoder, size_t list_symbols = 0);
    void destroy();
    void ensure_host_syms(size_t n);
    void ensure_dev_syms(size_t n);
    MCParams mc_params(int temporal_mc, int lossless) const;
    QuantCfg quant_cfg(int plane, int isP, int lossless, int do_psy, const DSV_MV *d_mvs) const;
    size_t nblocks() const { return (size_t) nbh * nbv; }
};

void block_geometry(int w, int h, int ovx, int ovy, int *blk_w, int *blk_h, int *nbh, int *nbv);



} // namespace dsv2
