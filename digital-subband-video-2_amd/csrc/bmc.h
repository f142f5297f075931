// bmc.h -- interfaces of the motion-compensation / in-loop filter kernels (bmc.hip).
#pragma once

#include "dev.h"
#include "quant.h"

namespace dsv2 {

struct MCParams { // the slice of DSV_PARAMS + DSV_META the MC stage reads
    int blk_w, blk_h, nbh, nbv;
    int hshift, vshift;
    int temporal_mc, lossless;
};

struct FilterParams {
    int blk_w, blk_h, nbh, nbv, hshift, vshift;
    int lossless, do_filter, sharpen;
    int q;       // compute_filter_q(quant), bmc.c:376
    int q_raw;   // the frame quantiser as transmitted (chroma filter thresholds, bmc.c:619)
    int fthresh; // 32 * (14 - lb2(q)), bmc.c:408
};

struct Planes3 {
    DPlane p[3];
};

// one stream's operands for the lockstep-batched MC / filter kernels
struct McJob {
    const DSV_MV *mvs;
    const uint8_t *bd;
    MCParams p;
    FilterParams f;
    Planes3 ref, pred, res;
    // predict + subtract: where the SOURCE pixels are read (null: from `res`, which then holds a copy of the source and is
    // overwritten in place -- the single-call seam); same strides as `res` (planes of dframe_alloc).  The batch encoder points
    // this at the padded source picture itself: no copy of the source into the working picture for P pictures.
    const uint8_t *src[3] = {nullptr, nullptr, nullptr};
};

inline int spatial_psy_factor_host(int bw, int bh, int nbh, int nbv, int sub) { return spatial_psy_factor(bw, bh, nbh, nbv, sub); }

FilterParams make_filter_params(const MCParams &p, int q, int do_filter, int inter_sharpen);

// dsv_sub_pred (bmc.c:1058): pred <- MC prediction from ref, resd <- resd - pred
void mc_sub_pred(hipStream_t s, const DSV_MV *d_mvs, const MCParams &p, const DFrame &pred, const DFrame &resd, const DFrame &ref);
// dsv_add_res (bmc.c:1073): resd <- pred + resd, then in-loop filters
void mc_add_res(hipStream_t s, const DSV_MV *d_mvs, const MCParams &p, int q, const DFrame &resd, const DFrame &pred, int do_filter,
                int inter_sharpen);
// dsv_add_pred (bmc.c:1094): out <- MC prediction from ref + resd, then in-loop filters
void mc_add_pred(hipStream_t s, const DSV_MV *d_mvs, const MCParams &p, int q, const DFrame &resd, const DFrame &out, const DFrame &ref,
                 int do_filter, int inter_sharpen);
// dsv_intra_filter (bmc.c:391), luma plane only
void intra_filter_luma(hipStream_t s, const uint8_t *d_bd, const MCParams &p, int q, const DPlane &luma);

// lockstep batches over n streams (job tables resident on the device)
void mc_sub_pred_batch(hipStream_t s, const McJob *d_tab, int n, int nbh, int nbv, int blk_w, int blk_h, bool c420);
void mc_add_res_batch(hipStream_t s, const McJob *d_tab, int n, int nbh, int nbv, bool any_filter, int luma_w, int luma_h, int blk_w, int blk_h);
void mc_add_pred_batch(hipStream_t s, const McJob *d_pred, const McJob *d_filt, int n, int nbh, int nbv, bool any_filter, int luma_w, int luma_h, int blk_w,
                       int blk_h, bool c420);
void intra_filter_batch(hipStream_t s, const McJob *d_tab, int n, int luma_w, int luma_h); // luma size: whether / how large the LDS ring

// dsv_post_process (bmc.c:340): de-gradient sharpen of every interior 4x4 cell
void post_process_plane(hipStream_t s, const DPlane &dp);

} // namespace dsv2
