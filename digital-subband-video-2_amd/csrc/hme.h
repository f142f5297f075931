// hme.h -- interfaces of the analysis / motion-estimation kernels (intra.hip, hme.hip).
#pragma once

#include "dev.h"

namespace dsv2 {

struct Planes3 {
    DPlane p[3];
};

struct AnalysisParams {
    int width, height; // luma picture size
    int blk_w, blk_h, nbh, nbv, hshift, vshift;
    int do_psy;
    int scale; // 2 * spatial_psy_factor(-1), hme.c:1851
};

// dsv_intra_analysis (hme.c:1836): flags-only DSV_MV field for an I frame
void intra_analysis(hipStream_t s, const DFrame &src, const AnalysisParams &p, DSV_MV *d_out);

} // namespace dsv2
