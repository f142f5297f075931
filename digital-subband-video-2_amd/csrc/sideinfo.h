// sideinfo.h -- per-block side information of a P picture coded on the GPU (sideinfo.hip).
#pragma once

#include "dev.h"

namespace dsv2 {

// the six sub-streams, in the order the packet carries them (dsv_encoder.c:797 encode_stable_blocks, :692 encode_motion)
enum { SIDE_STABLE = 0, SIDE_MODE, SIDE_MVX, SIDE_MVY, SIDE_SBIM, SIDE_EPRM, SIDE_SUBS };
constexpr int SIDE_IMG_BYTES = 3 * 2048 + 2 * 16384 + 8192; // all six images, side_image_offset(sub) apart
constexpr int SIDE_INFO_WORDS = 8;                           // [0] flag: 1 = the host must code this frame; [1 + sub] byte length

struct SideJob {
    const DSV_MV *raw;   // level-0 field as the search left it
    DSV_MV *final_mvs;   // out: the field as transmitted (a skipped block carries the zero vector, an intra block a full-pel one)
    uint8_t *bd;         // out: DSV_IS_* flag byte of every block (filters, quantiser)
    uint8_t *out;        // out: SIDE_IMG_BYTES of pinned host memory, sub-stream s at side_image_offset(s)
    int *info;           // out: SIDE_INFO_WORDS ints of pinned host memory
    int inv_stable, inv_mode, inv_eprm; // the majority vote inverts the plane (gather_stats, dsv_encoder.c:992)
};
void side_info_batch(hipStream_t s, const SideJob *d_jobs, int n, int nbh, int nbv);
int side_image_offset(int sub);

} // namespace dsv2
