/*
 * oracle/ -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the DSV2 (v2.8) per-frame hot path, written from
 * scratch in the *parallel formulation* the HIP kernels use (every output is a
 * pure function of the inputs; no in-place serial lifting), so that proving this
 * file bit-identical to the real reference (oracle/_ref, built from
 * /root/reference/src) also proves the kernel decomposition.  Nothing in the
 * product (digital-subband-video-2_amd/) may include, link or call this code:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED -- every entry point here is checked against
 * oracle/_ref/libdsv2ref.so on seeded inputs by tests/test_oracle_vs_ref.py and
 * against the committed golden vectors under tests/golden/.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))
#define ORC_CLAMP(x, a, b) ((x) < (a) ? (a) : ((x) > (b) ? (b) : (x)))
/* ceil(x / 2^s): reference dsv.h:68 DSV_ROUND_SHIFT */
#define ORC_RSHIFT_UP(x, s) (((x) + (1 << (s)) - 1) >> (s))
/* floor shift of a possibly negative int: reference dsv.h:72 DSV_SAR */
static inline int orc_sar(int v, int s) { return v < 0 ? ~(~v >> s) : v >> s; }

/* block flag bits in the per-block "blockdata" byte map (dsv_internal.h:96-110) */
#define ORC_BD_STABLE   (1 << 0)
#define ORC_BD_MAINTAIN (1 << 1)
#define ORC_BD_SKIP     (1 << 2)
#define ORC_BD_RINGING  (1 << 3)
#define ORC_BD_INTRA    (1 << 4)
#define ORC_BD_EPRM     (1 << 5)
#define ORC_BD_SIMCMPLX (1 << 6)

/* motion-vector flag bits (dsv.h:186-193) */
#define ORC_MV_INTRA    (1 << 0)
#define ORC_MV_EPRM     (1 << 1)
#define ORC_MV_MAINTAIN (1 << 2)
#define ORC_MV_SKIP     (1 << 3)
#define ORC_MV_RINGING  (1 << 4)
#define ORC_MV_NOXMITY  (1 << 5)
#define ORC_MV_NOXMITC  (1 << 6)
#define ORC_MV_SIMCMPLX (1 << 7)

#define ORC_BLOCK_P 14 /* dsv_internal.h:127 DSV_BLOCK_INTERP_P */
#define ORC_BORDER 32  /* dsv_internal.h:38 DSV_FRAME_BORDER */

/* 16-byte motion vector record, same memory layout as DSV_MV (dsv.h:171-216) */
typedef struct {
    int16_t x, y;
    uint32_t flags;
    uint16_t err;
    uint16_t dc;
    uint8_t submask;
    uint8_t pad_[3];
} orc_mv;

/* per-frame parameters shared by the stages */
typedef struct {
    int width, height;     /* luma picture size */
    int hshift, vshift;    /* chroma subsampling shifts */
    int blk_w, blk_h, nblocks_h, nblocks_v;
    int isP, lossless, do_psy, effort, temporal_mc, inter_sharpen;
} orc_params;

static inline int orc_lb2(unsigned n) /* ceil(log2 n): dsv.c:450 */
{
    unsigned i = 1;
    int l = 0;
    while (i < n) {
        i <<= 1;
        l++;
    }
    return l;
}

#endif
