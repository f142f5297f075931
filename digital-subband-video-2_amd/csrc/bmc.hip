// bmc.hip -- block motion compensation, residual formation / reconstruction and the in-loop
// 4x4 smoothing filters on gfx950.
//
// Replaces reference src/bmc.c: predict (:815) with luma_qp (:662), bilinear_sp (:773) and the
// intra DC fills, subtract (:990), reconstruct (:926), luma_filter (:460), chroma_filter (:605),
// dsv_intra_filter (:391) with ihfilter4x4 (:71), ivfilter4x4 (:131), artf4x4 (:253),
// dsff4x4 (:195), degrad4x4 (:277); entry points dsv_sub_pred (:1058), dsv_add_res (:1073),
// dsv_add_pred (:1094).
//
// Decomposition (proved bit-exact on the CPU by oracle/orc_bmc.c):
//   * prediction / subtraction / reconstruction: one workgroup per (block, plane); the 2-pass
//     quarter-pel filter is evaluated per output pixel from its 4x4 reference window (the
//     16-bit intermediate of the reference is a pure function of one reference row);
//   * the in-place filters are raster-order dependent in the reference.  Cell (i,j) only depends
//     on cells (i-1,j), (i-2,j), (i-1,j-1), (i,j-1), (i+1,j-1), so all cells with equal i + 2j
//     form a wavefront that is processed concurrently; one workgroup per plane sweeps the fronts
//     with a workgroup barrier between them (the plane stays in L2 / the CU's L1).
#include "dev.h"
#include "prio.h"
#include "blockstat.h"
#include "bmc.h"
#include "hme.h"

namespace dsv2 {

__device__ __forceinline__ int sar(int v, int s) { return v >> s; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ uint8_t clamp_u8(int v) { return (uint8_t) (v > 255 ? 255 : (v < 0 ? 0 : v)); }

// ---- prediction ---------------------------------------------------------------------

__device__ __forceinline__ int hp_tap(int a, int b, int c, int d, bool soft)
{
    return soft ? (19 * (b + c) - 3 * (a + d)) : (20 * (b + c) - 4 * (a + d)); // dsv_internal.h:130-133
}

__device__ __forceinline__ int qp_blend(int f, int b, int c, int frac) // bmc.c:702-715
{
    switch (frac) {
        case 0: return (64 * b + 32) >> 6;
        case 1: return (f + 32 * b + 32) >> 6;
        case 2: return (2 * f + 32) >> 6;
        default: return (f + 32 * c + 32) >> 6;
    }
}

// one luma pixel of a sub-pel block; r points at the reference sample (px-1, py-1) + (m, n)
__device__ __forceinline__ int luma_subpel_px(const uint8_t *r, int rs, int fx, int fy, bool soft_x, bool soft_y)
{
    int t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint8_t *q = r + k * rs;
        int a = q[0], b = q[1], c = q[2], d = q[3];
        t[k] = (int) (int16_t) qp_blend(hp_tap(a, b, c, d, soft_x), b, c, fx);
    }
    return qp_blend(hp_tap(t[0], t[1], t[2], t[3], soft_y), t[1], t[2], fy);
}

enum { MC_PREDICT_ONLY = 0, MC_SUBTRACT = 1, MC_RECONSTRUCT = 2 };

// ---- lockstep-batch forms: one WAVEFRONT per block (all three planes), four pixels per lane ---------------------
// grid = (ceil(nblocks_h / 4), nblocks_v, n streams), 256 threads = 4 wavefronts = 4 horizontally adjacent blocks.
// Pixels move as aligned dwords (block origins and widths are multiples of 4); the reference window of a
// fractional luma vector is staged per wavefront in LDS and filtered in two passes (rows, then columns: bmc.c:702-760).
struct __attribute__((packed)) U32u {
    uint32_t v;
};

// ---- packed 16-bit forms of the sub-pel taps (two pixels an instruction) -------------------------------------------------
// Exact: a 4-tap sum is at most 20 * 2 * 340 and a blended one 2 * that + 32, inside 16 bits with sign (the reference keeps the
// intermediate row in int16 as well, bmc.c:702-760).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk_of(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t u32_of(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 spl16s(int v) { return (s16x2){(short) v, (short) v}; }

struct SubpelAxis { // one axis of a quarter-pel vector, wave-uniform: tap = k1 (b + c) - k2 (a + d); out = (wf tap + wb b + wc c + 32) >> 6
    s16x2 k1, k2, wf, wb, wc;
};
__device__ __forceinline__ SubpelAxis subpel_axis(int frac, bool soft)
{
    SubpelAxis a;
    a.k1 = spl16s(soft ? 19 : 20); // dsv_internal.h:130-133
    a.k2 = spl16s(soft ? 3 : 4);
    a.wf = spl16s(frac == 0 ? 0 : (frac == 2 ? 2 : 1)); // bmc.c:702-715
    a.wb = spl16s(frac == 0 ? 64 : (frac == 1 ? 32 : 0));
    a.wc = spl16s(frac == 3 ? 32 : 0);
    return a;
}
__device__ __forceinline__ s16x2 subpel_pk(const SubpelAxis &A, s16x2 a, s16x2 b, s16x2 c, s16x2 d)
{
    const s16x2 f = A.k1 * (b + c) - A.k2 * (a + d);
    return (A.wf * f + A.wb * b + A.wc * c + spl16s(32)) >> spl16s(6);
}

// residual (MC_SUBTRACT) or reconstruction (MC_RECONSTRUCT) of four pixels, two an instruction: s4 = the four source / residual
// bytes, p01 / p23 = the prediction (0..255 in 16-bit halves); `flat` = a block whose residual is not transmitted (128)
template <int MODE>
__device__ __forceinline__ uint32_t resid4_pk(uint32_t s4, s16x2 p01, s16x2 p23, bool lossless, bool flat, bool eprm, bool plain)
{
    const s16x2 s01 = pk_of(__builtin_amdgcn_perm(0u, s4, 0x0c010c00u)), s23 = pk_of(__builtin_amdgcn_perm(0u, s4, 0x0c030c02u));
    const s16x2 lo = spl16s(0), hi = spl16s(255);
    s16x2 o01, o23;
    if (MODE == MC_SUBTRACT) { // residual_px (bmc.c:1015-1050)
        if (lossless) {
            o01 = (s01 - p01 + spl16s(128)) & hi;
            o23 = (s23 - p23 + spl16s(128)) & hi;
        } else if (flat) {
            return 0x80808080u;
        } else if (eprm) {
            o01 = __builtin_elementwise_min(__builtin_elementwise_max((s01 - p01 + spl16s(256)) >> spl16s(1), lo), hi);
            o23 = __builtin_elementwise_min(__builtin_elementwise_max((s23 - p23 + spl16s(256)) >> spl16s(1), lo), hi);
        } else {
            o01 = __builtin_elementwise_min(__builtin_elementwise_max(s01 - p01 + spl16s(128), lo), hi);
            o23 = __builtin_elementwise_min(__builtin_elementwise_max(s23 - p23 + spl16s(128), lo), hi);
        }
    } else { // recon_px (bmc.c:953-983)
        if (lossless) {
            o01 = (p01 + s01 - spl16s(128)) & hi;
            o23 = (p23 + s23 - spl16s(128)) & hi;
        } else if (plain) {
            o01 = __builtin_elementwise_min(__builtin_elementwise_max(p01 + s01 - spl16s(128), lo), hi);
            o23 = __builtin_elementwise_min(__builtin_elementwise_max(p23 + s23 - spl16s(128), lo), hi);
        } else {
            o01 = __builtin_elementwise_min(__builtin_elementwise_max(p01 + (s01 - spl16s(128)) * spl16s(2), lo), hi);
            o23 = __builtin_elementwise_min(__builtin_elementwise_max(p23 + (s23 - spl16s(128)) * spl16s(2), lo), hi);
        }
    }
    return __builtin_amdgcn_perm(u32_of(o23), u32_of(o01), 0x06040200u);
}

// A wavefront's scratch for the luma sub-pel path: the block's (bw + 3) x (bh + 3) window as row dwords (36-byte rows) and the
// horizontally filtered intermediate (32 int16 a row).  Sized BY THE LAUNCH's block height (dynamic LDS): 19 rows for 16-pixel
// blocks = 1.9 KB a wavefront, 7.6 KB a workgroup.  Until round 6 it was a static 35 rows (3.5 KB, 14 KB a workgroup) whatever the
// geometry -- and LDS is what the chip runs out of first under the headline's load: two in-loop filter sweeps (2 x 70 KB rings)
// and the motion search's twelve wavefronts (19 KB) fill a compute unit's 160 KB, and the kernels that need LDS (this one, the
// scatter, the entropy coder's emit) were the ones that stretched most beside them (profiles/r06_occupancy.txt).
struct WaveLds {
    uint8_t *win;
    int16_t *hz;
};
__host__ __device__ constexpr unsigned wave_lds_win_bytes(int rows) { return ((unsigned) rows * 36u + 15u) & ~15u; }
__host__ __device__ constexpr unsigned wave_lds_bytes(int rows) { return wave_lds_win_bytes(rows) + (unsigned) rows * 64u; }

__device__ __forceinline__ void wave_lds_sync()
{
    // one wavefront runs in lockstep: its LDS writes only have to land before the reads that follow
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_sum_i(int v)
{
    v = fold_xor<32>(v);
    v = fold_xor<16>(v);
    v = fold_xor<8>(v);
    v = fold_xor<4>(v);
    v = fold_xor<2>(v);
    v = fold_xor<1>(v);
    return __builtin_amdgcn_readfirstlane(v);
}

// one plane of one block.  CC >= 0 fixes the plane AND the geometry at compile time (16x16 luma blocks, 4:2:0: 8x8 chroma):
// loop bounds, shifts and the lane -> pixel mapping fold to constants, which removes about a third of the kernel's scalar and
// a fifth of its vector instructions; CC < 0 is the general form (plane c, sizes from the job).
// TILED (with CC >= 0): the 16 x 16 (8 x 8) piece at luma offset (ox, oy) of an INTER block of 32 x 32, 32 x 16 or 16 x 32 -- a block's prediction
// is per pixel once the block's reference position is known, so the piece takes the whole block's position (clamped with the
// whole block's size, bmc.c:700-707) plus its own offset.
template <int MODE, int CC, bool TILED = false>
__device__ __forceinline__ void predict_plane(const McJob &jb, const MCParams &p, const DSV_MV &mv, int i, int j, WaveLds &L, int c_rt, int ox = 0, int oy = 0)
{
    const int lane = threadIdx.x & 63;
    const int mvx = mv.u.mv.x, mvy = mv.u.mv.y;
    const uint32_t flags = mv.flags;
    const bool intra = flags & (1u << DSV_MV_BIT_INTRA);
    const bool skip = flags & (1u << DSV_MV_BIT_SKIP), eprm = flags & (1u << DSV_MV_BIT_EPRM);
    const int c = CC >= 0 ? CC : c_rt;
    {
        const int sh = CC >= 0 ? (CC ? 1 : 0) : (c ? p.hshift : 0), sv = CC >= 0 ? (CC ? 1 : 0) : (c ? p.vshift : 0);
        const int bw = CC >= 0 ? (CC ? 8 : 16) : (p.blk_w >> sh), bh = CC >= 0 ? (CC ? 8 : 16) : (p.blk_h >> sv);
        const DPlane rp = jb.ref.p[c], dp = jb.pred.p[c], sp = jb.res.p[c];
        const int Bw = TILED ? (p.blk_w >> sh) : bw, Bh = TILED ? (p.blk_h >> sv) : bh; // the block the vector belongs to
        const int tox = TILED ? (ox >> sh) : 0, toy = TILED ? (oy >> sv) : 0;
        const int limx = (dp.w - Bw) + kBorder - 1, limy = (dp.h - Bh) + kBorder - 1;
        const int x = i * Bw + tox, y = j * Bh + toy;
        int px = i * Bw + sar(mvx, 2 + sh), py = j * Bh + sar(mvy, 2 + sv);
        const int sbw = bw >> 1, sbh = bh >> 1;
        const bool subpel_luma = (c == 0) && !intra && ((mvx | mvy) & 3);
        px = clampi(subpel_luma ? px - 1 : px, -kBorder, limx) + tox;
        py = clampi(subpel_luma ? py - 1 : py, -kBorder, limy) + toy;
        const uint8_t *rbase = rp.data + (ptrdiff_t) py * rp.stride + px;
        // block sizes are powers of two (16 << e, halved by the chroma shifts): lane -> (row, column) by shifts, not by the
        // 32-bit divide sequence (a dozen of them per block otherwise)
        const int lbw = 31 - __builtin_clz((unsigned) bw), lgw = lbw - 2;
        const int ngroups = (bw * bh) >> 2, gw = bw >> 2; // groups of 4 pixels, gw per row
        int dcq[4] = {0, 0, 0, 0};
        if (intra) {
            if (!(c == 0 && mv.dc)) { // quadrant means of the reference block (bmc.c:845-900)
                int part[4] = {0, 0, 0, 0};
                for (int g = lane; g < ngroups; g += 64) {
                    int m = (g & (gw - 1)) * 4, n = g >> lgw;
                    uint32_t v = ((const U32u *) (rbase + (ptrdiff_t) n * rp.stride + m))->v;
                    int s4 = (int) ((v & 0xff) + ((v >> 8) & 0xff) + ((v >> 16) & 0xff) + (v >> 24));
                    int k = (m >= sbw ? 1 : 0) | (n >= sbh ? 2 : 0);
                    if (sbw >= 4) { // the four pixels share a quadrant
                        part[k] += s4;
                    } else {        // 4-pixel-wide chroma block: two pixels per quadrant column
                        part[k & 2] += (int) ((v & 0xff) + ((v >> 8) & 0xff));
                        part[(k & 2) | 1] += (int) (((v >> 16) & 0xff) + (v >> 24));
                    }
                }
                int q0 = wave_sum_i(part[0]), q1 = wave_sum_i(part[1]), q2 = wave_sum_i(part[2]), q3 = wave_sum_i(part[3]);
                if (mv.submask == DSV_MASK_ALL_INTRA) {
                    dcq[0] = dcq[1] = dcq[2] = dcq[3] = (q0 + q1 + q2 + q3) / (bw * bh); // bmc.c:857
                } else {
                    dcq[0] = q0 / (sbw * sbh); // bmc.c:884
                    dcq[1] = q1 / (sbw * sbh);
                    dcq[2] = q2 / (sbw * sbh);
                    dcq[3] = q3 / (sbw * sbh);
                }
            } else {
                dcq[0] = dcq[1] = dcq[2] = dcq[3] = mv.dc; // transmitted DC, luma only (bmc.c:854,881)
            }
        }
        int fx = 0, fy = 0;
        bool soft_x = false, soft_y = false;
        int f0 = 0, f1 = 0, f2 = 0, f3 = 0, sf = 0, af = 0;
        bool chroma_frac = false;
        if (subpel_luma) {
            bool large = abs(mvx) >= 8 || abs(mvy) >= 8; // bmc.c:674-679
            fx = mvx & 3;
            fy = mvy & 3;
            soft_x = large || !(fx & 1) || (p.temporal_mc & 1);
            soft_y = large || !(fy & 1) || (p.temporal_mc & 1);
            const int ww = bw + 3, wh = bh + 3;
            // the (bw+3) x (bh+3) window as row dwords, all loads of a lane issued before the first is used (a load in
            // a loop body is followed by its wait: one memory round trip per iteration)
            typedef const __attribute__((address_space(1))) uint8_t *gb_t;
            typedef const __attribute__((address_space(1))) U32u *gu32_t;
            const int ndw = (ww + 3) >> 2, total = wh * ndw;
            // k / ndw == (k * ndw_inv) >> 16 for k < 4096 (ndw_inv = ceil(65536 / ndw): 5 dwords a row for 16-pixel blocks)
            const unsigned ndw_inv = ndw == 5 ? 13108u : (65536u + (unsigned) ndw - 1u) / (unsigned) ndw;
            gb_t gbase = (gb_t) rbase;
            uint32_t *win32 = (uint32_t *) L.win;
            wave_lds_sync(); // the previous block's readers are done
            if (total <= 128) { // 16-pixel blocks: 19 rows of 5 dwords
                const int k0 = lane, k1 = lane + 64 < total ? lane + 64 : 0;
                const int r0 = (int) (((unsigned) k0 * ndw_inv) >> 16), c0 = k0 - r0 * ndw;
                const int r1 = (int) (((unsigned) k1 * ndw_inv) >> 16), c1 = k1 - r1 * ndw;
                // (per-lane offsets in 32 bits from a wave-uniform base: a 24-bit multiply instead of a 64-bit multiply-add)
                uint32_t d0 = ((gu32_t) (gbase + (__mul24(r0, rp.stride) + 4 * c0)))->v;
                uint32_t d1 = ((gu32_t) (gbase + (__mul24(r1, rp.stride) + 4 * c1)))->v;
                if (k0 < total) {
                    win32[r0 * 9 + c0] = d0;
                }
                if (lane + 64 < total) {
                    win32[r1 * 9 + c1] = d1;
                }
            } else {
                for (int idx = lane; idx < total; idx += 64) {
                    int r = (int) (((unsigned) idx * ndw_inv) >> 16), cc = idx - r * ndw;
                    win32[r * 9 + cc] = ((gu32_t) (gbase + (ptrdiff_t) r * rp.stride + 4 * cc))->v;
                }
            }
            wave_lds_sync();
            if (CC == 0) { // 16x16: four pixels of a window row per lane (19 rows x 4 groups: two rounds), two pixels an instruction
                const SubpelAxis AX = subpel_axis(fx, soft_x);
#pragma unroll
                for (int rnd = 0; rnd < 2; rnd++) {
                    const int it = lane + 64 * rnd, r = it >> 2, g = it & 3;
                    if (rnd == 0 || it < 19 * 4) {
                        const uint32_t w0 = win32[r * 9 + g], w1 = win32[r * 9 + g + 1]; // bytes B0..B7; output k takes Bk..Bk+3
                        const s16x2 p01 = pk_of(__builtin_amdgcn_perm(w1, w0, 0x0c010c00u)), p12 = pk_of(__builtin_amdgcn_perm(w1, w0, 0x0c020c01u)),
                                    p23 = pk_of(__builtin_amdgcn_perm(w1, w0, 0x0c030c02u)), p34 = pk_of(__builtin_amdgcn_perm(w1, w0, 0x0c040c03u)),
                                    p45 = pk_of(__builtin_amdgcn_perm(w1, w0, 0x0c050c04u)), p56 = pk_of(__builtin_amdgcn_perm(w1, w0, 0x0c060c05u));
                        const s16x2 o01 = subpel_pk(AX, p01, p12, p23, p34), o23 = subpel_pk(AX, p23, p34, p45, p56);
                        *(uint2 *) &L.hz[r * 32 + 4 * g] = make_uint2(u32_of(o01), u32_of(o23));
                    }
                }
            } else {
                for (int idx = lane; idx < wh * bw; idx += 64) {
                    int r = idx >> lbw, m = idx & (bw - 1);
                    const uint8_t *q = &L.win[r * 36 + m];
                    int a = q[0], b = q[1], cc = q[2], d = q[3];
                    L.hz[r * 32 + m] = (int16_t) qp_blend(hp_tap(a, b, cc, d, soft_x), b, cc, fx);
                }
            }
            wave_lds_sync();
        } else if (c != 0 && !intra) {
            int hb = 2 + sh, vb = 2 + sv, hf = 1 << hb, vf = 1 << vb; // bmc.c:778-798
            int dx = mvx & (hf - 1), dy = mvy & (vf - 1);
            chroma_frac = (dx | dy) != 0;
            f0 = (hf - dx) * (vf - dy);
            f1 = dx * (vf - dy);
            f2 = (hf - dx) * dy;
            f3 = dx * dy;
            sf = hb + vb;
            af = 1 << (sf - 1);
        }
        const bool noxmit = c == 0 ? (flags & (1u << DSV_MV_BIT_NOXMITY)) : (flags & (1u << DSV_MV_BIT_NOXMITC));
        typedef const __attribute__((address_space(1))) uint8_t *gbr_t;
        typedef const __attribute__((address_space(1))) U32u *gur_t;
        typedef __attribute__((address_space(1))) uint32_t *gw32_t;
        for (int g = lane; g < ngroups; g += 64) {
            int m = (g & (gw - 1)) * 4, n = g >> lgw;
            gbr_t r = (gbr_t) rbase + (__mul24(n, rp.stride) + m);
            // every load of the group up front (explicit global accesses), so that they share one round trip
            const uint8_t *srcd = (MODE == MC_SUBTRACT && jb.src[c]) ? jb.src[c] : sp.data;
            const uint32_t sv4 = *(gw32_t) (srcd + ((ptrdiff_t) y * sp.stride + x) + (__mul24(n, sp.stride) + m));
            int pv[4];
            if (intra) {
                uint32_t v = ((gur_t) r)->v;
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    int k = ((m + k4) >= sbw ? 1 : 0) | (n >= sbh ? 2 : 0);
                    bool fill = (mv.submask == DSV_MASK_ALL_INTRA) || (mv.submask & (1 << k));
                    pv[k4] = fill ? (dcq[k] & 0xff) : (int) ((v >> (8 * k4)) & 0xff);
                }
            } else if (subpel_luma && CC == 0) { // four rows of the intermediate image, two pixels an instruction
                const SubpelAxis AY = subpel_axis(fy, soft_y);
                const uint2 t0 = *(const uint2 *) &L.hz[n * 32 + m], t1 = *(const uint2 *) &L.hz[(n + 1) * 32 + m],
                            t2 = *(const uint2 *) &L.hz[(n + 2) * 32 + m], t3 = *(const uint2 *) &L.hz[(n + 3) * 32 + m];
                const s16x2 lo = spl16s(0), hi = spl16s(255);
                const s16x2 o01 = __builtin_elementwise_min(__builtin_elementwise_max(subpel_pk(AY, pk_of(t0.x), pk_of(t1.x), pk_of(t2.x), pk_of(t3.x)), lo), hi);
                const s16x2 o23 = __builtin_elementwise_min(__builtin_elementwise_max(subpel_pk(AY, pk_of(t0.y), pk_of(t1.y), pk_of(t2.y), pk_of(t3.y)), lo), hi);
                pv[0] = o01.x;
                pv[1] = o01.y;
                pv[2] = o23.x;
                pv[3] = o23.y;
            } else if (subpel_luma) {
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int16_t *t = &L.hz[n * 32 + m + k4];
                    pv[k4] = clamp_u8(qp_blend(hp_tap(t[0], t[32], t[64], t[96], soft_y), t[32], t[64], fy));
                }
            } else if (chroma_frac) {
                gbr_t r2 = r + rp.stride;
                const uint32_t va = ((gur_t) r)->v, vb = ((gur_t) r2)->v; // five pixels of two rows: a dword and a byte each
                const int a4 = r[4], b4 = r2[4];
                int a0 = va & 0xff, a1 = (va >> 8) & 0xff, a2 = (va >> 16) & 0xff, a3 = va >> 24;
                int b0 = vb & 0xff, b1 = (vb >> 8) & 0xff, b2 = (vb >> 16) & 0xff, b3 = vb >> 24;
                pv[0] = ((f0 * a0 + f1 * a1 + f2 * b0 + f3 * b1 + af) >> sf) & 0xff;
                pv[1] = ((f0 * a1 + f1 * a2 + f2 * b1 + f3 * b2 + af) >> sf) & 0xff;
                pv[2] = ((f0 * a2 + f1 * a3 + f2 * b2 + f3 * b3 + af) >> sf) & 0xff;
                pv[3] = ((f0 * a3 + f1 * a4 + f2 * b3 + f3 * b4 + af) >> sf) & 0xff;
            } else {
                uint32_t v = ((gur_t) r)->v;
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    pv[k4] = (int) ((v >> (8 * k4)) & 0xff);
                }
            }
            gw32_t dpx = (gw32_t) (dp.data + ((ptrdiff_t) y * dp.stride + x) + (__mul24(n, dp.stride) + m));
            gw32_t spx = (gw32_t) (sp.data + ((ptrdiff_t) y * sp.stride + x) + (__mul24(n, sp.stride) + m));
            const s16x2 p01 = (s16x2){(short) pv[0], (short) pv[1]}, p23 = (s16x2){(short) pv[2], (short) pv[3]};
            const uint32_t out = resid4_pk<MODE>(sv4, p01, p23, p.lossless, !intra && (skip || noxmit), eprm, !eprm || (!intra && skip));
            if (MODE == MC_SUBTRACT) {
                *dpx = __builtin_amdgcn_perm(u32_of(p23), u32_of(p01), 0x06040200u);
                *spx = out;
            } else {
                *dpx = out;
            }
        }
    }
}

// Both 8x8 chroma blocks of a 16x16 luma block in 4:2:0 in ONE pass: lanes 0..15 take U, 16..31 V (predict_plane gives a
// chroma plane 16 of the 64 lanes, twice).  Same arithmetic as predict_plane<MODE, 1 / 2>; the intra block's quadrant means
// are sums over the 16 lanes of a plane (a DPP row).  The two planes have one geometry (dframe_alloc).
__device__ __forceinline__ int row16_sum_i(int v)
{
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true); // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true); // row_mirror
    return v;
}

#ifndef DSV2_CHROMA_PAIR
#define DSV2_CHROMA_PAIR 1
#endif
constexpr bool kChromaPair = DSV2_CHROMA_PAIR != 0; // (build switch for A/B: make EXTRA=-DDSV2_CHROMA_PAIR=0)
template <int MODE, bool TILED = false>
__device__ __forceinline__ void predict_chroma_pair(const McJob &jb, const MCParams &p, const DSV_MV &mv, int i, int j, int ox = 0, int oy = 0)
{
    const int lane = threadIdx.x & 63;
    const int mvx = mv.u.mv.x, mvy = mv.u.mv.y;
    const uint32_t flags = mv.flags;
    const bool intra = flags & (1u << DSV_MV_BIT_INTRA);
    const bool skip = flags & (1u << DSV_MV_BIT_SKIP), eprm = flags & (1u << DSV_MV_BIT_EPRM);
    const bool v_plane = (lane & 16) != 0, on = lane < 32;
    const int bw = 8, bh = 8, sbw = 4, sbh = 4;
    const DPlane rp1 = jb.ref.p[1], dp1 = jb.pred.p[1], sp1 = jb.res.p[1];
    const uint8_t *rdata = v_plane ? jb.ref.p[2].data : rp1.data;
    uint8_t *ddata = v_plane ? jb.pred.p[2].data : dp1.data, *sdata = v_plane ? jb.res.p[2].data : sp1.data;
    const int Bw = TILED ? (p.blk_w >> 1) : bw, Bh = TILED ? (p.blk_h >> 1) : bh; // (see predict_plane)
    const int tox = TILED ? (ox >> 1) : 0, toy = TILED ? (oy >> 1) : 0;
    const int limx = (dp1.w - Bw) + kBorder - 1, limy = (dp1.h - Bh) + kBorder - 1;
    const int x = i * Bw + tox, y = j * Bh + toy;
    const int px = clampi(i * Bw + sar(mvx, 3), -kBorder, limx) + tox, py = clampi(j * Bh + sar(mvy, 3), -kBorder, limy) + toy;
    const uint8_t *rbase = rdata + (ptrdiff_t) py * rp1.stride + px;
    const int g = lane & 15, m = (g & 1) * 4, n = g >> 1; // 16 groups of 4 pixels, two per row
    typedef const __attribute__((address_space(1))) uint8_t *gbr_t;
    typedef const __attribute__((address_space(1))) U32u *gur_t;
    typedef __attribute__((address_space(1))) uint32_t *gw32_t;
    gbr_t r = (gbr_t) rbase + (__mul24(n, rp1.stride) + m);
    int dcq[4] = {0, 0, 0, 0};
    if (intra) { // quadrant means of the reference block (bmc.c:845-900); a transmitted DC is luma only
        const uint32_t v = ((gur_t) r)->v;
        const int s4 = (int) ((v & 0xff) + ((v >> 8) & 0xff) + ((v >> 16) & 0xff) + (v >> 24));
        const int k = (m >= sbw ? 1 : 0) | (n >= sbh ? 2 : 0);
        const int q0 = row16_sum_i(k == 0 ? s4 : 0), q1 = row16_sum_i(k == 1 ? s4 : 0), q2 = row16_sum_i(k == 2 ? s4 : 0),
                  q3 = row16_sum_i(k == 3 ? s4 : 0);
        if (mv.submask == DSV_MASK_ALL_INTRA) {
            dcq[0] = dcq[1] = dcq[2] = dcq[3] = (q0 + q1 + q2 + q3) / (bw * bh); // bmc.c:857
        } else {
            dcq[0] = q0 / (sbw * sbh); // bmc.c:884
            dcq[1] = q1 / (sbw * sbh);
            dcq[2] = q2 / (sbw * sbh);
            dcq[3] = q3 / (sbw * sbh);
        }
    }
    int f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    const int sf = 6, af = 32; // bmc.c:778-798 with eighth-pel chroma vectors
    bool chroma_frac = false;
    if (!intra) {
        const int dx = mvx & 7, dy = mvy & 7;
        chroma_frac = (dx | dy) != 0;
        f0 = (8 - dx) * (8 - dy);
        f1 = dx * (8 - dy);
        f2 = (8 - dx) * dy;
        f3 = dx * dy;
    }
    if (!on) {
        return;
    }
    const bool noxmit = (flags & (1u << DSV_MV_BIT_NOXMITC)) != 0;
    const uint8_t *srcd = sdata;
    if (MODE == MC_SUBTRACT && jb.src[1]) {
        srcd = v_plane ? jb.src[2] : jb.src[1];
    }
    const uint32_t sv4 = *(gw32_t) (srcd + ((ptrdiff_t) y * sp1.stride + x) + (__mul24(n, sp1.stride) + m));
    int pv[4];
    if (intra) {
        const uint32_t v = ((gur_t) r)->v;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const int k = ((m + k4) >= sbw ? 1 : 0) | (n >= sbh ? 2 : 0);
            const bool fill = (mv.submask == DSV_MASK_ALL_INTRA) || (mv.submask & (1 << k));
            pv[k4] = fill ? (dcq[k] & 0xff) : (int) ((v >> (8 * k4)) & 0xff);
        }
    } else if (chroma_frac) {
        // (f0 a + f1 a' + f2 b + f3 b' + 32) >> 6 with f0 + .. + f3 = 64: at most 64 * 255 + 32, two pixels an instruction
        gbr_t r2 = r + rp1.stride;
        const uint32_t va = ((gur_t) r)->v, vb = ((gur_t) r2)->v; // five pixels of two rows: a dword and a byte each
        const uint32_t a4 = r[4], b4 = r2[4];
        typedef unsigned short q16x2 __attribute__((ext_vector_type(2)));
        auto uq = [](uint32_t v) { return __builtin_bit_cast(q16x2, v); };
        const q16x2 a01 = uq(__builtin_amdgcn_perm(a4, va, 0x0c010c00u)), a12 = uq(__builtin_amdgcn_perm(a4, va, 0x0c020c01u)),
                    a23 = uq(__builtin_amdgcn_perm(a4, va, 0x0c030c02u)), a34 = uq(__builtin_amdgcn_perm(a4, va, 0x0c040c03u));
        const q16x2 b01 = uq(__builtin_amdgcn_perm(b4, vb, 0x0c010c00u)), b12 = uq(__builtin_amdgcn_perm(b4, vb, 0x0c020c01u)),
                    b23 = uq(__builtin_amdgcn_perm(b4, vb, 0x0c030c02u)), b34 = uq(__builtin_amdgcn_perm(b4, vb, 0x0c040c03u));
        const q16x2 F0 = (q16x2){(unsigned short) f0, (unsigned short) f0}, F1 = (q16x2){(unsigned short) f1, (unsigned short) f1},
                    F2 = (q16x2){(unsigned short) f2, (unsigned short) f2}, F3 = (q16x2){(unsigned short) f3, (unsigned short) f3},
                    AF = (q16x2){(unsigned short) af, (unsigned short) af}, SH = (q16x2){(unsigned short) sf, (unsigned short) sf};
        const q16x2 o01 = (F0 * a01 + F1 * a12 + F2 * b01 + F3 * b12 + AF) >> SH, o23 = (F0 * a23 + F1 * a34 + F2 * b23 + F3 * b34 + AF) >> SH;
        pv[0] = o01.x;
        pv[1] = o01.y;
        pv[2] = o23.x;
        pv[3] = o23.y;
    } else {
        const uint32_t v = ((gur_t) r)->v;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            pv[k4] = (int) ((v >> (8 * k4)) & 0xff);
        }
    }
    gw32_t dpx = (gw32_t) (ddata + ((ptrdiff_t) y * dp1.stride + x) + (__mul24(n, dp1.stride) + m));
    gw32_t spx = (gw32_t) (sdata + ((ptrdiff_t) y * sp1.stride + x) + (__mul24(n, sp1.stride) + m));
    const s16x2 p01 = (s16x2){(short) pv[0], (short) pv[1]}, p23 = (s16x2){(short) pv[2], (short) pv[3]};
    const uint32_t out = resid4_pk<MODE>(sv4, p01, p23, p.lossless, !intra && (skip || noxmit), eprm, !eprm || (!intra && skip));
    if (MODE == MC_SUBTRACT) {
        *dpx = __builtin_amdgcn_perm(u32_of(p23), u32_of(p01), 0x06040200u);
        *spx = out;
    } else {
        *dpx = out;
    }
}

// A block's record through the SCALAR cache (the address is wave-uniform and an earlier launch wrote the field): the vector, the
// flags and everything decoded from them then live in scalar registers, and every "is this block intra / skipped / sub-pel" is a
// scalar branch.  (Loaded through a generic pointer the compiler makes it a per-lane load of the same 16 bytes: the decode runs on
// the vector unit and each such test becomes a compare, an exec-mask save and a branch.)
__device__ __forceinline__ DSV_MV load_mv_uniform(const DSV_MV *p)
{
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) v4u_t *cv4u_t;
    const v4u_t w = *(cv4u_t) p;
    DSV_MV mv;
    mv.u.all = (int32_t) w.x;
    mv.flags = w.y;
    mv.err = (uint16_t) (w.z & 0xffffu);
    mv.dc = (uint16_t) (w.z >> 16);
    mv.submask = (uint8_t) (w.w & 0xffu);
    return mv;
}

// TILED (a kernel of its own: k_predict_w<MODE, PRED_TILED> -- the 16 x 16 kernel's code is what a 1080p launch keeps in the
// instruction cache beside every other group's kernels, and it measured 26 - 34 % slower under load with this form folded in):
// tiles = (tw | th << 8), a wavefront per 16 x 16 piece of a block of tw x th pieces (4:2:0, the host's choice: mc_tiles);
// (ti, tj) is then the piece's place in the picture.  Otherwise a wavefront per block.
// FORM: PRED_ANY = whatever the job's geometry asks for; PRED_16 = 16 x 16 blocks in 4:2:0 and nothing else (the host's promise: the kernel
// carries no other form's code or registers); PRED_TILED as above.
enum { PRED_ANY = 0, PRED_16 = 1, PRED_TILED = 2 };
template <int MODE, int FORM> __device__ __forceinline__ void predict_block_wave(const McJob &jb, int ti, int tj, WaveLds &L, int tiles)
{
    constexpr bool TILED = FORM == PRED_TILED;
    const MCParams p = jb.p;
    int i = ti, j = tj, ox = 0, oy = 0;
    if (TILED) {
        const int tw = tiles & 0xff, th = tiles >> 8; // 1 or 2 each
        i = tw == 2 ? ti >> 1 : ti;
        j = th == 2 ? tj >> 1 : tj;
        ox = tw == 2 ? 16 * (ti & 1) : 0;
        oy = th == 2 ? 16 * (tj & 1) : 0;
    }
    if (i >= p.nbh) {
        return;
    }
    const DSV_MV mv = load_mv_uniform(&jb.mvs[i + j * p.nbh]);
    if (FORM == PRED_16 || (FORM == PRED_ANY && p.blk_w == 16 && p.blk_h == 16 && p.hshift == 1 && p.vshift == 1)) { // (uniform over the launch)
        predict_plane<MODE, 0>(jb, p, mv, i, j, L, 0);
        if (kChromaPair) {
            predict_chroma_pair<MODE>(jb, p, mv, i, j);
        } else {
            predict_plane<MODE, 1>(jb, p, mv, i, j, L, 1);
            predict_plane<MODE, 2>(jb, p, mv, i, j, L, 2);
        }
    } else if (TILED && !(mv.flags & (1u << DSV_MV_BIT_INTRA))) {
        predict_plane<MODE, 0, true>(jb, p, mv, i, j, L, 0, ox, oy);
        predict_chroma_pair<MODE, true>(jb, p, mv, i, j, ox, oy);
    } else if (TILED && (ox | oy)) {
        // (an intra block -- sub-block means over the whole block -- is its first piece's work, in the general form below)
    } else if constexpr (FORM != PRED_16) {
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            predict_plane<MODE, -1>(jb, p, mv, i, j, L, c);
        }
    }
}

template <int MODE, int FORM> __global__ __launch_bounds__(256) void k_predict_w(const McJob *__restrict__ tab, int lds_rows, int tiles)
{
    DSV2_KERNEL_PRIO();
    extern __shared__ __align__(16) uint8_t predict_lds[];
    const McJob &jb = tab[blockIdx.z];
    // (the wavefront's index told to the compiler as wave-uniform: the block's origin, its vector and every base address
    // then live in scalar registers)
    const int w = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    int i = (int) blockIdx.x * 4 + w, j = blockIdx.y;
#ifdef DSV2_PRED_PAD // (experiment: the 16 x 16 kernel's code with the general kernel's register allocation)
    asm volatile("v_mov_b32 v55, 0" ::: "v55");
#endif
    {
        uint8_t *mine = predict_lds + (unsigned) w * wave_lds_bytes(lds_rows);
        WaveLds L{mine, (int16_t *) (mine + wave_lds_win_bytes(lds_rows))};
        predict_block_wave<MODE, FORM>(jb, i, j, L, tiles);
    }
}

// res <- recon(pred, res) over the block grid of every plane: grid = (x groups, rows, 3 n), one dword per thread
// (recon_px, bmc.c:953-983, two pixels an instruction: the residual's weight -- 1 for a plain block, 2 for an expanded-range one --
// is a per-lane multiplier, not a branch; (rv - 128) * 2 + pv stays inside 16 bits)
__device__ __forceinline__ uint32_t recon4(uint32_t rv4, uint32_t pv4, bool plain, int lossless)
{
    const s16x2 r01 = pk_of(__builtin_amdgcn_perm(0u, rv4, 0x0c010c00u)) - spl16s(128), r23 = pk_of(__builtin_amdgcn_perm(0u, rv4, 0x0c030c02u)) - spl16s(128);
    const s16x2 p01 = pk_of(__builtin_amdgcn_perm(0u, pv4, 0x0c010c00u)), p23 = pk_of(__builtin_amdgcn_perm(0u, pv4, 0x0c030c02u));
    const s16x2 lo = spl16s(0), hi = spl16s(255);
    s16x2 o01, o23;
    if (lossless) { // (uniform over the launch)
        o01 = (p01 + r01) & hi;
        o23 = (p23 + r23) & hi;
    } else {
        const s16x2 m = spl16s(plain ? 1 : 2);
        o01 = __builtin_elementwise_min(__builtin_elementwise_max(p01 + r01 * m, lo), hi);
        o23 = __builtin_elementwise_min(__builtin_elementwise_max(p23 + r23 * m, lo), hi);
    }
    return __builtin_amdgcn_perm(u32_of(o23), u32_of(o01), 0x06040200u);
}

// sixteen pixels of a row per thread (one 16-byte load of the residual and of the prediction, one 16-byte store); a
// group of four pixels never straddles a block (block widths are multiples of 8 in every plane), so each dword takes the
// flags of its own block.  Planes come from dframe_alloc: rows and origins are 16-byte aligned.
constexpr int kReconRows = 2; // rows per thread (y, y + 4): both rows' loads issued before the first is worked on (four rows: 17 % slower under load)
__global__ __launch_bounds__(256) void k_reconstruct_w(const McJob *__restrict__ tab)
{
    DSV2_KERNEL_PRIO();
    const McJob &jb = tab[blockIdx.z / 3];
    const MCParams p = jb.p;
    int c = blockIdx.z % 3;
    int sh = c ? p.hshift : 0, sv = c ? p.vshift : 0;
    int bw = p.blk_w >> sh, bh = p.blk_h >> sv;
    // (block sizes are powers of two -- 16 << e, halved by the chroma shifts: pixel -> block by a shift; the compiler's integer
    // division of a runtime divisor is ~30 instructions, and this kernel made five of them per thread)
    const int lbw = 31 - __builtin_clz((unsigned) bw), lbh = 31 - __builtin_clz((unsigned) bh);
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 16, y0 = blockIdx.y * (4 * kReconRows) + (threadIdx.x >> 6);
    const int xlim = p.nbh * bw, ylim = p.nbv * bh;
    if (x >= xlim || y0 >= ylim) {
        return;
    }
    const DPlane dp = jb.pred.p[c], sp = jb.res.p[c];
    const bool whole = x + 16 <= xlim;
    uint32_t rv[kReconRows][4], pv[kReconRows][4], fl[kReconRows][4];
#pragma unroll
    for (int r = 0; r < kReconRows; r++) { // (a row past the plane's end re-reads row y0: harmless, never stored)
        const int y = y0 + 4 * r < ylim ? y0 + 4 * r : y0;
        const DSV_MV *row = jb.mvs + (y >> lbh) * p.nbh;
        const uint8_t *spx = sp.data + (ptrdiff_t) y * sp.stride + x;
        const uint8_t *dpx = dp.data + (ptrdiff_t) y * dp.stride + x;
        if (whole) {
            const uint4 rr = *(const uint4 *) spx, q = *(const uint4 *) dpx;
            rv[r][0] = rr.x, rv[r][1] = rr.y, rv[r][2] = rr.z, rv[r][3] = rr.w;
            pv[r][0] = q.x, pv[r][1] = q.y, pv[r][2] = q.z, pv[r][3] = q.w;
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bool in = x + 4 * g < xlim;
                rv[r][g] = in ? *(const uint32_t *) (spx + 4 * g) : 0u;
                pv[r][g] = in ? *(const uint32_t *) (dpx + 4 * g) : 0u;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            fl[r][g] = row[min((x + 4 * g) >> lbw, p.nbh - 1)].flags;
        }
    }
#pragma unroll
    for (int r = 0; r < kReconRows; r++) {
        const int y = y0 + 4 * r;
        if (y >= ylim) {
            break;
        }
        uint8_t *spx = sp.data + (ptrdiff_t) y * sp.stride + x;
        uint32_t o[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t flags = fl[r][g];
            bool plain = !(flags & (1u << DSV_MV_BIT_EPRM)) || (!(flags & (1u << DSV_MV_BIT_INTRA)) && (flags & (1u << DSV_MV_BIT_SKIP)));
            o[g] = recon4(rv[r][g], pv[r][g], plain, p.lossless);
        }
        if (whole) {
            *(uint4 *) spx = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (x + 4 * g < xlim) {
                    *(uint32_t *) (spx + 4 * g) = o[g];
                }
            }
        }
    }
}

// ---- 4x4 edge smoothing primitives (bmc.c:53-191) ------------------------------------------

// Evaluated without branches: the six range tests are combined bitwise and the four outputs are always
// formed; callers select between old and new samples.  The wave executes every path of divergent code anyway,
// so predication costs nothing here and removes ~a thousand exec-mask branches per cell.
// |a - b| of two non-negative values in ONE instruction (v_sad_u32 with a zero accumulator; the compiler's own form of
// abs(a - b) is two subtractions and a maximum, and a cell evaluates some 130 of them)
__device__ __forceinline__ int absdiff(int a, int b)
{
    int d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ bool smooth6(int e2, int e1, int e0, int i0, int i1, int i2, int t, int o[4])
{
    int avg = (5 * (e0 + i0) + 3 * (e1 + i1) + 8) >> 4;
    // all six samples within t of the average <=> the largest deviation is (bmc.c:53-70: six range tests)
    int dev = max(max(max(absdiff(e0, avg), absdiff(i0, avg)), max(absdiff(e1, avg), absdiff(i1, avg))), max(absdiff(e2, avg), absdiff(i2, avg)));
    bool ok = dev < t;
    int a5 = avg * 5;
    o[0] = (3 * (avg + e1) + 2 * e2 + 4) >> 3;
    o[1] = (a5 + 2 * e1 + e2 + 4) >> 3;
    o[2] = avg;
    o[3] = (a5 + 2 * i1 + i2 + 4) >> 3;
    return ok;
}

// ---- the filters of one 4x4 cell, evaluated in registers ---------------------------------------
// Every pixel a cell's horizontal / vertical smoothing can touch is fetched once (19 aligned dword
// loads), all filter passes then run on registers and only modified rows go back (dword stores):
// two memory round trips per cell instead of one per 6-tap line.  t[r][c] = pixel (x - 4 + c,
// y - 3 + r): rows 3..6 hold 12 columns, rows 0..2 and 7..10 hold only the cell's own columns 4..7.
// None of these bytes is written by another cell of the same sweep front (cells (i-2, j+1) and
// (i+2, j-1) reach at most column x-5 / row y-1 of our rows), so whole-dword stores are safe.
// where a Tile's pixels live: the plane in global memory ...
struct GlobalView {
    static constexpr bool kStoreAll = false; // write back modified rows only
    DPlane dp;
    __device__ __forceinline__ uint32_t ld(int row, int col) const { return *(const uint32_t *) (dp.data + (ptrdiff_t) row * dp.stride + col); }
    __device__ __forceinline__ void st(int row, int col, uint32_t v) const { *(uint32_t *) (dp.data + (ptrdiff_t) row * dp.stride + col) = v; }
};
// ... or the LDS ring of a plane-resident sweep (ring_sweep below): one 64-byte ring per pixel row holding a
// sliding 64-column window, plus guard rows above and below the plane.
struct RingView {
    static constexpr bool kStoreAll = true; // LDS: cheaper to write the whole cross back than to branch per row
    static constexpr int kGuardTop = 3, kGuardBelow = 4, kGuardRows = 11; // a cell's cross reaches 3 rows above and 4 below the plane's rows; 4 parking rows (ring_sweep)
    uint8_t *ring; // row 0 of the plane (kGuardTop rows into the allocation)
    int h;
    // Rows outside the plane are never filtered (the filters skip y < 4 and y > h - 4): what is loaded from the guard rows
    // is never used and what is stored there never retired, so neither is clamped or tested.
    __device__ __forceinline__ uint32_t ld(int row, int col) const { return *(const uint32_t *) (ring + row * 64 + (col & 63)); }
    __device__ __forceinline__ void st(int row, int col, uint32_t v) const { *(uint32_t *) (ring + row * 64 + (col & 63)) = v; }
    // two pixels (col even)
    __device__ __forceinline__ uint32_t ld16(int row, int col) const { return *(const uint16_t *) (ring + row * 64 + (col & 63)); }
    __device__ __forceinline__ void st16(int row, int col, uint32_t v) const { *(uint16_t *) (ring + row * 64 + (col & 63)) = (uint16_t) v; }
};

struct Tile {
    int t[11][12];
    unsigned dirty; // bit r: row r modified

    template <class V> __device__ __forceinline__ void load(const V &view, int x, int y)
    {
#pragma unroll
        for (int r = 0; r < 11; r++) {
            if (r >= 3 && r <= 6) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    uint32_t v = view.ld(y - 3 + r, x - 4 + 4 * d);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        t[r][4 * d + k] = (int) ((v >> (8 * k)) & 0xffu);
                    }
                }
            } else {
                uint32_t v = view.ld(y - 3 + r, x);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    t[r][4 + k] = (int) ((v >> (8 * k)) & 0xffu);
                }
            }
        }
        dirty = 0;
    }

    template <class V> __device__ __forceinline__ void store(const V &view, int x, int y) const
    {
        if (!V::kStoreAll && !dirty) {
            return;
        }
#pragma unroll
        for (int r = 0; r < 11; r++) {
            if (!V::kStoreAll && !(dirty & (1u << r))) {
                continue;
            }
            if (V::kStoreAll && !__any((dirty >> r) & 1u)) { // LDS: a row no cell of the wavefront changed is not packed and written at all
                continue;
            }
            if (r >= 3 && r <= 6) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    view.st(y - 3 + r, x - 4 + 4 * d,
                            (uint32_t) t[r][4 * d] | ((uint32_t) t[r][4 * d + 1] << 8) | ((uint32_t) t[r][4 * d + 2] << 16) |
                                ((uint32_t) t[r][4 * d + 3] << 24));
                }
            } else {
                view.st(y - 3 + r, x, (uint32_t) t[r][4] | ((uint32_t) t[r][5] << 8) | ((uint32_t) t[r][6] << 16) | ((uint32_t) t[r][7] << 24));
            }
        }
    }
};

// one 11-sample line through the cell: l[3] is the cell's first sample (bmc.c:71-191)
__device__ __forceinline__ bool line_filter(int (&l)[11], bool in_edge, int tE, int tM)
{
    int o[4], p[4];
    bool h1 = smooth6(l[0], l[1], l[2], l[3], l[4], l[5], tE, o);
    bool h2 = in_edge & smooth6(l[10], l[9], l[8], l[7], l[6], l[5], tM, p); // reads l[5..10] only: untouched by the first half
    // (smooth6's outputs are weighted means of samples <= 255 with weights summing to 8 resp. 16: they fit a byte as they are)
    l[1] = h1 ? o[0] : l[1];
    l[2] = h1 ? o[1] : l[2];
    l[3] = h1 ? o[2] : l[3];
    l[4] = h1 ? o[3] : l[4];
    l[6] = h2 ? p[3] : l[6];
    l[7] = h2 ? p[2] : l[7];
    l[8] = h2 ? p[1] : l[8];
    l[9] = h2 ? p[0] : l[9];
    return h1 | h2;
}

// `on` = false turns the pass into a no-op (thresholds 0: no range test can pass)
__device__ __forceinline__ void hfilter(Tile &T, const DPlane &dp, int x, bool edge, int tE, int tM, bool on = true)
{
    on = on & !(x < 4 || x > dp.w - 4 || (edge && tE <= 0) || tM <= 0);
    tE = edge ? tE : tM;
    tE = on ? tE : 0;
    tM = on ? tM : 0;
    bool in_edge = x < dp.w - 8;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        int l[11];
#pragma unroll
        for (int k = 0; k < 11; k++) {
            l[k] = T.t[3 + n][1 + k];
        }
        bool hit = line_filter(l, in_edge, tE, tM);
#pragma unroll
        for (int k = 1; k < 10; k++) {
            T.t[3 + n][1 + k] = l[k];
        }
        T.dirty |= hit ? (1u << (3 + n)) : 0u;
    }
}

__device__ __forceinline__ void vfilter(Tile &T, const DPlane &dp, int y, bool edge, int tE, int tM, bool on = true)
{
    on = on & !(y < 4 || y > dp.h - 4 || (edge && tE <= 0) || tM <= 0);
    tE = edge ? tE : tM;
    tE = on ? tE : 0;
    tM = on ? tM : 0;
    bool in_edge = y < dp.h - 8;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        int l[11];
#pragma unroll
        for (int k = 0; k < 11; k++) {
            l[k] = T.t[k][4 + n];
        }
        bool hit = line_filter(l, in_edge, tE, tM);
#pragma unroll
        for (int k = 1; k < 10; k++) {
            T.t[k][4 + n] = l[k];
        }
        T.dirty |= hit ? 0x3deu : 0u; // rows 1..4 and 6..9
    }
}

// chroma: one horizontal (4 rows x 11 columns at (x, y)) or vertical (11 rows x 4 columns) pass on
// its own, fetched and written back as aligned dwords
__device__ __forceinline__ void hfilter_mem(const DPlane &dp, int x, int y, int tE, int tM)
{
    if (x < 4 || x > dp.w - 4 || tM <= 0) {
        return;
    }
    (void) tE;
    bool in_edge = x < dp.w - 8;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        uint32_t *row = (uint32_t *) (dp.data + (ptrdiff_t) (y + n) * dp.stride + (x - 4));
        uint32_t d[3] = {row[0], row[1], row[2]};
        int l[11];
#pragma unroll
        for (int k = 0; k < 11; k++) {
            l[k] = (int) ((d[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xffu);
        }
        if (line_filter(l, in_edge, tM, tM)) {
            int c0 = (int) (d[0] & 0xffu);
            row[0] = (uint32_t) c0 | ((uint32_t) l[0] << 8) | ((uint32_t) l[1] << 16) | ((uint32_t) l[2] << 24);
            row[1] = (uint32_t) l[3] | ((uint32_t) l[4] << 8) | ((uint32_t) l[5] << 16) | ((uint32_t) l[6] << 24);
            row[2] = (uint32_t) l[7] | ((uint32_t) l[8] << 8) | ((uint32_t) l[9] << 16) | ((uint32_t) l[10] << 24);
        }
    }
}

__device__ __forceinline__ void vfilter_mem(const DPlane &dp, int x, int y, int tE, int tM)
{
    if (y < 4 || y > dp.h - 4 || tM <= 0) {
        return;
    }
    (void) tE;
    bool in_edge = y < dp.h - 8;
    uint32_t d[11];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        d[k] = *(const uint32_t *) (dp.data + (ptrdiff_t) (y - 3 + k) * dp.stride + x);
    }
    unsigned dirty = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        int l[11];
#pragma unroll
        for (int k = 0; k < 11; k++) {
            l[k] = (int) ((d[k] >> (8 * n)) & 0xffu);
        }
        if (line_filter(l, in_edge, tM, tM)) {
#pragma unroll
            for (int k = 1; k < 10; k++) {
                d[k] = (d[k] & ~(0xffu << (8 * n))) | ((uint32_t) l[k] << (8 * n));
            }
            dirty = 1;
        }
    }
    if (dirty) {
#pragma unroll
        for (int k = 1; k < 10; k++) {
            if (k != 5) {
                *(uint32_t *) (dp.data + (ptrdiff_t) (y - 3 + k) * dp.stride + x) = d[k];
            }
        }
    }
}

// ---- 4x4 cell statistics; CELL(y, x) yields the cell's pixel ---------------------------------------
#define DS2X2(CELL, d)                                                        \
    do {                                                                      \
        d[0] = (CELL(0, 0) + CELL(0, 1) + CELL(1, 0) + CELL(1, 1) + 2) >> 2;  \
        d[1] = (CELL(0, 2) + CELL(0, 3) + CELL(1, 2) + CELL(1, 3) + 2) >> 2;  \
        d[2] = (CELL(2, 0) + CELL(2, 1) + CELL(3, 0) + CELL(3, 1) + 2) >> 2;  \
        d[3] = (CELL(2, 2) + CELL(2, 3) + CELL(3, 2) + CELL(3, 3) + 2) >> 2;  \
    } while (0)

__device__ __forceinline__ unsigned dsff_d(int d[4]) // bmc.c:194
{
    unsigned sh = (unsigned) absdiff(d[0] + d[1], d[3] + d[2]);
    unsigned sv = (unsigned) absdiff(d[2] + d[1], d[3] + d[0]);
    if (max(sh, sv) < 8) {
        return 0;
    }
    d[2] = 255 - d[2];
    d[3] = 255 - d[3];
    sh = (unsigned) abs(d[0] - d[1] + d[2] - d[3]);
    sv = (unsigned) abs(d[0] + d[1] - d[2] - d[3]) >> 2;
    return sh > sv ? (3 * sh + sv + 2) >> 2 : (3 * sv + sh + 2) >> 2;
}

__device__ __forceinline__ unsigned dsff(const Tile &T)
{
    int d[4];
#define TCELL(yy, xx) T.t[3 + (yy)][4 + (xx)]
    DS2X2(TCELL, d);
    return dsff_d(d);
}

__device__ __forceinline__ void artf(const Tile &T, int &sh, int &sv, int &slh, int &slv) // bmc.c:224-270
{
    sh = sv = 0;
#pragma unroll
    for (int y = 0; y < 4; y += 2) {
#pragma unroll
        for (int x = 0; x < 4; x += 2) {
            int x0 = TCELL(y, x), x1 = TCELL(y, x + 1), x2 = TCELL(y + 1, x), x3 = TCELL(y + 1, x + 1);
            int hh = absdiff(x0 + x3, x1 + x2) >> 1;
            sh += absdiff(x0 + x2, x1 + x3) + hh;
            sv += absdiff(x0 + x1, x2 + x3) + hh;
        }
    }
    int d[4];
    DS2X2(TCELL, d);
    int hh = absdiff(d[0] + d[3], d[1] + d[2]) >> 1;
    slh = absdiff(d[0] + d[2], d[1] + d[3]) + hh;
    slv = absdiff(d[0] + d[1], d[2] + d[3]) + hh;
}

// n / d for 0 <= n < 2^12, 1 <= d <= 16 (sums of at most 16 pixels over their count): reciprocal estimate + one correction
__device__ __forceinline__ int div_small(int n, int d)
{
    int q = (int) ((float) n * __builtin_amdgcn_rcpf((float) d));
    int r = n - q * d;
    return r < 0 ? q - 1 : (r >= d ? q + 1 : q);
}

// one pixel of the de-gradient step (bmc.c:318-330), without branches: pulled towards the mean of the darkest / brightest
// bin by count / 16 (C division: towards zero; |count * difference| < 2^12, a 24-bit multiply)
__device__ __forceinline__ int degrad_px(int os, int t, int nlo, int alo, int nhi, int ahi)
{
    bool lt = os < t;
    int v = __mul24(lt ? nlo : nhi, (lt ? alo : ahi) - os);
    int adj = (v + ((v >> 31) & 15)) >> 4;
    return os == t ? os : (os + adj) & 0xff;
}

// de-gradient sharpening of 16 pixels px[y * 4 + x] in place (bmc.c:276); returns true when it changed them
__device__ __forceinline__ bool degrad16(int (&px)[16])
{
    int lo = 16, hi = -1;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo = min(lo, px[k] >> 4);
        hi = max(hi, px[k] >> 4);
    }
    if (lo >= hi) {
        return false;
    }
    int nlo = 0, nhi = 0, slo = 0, shi = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int b = px[k] >> 4;
        if (b == lo) {
            nlo++;
            slo += px[k];
        }
        if (b == hi) {
            nhi++;
            shi += px[k];
        }
    }
    int alo = div_small(slo, nlo), ahi = div_small(shi, nhi);
    if (alo == 0) {
        alo = 1;
    }
    if (ahi == 0) {
        ahi = 1;
    }
    int t = (alo + ahi + 1) >> 1;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        px[k] = degrad_px(px[k], t, nlo, alo, nhi, ahi);
    }
    return true;
}

__device__ __forceinline__ void degrad(Tile &T)
{
    int px[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        px[k] = TCELL(k >> 2, k & 3);
    }
    if (degrad16(px)) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            T.t[3 + (k >> 2)][4 + (k & 3)] = px[k];
        }
        T.dirty |= 0x78u;
    }
}

// the same on memory (decoder post-processing: every cell is independent)
__device__ void degrad(uint8_t *a, int as)
{
    int px[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        px[k] = a[(k >> 2) * as + (k & 3)];
    }
    if (degrad16(px)) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a[(k >> 2) * as + (k & 3)] = (uint8_t) px[k];
        }
    }
}

__device__ __forceinline__ int curve_tex(int tt) // bmc.c:364
{
    if (tt < 8) {
        return (8 - tt) * 8;
    }
    if (tt > 192) {
        return 0;
    }
    return tt - 7;
}

__device__ __forceinline__ void neighbordif2(const DSV_MV *v, int nbh, int x, int y, int &dx, int &dy) // dsv.c:403
{
    const DSV_MV *c = &v[x + y * nbh];
    int cx = c->u.mv.x, cy = c->u.mv.y, lx = cx, ly = cy, tx = cx, ty = cy;
    if (abs(cx) < 2 && abs(cy) < 2) {
        dx = dy = 0;
        return;
    }
    if (x > 0) {
        const DSV_MV *m = c - 1;
        if (m->u.all && !(m->flags & (1u << DSV_MV_BIT_SKIP))) {
            lx = m->u.mv.x;
            ly = m->u.mv.y;
        }
    }
    if (y > 0) {
        const DSV_MV *m = c - nbh;
        if (m->u.all && !(m->flags & (1u << DSV_MV_BIT_SKIP))) {
            tx = m->u.mv.x;
            ty = m->u.mv.y;
        }
    }
    dx = abs(lx - cx) + abs(ly - cy);
    dy = abs(tx - cx) + abs(ty - cy);
}

// explicit global-memory accesses: a generic pointer would compile to FLAT instructions, which count against the
// LDS counter as well and so make every LDS wait of the sweep wait for global memory
typedef const __attribute__((address_space(1))) uint32_t *gu32_t;
typedef __attribute__((address_space(1))) uint32_t *gu32w_t;

// what a luma cell needs of the motion field: its block's record and the left / top neighbours' vectors
struct CellRec {
    uint32_t all, flags, submask; // this block: packed {x, y}, flags, sub-block intra mask
    uint32_t l_all, l_flags, t_all, t_flags;
};

// (opaque to the compiler, which would otherwise narrow the load to a byte and extend it right behind the load: a record
// fetched ahead would be waited for at once)
__device__ __forceinline__ uint32_t low_byte(uint32_t v)
{
    uint32_t r;
    asm("v_and_b32 %0, 0xff, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ CellRec fetch_cell_rec(const DSV_MV *vecs, int nbh, int fx, int fy)
{
    gu32_t c = (gu32_t) &vecs[fx + fy * nbh];
    CellRec r;
    r.all = c[0];
    r.flags = c[1];
    r.submask = c[3]; // raw: landed() masks it
    r.l_all = r.l_flags = r.t_all = r.t_flags = 0;
    if (fx > 0) {
        r.l_all = c[-4];
        r.l_flags = c[-3];
    }
    if (fy > 0) {
        r.t_all = c[-4 * nbh];
        r.t_flags = c[-4 * nbh + 1];
    }
    return r;
}

// a fetched record made usable (behind the wait for its loads)
__device__ __forceinline__ CellRec landed(CellRec r)
{
    r.submask = low_byte(r.submask);
    return r;
}

__device__ __forceinline__ int mvx_of(uint32_t all) { return (int) (int16_t) (all & 0xffffu); }
__device__ __forceinline__ int mvy_of(uint32_t all) { return (int) (int16_t) (all >> 16); }

__device__ __forceinline__ void neighbordif2(const CellRec &r, int fx, int fy, int &dx, int &dy) // dsv.c:403
{
    int cx = mvx_of(r.all), cy = mvy_of(r.all), lx = cx, ly = cy, tx = cx, ty = cy;
    if (abs(cx) < 2 && abs(cy) < 2) {
        dx = dy = 0;
        return;
    }
    if (fx > 0 && r.l_all && !(r.l_flags & (1u << DSV_MV_BIT_SKIP))) {
        lx = mvx_of(r.l_all);
        ly = mvy_of(r.l_all);
    }
    if (fy > 0 && r.t_all && !(r.t_flags & (1u << DSV_MV_BIT_SKIP))) {
        tx = mvx_of(r.t_all);
        ty = mvy_of(r.t_all);
    }
    dx = abs(lx - cx) + abs(ly - cy);
    dy = abs(tx - cx) + abs(ty - cy);
}

// (a * num) / den for the cell -> block mapping of the filters: single-precision estimate + multiply-back instead of the
// ~35-instruction integer divide, four to six times per cell.  The estimate is exact while the product stays below 2^20
// (a cell index below 2^10 times a block count below 2^10: every picture up to ~4096 pixels a side); larger pictures (the
// decoder accepts up to 16384 x 16384) take the wave-uniform exact divide below -- tests/test_gpu_bmc.py covers both.
__device__ __forceinline__ int scale_div(int a, int num, int den)
{
    const unsigned n = (unsigned) (a * num), d = (unsigned) den;
    unsigned est = (unsigned) ((float) n * __builtin_amdgcn_rcpf((float) d));
    const int r = (int) (n - est * d);
    est = r < 0 ? est - 1u : (r >= (int) d ? est + 1u : est);
    if (__builtin_expect(__any((n | d) >= (1u << 20)), 0)) { // large pictures: exact integer divide for the whole wave
        est = n / d;
    }
    return (int) est;
}

// ---- one cell of each filter (oracle/orc_bmc.c: intra_cell, luma_cell, chroma_block) ----------

template <class V>
__device__ void intra_cell(const V &view, const DPlane &dp, const FilterParams &f, const uint8_t *bd, int i, int j, int nsbx, int nsby)
{
    int x = i * 4, y = j * 4;
    bool live = !(y + 4 >= dp.h || x + 4 >= dp.w);
    int flags = live ? bd[scale_div(i, f.nbh, nsbx) + scale_div(j, f.nbv, nsby) * f.nbh] : DSV_IS_RINGING;
    live = live & !(flags & DSV_IS_RINGING);
    if (!__any(live)) { // nothing to do for the whole wavefront
        return;
    }
    Tile T;
    T.load(view, x, y);
    int sh, sv, shl, svl;
    artf(T, sh, sv, shl, svl);
    int mx = max(sh, sv);
    live = live & (mx < 256 && mx > 8);
    int tt = 32;
    {
        int td = (int) dsff(T);
        td = (flags & DSV_IS_STABLE) ? (td * 5 >> 2) : td;
        tt = (flags & (DSV_IS_MAINTAIN | DSV_IS_STABLE)) ? td : (tt >> 2);
    }
    tt = tt * 2 / 3;
    tt = (tt * f.q) >> 12;
    tt = clampi(tt, 0, f.fthresh);
    hfilter(T, dp, x, false, tt, tt, live);
    vfilter(T, dp, y, false, tt, tt, live);
    tt = sh > sv ? (3 * sh + sv) : (3 * sv + sh);
    tt = curve_tex(tt);
    tt = 16 + ((tt + 2) >> 2);
    tt = (tt * f.q) >> 12;
    tt = clampi(tt, 0, f.fthresh);
    hfilter(T, dp, x, false, tt, tt, live);
    vfilter(T, dp, y, false, tt, tt, live);
    T.store(view, x, y);
}

template <class V>
__device__ void luma_cell_rec(const V &view, const DPlane &dp, const FilterParams &f, const CellRec &rec, int fx, int fy, int i, int j)
{
    int x = i * 4, y = j * 4;
    uint32_t flags = rec.flags;
    bool live = !(y + 4 >= dp.h || (flags & (1u << DSV_MV_BIT_SKIP)) || x + 4 >= dp.w);
    // (block sizes are powers of two: 16 << e)
    bool edgeh = (x & (f.blk_w - 1)) == 0, edgehs = (x & (f.blk_w / 2 - 1)) == 0;
    bool edgev = (y & (f.blk_h - 1)) == 0, edgevs = (y & (f.blk_h / 2 - 1)) == 0;
    int mvx = mvx_of(rec.all), mvy = mvy_of(rec.all);
    int amx = abs(mvx), amy = abs(mvy);
    bool intra = (flags & (1u << DSV_MV_BIT_INTRA)) != 0;
    int ndx = 0, ndy = 0;
    if (f.do_filter) {
        neighbordif2(rec, fx, fy, ndx, ndy);
    }
    bool filt = !intra && (ndx || ndy);
    bool sharp = !intra && f.sharpen && (mvx & 3) && (mvy & 3) && ((mvx | mvy) & 1) && amx < 8 && amy < 8;
    live = live & (intra | filt | sharp);
    if (!__any(live)) { // nothing to do for the whole wavefront
        return;
    }
    Tile T;
    T.load(view, x, y);
    // the two passes and their thresholds, by block type (bmc.c:527-596)
    bool h_on, v_on, eh, ev;
    int hE, hM, vE, vM;
    {
        // intra block
        int tH = clampi((64 * f.q) >> 12, 2, 32), tL = clampi((32 * f.q) >> 12, 2, 32);
        bool part = rec.submask != DSV_MASK_ALL_INTRA;
        bool ieh = edgeh | (part & edgehs), iev = edgev | (part & edgevs);
        // inter block with a motion discontinuity
        bool eprm = (flags & (1u << DSV_MV_BIT_EPRM)) != 0;
        int tndc = (ndx + ndy + 1) >> 1;
        int sh, sv, shl, svl, tt;
        artf(T, sh, sv, shl, svl);
        int n_dx = ndx, n_dy = ndy;
        bool mixed = sh < 2 * sv && sv < 2 * sh;
        {
            int mdx = ndx < amx ? ndx >> 1 : ndx, mdy = ndy < amy ? ndy >> 1 : ndy;
            int shl2 = shl > 128 ? 0 : 128 - shl, svl2 = svl > 128 ? 0 : 128 - svl;
            int ix = min(amx, 32), iy = min(amy, 32);
            int tm = ((sh * (32 - iy) + shl2 * iy) + 16) >> 5;
            tm += ((sv * (32 - ix) + svl2 * ix) + 16) >> 5;
            tm = (tm + 1) >> 1;
            tm = (mdx < amy && mdy < amx) ? 0 : tm;
            tt = mixed ? tm : ((sh + sv + 1) >> 1);
            n_dx = mixed ? mdx : ndx;
            n_dy = mixed ? mdy : ndy;
        }
        tt = (tt * tndc + 4) >> 3;
        tt = (min(tt, f.fthresh) * f.q) >> 12;
        int addx = (min(n_dy, f.fthresh) * f.q) >> 12;
        int addy = (min(n_dx, f.fthresh) * f.q) >> 12;
        bool v_only = sh > 2 * sv || amy > 2 * amx;
        bool h_only = !v_only && (sv > 2 * sh || amx > 2 * amy);
        h_on = live & (intra | (filt & !v_only));
        v_on = live & (intra | (filt & !h_only));
        eh = intra ? ieh : (edgeh | eprm);
        ev = intra ? iev : (edgev | eprm);
        hE = intra ? tH : tt + addx;
        hM = intra ? tL : tt;
        vE = intra ? tH : tt + addy;
        vM = intra ? tL : tt;
    }
    // (a pass that no cell of the wavefront wants is a no-op by construction: skipping it is a wave-uniform branch)
    if (__any(h_on)) {
        hfilter(T, dp, x, eh, hE, hM, h_on);
    }
    if (__any(v_on)) {
        vfilter(T, dp, y, ev, vE, vM, v_on);
    }
    if (__any(live & sharp)) {
        if (live & sharp) {
            degrad(T);
        }
    }
    T.store(view, x, y);
}

template <class V>
__device__ void luma_cell(const V &view, const DPlane &dp, const FilterParams &f, const DSV_MV *vecs, int i, int j, int nsbx, int nsby)
{
    int fx = scale_div(i, f.nbh, nsbx), fy = scale_div(j, f.nbv, nsby);
    luma_cell_rec(view, dp, f, landed(fetch_cell_rec(vecs, f.nbh, fx, fy)), fx, fy, i, j);
}

__device__ void chroma_block(const DPlane &dp, const FilterParams &f, const DSV_MV *vecs, int i, int j)
{
    int bw = f.blk_w >> f.hshift, bh = f.blk_h >> f.vshift;
    int x = i * bw, y = j * bh;
    const DSV_MV *mv = &vecs[i + j * f.nbh];
    uint32_t flags = mv->flags;
    if (flags & (1u << DSV_MV_BIT_SKIP)) {
        return;
    }
    int it = clampi((64 * f.q_raw) >> 12, 2, 32);
    int tx = it, ty = it;
    if (!(flags & (1u << DSV_MV_BIT_INTRA))) {
        int ndx, ndy, amx = abs((int) mv->u.mv.x), amy = abs((int) mv->u.mv.y);
        neighbordif2(vecs, f.nbh, i, j, ndx, ndy);
        if (ndx < amy && ndy < amx) {
            tx = ty = 0;
        } else {
            tx = (min(ndy, 64) * f.q_raw) >> 12;
            ty = (min(ndx, 64) * f.q_raw) >> 12;
        }
    }
    for (int z = 0; z < bh; z += 4) {
        if (y + z + 4 < dp.h) {
            hfilter_mem(dp, x, y + z, tx, tx);
        }
    }
    for (int z = 0; z < bw; z += 4) {
        if (x + z + 4 < dp.w) {
            vfilter_mem(dp, x + z, y, ty, ty);
        }
    }
}

// ---- two lanes per cell (the batched luma sweeps) ------------------------------------------------------------------
// The sweep's critical path is one cell routine per front, run by one wavefront per SIMD at the 4 - 5 clocks per
// instruction a lone wavefront issues at.  Here an even / odd lane pair shares a cell: lane p owns the cell's pixel rows
// 2p, 2p + 1 for the horizontal pass and its pixel columns 2p, 2p + 1 for the vertical one (every 11-sample line is
// filtered exactly as before, by one lane), the 2 x 2 hand-over between the passes and the cell's joint statistics go
// through DPP lane swaps.  Half the instructions per lane, two wavefronts per SIMD.
#ifdef DSV2_FILTER_PROF
__device__ unsigned long long g_filt_cnt[8]; // (wavefront, front) pairs of the pair sweep: [0] all [1] any live [2] any horizontal [3] any vertical [4] any sharpen [5] live cells
#define FILT_COUNT(k, cond)                                                                  \
    do {                                                                                     \
        unsigned long long b_ = __ballot(cond);                                              \
        if (b_ && (int) (threadIdx.x & 63) == __ffsll((long long) __ballot(1)) - 1) {        \
            atomicAdd(&g_filt_cnt[k], (k) == 5 ? (unsigned long long) __popcll(b_) : 1ull);  \
        }                                                                                    \
    } while (0)
#else
#define FILT_COUNT(k, cond)                                                                  \
    do {                                                                                     \
    } while (0)
#endif
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); } // quad_perm [1,0,3,2]

// Two 11-sample lines at once: a lane's two lines as the 16-bit halves of one register (v_pk_*_u16).  Every intermediate
// of smooth6 fits 16 bits (samples <= 255, weights summing to <= 16, thresholds < 2^15).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk16(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t un16(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 splat16(int v) { return u16x2{(unsigned short) v, (unsigned short) v}; }
__device__ __forceinline__ u16x2 max16(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u16x2 min16(u16x2 a, u16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ uint32_t bsel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); } // v_bfi_b32

// smooth6 (bmc.c:53-70) on two lines; returns per half 0xffff where the six range tests pass
__device__ __forceinline__ uint32_t smooth6_pk(uint32_t e2_, uint32_t e1_, uint32_t e0_, uint32_t i0_, uint32_t i1_, uint32_t i2_, uint32_t t_, uint32_t o[4])
{
    u16x2 e2 = pk16(e2_), e1 = pk16(e1_), e0 = pk16(e0_), i0 = pk16(i0_), i1 = pk16(i1_), i2 = pk16(i2_);
    u16x2 avg = ((e0 + i0) * splat16(5) + (e1 + i1) * splat16(3) + splat16(8)) >> splat16(4);
    // largest deviation from the average = max(max6 - avg, avg - min6)
    u16x2 mx = max16(max16(max16(e0, i0), max16(e1, i1)), max16(e2, i2));
    u16x2 mn = min16(min16(min16(e0, i0), min16(e1, i1)), min16(e2, i2));
    u16x2 dev = max16(__builtin_elementwise_sub_sat(mx, avg), __builtin_elementwise_sub_sat(avg, mn));
    // dev < t per half -> 0xffff (kept from the compiler, which would turn it back into compares and selects per half)
    uint32_t m;
    asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\tv_pk_sub_u16 %0, 0, %0 op_sel_hi:[0,1]"
        : "=&v"(m)
        : "v"(t_), "v"(un16(dev)));
    u16x2 a5 = avg * splat16(5);
    o[0] = un16(((avg + e1) * splat16(3) + e2 * splat16(2) + splat16(4)) >> splat16(3));
    o[1] = un16((a5 + e1 * splat16(2) + e2 + splat16(4)) >> splat16(3));
    o[2] = un16(avg);
    o[3] = un16((a5 + i1 * splat16(2) + i2 + splat16(4)) >> splat16(3));
    return m;
}

// line_filter on two lines (tE, tM: the cell's thresholds, the same for both)
__device__ __forceinline__ bool line_filter_pk(uint32_t (&l)[11], bool in_edge, int tE, int tM)
{
    uint32_t o[4], q[4];
    uint32_t m1 = smooth6_pk(l[0], l[1], l[2], l[3], l[4], l[5], un16(splat16(tE)), o);
    uint32_t m2 = smooth6_pk(l[10], l[9], l[8], l[7], l[6], l[5], un16(splat16(in_edge ? tM : 0)), q);
    l[1] = bsel(m1, o[0], l[1]);
    l[2] = bsel(m1, o[1], l[2]);
    l[3] = bsel(m1, o[2], l[3]);
    l[4] = bsel(m1, o[3], l[4]);
    l[6] = bsel(m2, q[3], l[6]);
    l[7] = bsel(m2, q[2], l[7]);
    l[8] = bsel(m2, q[1], l[8]);
    l[9] = bsel(m2, q[0], l[9]);
    return (m1 | m2) != 0;
}

struct PairTile {
    uint32_t h[12]; // my two pixel rows y + 2p, y + 2p + 1 (low / high half), columns x - 4 .. x + 7
    uint32_t o[7];  // my two pixel columns x + 2p, x + 2p + 1 (low / high half) in the rows above (y - 3 .. y - 1) and below (y + 4 .. y + 7)
    bool rows_dirty, outer_dirty;

    __device__ __forceinline__ void load(const RingView &view, int x, int y, int p)
    {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            uint32_t r0 = view.ld(y + 2 * p, x - 4 + 4 * d), r1 = view.ld(y + 2 * p + 1, x - 4 + 4 * d);
            h[4 * d + 0] = __builtin_amdgcn_perm(r1, r0, 0x0c040c00u);
            h[4 * d + 1] = __builtin_amdgcn_perm(r1, r0, 0x0c050c01u);
            h[4 * d + 2] = __builtin_amdgcn_perm(r1, r0, 0x0c060c02u);
            h[4 * d + 3] = __builtin_amdgcn_perm(r1, r0, 0x0c070c03u);
        }
#pragma unroll
        for (int k = 0; k < 7; k++) {
            o[k] = __builtin_amdgcn_perm(0u, view.ld16(y + (k < 3 ? k - 3 : k + 1), x + 2 * p), 0x0c010c00u);
        }
        rows_dirty = outer_dirty = false;
    }

    __device__ __forceinline__ void store(const RingView &view, int x, int y, int p) const
    {
        if (__any(rows_dirty)) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                // (row0 px0, row0 px1, row1 px0, row1 px1) of two columns, then the rows' dwords from two such
                uint32_t t01 = __builtin_amdgcn_perm(h[4 * d + 1], h[4 * d], 0x06020400u), t23 = __builtin_amdgcn_perm(h[4 * d + 3], h[4 * d + 2], 0x06020400u);
                view.st(y + 2 * p, x - 4 + 4 * d, __builtin_amdgcn_perm(t23, t01, 0x05040100u));
                view.st(y + 2 * p + 1, x - 4 + 4 * d, __builtin_amdgcn_perm(t23, t01, 0x07060302u));
            }
        }
        if (__any(outer_dirty)) { // rows y - 2, y - 1, y + 4, y + 5, y + 6 (the vertical pass leaves y - 3 and y + 7 alone)
#pragma unroll
            for (int k = 1; k < 6; k++) {
                view.st16(y + (k < 3 ? k - 3 : k + 1), x + 2 * p, __builtin_amdgcn_perm(0u, o[k], 0x0c0c0200u));
            }
        }
    }
    // pixel (row a, column xx) of the cell's half I own
    __device__ __forceinline__ int px(int a, int xx) const { return (int) (a ? h[4 + xx] >> 16 : h[4 + xx] & 0xffffu); }
};

__device__ __forceinline__ void hfilter2(PairTile &T, const DPlane &dp, int x, bool edge, int tE, int tM, bool on)
{
    on = on & !(x < 4 || x > dp.w - 4 || (edge && tE <= 0) || tM <= 0);
    tE = edge ? tE : tM;
    tE = on ? tE : 0;
    tM = on ? tM : 0;
    bool in_edge = x < dp.w - 8;
    uint32_t l[11];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        l[k] = T.h[1 + k];
    }
    bool hit = line_filter_pk(l, in_edge, tE, tM);
#pragma unroll
    for (int k = 1; k < 10; k++) {
        T.h[1 + k] = l[k];
    }
    T.rows_dirty |= hit;
}

__device__ __forceinline__ void vfilter2(PairTile &T, const DPlane &dp, int y, int p, bool edge, int tE, int tM, bool on)
{
    on = on & !(y < 4 || y > dp.h - 4 || (edge && tE <= 0) || tM <= 0);
    tE = edge ? tE : tM;
    tE = on ? tE : 0;
    tM = on ? tM : 0;
    bool in_edge = y < dp.h - 8;
    // rows -> columns.  h[] pairs the two ROWS of one column; the vertical lines want the two COLUMNS of one row.
    // Of the cell's four rows I hold two; the partner's two, in MY columns, come over.
    uint32_t mc0 = p ? T.h[6] : T.h[4], mc1 = p ? T.h[7] : T.h[5]; // my columns (rows: mine)
    uint32_t oc0 = p ? T.h[4] : T.h[6], oc1 = p ? T.h[5] : T.h[7]; // the partner's columns (rows: mine)
    uint32_t mine0 = __builtin_amdgcn_perm(mc1, mc0, 0x05040100u), mine1 = __builtin_amdgcn_perm(mc1, mc0, 0x07060302u);
    uint32_t recv0 = (uint32_t) pair_swap((int) __builtin_amdgcn_perm(oc1, oc0, 0x05040100u));
    uint32_t recv1 = (uint32_t) pair_swap((int) __builtin_amdgcn_perm(oc1, oc0, 0x07060302u));
    uint32_t l[11];
    l[0] = T.o[0];
    l[1] = T.o[1];
    l[2] = T.o[2];
    l[3] = p ? recv0 : mine0;
    l[4] = p ? recv1 : mine1;
    l[5] = p ? mine0 : recv0;
    l[6] = p ? mine1 : recv1;
    l[7] = T.o[3];
    l[8] = T.o[4];
    l[9] = T.o[5];
    l[10] = T.o[6];
    bool hit = line_filter_pk(l, in_edge, tE, tM);
    T.o[1] = l[1];
    T.o[2] = l[2];
    T.o[3] = l[7];
    T.o[4] = l[8];
    T.o[5] = l[9];
    // columns -> rows
    mine0 = p ? l[5] : l[3];
    mine1 = p ? l[6] : l[4];
    uint32_t got0 = (uint32_t) pair_swap((int) (p ? l[3] : l[5])), got1 = (uint32_t) pair_swap((int) (p ? l[4] : l[6])); // my rows, the partner's columns
    mc0 = __builtin_amdgcn_perm(mine1, mine0, 0x05040100u);
    mc1 = __builtin_amdgcn_perm(mine1, mine0, 0x07060302u);
    oc0 = __builtin_amdgcn_perm(got1, got0, 0x05040100u);
    oc1 = __builtin_amdgcn_perm(got1, got0, 0x07060302u);
    T.h[4] = p ? oc0 : mc0;
    T.h[5] = p ? oc1 : mc1;
    T.h[6] = p ? mc0 : oc0;
    T.h[7] = p ? mc1 : oc1;
    hit |= (bool) pair_swap((int) hit);
    T.rows_dirty |= hit;
    T.outer_dirty |= hit;
}

#define PCELL(a, xx) T.px(a, xx)
__device__ __forceinline__ void artf2(const PairTile &T, int p, int &sh, int &sv, int &slh, int &slv) // bmc.c:224-270
{
    sh = sv = 0;
#pragma unroll
    for (int x = 0; x < 4; x += 2) {
        int x0 = PCELL(0, x), x1 = PCELL(0, x + 1), x2 = PCELL(1, x), x3 = PCELL(1, x + 1);
        int hh = absdiff(x0 + x3, x1 + x2) >> 1;
        sh += absdiff(x0 + x2, x1 + x3) + hh;
        sv += absdiff(x0 + x1, x2 + x3) + hh;
    }
    sh += pair_swap(sh);
    sv += pair_swap(sv);
    int dA = (PCELL(0, 0) + PCELL(0, 1) + PCELL(1, 0) + PCELL(1, 1) + 2) >> 2;
    int dB = (PCELL(0, 2) + PCELL(0, 3) + PCELL(1, 2) + PCELL(1, 3) + 2) >> 2;
    int oA = pair_swap(dA), oB = pair_swap(dB);
    int d0 = p ? oA : dA, d1 = p ? oB : dB, d2 = p ? dA : oA, d3 = p ? dB : oB;
    int hh = absdiff(d0 + d3, d1 + d2) >> 1;
    slh = absdiff(d0 + d2, d1 + d3) + hh;
    slv = absdiff(d0 + d1, d2 + d3) + hh;
}

__device__ __forceinline__ unsigned dsff2(const PairTile &T, int p)
{
    int dA = (PCELL(0, 0) + PCELL(0, 1) + PCELL(1, 0) + PCELL(1, 1) + 2) >> 2;
    int dB = (PCELL(0, 2) + PCELL(0, 3) + PCELL(1, 2) + PCELL(1, 3) + 2) >> 2;
    int oA = pair_swap(dA), oB = pair_swap(dB);
    int d[4] = {p ? oA : dA, p ? oB : dB, p ? dA : oA, p ? dB : oB};
    return dsff_d(d);
}

// de-gradient sharpening (bmc.c:276) of the cell, eight of its pixels per lane; both lanes of a pair take the same path
__device__ __forceinline__ void degrad2(PairTile &T)
{
    int px[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        px[k] = PCELL(k >> 2, k & 3);
    }
    int lo = 16, hi = -1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo = min(lo, px[k] >> 4);
        hi = max(hi, px[k] >> 4);
    }
    lo = min(lo, pair_swap(lo));
    hi = max(hi, pair_swap(hi));
    if (lo >= hi) {
        return;
    }
    int nlo = 0, nhi = 0, slo = 0, shi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int b = px[k] >> 4;
        if (b == lo) {
            nlo++;
            slo += px[k];
        }
        if (b == hi) {
            nhi++;
            shi += px[k];
        }
    }
    nlo += pair_swap(nlo);
    nhi += pair_swap(nhi);
    slo += pair_swap(slo);
    shi += pair_swap(shi);
    int alo = div_small(slo, nlo), ahi = div_small(shi, nhi);
    if (alo == 0) {
        alo = 1;
    }
    if (ahi == 0) {
        ahi = 1;
    }
    int t = (alo + ahi + 1) >> 1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        px[k] = degrad_px(px[k], t, nlo, alo, nhi, ahi);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T.h[4 + k] = (uint32_t) px[k] | ((uint32_t) px[4 + k] << 16);
    }
    T.rows_dirty = true;
}

// the cell as the routines below see it: a lane pair sharing it (PairCell, lane p) ...
struct PairCell {
    PairTile T;
    int p;
    __device__ __forceinline__ void load(const RingView &view, int x, int y) { T.load(view, x, y, p); }
    __device__ __forceinline__ void store(const RingView &view, int x, int y) const { T.store(view, x, y, p); }
    __device__ __forceinline__ void hfilter(const DPlane &dp, int x, bool edge, int tE, int tM, bool on) { hfilter2(T, dp, x, edge, tE, tM, on); }
    __device__ __forceinline__ void vfilter(const DPlane &dp, int y, bool edge, int tE, int tM, bool on) { vfilter2(T, dp, y, p, edge, tE, tM, on); }
    __device__ __forceinline__ void artf(int &sh, int &sv, int &slh, int &slv) const { artf2(T, p, sh, sv, slh, slv); }
    __device__ __forceinline__ unsigned dsff() const { return dsff2(T, p); }
    __device__ __forceinline__ void degrad() { degrad2(T); }
};

// ... or one lane holding both halves (launches that fill the chip several times over are bound by instructions per cell, not
// by the front's latency: the packed line filters without the second wavefront).  What the pair hands over by lane swaps
// moves between the halves' registers here.
struct DualCell {
    PairTile H[2];
    __device__ __forceinline__ void load(const RingView &view, int x, int y)
    {
        H[0].load(view, x, y, 0);
        H[1].load(view, x, y, 1);
    }
    __device__ __forceinline__ void store(const RingView &view, int x, int y) const
    {
        H[0].store(view, x, y, 0);
        H[1].store(view, x, y, 1);
    }
    __device__ __forceinline__ void hfilter(const DPlane &dp, int x, bool edge, int tE, int tM, bool on)
    {
        hfilter2(H[0], dp, x, edge, tE, tM, on);
        hfilter2(H[1], dp, x, edge, tE, tM, on);
    }
    __device__ __forceinline__ void vfilter(const DPlane &dp, int y, bool edge, int tE, int tM, bool on)
    {
        on = on & !(y < 4 || y > dp.h - 4 || (edge && tE <= 0) || tM <= 0);
        tE = edge ? tE : tM;
        tE = on ? tE : 0;
        tM = on ? tM : 0;
        bool in_edge = y < dp.h - 8;
        uint32_t l[2][11];
#pragma unroll
        for (int p = 0; p < 2; p++) { // half p's columns 2p, 2p + 1 through all four rows: rows 0, 1 sit in H[0], rows 2, 3 in H[1]
            l[p][0] = H[p].o[0];
            l[p][1] = H[p].o[1];
            l[p][2] = H[p].o[2];
            l[p][3] = __builtin_amdgcn_perm(H[0].h[5 + 2 * p], H[0].h[4 + 2 * p], 0x05040100u);
            l[p][4] = __builtin_amdgcn_perm(H[0].h[5 + 2 * p], H[0].h[4 + 2 * p], 0x07060302u);
            l[p][5] = __builtin_amdgcn_perm(H[1].h[5 + 2 * p], H[1].h[4 + 2 * p], 0x05040100u);
            l[p][6] = __builtin_amdgcn_perm(H[1].h[5 + 2 * p], H[1].h[4 + 2 * p], 0x07060302u);
            l[p][7] = H[p].o[3];
            l[p][8] = H[p].o[4];
            l[p][9] = H[p].o[5];
            l[p][10] = H[p].o[6];
        }
        bool hit = false;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            hit |= line_filter_pk(l[p], in_edge, tE, tM);
            H[p].o[1] = l[p][1];
            H[p].o[2] = l[p][2];
            H[p].o[3] = l[p][7];
            H[p].o[4] = l[p][8];
            H[p].o[5] = l[p][9];
            H[0].h[4 + 2 * p] = __builtin_amdgcn_perm(l[p][4], l[p][3], 0x05040100u);
            H[0].h[5 + 2 * p] = __builtin_amdgcn_perm(l[p][4], l[p][3], 0x07060302u);
            H[1].h[4 + 2 * p] = __builtin_amdgcn_perm(l[p][6], l[p][5], 0x05040100u);
            H[1].h[5 + 2 * p] = __builtin_amdgcn_perm(l[p][6], l[p][5], 0x07060302u);
        }
        H[0].rows_dirty |= hit;
        H[1].rows_dirty |= hit;
        H[0].outer_dirty |= hit;
        H[1].outer_dirty |= hit;
    }
    // the cell's pixel (yy, xx)
    __device__ __forceinline__ int cpx(int yy, int xx) const { return H[yy >> 1].px(yy & 1, xx); }
    __device__ __forceinline__ void artf(int &sh, int &sv, int &slh, int &slv) const // bmc.c:224-270
    {
        sh = sv = 0;
#pragma unroll
        for (int y = 0; y < 4; y += 2) {
#pragma unroll
            for (int x = 0; x < 4; x += 2) {
                int x0 = cpx(y, x), x1 = cpx(y, x + 1), x2 = cpx(y + 1, x), x3 = cpx(y + 1, x + 1);
                int hh = absdiff(x0 + x3, x1 + x2) >> 1;
                sh += absdiff(x0 + x2, x1 + x3) + hh;
                sv += absdiff(x0 + x1, x2 + x3) + hh;
            }
        }
        int d[4];
        DS2X2(cpx, d);
        int hh = absdiff(d[0] + d[3], d[1] + d[2]) >> 1;
        slh = absdiff(d[0] + d[2], d[1] + d[3]) + hh;
        slv = absdiff(d[0] + d[1], d[2] + d[3]) + hh;
    }
    __device__ __forceinline__ unsigned dsff() const
    {
        int d[4];
        DS2X2(cpx, d);
        return dsff_d(d);
    }
    __device__ __forceinline__ void degrad()
    {
        int px[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            px[k] = cpx(k >> 2, k & 3);
        }
        if (degrad16(px)) {
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    H[hh].h[4 + k] = (uint32_t) px[8 * hh + k] | ((uint32_t) px[8 * hh + 4 + k] << 16);
                }
            }
            H[0].rows_dirty = H[1].rows_dirty = true;
        }
    }
};

template <class Cell>
__device__ void intra_cell_pk(Cell &T, const RingView &view, const DPlane &dp, const FilterParams &f, const uint8_t *bd, int i, int j, int nsbx, int nsby)
{
    int x = i * 4, y = j * 4;
    bool live = !(y + 4 >= dp.h || x + 4 >= dp.w);
    int flags = live ? bd[scale_div(i, f.nbh, nsbx) + scale_div(j, f.nbv, nsby) * f.nbh] : DSV_IS_RINGING;
    live = live & !(flags & DSV_IS_RINGING);
    if (!__any(live)) { // nothing to do for the whole wavefront
        return;
    }
    T.load(view, x, y);
    int sh, sv, shl, svl;
    T.artf(sh, sv, shl, svl);
    int mx = max(sh, sv);
    live = live & (mx < 256 && mx > 8);
    int tt = 32;
    {
        int td = (int) T.dsff();
        td = (flags & DSV_IS_STABLE) ? (td * 5 >> 2) : td;
        tt = (flags & (DSV_IS_MAINTAIN | DSV_IS_STABLE)) ? td : (tt >> 2);
    }
    tt = tt * 2 / 3;
    tt = (tt * f.q) >> 12;
    tt = clampi(tt, 0, f.fthresh);
    T.hfilter(dp, x, false, tt, tt, live);
    T.vfilter(dp, y, false, tt, tt, live);
    tt = sh > sv ? (3 * sh + sv) : (3 * sv + sh);
    tt = curve_tex(tt);
    tt = 16 + ((tt + 2) >> 2);
    tt = (tt * f.q) >> 12;
    tt = clampi(tt, 0, f.fthresh);
    T.hfilter(dp, x, false, tt, tt, live);
    T.vfilter(dp, y, false, tt, tt, live);
    T.store(view, x, y);
}

template <class Cell>
__device__ void luma_cell_rec_pk(Cell &T, const RingView &view, const DPlane &dp, const FilterParams &f, const CellRec &rec, int fx, int fy, int i, int j)
{
    int x = i * 4, y = j * 4;
    uint32_t flags = rec.flags;
    bool live = !(y + 4 >= dp.h || (flags & (1u << DSV_MV_BIT_SKIP)) || x + 4 >= dp.w);
    bool edgeh = (x & (f.blk_w - 1)) == 0, edgehs = (x & (f.blk_w / 2 - 1)) == 0;
    bool edgev = (y & (f.blk_h - 1)) == 0, edgevs = (y & (f.blk_h / 2 - 1)) == 0;
    int mvx = mvx_of(rec.all), mvy = mvy_of(rec.all);
    int amx = abs(mvx), amy = abs(mvy);
    bool intra = (flags & (1u << DSV_MV_BIT_INTRA)) != 0;
    int ndx = 0, ndy = 0;
    if (f.do_filter) {
        neighbordif2(rec, fx, fy, ndx, ndy);
    }
    bool filt = !intra && (ndx || ndy);
    bool sharp = !intra && f.sharpen && (mvx & 3) && (mvy & 3) && ((mvx | mvy) & 1) && amx < 8 && amy < 8;
    live = live & (intra | filt | sharp);
    FILT_COUNT(0, true);
    FILT_COUNT(1, live);
    FILT_COUNT(5, live);
    if (!__any(live)) { // nothing to do for the whole wavefront
        return;
    }
    T.load(view, x, y);
    // the two passes and their thresholds, by block type (bmc.c:527-596): as luma_cell_rec
    bool h_on, v_on, eh, ev;
    int hE, hM, vE, vM;
    {
        int tH = clampi((64 * f.q) >> 12, 2, 32), tL = clampi((32 * f.q) >> 12, 2, 32);
        bool part = rec.submask != DSV_MASK_ALL_INTRA;
        bool ieh = edgeh | (part & edgehs), iev = edgev | (part & edgevs);
        bool eprm = (flags & (1u << DSV_MV_BIT_EPRM)) != 0;
        int tndc = (ndx + ndy + 1) >> 1;
        int sh, sv, shl, svl, tt;
        T.artf(sh, sv, shl, svl);
        int n_dx = ndx, n_dy = ndy;
        bool mixed = sh < 2 * sv && sv < 2 * sh;
        {
            int mdx = ndx < amx ? ndx >> 1 : ndx, mdy = ndy < amy ? ndy >> 1 : ndy;
            int shl2 = shl > 128 ? 0 : 128 - shl, svl2 = svl > 128 ? 0 : 128 - svl;
            int ix = min(amx, 32), iy = min(amy, 32);
            int tm = ((sh * (32 - iy) + shl2 * iy) + 16) >> 5;
            tm += ((sv * (32 - ix) + svl2 * ix) + 16) >> 5;
            tm = (tm + 1) >> 1;
            tm = (mdx < amy && mdy < amx) ? 0 : tm;
            tt = mixed ? tm : ((sh + sv + 1) >> 1);
            n_dx = mixed ? mdx : ndx;
            n_dy = mixed ? mdy : ndy;
        }
        tt = (tt * tndc + 4) >> 3;
        tt = (min(tt, f.fthresh) * f.q) >> 12;
        int addx = (min(n_dy, f.fthresh) * f.q) >> 12;
        int addy = (min(n_dx, f.fthresh) * f.q) >> 12;
        bool v_only = sh > 2 * sv || amy > 2 * amx;
        bool h_only = !v_only && (sv > 2 * sh || amx > 2 * amy);
        h_on = live & (intra | (filt & !v_only));
        v_on = live & (intra | (filt & !h_only));
        eh = intra ? ieh : (edgeh | eprm);
        ev = intra ? iev : (edgev | eprm);
        hE = intra ? tH : tt + addx;
        hM = intra ? tL : tt;
        vE = intra ? tH : tt + addy;
        vM = intra ? tL : tt;
    }
    FILT_COUNT(2, h_on);
    FILT_COUNT(3, v_on);
    FILT_COUNT(4, live & sharp);
    if (__any(h_on)) {
        T.hfilter(dp, x, eh, hE, hM, h_on);
    }
    if (__any(v_on)) {
        T.vfilter(dp, y, ev, vE, vM, v_on);
    }
    if (__any(live & sharp)) {
        if (live & sharp) {
            T.degrad();
        }
    }
    T.store(view, x, y);
}

// wavefront sweep helper: front t holds the cells (i, j) with i + 2j == t
template <class Body> __device__ __forceinline__ void sweep_fronts(int nx, int ny, Body body)
{
    int last = (nx - 1) + 2 * (ny - 1);
    for (int t = 0; t <= last; t++) {
        int jmax = min(ny - 1, t >> 1);
        int jmin = max(0, (t - (nx - 1) + 1) >> 1);
        for (int j = jmin + (int) threadIdx.x; j <= jmax; j += (int) blockDim.x) {
            body(t - 2 * j, j);
        }
        __syncthreads(); // workgroup-scope release/acquire: the next front sees this front's pixels
    }
}

// ---- plane-resident sweep: the same fronts, but the pixels a front can touch live in LDS -----------------
// Every pixel row owns a 64-byte ring holding columns [4*ic - 12, 4*ic + 52) of that row, where ic = t - 2j
// is the cell column its cell row j is at on front t.  Cell (ic, j) touches its own rows at columns
// 4ic-4 .. 4ic+7 and the rows of cell rows j-1 / j+1 at columns 4ic .. 4ic+3, i.e. (seen from those
// rows, which are two cells ahead / behind) their columns 4ic'-8 .. 4ic'-5 and 4ic'+8 .. 4ic'+11: all inside the
// window.  Per front a row fetches the 4 columns entering its window (registers now, LDS next front,
// first needed nine fronts later), and retires the 4 columns leaving it to global memory; nothing on
// the dependent path of a front goes to global memory.  Thread = cell row mod blockDim (the band of
// rows that are active on a front is at most (nsbx + 14) / 2 + 1 rows wide).
__device__ __forceinline__ bool ring_eligible(const DPlane &dp, int nthreads, size_t lds_bytes)
{
    return (dp.w & 3) == 0 && (dp.h & 3) == 0 && dp.w >= 64 && (size_t) (dp.h + RingView::kGuardRows) * 64 <= lds_bytes && (dp.w / 4 + 14) / 2 + 1 <= nthreads;
}

// cell(ic, j) filters one cell; ahead(ic, j) is told which cell this thread will filter four fronts later so
// that it can fetch that cell's side information off the dependent path; land() runs at the top of every front, behind
// the wait for the previous front's vector-memory operations: what ahead() fetched is copied out of its load registers there
// phase clock of a debugging build (make prof): shader-clock ticks wave 0 of a luma sweep spends in each part of a front
#ifdef DSV2_FILTER_PROF
__device__ unsigned long long g_filt_prof[8];
#define FILT_MARK(k)                                                     \
    do {                                                                 \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();           \
        prof_acc[k] += t_ - prof_t;                                      \
        prof_t = t_;                                                     \
    } while (0)
#else
#define FILT_MARK(k)                                                     \
    do {                                                                 \
    } while (0)
#endif

template <class CellFn, class AheadFn, class LandFn>
__device__ __forceinline__ void ring_sweep(const DPlane &dp, uint8_t *ring, CellFn cell, AheadFn ahead, LandFn land)
{
#ifdef DSV2_FILTER_PROF
    unsigned long long prof_acc[4] = {0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
    const int nsbx = dp.w / 4, nsby = dp.h / 4;
    const int tid = (int) threadIdx.x, nthr = (int) blockDim.x;
    const int t_last = (nsbx + 2) + 2 * (nsby - 1);
    uint32_t pend[4] = {0, 0, 0, 0};
    int pend_row = -1, pend_col = 0;
    for (int t = -12; t <= t_last; t++) {
        // the rows in flight: -12 <= ic <= nsbx + 2
        int jlo = t - (nsbx + 2) > 0 ? (t - (nsbx + 2) + 1) >> 1 : 0;
        int jhi = (t + 12) >> 1;
        jhi = jhi < nsby - 1 ? jhi : nsby - 1;
        int j = jlo + ((tid - jlo) & (nthr - 1)); // (tid - jlo) mod nthr: the workgroup size is a power of two (256)
        bool active = j <= jhi;
        int ic = t - 2 * j;
        { // Columns fetched on the previous front enter the window.  A thread with nothing pending parks stale registers below
          // the plane: the wait for the fetch then stands at the top of EVERY front for every thread, and whatever was fetched
          // ahead on earlier fronts (land()) is used without a wait of its own.
            const int prow = pend_row >= 0 ? pend_row : dp.h + RingView::kGuardBelow, pcol = pend_row >= 0 ? (pend_col & 63) : 4 * (tid & 15);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                *(uint32_t *) (ring + (prow + r) * 64 + pcol) = pend[r];
            }
            pend_row = -1;
            land();
        }
        FILT_MARK(0);
        if (active) {
            // Every global-memory operation of a front is ISSUED before its cell is filtered and first waited for at the top
            // of the next front (vector-memory operations complete in order, stores included: retirements issued behind the
            // cell would be waited out at once by the next front's column hand-over).
            int g = 4 * ic - 12;
            if (g >= 0 && g < dp.w) { // these columns are final: no cell of this or a later front reaches them
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    *(gu32w_t) (dp.data + (ptrdiff_t) (4 * j + r) * dp.stride + g) = *(const uint32_t *) (ring + (4 * j + r) * 64 + (g & 63));
                }
            }
            if (ic + 4 >= 0 && ic + 4 < nsbx) { // one block (four cells) of lead: a global fetch has that long to arrive
                ahead(ic + 4, j);
            }
            g = 4 * ic + 48;
            if (g >= 0 && g < dp.w) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pend[r] = *(gu32_t) (dp.data + (ptrdiff_t) (4 * j + r) * dp.stride + g);
                }
                pend_row = 4 * j;
                pend_col = g;
            }
            FILT_MARK(1);
            if (ic >= 0 && ic < nsbx) {
                cell(ic, j);
            }
        }
        FILT_MARK(2);
        // fronts hand over through LDS only: wait for this wave's LDS traffic and meet the others, but leave
        // the column fetches and retirements in flight (__syncthreads() would drain them every front)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        FILT_MARK(3);
    }
#ifdef DSV2_FILTER_PROF
    if (tid == 0) {
        for (int k = 0; k < 4; k++) {
            atomicAdd(&g_filt_prof[k], prof_acc[k]);
        }
        atomicAdd(&g_filt_prof[4], 1ull);
    }
#endif
    __syncthreads();
}

// the same sweep with a lane pair per cell (cell(ic, j, p): lane p of the pair): thread pair = cell row mod (blockDim / 2),
// each lane retires and fetches the columns of its own two pixel rows
template <class CellFn, class AheadFn, class LandFn>
__device__ __forceinline__ void ring_sweep2(const DPlane &dp, uint8_t *ring, CellFn cell, AheadFn ahead, LandFn land)
{
#ifdef DSV2_FILTER_PROF
    unsigned long long prof_acc[4] = {0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
    const int nsbx = dp.w / 4, nsby = dp.h / 4;
    const int tid = (int) threadIdx.x >> 1, nthr = (int) blockDim.x >> 1, p = (int) threadIdx.x & 1;
    const int t_last = (nsbx + 2) + 2 * (nsby - 1);
    uint32_t pend[2] = {0, 0};
    int pend_row = -1, pend_col = 0;
    for (int t = -12; t <= t_last; t++) {
        int jlo = t - (nsbx + 2) > 0 ? (t - (nsbx + 2) + 1) >> 1 : 0;
        int jhi = (t + 12) >> 1;
        jhi = jhi < nsby - 1 ? jhi : nsby - 1;
        int j = jlo + ((tid - jlo) & (nthr - 1));
        bool active = j <= jhi;
        int ic = t - 2 * j;
        { // (as in ring_sweep: unconditional, so that the wait for the fetch stands at the top of every front)
            const int prow = pend_row >= 0 ? pend_row : dp.h + RingView::kGuardBelow, pcol = pend_row >= 0 ? (pend_col & 63) : 4 * (tid & 15);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                *(uint32_t *) (ring + (prow + r) * 64 + pcol) = pend[r];
            }
            pend_row = -1;
            land();
        }
        FILT_MARK(0);
        if (active) {
            const int row0 = 4 * j + 2 * p;
            int g = 4 * ic - 12;
            if (g >= 0 && g < dp.w) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    *(gu32w_t) (dp.data + (ptrdiff_t) (row0 + r) * dp.stride + g) = *(const uint32_t *) (ring + (row0 + r) * 64 + (g & 63));
                }
            }
            if (ic + 4 >= 0 && ic + 4 < nsbx) {
                ahead(ic + 4, j);
            }
            g = 4 * ic + 48;
            if (g >= 0 && g < dp.w) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    pend[r] = *(gu32_t) (dp.data + (ptrdiff_t) (row0 + r) * dp.stride + g);
                }
                pend_row = row0;
                pend_col = g;
            }
            FILT_MARK(1);
            if (ic >= 0 && ic < nsbx) {
                cell(ic, j, p);
            }
        }
        FILT_MARK(2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        FILT_MARK(3);
    }
#ifdef DSV2_FILTER_PROF
    if (threadIdx.x == 0) {
        for (int k = 0; k < 4; k++) {
            atomicAdd(&g_filt_prof[k], prof_acc[k]);
        }
        atomicAdd(&g_filt_prof[4], 1ull);
    }
#endif
    __syncthreads();
}

// grid = 3 workgroups: luma filter, U chroma filter, V chroma filter
__global__ __launch_bounds__(256) void k_inter_filters(const DSV_MV *__restrict__ vecs, FilterParams f, Planes3 pl)
{
    int c = blockIdx.x;
    const DPlane dp = pl.p[c];
    if (f.lossless) {
        return;
    }
    if (c == 0) {
        int nsbx = dp.w / 4, nsby = dp.h / 4;
        sweep_fronts(nsbx, nsby, [&](int i, int j) { luma_cell(GlobalView{dp}, dp, f, vecs, i, j, nsbx, nsby); });
    } else {
        sweep_fronts(f.nbh, f.nbv, [&](int i, int j) { chroma_block(dp, f, vecs, i, j); });
    }
}

// stream-batched filters: grid = (n jobs, 3 planes) / (n jobs).  A luma sweep holds ~70 KB of LDS -- two fit a CU, 512 the
// chip -- and the launch reserves that for the chroma workgroups too: with the plane as the SLOW grid index all luma sweeps of
// a 192-picture launch (576 workgroups) are dispatched first and start at once; the short chroma sweeps fill in behind them.
// (Plane-major order had a fifth of the luma sweeps start when the first chroma sweeps had finished.)
__device__ __forceinline__ void inter_filters_b_body(const McJob *__restrict__ tab, unsigned lds_bytes)
{
    DSV2_KERNEL_PRIO();
    extern __shared__ uint8_t dyn_lds[];
    const McJob &jb = tab[blockIdx.x];
    int c = blockIdx.y; // (the luma sweeps -- the long ones, and the ones that need the LDS -- are dispatched first)
    const DPlane dp = jb.res.p[c];
    const FilterParams f = jb.f;
    const DSV_MV *vecs = jb.mvs;
    if (f.lossless) {
        return;
    }
    if (c == 0) {
        int nsbx = dp.w / 4, nsby = dp.h / 4;
        { // (the host launches this kernel only where the ring fits: ring_fits())
            uint8_t *ring0 = dyn_lds + RingView::kGuardTop * 64;
            RingView view{ring0, dp.h};
            // the block record of the cell at hand and, fetched one front early, of the next one
            CellRec cur = {}, nxt = {}, raw = {}; // raw: the load registers of ahead()
            int cur_key = -1, nxt_key = -1, raw_key = -1;
            // cell -> block (bmc.c:505-506 scales the cell index): a thread stays on its cell row, and with 16-pixel blocks
            // on a width that is a multiple of 16 four cells make a block
            const bool quarter_x = f.nbh * 4 == nsbx;
            int fy_j = -1, fy_v = 0;
            auto row_block = [&](int j) {
                if (j != fy_j) {
                    fy_v = scale_div(j, f.nbv, nsby);
                    fy_j = j;
                }
                return fy_v;
            };
            ring_sweep(
                dp, ring0,
                [&](int i, int j) {
                    int fx = quarter_x ? i >> 2 : scale_div(i, f.nbh, nsbx), fy = row_block(j), key = fx + fy * f.nbh;
                    if (key != cur_key) {
                        cur = key == nxt_key ? nxt : landed(fetch_cell_rec(vecs, f.nbh, fx, fy));
                        cur_key = key;
                    }
                    DualCell C;
                    luma_cell_rec_pk(C, view, dp, f, cur, fx, fy, i, j);
                },
                [&](int i, int j) {
                    int fx = quarter_x ? i >> 2 : scale_div(i, f.nbh, nsbx), fy = row_block(j), key = fx + fy * f.nbh;
                    if (key != cur_key && key != nxt_key) {
                        raw = fetch_cell_rec(vecs, f.nbh, fx, fy);
                        raw_key = key;
                    }
                },
                [&]() {
                    nxt = landed(raw);
                    nxt_key = raw_key;
                });
        }
    } else {
        sweep_fronts(f.nbh, f.nbv, [&](int i, int j) { chroma_block(dp, f, vecs, i, j); });
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_inter_filters_b(const McJob *__restrict__ tab, unsigned lds_bytes)
{
    inter_filters_b_body(tab, lds_bytes);
}
// 512 threads, still a lane per cell: pictures so wide that more than 256 cell rows are in flight on a front (3840 pixels: 488)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_inter_filters_bw(const McJob *__restrict__ tab, unsigned lds_bytes)
{
    inter_filters_b_body(tab, lds_bytes);
}

// the same with a lane pair per luma cell: 512 threads (the chroma workgroups use all of them as block rows)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_inter_filters_b2(const McJob *__restrict__ tab, unsigned lds_bytes)
{
    DSV2_KERNEL_PRIO();
    extern __shared__ uint8_t dyn_lds[];
    const McJob &jb = tab[blockIdx.x];
    int c = blockIdx.y; // (the luma sweeps -- the long ones, and the ones that need the LDS -- are dispatched first)
    const DPlane dp = jb.res.p[c];
    const FilterParams f = jb.f;
    const DSV_MV *vecs = jb.mvs;
    if (f.lossless) {
        return;
    }
    if (c == 0) {
        int nsbx = dp.w / 4, nsby = dp.h / 4;
        { // (the host launches this kernel only where the ring fits: ring_fits())
            uint8_t *ring0 = dyn_lds + RingView::kGuardTop * 64;
            RingView view{ring0, dp.h};
            CellRec cur = {}, nxt = {}, raw = {}; // raw: the load registers of ahead()
            int cur_key = -1, nxt_key = -1, raw_key = -1;
            // cell -> block (bmc.c:505-506 scales the cell index): a thread stays on its cell row, and with 16-pixel blocks
            // on a width that is a multiple of 16 four cells make a block
            const bool quarter_x = f.nbh * 4 == nsbx;
            int fy_j = -1, fy_v = 0;
            auto row_block = [&](int j) {
                if (j != fy_j) {
                    fy_v = scale_div(j, f.nbv, nsby);
                    fy_j = j;
                }
                return fy_v;
            };
            ring_sweep2(
                dp, ring0,
                [&](int i, int j, int p) {
                    int fx = quarter_x ? i >> 2 : scale_div(i, f.nbh, nsbx), fy = row_block(j), key = fx + fy * f.nbh;
                    if (key != cur_key) {
                        cur = key == nxt_key ? nxt : landed(fetch_cell_rec(vecs, f.nbh, fx, fy));
                        cur_key = key;
                    }
                    PairCell C;
                    C.p = p;
                    luma_cell_rec_pk(C, view, dp, f, cur, fx, fy, i, j);
                },
                [&](int i, int j) {
                    int fx = quarter_x ? i >> 2 : scale_div(i, f.nbh, nsbx), fy = row_block(j), key = fx + fy * f.nbh;
                    if (key != cur_key && key != nxt_key) {
                        raw = fetch_cell_rec(vecs, f.nbh, fx, fy);
                        raw_key = key;
                    }
                },
                [&]() {
                    nxt = landed(raw);
                    nxt_key = raw_key;
                });
        }
    } else {
        sweep_fronts(f.nbh, f.nbv, [&](int i, int j) { chroma_block(dp, f, vecs, i, j); });
    }
}

__global__ __launch_bounds__(512) void k_intra_filter_b2(const McJob *__restrict__ tab, unsigned lds_bytes)
{
    DSV2_KERNEL_PRIO();
    extern __shared__ uint8_t dyn_lds[];
    const McJob &jb = tab[blockIdx.x];
    const DPlane dp = jb.res.p[0];
    const FilterParams f = jb.f;
    const uint8_t *bd = jb.bd;
    int nsbx = dp.w / 4, nsby = dp.h / 4;
    { // (launched only where the ring fits: ring_fits())
        uint8_t *ring0 = dyn_lds + RingView::kGuardTop * 64;
            RingView view{ring0, dp.h};
        ring_sweep2(dp, ring0, [&](int i, int j, int p) {
            PairCell C;
            C.p = p;
            intra_cell_pk(C, view, dp, f, bd, i, j, nsbx, nsby);
        }, [](int, int) {}, []() {});
    }
}

__device__ __forceinline__ void intra_filter_b_body(const McJob *__restrict__ tab, unsigned lds_bytes)
{
    DSV2_KERNEL_PRIO();
    extern __shared__ uint8_t dyn_lds[];
    const McJob &jb = tab[blockIdx.x];
    const DPlane dp = jb.res.p[0];
    const FilterParams f = jb.f;
    const uint8_t *bd = jb.bd;
    int nsbx = dp.w / 4, nsby = dp.h / 4;
    { // (launched only where the ring fits: ring_fits())
        uint8_t *ring0 = dyn_lds + RingView::kGuardTop * 64;
            RingView view{ring0, dp.h};
        ring_sweep(
            dp, ring0,
            [&](int i, int j) {
                DualCell C;
                intra_cell_pk(C, view, dp, f, bd, i, j, nsbx, nsby);
            },
            [](int, int) {}, []() {});
    }
}

__global__ __launch_bounds__(256) void k_intra_filter_b(const McJob *__restrict__ tab, unsigned lds_bytes) { intra_filter_b_body(tab, lds_bytes); }
__global__ __launch_bounds__(512) void k_intra_filter_bw(const McJob *__restrict__ tab, unsigned lds_bytes) { intra_filter_b_body(tab, lds_bytes); }

// the batched filters where the plane-resident ring does not fit (width not a multiple of 4, more cell rows in flight than
// threads, taller than the LDS): the same fronts through global memory; grid = (n jobs, 3 planes) / (n jobs)
__global__ __launch_bounds__(256) void k_inter_filters_g(const McJob *__restrict__ tab)
{
    const McJob &jb = tab[blockIdx.x];
    int c = blockIdx.y;
    const DPlane dp = jb.res.p[c];
    const FilterParams f = jb.f;
    const DSV_MV *vecs = jb.mvs;
    if (f.lossless) {
        return;
    }
    if (c == 0) {
        int nsbx = dp.w / 4, nsby = dp.h / 4;
        sweep_fronts(nsbx, nsby, [&](int i, int j) { luma_cell(GlobalView{dp}, dp, f, vecs, i, j, nsbx, nsby); });
    } else {
        sweep_fronts(f.nbh, f.nbv, [&](int i, int j) { chroma_block(dp, f, vecs, i, j); });
    }
}

__global__ __launch_bounds__(256) void k_intra_filter_g(const McJob *__restrict__ tab)
{
    const McJob &jb = tab[blockIdx.x];
    const DPlane dp = jb.res.p[0];
    const FilterParams f = jb.f;
    const uint8_t *bd = jb.bd;
    int nsbx = dp.w / 4, nsby = dp.h / 4;
    sweep_fronts(nsbx, nsby, [&](int i, int j) { intra_cell(GlobalView{dp}, dp, f, bd, i, j, nsbx, nsby); });
}

__global__ __launch_bounds__(256) void k_intra_filter(const uint8_t *__restrict__ bd, FilterParams f, DPlane dp)
{
    int nsbx = dp.w / 4, nsby = dp.h / 4;
    sweep_fronts(nsbx, nsby, [&](int i, int j) { intra_cell(GlobalView{dp}, dp, f, bd, i, j, nsbx, nsby); });
}

// decoder-side sharpening (dsv_post_process, bmc.c:340): every 4x4 cell is independent
__global__ __launch_bounds__(256) void k_post_process(DPlane dp)
{
    int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    int x = i * 4, y = j * 4;
    if (i >= dp.w / 4 || j >= dp.h / 4 || x + 4 >= dp.w || y + 4 >= dp.h) {
        return;
    }
    degrad(dp.data + (ptrdiff_t) y * dp.stride + x, dp.stride);
}

void post_process_plane(hipStream_t s, const DPlane &dp)
{
    DSV2_LAUNCH(k_post_process, dim3((dp.w / 4 + 63) / 64, (dp.h / 4 + 3) / 4), dim3(64, 4), 0, s, dp);
    HIPCHK(hipGetLastError());
}

#ifdef DSV2_FILTER_PROF
} // namespace dsv2
#ifdef DSV2_FILTER_PROF
extern "C" void dsv2hip_debug_filter_counts(unsigned long long out[8])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(dsv2::g_filt_cnt), 8 * sizeof(unsigned long long)));
}
#endif
extern "C" void dsv2hip_debug_filter_prof(unsigned long long out[8])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(dsv2::g_filt_prof), 8 * sizeof(unsigned long long)));
}
namespace dsv2 {
#endif

// ---- host drivers --------------------------------------------------------------------------

static int host_lb2(unsigned n)
{
    unsigned i = 1;
    int l = 0;
    while (i < n) {
        i <<= 1;
        l++;
    }
    return l;
}

FilterParams make_filter_params(const MCParams &p, int q, int do_filter, int inter_sharpen)
{
    FilterParams f;
    f.blk_w = p.blk_w;
    f.blk_h = p.blk_h;
    f.nbh = p.nbh;
    f.nbv = p.nbv;
    f.hshift = p.hshift;
    f.vshift = p.vshift;
    f.lossless = p.lossless;
    f.do_filter = do_filter;
    f.sharpen = inter_sharpen ? p.temporal_mc : 0; // bmc.c:470-474
    f.q_raw = q;
    // compute_filter_q (bmc.c:376) and fthresh (:408,:481)
    int psyf = spatial_psy_factor_host(p.blk_w, p.blk_h, p.nbh, p.nbv, -1);
    int fq = q > 1536 ? 1536 : q;
    fq += fq * psyf >> 10;
    if (fq < 1024) {
        fq = 512 + fq / 2;
    }
    f.q = fq;
    f.fthresh = 32 * (14 - host_lb2((unsigned) fq));
    return f;
}

static Planes3 planes_of(const DFrame &f)
{
    Planes3 p;
    for (int c = 0; c < 3; c++) {
        p.p[c] = f.p[c];
    }
    return p;
}

// Blocks larger than 16 x 16 in 4:2:0 (32 x 32: 2160p; 32 x 16: 1920 x 800, 2560 x 1080) are predicted a 16 x 16 piece per wavefront through
// the 16 x 16 routine (predict_plane<.., TILED>): tw | th << 8 pieces a block, 0 = a wavefront per block (16 x 16 itself; other formats)
static int mc_tiles(int blk_w, int blk_h, bool c420)
{
    const bool big = (blk_w == 32 || blk_h == 32) && (blk_w == 16 || blk_w == 32) && (blk_h == 16 || blk_h == 32);
    return (c420 && big) ? ((blk_w / 16) | ((blk_h / 16) << 8)) : 0;
}
template <int MODE> static void launch_predict(hipStream_t s, const McJob *d_tab, int n, int nbh, int nbv, int blk_w, int blk_h, bool c420)
{
    const int tiles = mc_tiles(blk_w, blk_h, c420);
    const int tw = tiles ? (tiles & 0xff) : 1, th = tiles ? (tiles >> 8) : 1;
    // the luma window of a wavefront's piece: the launch's block height + 3, or a 16-high piece's 19 (an intra block of a tiled
    // launch takes the general form, which only stages a window for sub-pel vectors -- an intra block has none)
    const int rows = (tiles ? 16 : blk_h) + 3;
    if (tiles) {
        DSV2_LAUNCH((k_predict_w<MODE, PRED_TILED>), dim3((nbh * tw + 3) / 4, nbv * th, n), dim3(256), 4 * wave_lds_bytes(rows), s, d_tab, rows, tiles);
    } else if (blk_w == 16 && blk_h == 16 && c420) {
        DSV2_LAUNCH((k_predict_w<MODE, PRED_16>), dim3((nbh + 3) / 4, nbv, n), dim3(256), 4 * wave_lds_bytes(rows), s, d_tab, rows, 0);
    } else {
        DSV2_LAUNCH((k_predict_w<MODE, PRED_ANY>), dim3((nbh + 3) / 4, nbv, n), dim3(256), 4 * wave_lds_bytes(rows), s, d_tab, rows, 0);
    }
}
// the single-call seam (dsv_sub_pred, dsv_add_pred): the batch kernel over a table of ONE job, so that the stage tests exercise
// the kernel the encoder and the decoder run.  (The seam serialises its callers and drains the stream before it returns: one
// table is enough.)
template <int MODE> static void predict_one(hipStream_t s, const DSV_MV *d_mvs, const MCParams &p, const DFrame &ref, const DFrame &pred, const DFrame &resd)
{
    static McJob *d_job = nullptr;
    if (d_job == nullptr) {
        HIPCHK(hipMalloc(&d_job, sizeof(McJob)));
    }
    McJob jb{};
    jb.mvs = d_mvs;
    jb.p = p;
    jb.ref = planes_of(ref);
    jb.pred = planes_of(pred);
    jb.res = planes_of(resd);
    HIPCHK(hipMemcpyAsync(d_job, &jb, sizeof(McJob), hipMemcpyHostToDevice, s)); // (pageable source: staged before the call returns)
    launch_predict<MODE>(s, d_job, 1, p.nbh, p.nbv, p.blk_w, p.blk_h, p.hshift == 1 && p.vshift == 1);
}

void mc_sub_pred(hipStream_t s, const DSV_MV *d_mvs, const MCParams &p, const DFrame &pred, const DFrame &resd, const DFrame &ref)
{
    predict_one<MC_SUBTRACT>(s, d_mvs, p, ref, pred, resd);
    HIPCHK(hipGetLastError());
}
void mc_add_res(hipStream_t s, const DSV_MV *d_mvs, const MCParams &p, int q, const DFrame &resd, const DFrame &pred, int do_filter,
                int inter_sharpen)
{
    { // (the batch kernel over a table of one: see predict_one)
        static McJob *d_job = nullptr;
        if (d_job == nullptr) {
            HIPCHK(hipMalloc(&d_job, sizeof(McJob)));
        }
        McJob jb{};
        jb.mvs = d_mvs;
        jb.p = p;
        jb.pred = planes_of(pred);
        jb.res = planes_of(resd);
        HIPCHK(hipMemcpyAsync(d_job, &jb, sizeof(McJob), hipMemcpyHostToDevice, s));
        DSV2_LAUNCH(k_reconstruct_w, dim3((p.nbh * p.blk_w / 16 + 63) / 64, (p.nbv * p.blk_h + 4 * kReconRows - 1) / (4 * kReconRows), 3), dim3(256), 0, s, d_job);
    }
    if (!p.lossless) {
        DSV2_LAUNCH(k_inter_filters, dim3(3), dim3(256), 0, s, d_mvs, make_filter_params(p, q, do_filter, inter_sharpen),
                           planes_of(resd));
    }
    HIPCHK(hipGetLastError());
}

void mc_add_pred(hipStream_t s, const DSV_MV *d_mvs, const MCParams &p, int q, const DFrame &resd, const DFrame &out, const DFrame &ref,
                 int do_filter, int inter_sharpen)
{
    predict_one<MC_RECONSTRUCT>(s, d_mvs, p, ref, out, resd);
    if (!p.lossless) {
        DSV2_LAUNCH(k_inter_filters, dim3(3), dim3(256), 0, s, d_mvs, make_filter_params(p, q, do_filter, inter_sharpen),
                           planes_of(out));
    }
    HIPCHK(hipGetLastError());
}

// dynamic LDS of the plane-resident luma sweep: one 64-byte ring per pixel row + guard rows (0 = the ring does not fit this
// picture: the global-memory kernels run).  Host mirror of what the ring sweeps assume (ring_eligible).
// bytes of LDS of the plane-resident sweep, or 0 where it does not fit; *wide: more than 256 cell rows are in flight on a front
// (the 512-thread lane-per-cell kernels; no lane-pair form for those)
static unsigned ring_lds_bytes(int luma_w, int luma_h, bool *wide = nullptr)
{
    // The ring needs more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU): asked for once, only when the ring
    // is selected at all, and a device that refuses (64 KB parts) gets the global-memory kernels instead of an abort.
    static const int on = [] {
        if (getenv("DSV2_FILTER_RING") && atoi(getenv("DSV2_FILTER_RING")) == 0) {
            return 0;
        }
        const void *ks[6] = {(const void *) k_inter_filters_b, (const void *) k_intra_filter_b, (const void *) k_inter_filters_b2, (const void *) k_intra_filter_b2,
                             (const void *) k_inter_filters_bw, (const void *) k_intra_filter_bw};
        for (const void *k : ks) {
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
                (void) hipGetLastError();
                return 0;
            }
        }
        return 1;
    }();
    size_t b = (size_t) (luma_h + RingView::kGuardRows) * 64;
    const int rows_in_flight = (luma_w / 4 + 14) / 2 + 1;
    const bool fits = on && (luma_w & 3) == 0 && (luma_h & 3) == 0 && luma_w >= 64 && b <= 150 * 1024 && rows_in_flight <= 512;
    if (wide) {
        *wide = rows_in_flight > 256;
    }
    return fits ? (unsigned) b : 0u;
}

// Which luma sweep a launch of n pictures gets.  The lane-pair kernels shorten a sweep's critical path (one picture:
// 4.3 -> 3.2 ms) and cost more instructions per cell; launches that fill the chip several times over (192 pictures of the
// headline: 7 060 against 6 700 - 7 010 frames/s) are bound by those and keep a lane per cell.
// DSV2_FILTER_PAIR_MAX = largest n that takes the pair kernels (0: never).
static bool filter_pair(int n)
{
    static int nmax = getenv("DSV2_FILTER_PAIR_MAX") ? atoi(getenv("DSV2_FILTER_PAIR_MAX")) : 64;
    return n <= nmax;
}

// ---- lockstep batch drivers: `d_tab` holds n McJob records already resident on the device ----
void mc_sub_pred_batch(hipStream_t s, const McJob *d_tab, int n, int nbh, int nbv, int blk_w, int blk_h, bool c420)
{
    if (n > 0) {
        launch_predict<MC_SUBTRACT>(s, d_tab, n, nbh, nbv, blk_w, blk_h, c420);
    }
}

void mc_add_res_batch(hipStream_t s, const McJob *d_tab, int n, int nbh, int nbv, bool any_filter, int luma_w, int luma_h, int blk_w, int blk_h)
{
    if (n > 0) {
        DSV2_LAUNCH(k_reconstruct_w, dim3((nbh * blk_w / 16 + 63) / 64, (nbv * blk_h + 4 * kReconRows - 1) / (4 * kReconRows), 3 * n), dim3(256), 0, s, d_tab);
        if (any_filter) {
            bool wide = false;
            const unsigned lds = ring_lds_bytes(luma_w, luma_h, &wide);
            if (!lds) {
                DSV2_LAUNCH(k_inter_filters_g, dim3(n, 3), dim3(256), 0, s, d_tab);
            } else if (wide) {
                DSV2_LAUNCH(k_inter_filters_bw, dim3(n, 3), dim3(512), lds, s, d_tab, lds);
            } else if (filter_pair(n)) {
                DSV2_LAUNCH(k_inter_filters_b2, dim3(n, 3), dim3(512), lds, s, d_tab, lds);
            } else {
                DSV2_LAUNCH(k_inter_filters_b, dim3(n, 3), dim3(256), lds, s, d_tab, lds);
            }
        }
    }
}

// decoder: d_pred jobs {ref, pred = output picture, res = residual}; d_filt jobs {res = output picture}
void mc_add_pred_batch(hipStream_t s, const McJob *d_pred, const McJob *d_filt, int n, int nbh, int nbv, bool any_filter, int luma_w, int luma_h, int blk_w,
                       int blk_h, bool c420)
{
    if (n > 0) {
        launch_predict<MC_RECONSTRUCT>(s, d_pred, n, nbh, nbv, blk_w, blk_h, c420);
        if (any_filter) {
            bool wide = false;
            const unsigned lds = ring_lds_bytes(luma_w, luma_h, &wide);
            if (!lds) {
                DSV2_LAUNCH(k_inter_filters_g, dim3(n, 3), dim3(256), 0, s, d_filt);
            } else if (wide) {
                DSV2_LAUNCH(k_inter_filters_bw, dim3(n, 3), dim3(512), lds, s, d_filt, lds);
            } else if (filter_pair(n)) {
                DSV2_LAUNCH(k_inter_filters_b2, dim3(n, 3), dim3(512), lds, s, d_filt, lds);
            } else {
                DSV2_LAUNCH(k_inter_filters_b, dim3(n, 3), dim3(256), lds, s, d_filt, lds);
            }
        }
    }
}

void intra_filter_batch(hipStream_t s, const McJob *d_tab, int n, int luma_w, int luma_h)
{
    if (n > 0) {
        bool wide = false;
        const unsigned lds = ring_lds_bytes(luma_w, luma_h, &wide);
        if (!lds) {
            DSV2_LAUNCH(k_intra_filter_g, dim3(n), dim3(256), 0, s, d_tab);
        } else if (wide) {
            DSV2_LAUNCH(k_intra_filter_bw, dim3(n), dim3(512), lds, s, d_tab, lds);
        } else if (filter_pair(n)) {
            DSV2_LAUNCH(k_intra_filter_b2, dim3(n), dim3(512), lds, s, d_tab, lds);
        } else {
            DSV2_LAUNCH(k_intra_filter_b, dim3(n), dim3(256), lds, s, d_tab, lds);
        }
    }
}

void intra_filter_luma(hipStream_t s, const uint8_t *d_bd, const MCParams &p, int q, const DPlane &luma)
{
    if (p.lossless) {
        return;
    }
    DSV2_LAUNCH(k_intra_filter, dim3(1), dim3(256), 0, s, d_bd, make_filter_params(p, q, 1, 0), luma);
    HIPCHK(hipGetLastError());
}

} // namespace dsv2
