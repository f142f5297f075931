// quant.hip -- per-subband adaptive quantisation / dequantisation ("HZCC" quant half)
// and ordered nonzero compaction on gfx950.
//
// Replaces the arithmetic of reference src/hzcc.c: hzcc_enc (:235) quantise+dequantise in
// place, lfquant (:89), hfquant (:108), TMQ4POS_P/I (:164/:171), quantSUB (:209),
// dequantS/D (:217/:224), and the dequantisation half of hzcc_dec (:451).  The serial
// adaptive entropy coder (zero runs + NEG / adaptive Rice) stays on the host (entropy.cpp):
// the GPU hands it the nonzero symbols already compacted in scan order.
//
// Decomposition (proved bit-exact on the CPU by oracle/orc_hzcc.c):
//   pass LL, then passes l = 0,1,2; within a pass every coefficient is independent except
//   the "dependent" last column / row of a level whose parent cell lies in the first
//   column / row of the same level when subband sizes are odd (the reference rounds every
//   scanned size up, hzcc.c:40-57, so adjacent levels overlap by one line).  Dependents
//   run as a tiny second launch of the pass.
#include "dev.h"
#include "prio.h"
#include "quant.h"

namespace dsv2 {

static inline int rshift_up(int x, int s) { return (x + (1 << s) - 1) >> s; }
static inline int h_dimat(int level, int v) { return rshift_up(v, 3 - level); }
static inline int h_subband_off(int level, int sub, int w, int h)
{
    int o = 0;
    if (sub & 1) {
        o += rshift_up(w, 3 - level);
    }
    if (sub & 2) {
        o += rshift_up(h, 3 - level) * w;
    }
    return o;
}

static int udiv_up(int a, int b) { return (a + b - 1) / b; }

int spatial_psy_factor(int blk_w, int blk_h, int nbh, int nbv, int sub) // hzcc.c:67
{
    int scale, lo, hi;
    if (sub == 1) {
        lo = udiv_up(352, blk_w);
        hi = udiv_up(1920, blk_w);
        scale = nbh;
    } else if (sub == 2) {
        lo = udiv_up(288, blk_h);
        hi = udiv_up(1080, blk_h);
        scale = nbv;
    } else {
        lo = udiv_up(352, blk_w) * udiv_up(288, blk_h);
        hi = udiv_up(1920, blk_w) * udiv_up(1080, blk_h);
        scale = nbh * nbv;
    }
    scale = scale - lo > 0 ? scale - lo : 0;
    return (scale << 7) / (hi - lo);
}

static int lfquant(const QuantCfg &c, int q) // hzcc.c:89
{
    int pf = spatial_psy_factor(c.blk_w, c.blk_h, c.nbh, c.nbv, 3);
    q -= (q * pf >> 10);
    q = q > 8 ? q : 8;
    if (c.plane) {
        if (q > 256) {
            q = 256 + q / 4;
        }
        return q < 768 ? q : 768;
    }
    return q < 3072 ? q : 3072;
}

static int hfquant(const QuantCfg &c, int q, int s, int l) // hzcc.c:108
{
    bool chroma = c.plane != 0;
    int pf = spatial_psy_factor(c.blk_w, c.blk_h, c.nbh, c.nbv, s);
    q /= 2;
    pf = q * pf >> (7 + (c.isP ? 0 : 1));
    if (chroma) {
        int tl = l - 2;
        if (s == 1) {
            tl += c.hshift;
        } else if (s == 2) {
            tl += c.vshift;
        }
        q = (q * 6) / (4 - tl);
    } else if (l == 1) {
        q += pf / 2;
    } else if (l == 2) {
        q += pf;
    }
    if (c.isP) {
        if (l == 0) {
            q = q * 2 - pf;
        } else if (l == 1) {
            q -= pf / 2;
        }
        q /= 4;
        return q > 8 ? q : 8;
    }
    q = q * (15 + 3 * l) / 16;
    if (!chroma) {
        if (l == 0) {
            q = (q * 3) / 8;
        } else if (s == 3) {
            q *= 2;
        }
    } else {
        q /= 4;
        if (s == 3) {
            q *= 2;
        }
    }
    return q > 8 ? q : 8;
}

// (make_scan: scan.cpp -- host geometry without any device code, shared with the CPU-side parser fuzzing harness)

// ---- device arithmetic ------------------------------------------------------------
struct LevelArgs {
    int l;
    int sw, sh;
    int dbx, dby;
    int xdep, ydep;
    int off[3], par[3], gpar[3], base[3];
    int vec; // geometry allows the four-coefficients-per-thread kernel (every row / subband start a multiple of four)
};

__device__ __forceinline__ int quant_sub(int v, int q, int sub) { return (v >= 0 ? v - sub : v + sub) / q; }
__device__ __forceinline__ int32_t dequant_S(int v, unsigned q)
{
    return (int32_t) ((unsigned) v * q + ((v < 0) ? 0u - (q * 2 / 3) : (q * 2 / 3)));
}
__device__ __forceinline__ int32_t dequant_D(int v, unsigned q)
{
    return (int32_t) ((unsigned) v * q + ((v < 0) ? 0u - (q / 2) : (q / 2)));
}
// v ? dequant_D(v, q) : 0 without the branch the compiler makes of it
__device__ __forceinline__ int32_t dequant_D0(int v, unsigned q)
{
    const unsigned h = q / 2;
    return (int32_t) ((unsigned) v * q + (v < 0 ? 0u - h : (v > 0 ? h : 0u)));
}
__device__ __forceinline__ int sgn(int x) { return x < 0 ? -1 : (x > 0 ? 1 : 0); }

__device__ __forceinline__ int tmq_for_P(int tmq, int flags, int parc) // hzcc.c:164
{
    if (parc || (flags & (DSV_IS_STABLE | DSV_IS_EPRM))) {
        return tmq * 7 >> 3;
    }
    if (flags & DSV_IS_INTRA) {
        return tmq * 6 >> 3;
    }
    return tmq;
}

__device__ __forceinline__ int tmq_for_I(int tmq, int flags, int parc, int l) // hzcc.c:171
{
    int sm = flags & (DSV_IS_STABLE | DSV_IS_MAINTAIN);
    if (l == 0) {
        return tmq;
    }
    if (sm == DSV_IS_STABLE) {
        return (l == 2) ? (tmq >> 2) : (tmq / 3);
    }
    if (sm == DSV_IS_MAINTAIN) {
        return tmq >> ((flags & DSV_IS_RINGING) ? 2 : !parc);
    }
    if (sm == (DSV_IS_STABLE | DSV_IS_MAINTAIN)) {
        return (l == 2) ? (tmq >> (2 + !parc)) : (tmq >> 2);
    }
    return tmq;
}

// Truncating division by a positive step without the 32-bit divide sequence: for |n| < 2^20 the quotient estimated in
// single precision (v_rcp_f32: 1 ulp) is within one of the exact one, which a multiply-back settles; anything larger
// takes the integer divide.
__device__ __forceinline__ int div_trunc_pos(int n, int q)
{
    const unsigned a = (unsigned) abs(n);
    unsigned est = (unsigned) ((float) a * __builtin_amdgcn_rcpf((float) q));
    const int r = (int) (a - est * (unsigned) q); // (wraps harmlessly for the large values that take the divide below)
    est = r < 0 ? est - 1u : (r >= q ? est + 1u : est);
    if (__builtin_expect(__any(a >= (1u << 20)), 0)) { // wave-uniform: no exec-mask juggling on the usual path
        est = a >= (1u << 20) ? a / (unsigned) q : est;
    }
    return n < 0 ? -(int) est : (int) est;
}

// One divide per coefficient: the visual-masking rules of hzcc.c:364-414 only choose the step and the dead-zone offset.
struct MvBits { // what the P-picture masking rule reads of a block's vector
    int x, y;
    uint32_t flags;
};
__device__ __forceinline__ bool needs_mv(const QuantCfg &c) { return c.isP && (c.do_psy & DSV_PSY_P_VISUAL_MASKING) && c.plane == 0 && !c.lossless; }
__device__ __forceinline__ MvBits load_mv(const DSV_MV *mvs, int bi)
{
    const DSV_MV mv = mvs[bi];
    return MvBits{(int) mv.u.mv.x, (int) mv.u.mv.y, mv.flags};
}

// which of the rule sets applies is a property of the launch: QM_* fixes it at compile time for the hot kernel
enum { QM_P_PLAIN = 0, QM_P_PSY = 1, QM_I_PSY = 2, QM_I_CHROMA = 3, QM_I_PLAIN = 4 };
__host__ __device__ __forceinline__ int quant_mode(const QuantCfg &c)
{
    if (c.isP) {
        return ((c.do_psy & DSV_PSY_P_VISUAL_MASKING) && c.plane == 0) ? QM_P_PSY : QM_P_PLAIN;
    }
    return ((c.do_psy & DSV_PSY_I_VISUAL_MASKING) && c.plane == 0) ? QM_I_PSY : (c.plane ? QM_I_CHROMA : QM_I_PLAIN);
}

template <int MODE>
__device__ __forceinline__ int quant_detail_m(const MvBits &mv, int val, int qp, int l, int flags, int parc, int gparc, int &tmq_out)
{
    int tmq = qp, sub = 0;
    const bool texture = !parc, gtexture = !gparc;
    if (MODE == QM_P_PLAIN || MODE == QM_P_PSY) {
        tmq = tmq_for_P(tmq, flags, parc);
        if (MODE == QM_P_PSY) { // hzcc.c:371-380
            const bool small_mv = abs(mv.x) < 32 && abs(mv.y) < 32;
            const bool fine = (gtexture & texture) | ((mv.flags & (1u << DSV_MV_BIT_EPRM)) != 0) |
                              (((mv.flags & (1u << DSV_MV_BIT_MAINTAIN)) != 0) & small_mv);
            const bool mid = texture | !(flags & DSV_IS_SIMCMPLX);
            sub = fine ? tmq >> 3 : (mid ? tmq / 6 : tmq >> 2);
        }
    } else {
        tmq = tmq_for_I(tmq, flags, parc, l);
        if (MODE == QM_I_PSY) { // hzcc.c:387-414
            const int smf = flags & (DSV_IS_MAINTAIN | DSV_IS_STABLE);
            const bool edge = sgn(parc) == sgn(val);
            const int stp = smf == 0 ? -tmq / 3 : ((edge && smf == DSV_IS_STABLE) ? tmq >> 3 : -tmq / 6);
            sub = (flags & DSV_IS_RINGING) ? -(tmq / 6) : (l == 0 ? -(tmq >> 3) : stp);
        } else if (MODE == QM_I_CHROMA) {
            sub = -(tmq >> 3);
        }
    }
    tmq_out = tmq;
    return div_trunc_pos(val >= 0 ? val - sub : val + sub, tmq);
}

__device__ __forceinline__ int quant_detail(const QuantCfg &c, const MvBits &mv, int val, int qp, int l, int flags, int parc, int gparc,
                                            int &tmq_out)
{
    switch (quant_mode(c)) {
        case QM_P_PLAIN: return quant_detail_m<QM_P_PLAIN>(mv, val, qp, l, flags, parc, gparc, tmq_out);
        case QM_P_PSY: return quant_detail_m<QM_P_PSY>(mv, val, qp, l, flags, parc, gparc, tmq_out);
        case QM_I_PSY: return quant_detail_m<QM_I_PSY>(mv, val, qp, l, flags, parc, gparc, tmq_out);
        case QM_I_CHROMA: return quant_detail_m<QM_I_CHROMA>(mv, val, qp, l, flags, parc, gparc, tmq_out);
        default: return quant_detail_m<QM_I_PLAIN>(mv, val, qp, l, flags, parc, gparc, tmq_out);
    }
}

// Kernels work through PlaneJob records (dev.h): tab == nullptr runs the single job `one`,
// otherwise a grid dimension indexes a device table whose entries share the geometry in `c`.

// Nonzero bookkeeping for the compaction that follows (saves it a pass over the dense values): the wavefront's
// nonzeros are tallied per 1024-position tile of the stream's symbol list; a wavefront writes consecutive scan
// positions, so one atomic per wavefront is the rule and a tile boundary inside it the exception.
__device__ __forceinline__ void count_nonzero(const PlaneJob &J, size_t pos, int v)
{
    if (J.tile_count == nullptr) {
        return;
    }
    const bool nz = v != 0;
    const unsigned tile = (unsigned) ((J.qv_base + pos) >> 10);
    const unsigned t0 = (unsigned) __builtin_amdgcn_readfirstlane((int) tile);
    const unsigned long long same = __ballot(nz && tile == t0);
    const unsigned long long active = __ballot(true);
    const int lane = (int) (threadIdx.x + threadIdx.y * blockDim.x) & 63;
    if (same && lane == __ffsll((long long) active) - 1) {
        atomicAdd(&J.tile_count[t0], __popcll(same));
    }
    if (nz && tile != t0) {
        atomicAdd(&J.tile_count[tile], 1);
    }
}

// LL region: hzcc.c:308-328
__global__ __launch_bounds__(256) void k_quant_ll(const PlaneJob *__restrict__ tab, PlaneJob one, QuantCfg c, int sw, int sh)
{
    DSV2_KERNEL_PRIO();
    const PlaneJob &J = tab ? tab[blockIdx.z] : one;
    int x = blockIdx.x * 64 + threadIdx.x;
    int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= sw || y >= sh) {
        return;
    }
    int qp = J.qll;
    int v = 0;
    if (x | y) { // the global DC is transmitted separately and never quantised (hzcc.c:265,599-602)
        int32_t *cell = J.coefs + (size_t) y * c.w + x;
        if (c.lossless) {
            v = *cell;
        } else {
            int val = *cell;
            v = c.isP ? (val / qp) : quant_sub(val, qp, -(qp / 6));
            *cell = v ? (c.isP ? dequant_D(v, (unsigned) qp) : dequant_S(v, (unsigned) qp)) : 0;
        }
    }
    J.qv[(size_t) y * sw + x] = v;
    count_nonzero(J, (size_t) y * sw + x, v);
}

__device__ __forceinline__ void quant_cell(const PlaneJob &J, const QuantCfg &c, const LevelArgs &a, int si, int x, int y)
{
    int32_t *coefs = J.coefs;
    int32_t *cell = coefs + a.off[si] + (size_t) y * c.w + x;
    int v;
    if (c.lossless) {
        v = *cell;
    } else {
        int bi = ((y * a.dby) >> kBlockP) * c.nbh + ((x * a.dbx) >> kBlockP);
        int parc = coefs[a.par[si] + (size_t) (y >> 1) * c.w + (x >> 1)];
        int gparc = coefs[a.gpar[si] + (size_t) (y >> 2) * c.w + (x >> 2)];
        int tmq;
        const MvBits mv = needs_mv(c) ? load_mv(J.mvs, bi) : MvBits{0, 0, 0u};
        v = quant_detail(c, mv, *cell, J.qp[a.l][si], a.l, J.bd[bi], parc, gparc, tmq);
        *cell = v ? dequant_D(v, (unsigned) tmq) : 0;
    }
    J.qv[a.base[si] + (size_t) y * a.sw + x] = v;
    count_nonzero(J, a.base[si] + (size_t) y * a.sw + x, v);
}

__device__ __forceinline__ bool is_dependent(const LevelArgs &a, int s, int x, int y)
{
    return ((s & 1) && a.xdep && x == a.sw - 1) || ((s & 2) && a.ydep && y == a.sh - 1);
}

// phase A of a detail level: every cell that is not a dependent; blockIdx.z = 3 * job + (subband - 1)
__global__ __launch_bounds__(256) void k_quant_level(const PlaneJob *__restrict__ tab, PlaneJob one, QuantCfg c, LevelArgs a)
{
    int x = blockIdx.x * 64 + threadIdx.x;
    int y = blockIdx.y * 4 + threadIdx.y;
    int si = blockIdx.z % 3;
    const PlaneJob &J = tab ? tab[blockIdx.z / 3] : one;
    if (x >= a.sw || y >= a.sh || is_dependent(a, si + 1, x, y)) {
        return;
    }
    quant_cell(J, c, a, si, x, y);
}

// The same, four coefficients of a row per thread: 16-byte loads and stores of the coefficients and of the dense symbol
// values, one parent pair and one grandparent for the four, the block's flags and vector fetched once when the four share
// a block (levels 0 and 1 of the 16-pixel-block geometries).  A thread whose fourth coefficient is a dependent of the
// last column, and any job whose buffers are not 16-byte aligned, goes cell by cell.
// kQRows rows per thread (rows y, y + 4, ... of the workgroup's band), every row's 16-byte load issued before the first is looked at:
// beside three other lockstep groups a streaming kernel holds one or two wavefronts per SIMD, not the six it holds alone
// (profiles/r06_occupancy.txt), and a workgroup's life is a chain of round trips -- kernel arguments, job record, data; throughput
// under load is (resident wavefronts) x (bytes in flight per wavefront) / latency, and the middle factor is the kernel's to choose.
constexpr int kQRows = 4;
template <int MODE>
__device__ __forceinline__ void quant_row4(const PlaneJob &J, const QuantCfg &c, const LevelArgs &a, int si, int x, int y, int4 *cell, const int4 cv)
{
    int32_t *coefs = J.coefs;
    if (MODE == QM_P_PLAIN || MODE == QM_P_PSY) {
        // A P picture's step is at least 6/8 of the subband's (tmq_for_P) and its dead-zone offset is never negative (hzcc.c:364-380):
        // a coefficient below 6/8 of the step quantises to zero whatever its parent, its block's flags and vector say.  Most of a
        // residual's detail coefficients are that small, and they come in runs: a wavefront whose 256 are all below the bound stores
        // its zeros and is done -- no parent / grandparent / flag / vector loads, no rule evaluation, no divisions, nothing to count.
        const int zmin = J.qp[a.l][si] * 6 >> 3;
        const bool small = abs(cv.x) < zmin && abs(cv.y) < zmin && abs(cv.z) < zmin && abs(cv.w) < zmin;
        if (__all(small)) {
            *cell = make_int4(0, 0, 0, 0);
            *(int4 *) (J.qv + a.base[si] + (size_t) y * a.sw + x) = make_int4(0, 0, 0, 0);
            return;
        }
    }
    const int2 pc = *(const int2 *) (coefs + a.par[si] + (size_t) (y >> 1) * c.w + (x >> 1));
    const int gparc = coefs[a.gpar[si] + (size_t) (y >> 2) * c.w + (x >> 2)];
    const int rowb = ((y * a.dby) >> kBlockP) * c.nbh;
    int bk[4], flags[4];
    MvBits mv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        bk[k] = rowb + (((x + k) * a.dbx) >> kBlockP);
    }
    const bool mvq = MODE == QM_P_PSY;
    const int qp = J.qp[a.l][si];
    const int val[4] = {cv.x, cv.y, cv.z, cv.w};
    int v[4], dq[4], nzc = 0;
    flags[0] = J.bd[bk[0]];
    mv[0] = mvq ? load_mv(J.mvs, bk[0]) : MvBits{0, 0, 0u};
    if (__any(bk[3] != bk[0])) {
#pragma unroll
        for (int k = 1; k < 4; k++) {
            flags[k] = J.bd[bk[k]];
            mv[k] = mvq ? load_mv(J.mvs, bk[k]) : MvBits{0, 0, 0u};
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int tmq;
            v[k] = quant_detail_m<MODE>(mv[k], val[k], qp, a.l, flags[k], k < 2 ? pc.x : pc.y, gparc, tmq);
            dq[k] = dequant_D0(v[k], (unsigned) tmq);
        }
    } else { // one block for the four: the step / dead-zone rules are evaluated once per parent (the same operands twice)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int tmq;
            v[k] = quant_detail_m<MODE>(mv[0], val[k], qp, a.l, flags[0], k < 2 ? pc.x : pc.y, gparc, tmq);
            dq[k] = dequant_D0(v[k], (unsigned) tmq);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        nzc += v[k] != 0;
    }
    *cell = make_int4(dq[0], dq[1], dq[2], dq[3]);
    const size_t pos = a.base[si] + (size_t) y * a.sw + x;
    *(int4 *) (J.qv + pos) = make_int4(v[0], v[1], v[2], v[3]);
    if (J.tile_count != nullptr) { // as count_nonzero, with up to four nonzeros a lane (its four positions share a tile)
        const unsigned tile = (unsigned) ((J.qv_base + pos) >> 10);
        const unsigned t0 = (unsigned) __builtin_amdgcn_readfirstlane((int) tile);
        const bool here = tile == t0;
        const int sum = __popcll(__ballot(here && (nzc & 1))) + 2 * __popcll(__ballot(here && (nzc & 2))) + 4 * __popcll(__ballot(here && (nzc & 4)));
        const unsigned long long active = __ballot(true);
        const int lane = (int) (threadIdx.x + threadIdx.y * blockDim.x) & 63;
        if (sum && lane == __ffsll((long long) active) - 1) {
            atomicAdd(&J.tile_count[t0], sum);
        }
        if (nzc && !here) {
            atomicAdd(&J.tile_count[tile], nzc);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_quant_level4(const PlaneJob *__restrict__ tab, PlaneJob one, QuantCfg c, LevelArgs a)
{
    DSV2_KERNEL_PRIO();
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int yb = blockIdx.y * (4 * kQRows) + threadIdx.y;
    const int si = blockIdx.z % 3;
    const PlaneJob &J = tab ? tab[blockIdx.z / 3] : one;
    if (x >= a.sw || yb >= a.sh) {
        return;
    }
    const int ylim = (((si + 1) & 2) && a.ydep) ? a.sh - 1 : a.sh; // (the last row's dependents are phase B's)
    const bool coldep = ((si + 1) & 1) && a.xdep && x + 4 == a.sw;
    const bool aligned = ((((uintptr_t) J.coefs) | ((uintptr_t) J.qv)) & 15) == 0 && (J.qv_base & 3u) == 0;
    if (c.lossless || coldep || !aligned) {
        for (int r = 0; r < kQRows; r++) {
            const int y = yb + 4 * r;
            if (y < ylim) {
                for (int k = 0; k < (coldep ? 3 : 4); k++) {
                    quant_cell(J, c, a, si, x + k, y);
                }
            }
        }
        return;
    }
    int4 *cell[kQRows];
    int4 cv[kQRows];
#pragma unroll
    for (int r = 0; r < kQRows; r++) { // (wave-uniform: a wavefront is one row segment)
        const int y = yb + 4 * r;
        cell[r] = (int4 *) (J.coefs + a.off[si] + (size_t) (y < ylim ? y : yb) * c.w + x);
        cv[r] = *cell[r];
    }
#pragma unroll
    for (int r = 0; r < kQRows; r++) {
        const int y = yb + 4 * r;
        if (y < ylim) {
            quant_row4<MODE>(J, c, a, si, x, y, cell[r], cv[r]);
        }
    }
}

// phase B: the dependents (last column, then last row without the shared corner); blockIdx.y = subband - 1
__global__ __launch_bounds__(256) void k_quant_level_dep(const PlaneJob *__restrict__ tab, PlaneJob one, QuantCfg c, LevelArgs a)
{
    const PlaneJob &J = tab ? tab[blockIdx.z] : one;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int si = blockIdx.y;
    int x, y;
    if (t < a.sh) {
        x = a.sw - 1;
        y = t;
    } else if (t < a.sh + a.sw - 1) {
        x = t - a.sh;
        y = a.sh - 1;
    } else {
        return;
    }
    if (!is_dependent(a, si + 1, x, y)) {
        return;
    }
    quant_cell(J, c, a, si, x, y);
}

void quant_steps(PlaneJob *job, const QuantCfg &cfg, int q)
{
    int qf = q * 3 / 2; // fix_quant, hzcc.c:59
    job->q = q;
    job->qll = cfg.lossless ? 1 : lfquant(cfg, qf);
    for (int l = 0; l < 3; l++) {
        for (int si = 0; si < 3; si++) {
            job->qp[l][si] = cfg.lossless ? 1 : hfquant(cfg, qf, si + 1, l);
        }
    }
}

static void quant_launch(hipStream_t s, const PlaneJob *tab, const PlaneJob &one, int n, const QuantCfg &cfg)
{
    ScanGeom g;
    make_scan(&g, cfg.w, cfg.h);
    const dim3 blk(64, 4);
    unsigned nz = tab ? (unsigned) n : 1u;
    DSV2_LAUNCH(k_quant_ll, dim3((g.sw[0] + 63) / 64, (g.sh[0] + 3) / 4, nz), blk, 0, s, tab, one, cfg, g.sw[0], g.sh[0]);
    for (int l = 0; l < 3; l++) {
        LevelArgs a;
        a.l = l;
        a.sw = h_dimat(l, cfg.w);
        a.sh = h_dimat(l, cfg.h);
        a.dbx = (cfg.nbh << kBlockP) / a.sw;
        a.dby = (cfg.nbv << kBlockP) / a.sh;
        a.xdep = 2 * h_dimat(l - 1, cfg.w) > a.sw;
        a.ydep = 2 * h_dimat(l - 1, cfg.h) > a.sh;
        for (int si = 0; si < 3; si++) {
            a.off[si] = g.off[1 + 3 * l + si];
            a.base[si] = g.base[1 + 3 * l + si];
            a.par[si] = h_subband_off(l - 1, si + 1, cfg.w, cfg.h);
            a.gpar[si] = h_subband_off(l - 2, si + 1, cfg.w, cfg.h);
        }
        a.vec = (cfg.w & 3) == 0 && (a.sw & 3) == 0;
        for (int si = 0; si < 3; si++) {
            a.vec = a.vec && (a.off[si] & 3) == 0 && (a.base[si] & 3) == 0 && (a.par[si] & 1) == 0;
        }
        if (a.vec) {
            const dim3 grid4((a.sw / 4 + 63) / 64, (a.sh + 4 * kQRows - 1) / (4 * kQRows), 3 * nz);
            switch (quant_mode(cfg)) { // (a lossless launch goes cell by cell inside the kernel: any instance will do)
                case QM_P_PLAIN: DSV2_LAUNCH(k_quant_level4<QM_P_PLAIN>, grid4, blk, 0, s, tab, one, cfg, a); break;
                case QM_P_PSY: DSV2_LAUNCH(k_quant_level4<QM_P_PSY>, grid4, blk, 0, s, tab, one, cfg, a); break;
                case QM_I_PSY: DSV2_LAUNCH(k_quant_level4<QM_I_PSY>, grid4, blk, 0, s, tab, one, cfg, a); break;
                case QM_I_CHROMA: DSV2_LAUNCH(k_quant_level4<QM_I_CHROMA>, grid4, blk, 0, s, tab, one, cfg, a); break;
                default: DSV2_LAUNCH(k_quant_level4<QM_I_PLAIN>, grid4, blk, 0, s, tab, one, cfg, a); break;
            }
        } else {
            DSV2_LAUNCH(k_quant_level, dim3((a.sw + 63) / 64, (a.sh + 3) / 4, 3 * nz), blk, 0, s, tab, one, cfg, a);
        }
        if (a.xdep || a.ydep) {
            DSV2_LAUNCH(k_quant_level_dep, dim3((a.sw + a.sh + 255) / 256, 3, nz), dim3(256), 0, s, tab, one, cfg, a);
        }
    }
    HIPCHK(hipGetLastError());
}

void quant_plane(hipStream_t s, DCoefs coefs, int32_t *qv, const QuantCfg &cfg, int q)
{
    PlaneJob one = {};
    one.coefs = coefs.data;
    one.qv = qv;
    one.bd = cfg.bd;
    one.mvs = cfg.mvs;
    quant_steps(&one, cfg, q);
    quant_launch(s, nullptr, one, 1, cfg);
}

void quant_jobs(hipStream_t s, const PlaneJob *d_jobs, int n, const QuantCfg &cfg)
{
    if (n > 0) {
        quant_launch(s, d_jobs, PlaneJob{}, n, cfg);
    }
}

// ---- ordered compaction of the nonzero symbols ---------------------------------------
// three small kernels: per-tile counts, one-workgroup exclusive scan of the tile counts,
// ordered scatter.  Tiles are 1024 consecutive scan positions handled by 256 threads.
constexpr int kTile = 1024;

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) {
            v += t;
        }
    }
    return v;
}

__global__ __launch_bounds__(256) void k_count(const CompactJob *__restrict__ tab, CompactJob one)
{
    __shared__ int wsum[4];
    const CompactJob J = job_of(tab, blockIdx.y, one);
    const int32_t *qv = J.qv;
    int n = J.n;
    int *tile_count = J.tile_count;
    int base = blockIdx.x * kTile;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int i = base + threadIdx.x * 4 + j;
        cnt += (i < n && qv[i] != 0);
    }
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = wave_incl_scan(cnt, lane);
    if (lane == 63) {
        wsum[wv] = inc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// exclusive scan of up to 1024*ntile_per_thread tile counts by one workgroup; also emits the total
// one workgroup of 256 per job (a 1024-thread workgroup needs sixteen free wavefront slots on ONE compute unit at once: beside
// the search's resident wavefronts its placement alone took ~1 ms in the four-group bench)
constexpr int kScanThreads = 256;
__global__ __launch_bounds__(kScanThreads) void k_scan_tiles(const CompactJob *__restrict__ tab, CompactJob one, int reset)
{
    DSV2_KERNEL_PRIO();
    __shared__ int wsum[kScanThreads / 64];
    const CompactJob J = job_of(tab, blockIdx.y, one);
    const int *tile_count = J.tile_count;
    int ntiles = (J.n + kTile - 1) / kTile;
    int *tile_base = J.tile_base, *total = J.total;
    __shared__ int carry;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        carry = 0;
    }
    __syncthreads();
    for (int start = 0; start < ntiles; start += kScanThreads) {
        int i = start + threadIdx.x;
        int v = i < ntiles ? tile_count[i] : 0;
        if (reset && i < ntiles) {
            J.tile_count[i] = 0; // the quantiser of the next frame accumulates into it
        }
        int inc = wave_incl_scan(v, lane);
        if (lane == 63) {
            wsum[wv] = inc;
        }
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wv; k++) {
            woff += wsum[k];
        }
        int c0 = carry;
        if (i < ntiles) {
            tile_base[i] = c0 + woff + inc - v;
        }
        __syncthreads();
        if (threadIdx.x == kScanThreads - 1) {
            carry = c0 + woff + inc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = carry;
    }
}

// kScatterTiles tiles per workgroup.  Three tiles in four of a P picture hold nothing, and a workgroup per tile spent its life on the
// round trips that tell it so (kernel arguments, job record, two tile bases) -- 3 038 workgroups a picture, 9 x slower under load than
// alone (round 6).  Now the first nine threads fetch the nine bases of eight tiles at once and the workgroup walks the tiles that hold something.
constexpr int kScatterTiles = 8;
__global__ __launch_bounds__(256) void k_scatter(const CompactJob *__restrict__ tab, CompactJob one)
{
    DSV2_KERNEL_PRIO();
    __shared__ int wsum[4];
    __shared__ uint32_t spos[kTile];
    __shared__ int32_t sval[kTile];
    __shared__ int tbs[kScatterTiles + 1];
    const CompactJob J = job_of(tab, blockIdx.y, one);
    const int32_t *qv = J.qv;
    int n = J.n;
    uint32_t *out_pos = J.pos;
    int32_t *out_val = J.val;
    uint32_t *host_pos = J.host_pos;
    int32_t *host_val = J.host_val;
    int host_cap = host_pos ? J.host_cap : 0;
    const int ntiles = (n + kTile - 1) / kTile;
    const int t0 = blockIdx.x * kScatterTiles;
    if (t0 >= ntiles) {
        return;
    }
    // the scan of the tile counts tells which tiles hold nothing (most of the finest level of a P picture): not read at all
    if (threadIdx.x <= kScatterTiles) {
        const int t = t0 + (int) threadIdx.x;
        tbs[threadIdx.x] = t < ntiles ? J.tile_base[t] : *J.total;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < kScatterTiles && t0 + k < ntiles; k++) {
        const int tb = tbs[k], tb_next = tbs[k + 1];
        if (tb_next == tb) {
            continue; // (uniform over the workgroup)
        }
        const int base = (t0 + k) * kTile;
        int vals[4];
        int cnt = 0;
        const int i0 = base + threadIdx.x * 4;
        if (i0 + 3 < n && (((uintptr_t) qv) & 15) == 0) {
            const int4 q4 = *(const int4 *) (qv + i0);
            vals[0] = q4.x, vals[1] = q4.y, vals[2] = q4.z, vals[3] = q4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                vals[j] = (i0 + j < n) ? qv[i0 + j] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cnt += vals[j] != 0;
        }
        int inc = wave_incl_scan(cnt, lane);
        if (lane == 63) {
            wsum[wv] = inc;
        }
        __syncthreads();
        int o = inc - cnt;
        for (int w = 0; w < wv; w++) {
            o += wsum[w];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (vals[j] != 0) {
                spos[o] = (uint32_t) (base + threadIdx.x * 4 + j);
                sval[o] = vals[j];
                o++;
            }
        }
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
        for (int q = threadIdx.x; q < total; q += 256) {
            uint32_t p = spos[q];
            int32_t v = sval[q];
            if (tb + q < J.list_cap) { // (a picture with more symbols than the lists hold: the host sees it in *total and has it redone)
                out_pos[tb + q] = p;
                out_val[tb + q] = v;
            }
            if (tb + q < host_cap) {
                host_pos[tb + q] = p;
                host_val[tb + q] = v;
            }
        }
        __syncthreads(); // (wsum / spos / sval are the next tile's)
    }
}

void Compactor::ensure(size_t n) { ensure_lists(n, n); }

void Compactor::ensure_lists(size_t n, size_t symbols)
{
    if (n <= cap) {
        return;
    }
    release();
    symbols = symbols < n ? symbols : n;
    size_t ntiles = (n + kTile - 1) / kTile;
    HIPCHK(dev_alloc((void **) &tile_count, ntiles * sizeof(int)));
    dev_zero(tile_count, ntiles * sizeof(int));
    HIPCHK(dev_alloc((void **) &tile_base, ntiles * sizeof(int)));
    HIPCHK(dev_alloc((void **) &d_total, sizeof(int)));
    HIPCHK(dev_alloc((void **) &d_pos, symbols * sizeof(uint32_t)));
    HIPCHK(dev_alloc((void **) &d_val, symbols * sizeof(int32_t)));
    HIPCHK(hipHostMalloc((void **) &h_total, sizeof(int), hipHostMallocDefault));
    cap = n;
    list_cap = symbols;
}

void Compactor::grow_lists(size_t symbols)
{
    symbols = symbols < cap ? symbols : cap;
    if (symbols <= list_cap) {
        return;
    }
    dev_release(d_pos); // (inside the instance's arena: stays there unused; outside: freed)
    dev_release(d_val);
    HIPCHK(hipMalloc((void **) &d_pos, symbols * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **) &d_val, symbols * sizeof(int32_t)));
    list_cap = symbols;
}

void Compactor::release()
{
    if (!cap) {
        return;
    }
    dev_release(tile_count);
    dev_release(tile_base);
    dev_release(d_total);
    dev_release(d_pos);
    dev_release(d_val);
    HIPCHK(hipHostFree(h_total));
    cap = 0;
}

CompactJob Compactor::job(const int32_t *qv, size_t n)
{
    ensure(n);
    return CompactJob{qv, (int) n, tile_count, tile_base, d_total, d_pos, d_val, nullptr, nullptr, 0, (int) list_cap};
}

void Compactor::run(hipStream_t s, const int32_t *qv, size_t n)
{
    CompactJob one = job(qv, n);
    int ntiles = (int) ((n + kTile - 1) / kTile);
    DSV2_LAUNCH(k_count, dim3(ntiles), dim3(256), 0, s, nullptr, one);
    DSV2_LAUNCH(k_scan_tiles, dim3(1), dim3(kScanThreads), 0, s, nullptr, one, 1); // counts always left at zero
    DSV2_LAUNCH(k_scatter, dim3((ntiles + kScatterTiles - 1) / kScatterTiles), dim3(256), 0, s, nullptr, one);
    HIPCHK(hipMemcpyAsync(h_total, d_total, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipGetLastError());
}

void compact_jobs(hipStream_t s, const CompactJob *d_jobs, int njobs, size_t n, bool counted)
{
    if (njobs <= 0) {
        return;
    }
    int ntiles = (int) ((n + kTile - 1) / kTile);
    if (!counted) { // else the quantiser tallied the tiles while writing the values (count_nonzero)
        DSV2_LAUNCH(k_count, dim3(ntiles, njobs), dim3(256), 0, s, d_jobs, CompactJob{});
    }
    DSV2_LAUNCH(k_scan_tiles, dim3(1, njobs), dim3(kScanThreads), 0, s, d_jobs, CompactJob{}, counted ? 1 : 0);
    DSV2_LAUNCH(k_scatter, dim3((ntiles + kScatterTiles - 1) / kScatterTiles, njobs), dim3(256), 0, s, d_jobs, CompactJob{});
    HIPCHK(hipGetLastError());
}

// ---- decoder side: scatter decoded symbols and dequantise (hzcc.c:451-583) -------------------
// One DequantJob per plane (and stream): its symbol list sorted by scan position, split by seg[] into
// {LL, level 0, level 1, level 2}.  tab == nullptr: the single job `one`; else blockIdx.y indexes the table.
struct DequantArgs {
    int l;
    int sw, sh, dbx, dby;
    int off[3], par[3], base[3];
};

// LL symbols: dequantL (hzcc.c:530); thread 0 also plants the separately transmitted DC (hzcc.c:599-602)
__global__ __launch_bounds__(256) void k_dequant_ll(const DequantJob *__restrict__ tab, DequantJob one, QuantCfg c, int sw)
{
    const DequantJob &J = tab ? tab[blockIdx.y] : one;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        J.coefs[0] = J.LL;
    }
    if (i >= J.seg[0]) {
        return;
    }
    int p = (int) J.pos[i], v = J.val[i];
    int x = p % sw, y = p / sw;
    unsigned qp = (unsigned) J.qll;
    int32_t out = c.lossless ? v : (c.isP ? dequant_D(v, qp) : dequant_S(v, qp));
    J.coefs[(size_t) y * c.w + x] = out;
}

// detail symbols of one level; `dep` selects the dependents phase (see header comment)
__global__ __launch_bounds__(256) void k_dequant_level(const DequantJob *__restrict__ tab, DequantJob one, QuantCfg c, DequantArgs a,
                                                       int xdep, int ydep, int dep)
{
    const DequantJob &J = tab ? tab[blockIdx.y] : one;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= J.seg[1 + a.l]) {
        return;
    }
    int first = J.seg[0] + (a.l > 0 ? J.seg[1] : 0) + (a.l > 1 ? J.seg[2] : 0);
    int p = (int) J.pos[first + i], v = J.val[first + i];
    int32_t *coefs = J.coefs;
    int si = p >= a.base[2] ? 2 : (p >= a.base[1] ? 1 : 0);
    int local = p - a.base[si];
    int x = local % a.sw, y = local / a.sw;
    int s = si + 1;
    bool is_dep = ((s & 1) && xdep && x == a.sw - 1) || ((s & 2) && ydep && y == a.sh - 1);
    if ((int) is_dep != dep) {
        return;
    }
    int32_t out;
    if (c.lossless) {
        out = v;
    } else {
        int flags = J.bd[((y * a.dby) >> kBlockP) * c.nbh + ((x * a.dbx) >> kBlockP)];
        int parc = coefs[a.par[si] + (size_t) (y >> 1) * c.w + (x >> 1)];
        int qp = J.qp[a.l][si];
        int tmq = c.isP ? tmq_for_P(qp, flags, parc) : tmq_for_I(qp, flags, parc, a.l);
        out = dequant_D(v, (unsigned) tmq);
    }
    coefs[a.off[si] + (size_t) y * c.w + x] = out;
}

void dequant_steps(DequantJob *job, const QuantCfg &cfg, int q)
{
    int qf = q * 3 / 2;
    job->qll = cfg.lossless ? 1 : lfquant(cfg, qf);
    for (int l = 0; l < 3; l++) {
        for (int si = 0; si < 3; si++) {
            job->qp[l][si] = cfg.lossless ? 1 : hfquant(cfg, qf, si + 1, l);
        }
    }
}

// max_seg[k] bounds seg[k] over the jobs (grid size)
static void dequant_launch(hipStream_t s, const DequantJob *tab, const DequantJob &one, int n, const int max_seg[4], const QuantCfg &cfg)
{
    ScanGeom g;
    make_scan(&g, cfg.w, cfg.h);
    unsigned ny = tab ? (unsigned) n : 1u;
    DSV2_LAUNCH(k_dequant_ll, dim3((unsigned) (max_seg[0] + 255) / 256 + (max_seg[0] == 0), ny), dim3(256), 0, s, tab, one, cfg, g.sw[0]);
    for (int l = 0; l < 3; l++) {
        int nmax = max_seg[1 + l];
        if (nmax <= 0) {
            continue;
        }
        DequantArgs a;
        a.l = l;
        a.sw = h_dimat(l, cfg.w);
        a.sh = h_dimat(l, cfg.h);
        a.dbx = (cfg.nbh << kBlockP) / a.sw;
        a.dby = (cfg.nbv << kBlockP) / a.sh;
        int xdep = 2 * h_dimat(l - 1, cfg.w) > a.sw, ydep = 2 * h_dimat(l - 1, cfg.h) > a.sh;
        for (int si = 0; si < 3; si++) {
            a.off[si] = g.off[1 + 3 * l + si];
            a.base[si] = g.base[1 + 3 * l + si];
            a.par[si] = h_subband_off(l - 1, si + 1, cfg.w, cfg.h);
        }
        DSV2_LAUNCH(k_dequant_level, dim3((unsigned) (nmax + 255) / 256, ny), dim3(256), 0, s, tab, one, cfg, a, xdep, ydep, 0);
        if (xdep || ydep) {
            DSV2_LAUNCH(k_dequant_level, dim3((unsigned) (nmax + 255) / 256, ny), dim3(256), 0, s, tab, one, cfg, a, xdep, ydep, 1);
        }
    }
    HIPCHK(hipGetLastError());
}

void dequant_plane(hipStream_t s, DCoefs coefs, const uint32_t *d_pos, const int32_t *d_val, const int seg_count[4], int32_t LL,
                   const QuantCfg &cfg, int q)
{
    DequantJob one = {};
    one.coefs = coefs.data;
    one.pos = d_pos;
    one.val = d_val;
    for (int k = 0; k < 4; k++) {
        one.seg[k] = seg_count[k];
    }
    one.bd = cfg.bd;
    one.LL = LL;
    dequant_steps(&one, cfg, q);
    dequant_launch(s, nullptr, one, 1, seg_count, cfg);
}

void dequant_jobs(hipStream_t s, const DequantJob *d_jobs, int n, const int max_seg[4], const QuantCfg &cfg)
{
    if (n > 0) {
        dequant_launch(s, d_jobs, DequantJob{}, n, max_seg, cfg);
    }
}

} // namespace dsv2
