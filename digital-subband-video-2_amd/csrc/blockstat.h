// blockstat.h -- wavefront-cooperative per-block statistics (device code).
//
// Replaces the scalar block measures of reference src/hme.c: iisqrt (:99), block_avg (:436),
// block_tex (:492), block_var (:518), block_detail (:546), quant_tex (:586), block_peaks (:624),
// block_hist_var (:711), c_average (:751), chroma_analysis (:69).
//
// Calling convention: all 64 lanes of ONE wavefront call each function with the same
// arguments; pixels of the block are dealt round-robin to the lanes, partial sums are
// combined with a butterfly of cross-lane shuffles (6 steps) and every lane receives the
// result.  Kernels that use them run one wavefront per workgroup, so __syncthreads() is a
// single-wave barrier that only orders the LDS histogram accesses.
#pragma once

#include "dev.h"

namespace dsv2 {

// ---- cross-lane exchange without the LDS crossbar ------------------------------------------------
// lane_xor<B>(v) = v of lane (lane ^ B).  gfx950: the 32- and 16-lane halves swap with
// v_permlane32_swap / v_permlane16_swap, everything inside a row of 16 is a DPP move.
typedef unsigned uint2v_t __attribute__((ext_vector_type(2)));

template <int CTRL> __device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

template <int B> __device__ __forceinline__ int lane_xor(int v)
{
    if (B == 32) {
        uint2v_t r = __builtin_amdgcn_permlane32_swap((unsigned) v, (unsigned) v, false, false);
        return (int) ((threadIdx.x & 32) ? r[0] : r[1]); // r0 = {lo, lo}, r1 = {hi, hi}
    } else if (B == 16) {
        uint2v_t r = __builtin_amdgcn_permlane16_swap((unsigned) v, (unsigned) v, false, false);
        return (int) ((threadIdx.x & 16) ? r[0] : r[1]);
    } else if (B == 8) {
        return dpp_mov<0x128>(v); // row_ror:8
    } else if (B == 4) {
        int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false); // row_shl:4 into banks 0,2
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);  // row_shr:4 into banks 1,3
    } else if (B == 2) {
        return dpp_mov<0x4e>(v); // quad_perm [2,3,0,1]
    } else {
        return dpp_mov<0xb1>(v); // quad_perm [1,0,3,2]
    }
}

// v[lane] + v[lane ^ B] in every lane (one all-reduce step)
template <int B> __device__ __forceinline__ int fold_xor(int v)
{
    if (B == 32) {
        uint2v_t r = __builtin_amdgcn_permlane32_swap((unsigned) v, (unsigned) v, false, false);
        return (int) (r[0] + r[1]);
    } else if (B == 16) {
        uint2v_t r = __builtin_amdgcn_permlane16_swap((unsigned) v, (unsigned) v, false, false);
        return (int) (r[0] + r[1]);
    }
    return v + lane_xor<B>(v);
}

__device__ __forceinline__ int wave_sum(int v)
{
    v = fold_xor<32>(v);
    v = fold_xor<16>(v);
    v = fold_xor<8>(v);
    v = fold_xor<4>(v);
    v = fold_xor<2>(v);
    v = fold_xor<1>(v);
    // every lane holds the total: hand it out as a wave-uniform scalar (SGPR), which keeps the many
    // block statistics out of the vector register file
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) { return (unsigned) wave_sum((int) v); }

// floor(sqrt(n)) -- what the bit-by-bit loop of hme.c:99 computes -- from the hardware's float square root plus one
// exact integer correction step.  (float) n and v_sqrt_f32 are each within a relative 2^-23: the truncated estimate is off
// by at most one either way for every 32-bit n (root <= 65535, so r * r never overflows).  The loop was ~100 instructions
// with a data-dependent trip count, and the block metric calls this once per scored vector: a quarter of the search's
// vector instructions.
__device__ __forceinline__ unsigned isqrt_u32(unsigned n)
{
    unsigned r = (unsigned) __builtin_amdgcn_sqrtf((float) n); // v_sqrt_f32: one instruction, 1 ulp
    r = r > 65535u ? 65535u : r;
    if (r * r > n) {
        r--;
    } else if (r < 65535u && (r + 1u) * (r + 1u) <= n) {
        r++;
    }
    return r;
}

// idx -> (x, y) in a w-wide block, for the wave-cooperative loops below: block widths are powers of two except where the
// picture's edge clips one, and a 32-bit division costs some twenty instructions per pixel
struct RowSplit {
    int w, sh;
    bool p2;
    __device__ __forceinline__ explicit RowSplit(int w_) : w(w_), sh(31 - __builtin_clz((unsigned) (w_ | 1))), p2((w_ & (w_ - 1)) == 0) {}
    __device__ __forceinline__ void operator()(int idx, int &x, int &y) const
    {
        if (p2) {
            x = idx & (w - 1);
            y = idx >> sh;
        } else {
            y = idx / w;
            x = idx - y * w;
        }
    }
};

struct Grad {
    unsigned sh, sv;
    int sum;
};

__device__ __forceinline__ Grad ws_gradients(const uint8_t *a, int as, int w, int h)
{
    int lane = threadIdx.x & 63;
    unsigned sh = 0, sv = 0;
    int s = 0;
    const RowSplit split(w);
    for (int idx = lane; idx < w * h; idx += 64) {
        int x, y;
        split(idx, x, y);
        const uint8_t *p = a + (ptrdiff_t) y * as + x;
        int v = p[0];
        s += v;
        if (y) {
            sv += (unsigned) abs(v - (int) p[-as]);
        }
        if (x) {
            sh += (unsigned) abs(v - (int) p[-1]);
        }
    }
    Grad g;
    g.sh = wave_sum(sh);
    g.sv = wave_sum(sv);
    g.sum = wave_sum(s);
    return g;
}

__device__ __forceinline__ int ws_abs_dev(const uint8_t *a, int as, int w, int h, int mean)
{
    int lane = threadIdx.x & 63, v = 0;
    const RowSplit split(w);
    for (int idx = lane; idx < w * h; idx += 64) {
        int x, y;
        split(idx, x, y);
        v += abs((int) a[(ptrdiff_t) y * as + x] - mean);
    }
    return wave_sum(v);
}

__device__ __forceinline__ int ws_block_sum(const uint8_t *a, int as, int w, int h)
{
    int lane = threadIdx.x & 63, v = 0;
    const RowSplit split(w);
    for (int idx = lane; idx < w * h; idx += 64) {
        int x, y;
        split(idx, x, y);
        v += a[(ptrdiff_t) y * as + x];
    }
    return wave_sum(v);
}

__device__ __forceinline__ int ws_block_avg(const uint8_t *a, int as, int w, int h) { return ws_block_sum(a, as, w, h) / (w * h); }

__device__ __forceinline__ unsigned ws_block_tex(const uint8_t *a, int as, int w, int h)
{
    Grad g = ws_gradients(a, as, w, h);
    return max(g.sh, g.sv);
}

__device__ __forceinline__ int ws_block_var(const uint8_t *a, int as, int w, int h, unsigned &avg)
{
    int s = ws_block_avg(a, as, w, h);
    avg = (unsigned) s;
    return ws_abs_dev(a, as, w, h, s);
}

__device__ __forceinline__ int ws_block_detail(const uint8_t *a, int as, int w, int h, unsigned &avg) // hme.c:546
{
    Grad g = ws_gradients(a, as, w, h);
    int s = g.sum / (w * h);
    avg = (unsigned) s;
    int var = ws_abs_dev(a, as, w, h, s) >> 1;
    int tex = (int) (max(g.sh, g.sv) - (unsigned) var);
    return var + max(tex, 0);
}

__device__ __forceinline__ int ws_quant_tex(const uint8_t *a, int as, int w, int h) // hme.c:586
{
    int lane = threadIdx.x & 63;
    unsigned sh = 0, sv = 0;
    const RowSplit split(w);
    for (int idx = lane; idx < w * h; idx += 64) {
        int x, y;
        split(idx, x, y);
        const uint8_t *p = a + (ptrdiff_t) y * as + x;
        int px = p[0] >> 4;
        int right = (x + 1 < w) ? (p[1] >> 4) : px;
        int up = y ? (p[-as] >> 4) : px;
        sh += (unsigned) ((px - right) * (px - right));
        sv += (unsigned) ((px - up) * (px - up));
    }
    sh = wave_sum(sh);
    sv = wave_sum(sv);
    return (int) (isqrt_u32(max(sh, sv)) / (unsigned) ((w + h + 1) >> 1));
}

// hist: 16 ints of LDS private to this wavefront
__device__ __forceinline__ unsigned ws_hist_var(const uint8_t *a, int as, int w, int h, int *hist) // hme.c:711
{
    int lane = threadIdx.x & 63;
    unsigned avg = (unsigned) ws_block_avg(a, as, w, h);
    if (avg == 0) {
        avg = 1;
    }
    unsigned q16 = (8u << 16) / avg;
    if (lane < 16) {
        hist[lane] = 0;
    }
    __syncthreads();
    const RowSplit split(w);
    for (int idx = lane; idx < w * h; idx += 64) {
        int x, y;
        split(idx, x, y);
        int hi = (int) ((unsigned) a[(ptrdiff_t) y * as + x] * q16 >> 16);
        atomicAdd(&hist[min(max(hi, 0), 15)], 1);
    }
    __syncthreads();
    avg = (unsigned) (w * h) / 16;
    unsigned var = 0;
    if (lane < 16) {
        unsigned d = (unsigned) hist[lane] - avg;
        var = d * d;
    }
    var = wave_sum(var);
    __syncthreads();
    return (var * 16 * 16) / (16u * (unsigned) (w * h * w * h));
}

__device__ __forceinline__ int ws_peaks(const uint8_t *a, int as, int w, int h, int bavg, int *hist) // hme.c:624
{
    int lane = threadIdx.x & 63;
    int avg = bavg ? bavg : 1;
    int q16 = (8 << 16) / avg;
    if (lane < 16) {
        hist[lane] = 0;
    }
    __syncthreads();
    int w2 = w / 2, h2 = h / 2;
    const RowSplit split(w2);
    for (int idx = lane; idx < w2 * h2; idx += 64) {
        int x, y;
        split(idx, x, y);
        const uint8_t *p = a + (ptrdiff_t) (2 * y) * as + 2 * x;
        int ds = (int) ((unsigned) (p[0] + p[1] + p[as] + p[as + 1] + 2) >> 2);
        int hi = ds * q16 >> 16;
        atomicAdd(&hist[min(hi, 15)], 1);
    }
    __syncthreads();
    int c = lane < 16 ? hist[lane] : 0;
    int total = wave_sum(c);
    int maxv = c;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        maxv = max(maxv, __shfl_xor(maxv, m, 64));
    }
    maxv = __builtin_amdgcn_readlane(maxv, 0) >> 2; // lanes 0..15 hold the maximum of the 16 bins
    int left = __shfl_up(c, 1, 64), right = __shfl_down(c, 1, 64);
    int pk = 0;
    if (lane < 16) {
        pk = 1;
        if (lane > 0) {
            pk &= c > left;
        }
        if (lane < 15) {
            pk &= c > right;
        }
        pk &= (c > maxv) || (c > total / 16);
    }
    int np = wave_sum(pk);
    __syncthreads();
    return np;
}

struct ChromaPsy {
    bool nature, hifreq, greyish, skinnish;
};

__device__ __forceinline__ ChromaPsy chroma_analysis(int y, int u, int v) // hme.c:69
{
    ChromaPsy c;
    c.nature = u < 128 && v < 160;
    c.greyish = abs(u - 128) < 8 && abs(v - 128) < 8;
    c.skinnish = y > 80 && y < 230 && abs(u - 108) < 24 && abs(v - 148) < 24;
    c.hifreq = u > 160 && !c.greyish && !c.skinnish;
    return c;
}

} // namespace dsv2
