// hme_fast.h -- latency-optimised evaluation of one motion-estimation block by one wavefront.
// Included by hme.hip (uses its primitives).  Same results as hme_block(); what changes is how the
// work is laid out on the 64 lanes:
//   * the block's 2x2 quads are register-resident (lane = 8*qj + qi owns quad (qi,qj) of the
//     source block, of each reference candidate, of the chroma blocks ...), so a metric needs
//     four byte loads per lane and no loop;
//   * independent sums are reduced TOGETHER: reduceN<16> folds 16 per-lane partial sums across
//     the wave with 17 cross-lane exchanges (a transpose-style butterfly: each step halves the
//     number of live values per lane) instead of 16 x 6 -- candidates are scored 16 at a time, a
//     refinement round scores the whole 3x3 neighbourhood at once, and the mode decision gathers
//     its ~60 block / sub-block sums into six such rounds;
//   * candidate bookkeeping (gather, qpel->fpel rounding, level scaling, first-occurrence
//     de-duplication, cost, arg-min with first-wins ties) is lane-parallel: lane k owns candidate
//     k; order-preserving compaction uses ballots.
// Preconditions (else hme_block() is used): 16x16 blocks, 4:2:0; clipped block sizes even at levels 0 and 1 (any size
// at the squared-error levels above), at level 0 additionally block width/height multiples of 8.
#pragma once

// a 2x2 pixel quad packed in one register: byte 0 = top-left, 1 = top-right, 2 = bottom-left,
// 3 = bottom-right, so the packed-byte instructions (v_sad_u8, v_dot4_u32_u8) do the per-quad
// arithmetic of METR_CALC (hme.c:126) in a handful of operations
struct Quad {
    uint32_t w;
    __device__ __forceinline__ int p1() const { return (int) (w & 0xff); }
    __device__ __forceinline__ int p2() const { return (int) ((w >> 8) & 0xff); }
    __device__ __forceinline__ int p3() const { return (int) ((w >> 16) & 0xff); }
    __device__ __forceinline__ int p4() const { return (int) (w >> 24); }
};

struct __attribute__((packed)) U16u { // possibly unaligned 16-bit load (one global_load_ushort)
    uint16_t v;
};

// the two 16-bit loads are explicit global-memory accesses (a generic pointer would make them flat loads,
// which also tie up the LDS counter)
typedef const __attribute__((address_space(1))) uint8_t *gbytes_t;
typedef const __attribute__((address_space(1))) U16u *gu16_t;

// Branch-free on purpose: a load inside a conditional block ends that block with a wait for it, which turns
// a batch of independent loads into as many memory round trips.  Lanes outside the block read its first quad.
__device__ __forceinline__ Quad ldq(const uint8_t *blk, int stride, int qi, int qj, bool act)
{
    gbytes_t g = (gbytes_t) blk;
    unsigned off = act ? (unsigned) ((2 * qj) * stride + 2 * qi) : 0u;
    uint32_t top = ((gu16_t) (g + off))->v, bot = ((gu16_t) (g + off + (unsigned) stride))->v;
    Quad q;
    q.w = act ? (top | (bot << 16)) : 0u;
    return q;
}

// the same in two steps, for call sites that fence a group of loads off from their first use with
// __builtin_amdgcn_sched_barrier (the scheduler otherwise pairs each load with its wait)
struct QuadRaw {
    uint32_t top, bot;
};
__device__ __forceinline__ QuadRaw ldq_raw(const uint8_t *blk, int stride, int qi, int qj, bool act)
{
    gbytes_t g = (gbytes_t) blk;
    unsigned off = act ? (unsigned) ((2 * qj) * stride + 2 * qi) : 0u;
    QuadRaw r;
    r.top = ((gu16_t) (g + off))->v;
    r.bot = ((gu16_t) (g + off + (unsigned) stride))->v;
    return r;
}
__device__ __forceinline__ Quad ldq_finish(const QuadRaw &r, bool act)
{
    Quad q;
    q.w = act ? (r.top | (r.bot << 16)) : 0u;
    return q;
}

// one pixel, same rules
__device__ __forceinline__ int ldpx(const uint8_t *p, int off, bool act)
{
    gbytes_t g = (gbytes_t) p;
    int v = g[act ? off : 0];
    return act ? v : 0;
}

__device__ __forceinline__ Quad mkq(int s1, int s2, int s3, int s4)
{
    Quad q;
    q.w = (uint32_t) s1 | ((uint32_t) s2 << 8) | ((uint32_t) s3 << 16) | ((uint32_t) s4 << 24);
    return q;
}

__device__ __forceinline__ uint32_t sad4(uint32_t a, uint32_t b) { return __builtin_amdgcn_sad_u8(a, b, 0u); }
__device__ __forceinline__ uint32_t rot8(uint32_t a) { return (a >> 8) | (a << 24); } // (p1,p2,p3,p4) -> (p2,p3,p4,p1)
__device__ __forceinline__ uint32_t rep4(int v) { return (uint32_t) v * 0x01010101u; }

// METR_CALC (hme.c:126): UAVG4 of four absolute differences == (sad + 2) >> 2, so every squared term is at most 255
__device__ __forceinline__ unsigned qmetric(const Quad &a, const Quad &b, const Psy &psy)
{
    int se = (int) ((sad4(a.w, b.w) + 2) >> 2);
    int ta = (int) ((sad4(a.w, rot8(a.w)) + 2) >> 2), tb = (int) ((sad4(b.w, rot8(b.w)) + 2) >> 2);
    int s0 = (int) ((sad4(a.w, 0) + 2) >> 2), s1 = (int) ((sad4(b.w, 0) + 2) >> 2);
    return (unsigned) (sq24(se) << psy.err_weight) + (unsigned) (sq24(ta - tb) << psy.tex_weight) +
           (unsigned) (sq24(s0 - s1) << psy.avg_weight);
}

__device__ __forceinline__ unsigned qsse(const Quad &a, const Quad &b) // sum of squared differences of the 4 bytes
{
    unsigned aa = __builtin_amdgcn_udot4(a.w, a.w, 0u, false), bb = __builtin_amdgcn_udot4(b.w, b.w, 0u, false);
    unsigned ab = __builtin_amdgcn_udot4(a.w, b.w, 0u, false);
    return aa + bb - 2u * ab;
}

// Folds N (power of two <= 16) per-lane partial sums across the wavefront.  On return lane L
// holds the total of entry (L >> (6 - log2 N)); fetch entry e with bcastN<N>(r, e).
// one transposing step at lane distance B over the first 2*HALF entries: lanes with bit B clear end up
// with a[lane] + a[lane ^ B] of entry t, the others with the same of entry t + HALF
template <int B, int HALF, int N> __device__ __forceinline__ void fold_pairs(int (&v)[N])
{
#pragma unroll
    for (int t = 0; t < HALF; t++) {
        int a = v[t], b = v[t + HALF];
        if (B == 32) {
            uint2v_t r = __builtin_amdgcn_permlane32_swap((unsigned) a, (unsigned) b, false, false);
            v[t] = (int) (r[0] + r[1]);
        } else if (B == 16) {
            uint2v_t r = __builtin_amdgcn_permlane16_swap((unsigned) a, (unsigned) b, false, false);
            v[t] = (int) (r[0] + r[1]);
        } else {
            bool sel = (threadIdx.x & B) != 0;
            int mine = sel ? b : a, other = sel ? a : b;
            v[t] = mine + lane_xor<B>(other);
        }
    }
}

template <int N> __device__ __forceinline__ int reduceN(int (&v)[N])
{
    static_assert(N == 2 || N == 4 || N == 8 || N == 16, "reduceN: N must be 2, 4, 8 or 16");
    // transposing butterfly: each step halves the number of live entries
    fold_pairs<32, N / 2, N>(v);
    if constexpr (N >= 4) {
        fold_pairs<16, N / 4, N>(v);
    }
    if constexpr (N >= 8) {
        fold_pairs<8, N / 8, N>(v);
    }
    if constexpr (N >= 16) {
        fold_pairs<4, 1, N>(v);
    }
    int r = v[0];
    // plain all-reduce over the remaining lane bits
    if constexpr (N < 4) {
        r = fold_xor<16>(r);
    }
    if constexpr (N < 8) {
        r = fold_xor<8>(r);
    }
    if constexpr (N < 16) {
        r = fold_xor<4>(r);
    }
    r = fold_xor<2>(r);
    r = fold_xor<1>(r);
    return r;
}

// entry e (wave-uniform e): a scalar read of the owning lane -- the result lives in an SGPR
template <int N> __device__ __forceinline__ int bcastN(int r, int e)
{
    constexpr int sh = N == 16 ? 2 : (N == 8 ? 3 : (N == 4 ? 4 : (N == 2 ? 5 : 6)));
    return __builtin_amdgcn_readlane(r, e << sh);
}

// entry e chosen per lane
template <int N> __device__ __forceinline__ int bcastL(int r, int e)
{
    constexpr int sh = N == 16 ? 2 : (N == 8 ? 3 : (N == 4 ? 4 : (N == 2 ? 5 : 6)));
    return __shfl(r, e << sh, 64);
}

__device__ __forceinline__ unsigned wave_min_u(unsigned v)
{
    v = min(v, (unsigned) lane_xor<32>((int) v));
    v = min(v, (unsigned) lane_xor<16>((int) v));
    v = min(v, (unsigned) lane_xor<8>((int) v));
    v = min(v, (unsigned) lane_xor<4>((int) v));
    v = min(v, (unsigned) lane_xor<2>((int) v));
    v = min(v, (unsigned) lane_xor<1>((int) v));
    return (unsigned) __builtin_amdgcn_readfirstlane((int) v);
}

// n / d for n >= 0, d > 0: block areas are powers of two for every interior block, and an integer
// division costs ~30 vector instructions on this machine
__device__ __forceinline__ int div_nn(int n, int d)
{
    if ((d & (d - 1)) == 0) {
        return n >> (31 - __clz(d));
    }
    return n / d;
}
__device__ __forceinline__ unsigned div_nn(unsigned n, unsigned d)
{
    if ((d & (d - 1)) == 0) {
        return n >> (31 - __clz((int) d));
    }
    return n / d;
}

// n / d (d > 0) without the ~35-instruction integer sequence: below 2^20 both operands are exact in single precision and the
// quotient estimated with v_rcp_f32 (1 ulp) is within one of the true one, which a multiply-back settles; larger operands
// (a choice made for the whole wavefront: the callers' operands are wave-uniform) take the integer divide.
__device__ __forceinline__ unsigned udiv_fast(unsigned n, unsigned d)
{
    unsigned est = (unsigned) ((float) n * __builtin_amdgcn_rcpf((float) d));
    const int r = (int) (n - est * d);
    est = r < 0 ? est - 1u : (r >= (int) d ? est + 1u : est);
    const bool big = (n | d) >= (1u << 20);
    if (__builtin_expect(__any(big), 0)) {
        est = big ? n / d : est;
    }
    return est;
}
// truncating signed / positive
__device__ __forceinline__ int sdiv_fast(int n, int d)
{
    const unsigned q = udiv_fast((unsigned) abs(n), (unsigned) d);
    return n < 0 ? -(int) q : (int) q;
}

// The lane number as a value the compiler cannot trace back to threadIdx: every routine asks for its own.  Lane predicates
// ("lane == 3", "lane < nq") are loop-invariant 64-bit masks; traced, the compiler hoists all of them out of the block loop
// into scalar registers it does not have, and every use becomes two v_readlane reloads of a spilled pair -- recomputing the
// compare where it is used is one instruction.
__device__ __forceinline__ int hme_lane()
{
    int l = (int) (threadIdx.x & 63);
    asm volatile("" : "+v"(l));
    return l;
}

struct FastLds {
    int hist[16];
    SubpelLds sp;
#ifdef DSV2_HME_PROF
    unsigned long long prof_t, prof_acc[32];
    int prof_on;
#endif
};

// phase clock of a debugging build (-DDSV2_HME_PROF): shader-clock ticks per phase of the level-0 search,
// summed over all wavefronts into g_hme_prof[]
#ifdef DSV2_HME_PROF
__device__ unsigned long long g_hme_prof[32];
#define HME_MARK(S, k)                                                                                                \
    do {                                                                                                              \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                        \
        if ((threadIdx.x & 63) == 0 && (S).prof_on) {                                                                 \
            (S).prof_acc[k] += t_ - (S).prof_t;                                                                       \
            (S).prof_t = t_;                                                                                          \
        }                                                                                                             \
    } while (0)
#define HME_COUNT(S, k, n)                                                                                            \
    do {                                                                                                              \
        if ((threadIdx.x & 63) == 0 && (S).prof_on) {                                                                 \
            (S).prof_acc[k] += (unsigned long long) (n);                                                              \
        }                                                                                                             \
    } while (0)
#else
#define HME_MARK(S, k)                                                                                                \
    do {                                                                                                              \
    } while (0)
#define HME_COUNT(S, k, n)                                                                                            \
    do {                                                                                                              \
    } while (0)
#endif

// sums needed by block_detail (hme.c:546) from a register-resident block: pixel sum and the
// horizontal / vertical first-difference sums; partials only, the caller reduces them
__device__ __forceinline__ void quad_grad_partials(const Quad &q, bool act, int qi, int qj, int qi0, int qj0, int &sum, int &sh, int &sv)
{
    // (qi0, qj0): first quad column / row of the (sub-)block this lane's quad belongs to
    int l2 = __shfl_up(q.p2(), 1, 64), l4 = __shfl_up(q.p4(), 1, 64);
    int u3 = __shfl_up(q.p3(), 8, 64), u4 = __shfl_up(q.p4(), 8, 64);
    sum = sh = sv = 0;
    if (act) {
        sum = q.p1() + q.p2() + q.p3() + q.p4();
        sh = abs(q.p2() - q.p1()) + abs(q.p4() - q.p3());
        sv = abs(q.p3() - q.p1()) + abs(q.p4() - q.p2());
        if (qi > qi0) {
            sh += abs(q.p1() - l2) + abs(q.p3() - l4);
        }
        if (qj > qj0) {
            sv += abs(q.p1() - u3) + abs(q.p2() - u4);
        }
    }
}

// the same on a quad grid W lanes wide (W = 4: the 8 x 16 chroma block of 4:2:2 -- 4 x 8 quads on 32 lanes)
template <int W> __device__ __forceinline__ void quad_grad_partials_w(const Quad &q, bool act, int qi, int qj, int &sum, int &sh, int &sv)
{
    int l2 = __shfl_up(q.p2(), 1, 64), l4 = __shfl_up(q.p4(), 1, 64);
    int u3 = __shfl_up(q.p3(), W, 64), u4 = __shfl_up(q.p4(), W, 64);
    sum = sh = sv = 0;
    if (act) {
        sum = q.p1() + q.p2() + q.p3() + q.p4();
        sh = abs(q.p2() - q.p1()) + abs(q.p4() - q.p3());
        sv = abs(q.p3() - q.p1()) + abs(q.p4() - q.p2());
        if (qi > 0) {
            sh += abs(q.p1() - l2) + abs(q.p3() - l4);
        }
        if (qj > 0) {
            sv += abs(q.p1() - u3) + abs(q.p2() - u4);
        }
    }
}

__device__ __forceinline__ int quad_absdev(const Quad &q, bool act, int mean)
{
    return act ? (int) sad4(q.w, rep4(mean)) : 0;
}

// hist_var / quant_tex / peaks of the source block from its register-resident quads (hme.c:586-749)
__device__ int src_hist_var(const Quad &q, bool act, int sum, int w, int h, int *hist)
{
    int lane = hme_lane();
    unsigned avg = (unsigned) div_nn(sum, w * h);
    if (avg == 0) {
        avg = 1;
    }
    unsigned q16 = udiv_fast(8u << 16, avg);
    if (lane < 16) {
        hist[lane] = 0;
    }
    __syncthreads();
    if (act) {
        atomicAdd(&hist[min((int) ((unsigned) q.p1() * q16 >> 16), 15)], 1);
        atomicAdd(&hist[min((int) ((unsigned) q.p2() * q16 >> 16), 15)], 1);
        atomicAdd(&hist[min((int) ((unsigned) q.p3() * q16 >> 16), 15)], 1);
        atomicAdd(&hist[min((int) ((unsigned) q.p4() * q16 >> 16), 15)], 1);
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
    return (int) div_nn(var * 16 * 16, 16u * (unsigned) (w * h * w * h));
}

__device__ int src_quant_tex(const Quad &q, bool act, int qi, int qj, int qw, int w, int h)
{
    int a1 = q.p1() >> 4, a2 = q.p2() >> 4, a3 = q.p3() >> 4, a4 = q.p4() >> 4;
    int r1 = __shfl_down(a1, 1, 64), r3 = __shfl_down(a3, 1, 64); // right neighbour quad's left column
    int u3 = __shfl_up(a3, 8, 64), u4 = __shfl_up(a4, 8, 64);     // upper neighbour quad's bottom row
    unsigned sh = 0, sv = 0;
    if (act) {
        int e2 = (qi + 1 < qw) ? r1 : a2, e4 = (qi + 1 < qw) ? r3 : a4; // past the last column: the pixel itself
        sh = (unsigned) ((a1 - a2) * (a1 - a2) + (a2 - e2) * (a2 - e2) + (a3 - a4) * (a3 - a4) + (a4 - e4) * (a4 - e4));
        int t1 = qj > 0 ? u3 : a1, t2 = qj > 0 ? u4 : a2;
        sv = (unsigned) ((a1 - t1) * (a1 - t1) + (a2 - t2) * (a2 - t2) + (a3 - a1) * (a3 - a1) + (a4 - a2) * (a4 - a2));
    }
    sh = wave_sum(sh);
    sv = wave_sum(sv);
    return (int) div_nn(isqrt_u32(max(sh, sv)), (unsigned) ((w + h + 1) >> 1));
}

__device__ int src_peaks(const Quad &q, bool act, int bavg, int *hist)
{
    int lane = hme_lane();
    int avg = bavg ? bavg : 1;
    int q16 = (int) udiv_fast(8u << 16, (unsigned) avg);
    if (lane < 16) {
        hist[lane] = 0;
    }
    __syncthreads();
    if (act) {
        int ds = (int) ((unsigned) (q.p1() + q.p2() + q.p3() + q.p4() + 2) >> 2);
        atomicAdd(&hist[min(ds * q16 >> 16, 15)], 1);
    }
    __syncthreads();
    int c = lane < 16 ? hist[lane] : 0;
    int total = wave_sum(c);
    int maxv = c;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        maxv = max(maxv, __shfl_xor(maxv, m, 64));
    }
    maxv = __builtin_amdgcn_readlane(maxv, 0) >> 2;
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

// ---- what the search needs to know about a SOURCE block before it looks at a single candidate (hme.c:1392-1441) --------
// A function of the block's own pixels and of the frame's quantiser alone: not of any neighbour, not of the reference.
// The row pipeline of the search is serial along a row and down the rows; this is not, so the batched driver works it out
// for every block of levels 0 and 1 in a launch of its own (k_hme_src_stats_b: one wavefront per block, no order) and a
// block of the search starts from 12 bytes instead of ~600 instructions and two LDS histograms.
struct SrcStats {
    int bias_raw;     // motion_bias before it is divided by the global motion (known only when the level above is done)
    unsigned var_src; // block_detail of the source
    unsigned avg_src;
    unsigned zoscore; // pre-pass only: the metric against the co-located block of the previous SOURCE picture ("good enough" test, hme.c:1559)
};

// the metric's weights for a source block of this detail (hme.c:1431-1441)
__device__ __forceinline__ Psy psy_of_source(unsigned var_src, int bw, int bh, int quant)
{
    Psy psy = var_src <= (unsigned) (8 * bw * bh * quant >> 9) ? Psy{2, 1, 2} : Psy{1, 2, 1};
    if (var_src > (unsigned) (24 * bw * bh)) {
        psy.avg_weight = 0;
    }
    return psy;
}

__device__ __forceinline__ SrcStats source_analysis(const Quad &a, bool act, int qi, int qj, int qw, int bw, int bh, int quant, int *hist)
{
    SrcStats st;
    int ps, ph, pv;
    quad_grad_partials(a, act, qi, qj, 0, 0, ps, ph, pv);
    int v4[4] = {ps, ph, pv, 0};
    int r = reduceN<4>(v4);
    int sum = bcastN<4>(r, 0);
    unsigned sh = (unsigned) bcastN<4>(r, 1), sv = (unsigned) bcastN<4>(r, 2);
    int mean = div_nn(sum, bw * bh);
    st.avg_src = (unsigned) mean;
    int var = wave_sum(quad_absdev(a, act, mean)) >> 1;
    int tex = (int) (max(sh, sv) - (unsigned) var);
    st.var_src = (unsigned) (var + max(tex, 0));
    int tvar = (int) (st.var_src + SQR(st.var_src >> 10));
    tvar = div_nn(8 * tvar * quant >> 9, bw * bh);
    st.bias_raw = 16 * 16;
    if (tvar) {
        int hvar = src_hist_var(a, act, sum, bw, bh, hist);
        int qtex = src_quant_tex(a, act, qi, qj, qw, bw, bh);
        int npeaks = src_peaks(a, act, (int) st.avg_src, hist);
        st.bias_raw += tvar * (hvar - qtex) * npeaks;
    }
    return st;
}

// ---- scoring rounds: up to 16 displacement vectors against the register-resident source block, ONE load round -----------------
// The vectors of a round are wave-uniform (dx, dy) pairs (entries beyond `cnt`: the zero vector); on return lane L holds the raw
// wave total of entry L & (NR - 1), NR = 4 / 8 / 16 (SSE for level > 1, psy accumulator otherwise); a vector whose block leaves
// the padded plane gives 0.
// Addressing: a block of the padded plane at any valid position is (plane origin) + a non-negative 32-bit offset, so the loads use
// ONE wave-uniform base -- the plane's first padded pixel -- plus a vector offset that the hardware adds (scalar base + vector
// offset form) instead of the wavefront (64-bit scalar arithmetic per vector and a 64-bit vector add per load).
// safe: every vector of the round is known to lie within +-31 pixels.  The planes carry a 32-pixel border, so such a block is
// inside the padded plane WHEREVER the block sits: no per-vector bounds test.
// smask: bytes of a quad that belong to the block (all four, except in the half quads of an odd last row / column at the
// squared-error levels)
template <int NT> struct VecSet {
    int dx[NT], dy[NT];
};

// A lane's share of the source block.  NQ = 1: a 16 x 16 block, the lane (qi, qj) owns its 2x2 quad (qi, qj).  NQ = 4: a
// 32 x 32 block (dsv_encoder.c:1203-1211: every picture of 2160p and up) as FOUR 16 x 16 quadrants -- quadrant k at (16 (k & 1),
// 16 (k >> 1)) -- of which the lane owns quad (qi, qj) each (NQ = 2: the upper two of them, a 32 x 16 block): every per-quad primitive of the 16 x 16 routine is used as it is,
// once per quadrant, with the sums combined before the metric's square root; the mode decision's four sub-block sums are the
// quadrants' own.  act[k]: the quad takes part in the level's metric (a clipped block's quads beyond its edge do not);
// smask[k]: bytes of the quad inside the block (all four, except in the half quads of an odd last row / column at the
// squared-error levels).
template <int NQ> struct SrcBlk {
    Quad a[NQ];
    bool act[NQ];
    uint32_t smask[NQ];
    int bx, by, bw, bh, qi, qj;
};
// byte offset of quadrant k's origin from the block's, for a plane of row stride `stride`
__device__ __forceinline__ unsigned quadrant_off(int k, int stride) { return (unsigned) (16 * (k & 1) + 16 * (k >> 1) * stride); }
// nominal block size of a SrcBlk<NQ>: 16 x 16 (one quadrant), 32 x 16 (two, side by side: the 32 x 16 blocks dsv_encoder.c:1203-1211
// gives wide pictures -- 1920 x 800, 2560 x 1080), 32 x 32 (four)
template <int NQ> struct BlkDim {
    static_assert(NQ == 1 || NQ == 2 || NQ == 4, "a block is one, two or four 16 x 16 quadrants");
    static constexpr int W = NQ == 1 ? 16 : 32, H = NQ == 4 ? 32 : 16;
};

template <int NT, int NQ>
__device__ __forceinline__ unsigned score_set(const VecSet<NT> &vs, int cnt, bool safe, const DPlane &ref, const SrcBlk<NQ> &B, int level, const Psy &psy)
{
    const int bx = B.bx, by = B.by, bw = B.bw, bh = B.bh;
    int ok[NT];
    if (safe) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            ok[t] = t < cnt;
        }
    } else {
        // valid <=> -32 <= bx + dx and bx + dx + bw < w + 32 (invalid_block, pad 0): one unsigned compare per axis
        const unsigned lx = (unsigned) (ref.w + 2 * kBorder - bw), ly = (unsigned) (ref.h + 2 * kBorder - bh);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            ok[t] = t < cnt && (unsigned) (bx + kBorder + vs.dx[t]) < lx && (unsigned) (by + kBorder + vs.dy[t]) < ly;
        }
    }
    // all loads first, back to back (one round trip); a vector that may not be read is replaced by the zero vector, whose block
    // always lies inside the frame
    gbytes_t base = (gbytes_t) uni_ptr(ref.data - (ptrdiff_t) kBorder * ref.stride - kBorder);
    const unsigned o0 = (unsigned) ((by + kBorder) * ref.stride + bx + kBorder);
    unsigned loff[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        loff[k] = B.act[k] ? (unsigned) ((2 * B.qj) * ref.stride + 2 * B.qi) + quadrant_off(k, ref.stride) : 0u;
    }
    Quad b[NT][NQ];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const unsigned o = o0 + (unsigned) (ok[t] ? vs.dy[t] * ref.stride + vs.dx[t] : 0);
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            const uint32_t top = ((gu16_t) (base + (loff[k] + o)))->v, bot = ((gu16_t) (base + (loff[k] + o + (unsigned) ref.stride)))->v;
            b[t][k].w = top | (bot << 16);
        }
    }
    constexpr int NR = NT <= 4 ? 4 : (NT <= 8 ? 8 : 16); // width of the joint reduction
    int v[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) {
        v[t] = 0;
        if (t < NT && ok[t]) { // wave-uniform: the arithmetic of an absent vector is skipped, its (dummy) load was not
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                b[t][k].w &= B.smask[k];
                int m = (int) (level > 1 ? qsse(B.a[k], b[t][k]) : qmetric(B.a[k], b[t][k], psy));
                v[t] += B.act[k] ? m : 0;
            }
        }
    }
    int r = reduceN<NR>(v);
    return (unsigned) bcastL<NR>(r, threadIdx.x & (NR - 1));
}

// the next NT vectors of the lanes in `rest` (lowest lanes first; key = x | y << 16, int16 each), popped off it
template <int NT> __device__ __forceinline__ VecSet<NT> pop_vecs(unsigned long long &rest, int key)
{
    VecSet<NT> vs;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int k = 0;
        if (rest) {
            k = __builtin_amdgcn_readlane(key, __ffsll((long long) rest) - 1);
            rest &= rest - 1;
        }
        vs.dx[t] = (int) (int16_t) (k & 0xffff);
        vs.dy[t] = k >> 16;
    }
    return vs;
}

// raw scores of the vectors held (as keys) by the lanes of `mask`, returned on those lanes: four vectors a load round (after
// de-duplication a list has ~3.5 entries; the rare longer one takes further rounds of the same code)
template <int NQ>
__device__ __forceinline__ unsigned score_lanes(unsigned long long mask, int key, const DPlane &ref, const SrcBlk<NQ> &B, int level, const Psy &psy)
{
    const int lane = hme_lane();
    const int n = __popcll(mask);
    const int widx = __popcll(mask & ((1ull << lane) - 1)); // this lane's place among them
    const int mx = (int) (int16_t) (key & 0xffff), my = key >> 16;
    const bool mine = ((mask >> lane) & 1ull) != 0;
    const bool safe = !__any(mine && ((unsigned) (mx + 31) > 62u || (unsigned) (my + 31) > 62u));
    unsigned long long rest = mask;
    unsigned raw = 0;
    for (int first = 0; first < n; first += 4) {
        const VecSet<4> vs = pop_vecs<4>(rest, key);
        const unsigned r = (unsigned) __shfl((int) score_set<4, NQ>(vs, min(4, n - first), safe, ref, B, level, psy), widx & 3, 64);
        if ((widx >> 2) == (first >> 2)) {
            raw = r;
        }
    }
    return raw;
}

// psy accumulator of one 2x2 quad pair for the three predictions compared by err_intra (hme.c:839)
__device__ __forceinline__ void quad_err_intra(const Quad &a, const Quad &b, int avg_sb, int dc, int ratio, unsigned &inter, unsigned &isb,
                                               unsigned &isrc)
{
    const Psy psy = {0, 1, 2};
    int s0 = (int) ((sad4(a.w, 0) + 2) >> 2), s1 = (int) ((sad4(b.w, 0) + 2) >> 2);
    int ae = (int) ((sad4(a.w, b.w) + 2) >> 2);
    int ta = (int) ((sad4(a.w, rot8(a.w)) + 2) >> 2), tb = (int) ((sad4(b.w, rot8(b.w)) + 2) >> 2);
    inter = (unsigned) (sq24(ae) * ratio >> (5 - psy.err_weight)) + (unsigned) (sq24(ta - tb) << psy.tex_weight) +
            (unsigned) (sq24(s0 - s1) << psy.avg_weight);
    ae = (int) ((sad4(a.w, rep4(avg_sb)) + 2) >> 2);
    isb = (unsigned) (sq24(ae) << psy.err_weight) + (unsigned) (sq24(ta) << psy.tex_weight) + (unsigned) (sq24(s0 - avg_sb) << (psy.avg_weight + 1));
    ae = (int) ((sad4(a.w, rep4(dc)) + 2) >> 2);
    isrc = (unsigned) (sq24(ae) << psy.err_weight) + (unsigned) (sq24(ta) << psy.tex_weight) + (unsigned) (sq24(s0 - dc) << (psy.avg_weight + 1));
}

// ---- sub-pel search (hme.c:1051), in two halves ----------------------------------------------------------------------------
// subpel_probes(): everything that reads pixels -- the squared errors of the four full-pel neighbours, the 34x34 half-pel image,
// the seven probes' metrics.  A function of the block, the reference and the full-pel centre alone: for the search around the
// parent average (known as soon as the level above is done) it runs in the unordered pre-pass (k_hme_l0_pre_b), not in the row
// pipeline.  subpel_decide(): the comparison of the probes' scores + vector costs with the best full-pel score -- needs the cost
// predictor (the block's same-level neighbours) and is a few dozen instructions.
// Probe directions: (pri, sec) in {-1, 0, 1}^2 each, packed two bits per component (value + 1): pri0 | pri1 << 2 | sec0 << 4 | sec1 << 6.
struct SubpelLoads {
    QuadRaw b4[4];
    QuadRaw awr;
    HpelWin hw;
};
template <class Ctx>
__device__ __forceinline__ SubpelLoads subpel_issue_loads(const Ctx &c, int fpelx, int fpely, int bx, int by, int bw, int bh, int qi, int qj, bool act)
{
    const DPlane &src = c.src[0], &ref = c.ref[0];
    const int dxs[4] = {1, -1, 0, 0}, dys[4] = {0, 0, 1, -1};
    SubpelLoads L;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        L.b4[n] = ldq_raw(at(ref, bx + fpelx + dxs[n], by + fpely + dys[n]), ref.stride, qi, qj, act);
    }
    int xx = bx + ((bw >> 1) - 8), yy = by + ((bh >> 1) - 8);
    L.awr = ldq_raw(at(src, xx, yy), src.stride, qi, qj, true); // the centred 16x16 source window
    L.hw = load_hpel_window(at(ref, xx + fpelx - 1, yy + fpely - 1), ref.stride);
    return L;
}

// the probe offsets (quarter-pel) of probe n for packed directions `dirs` (all wave-uniform)
__device__ __forceinline__ void subpel_probe_offset(unsigned dirs, int n, int &t0, int &t1)
{
    const int pri0 = (int) (dirs & 3u) - 1, pri1 = (int) ((dirs >> 2) & 3u) - 1, sec0 = (int) ((dirs >> 4) & 3u) - 1, sec1 = (int) ((dirs >> 6) & 3u) - 1;
    const int diag0 = pri0 + sec0, diag1 = pri1 + sec1;
    t0 = t1 = 0;
    if (n == 6) {
        t0 = pri0 + diag0;
        t1 = pri1 + diag1;
    } else if (n < 6) {
        int v0 = (n >> 1) == 0 ? pri0 : ((n >> 1) == 1 ? sec0 : diag0);
        int v1 = (n >> 1) == 0 ? pri1 : ((n >> 1) == 1 ? sec1 : diag1);
        int hp = !(n & 1);
        t0 = v0 * (1 << hp);
        t1 = v1 * (1 << hp);
    }
}

// returns on every lane L the normalised metric (metric_return) of probe L & 7 (probe 7 does not exist: 0); dirs: see above
template <class Ctx>
__device__ __forceinline__ unsigned subpel_probes(const Ctx &c, FastLds &S, int fpelx, int fpely, int bx, int by, int bw, int bh, const Quad &a, bool act, int qi,
                                                  int qj, const Psy &psy, unsigned &dirs)
{
    const int lane = hme_lane();
    int v4[4];
    Quad aw;
    {
        SubpelLoads L = subpel_issue_loads(c, fpelx, fpely, bx, by, bw, bh, qi, qj, act);
        __builtin_amdgcn_sched_barrier(0); // all twelve loads are in flight before the first is waited for
        aw = ldq_finish(L.awr, true);
#pragma unroll
        for (int n = 0; n < 4; n++) {
            v4[n] = act ? (int) qsse(a, ldq_finish(L.b4[n], act)) : 0;
        }
        // park the 20x20 window in LDS now: the registers that hold it are free during the reduction below
        uint32_t *win32 = (uint32_t *) S.sp.win;
        win32[lane] = L.hw.d0;
        if (lane + 64 < 100) {
            win32[lane + 64] = L.hw.d1;
        }
    }
    int r4 = reduceN<4>(v4);
    unsigned quad0 = (unsigned) bcastN<4>(r4, 0), quad1 = (unsigned) bcastN<4>(r4, 1), quad2 = (unsigned) bcastN<4>(r4, 2),
             quad3 = (unsigned) bcastN<4>(r4, 3);
    __syncthreads();
    build_hpel_at<20>(S.sp, S.sp.win);

    int pri0 = 0, pri1 = -1, sec0 = -1, sec1 = 0;
    unsigned ms1 = quad1, ms2 = quad3;
    if (quad3 >= quad2) {
        pri1 = 1;
        ms2 = quad2;
    }
    if (quad1 >= quad0) {
        sec0 = 1;
        ms1 = quad0;
    }
    if (ms2 > ms1) {
        int t0 = sec0, t1 = sec1;
        sec0 = pri0, sec1 = pri1;
        pri0 = t0, pri1 = t1;
    }
    dirs = (unsigned) (pri0 + 1) | ((unsigned) (pri1 + 1) << 2) | ((unsigned) (sec0 + 1) << 4) | ((unsigned) (sec1 + 1) << 6);
    int v8[8];
#pragma unroll
    for (int n = 0; n < 8; n++) {
        int t0, t1;
        subpel_probe_offset(dirs, n, t0, t1);
        int X = 4 + 8 * qi + t0, Y = 4 + 8 * qj + t1;
        // every lane samples at the same quarter-pel phase (its X, Y differ from the probe offset by multiples of 4)
        const int ph = __builtin_amdgcn_readfirstlane((t0 & 1) | ((t1 & 1) << 1));
        Quad qs;
        qs.w = n < 7 ? qquad_ph(S.sp.h, X, Y, ph) : 0u;
        v8[n] = n < 7 ? (int) qmetric(aw, qs, psy) : 0;
    }
    int r8 = reduceN<8>(v8);
    unsigned acc = (unsigned) bcastL<8>(r8, lane & 7);
    __syncthreads(); // (the LDS image may be rebuilt by a second search)
    return metric_return(acc, 16, 16);
}

// ---- the same for a WHOLE 16x16 block, out of registers ------------------------------------------------------------------------
// The 34x34 half-pel image costs a wavefront ~250 vector instructions, 50 LDS accesses (unaligned: bank conflicts) and two
// barriers -- and of its 1 156 samples the seven probes read one quadrant's worth.  Which quadrant is known once the four
// neighbours' squared errors are: with s0 / s1 the signs of the better horizontal / vertical neighbour, every probe is made of
//   F (x, y)   H (x + s0/2, y)   V (x, y + s1/2)   C (x + s0/2, y + s1/2)            (hme.c:787-835, qsample: AVG2 / AVG4 of those)
// of the lane's own four pixels.  A lane therefore loads a 6-row x 8-byte PATCH of the reference around its quad -- rows y0 - 2 ..
// y0 + 3, columns x0 - 2 .. x0 + 5: six loads, which also hold the four neighbour blocks' quads (twelve loads and an LDS round
// before) -- and filters it in registers: the 5,5,-1,-1 taps along a row as two byte dot products, down the columns in packed
// 16-bit arithmetic (no intermediate exceeds 26 520).
typedef short hs16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hs16x2 hpk(uint32_t v) { return __builtin_bit_cast(hs16x2, v); }
__device__ __forceinline__ uint32_t hu32(hs16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ hs16x2 hspl(int v) { return (hs16x2){(short) v, (short) v}; }
// 5 (b + c) - (a + d), two columns at once
__device__ __forceinline__ hs16x2 hpf_pk(hs16x2 a, hs16x2 b, hs16x2 c, hs16x2 d) { return hspl(5) * (b + c) - (a + d); }
// clamp_u8((v + r) >> s) of both halves, as bytes 0 and 2
__device__ __forceinline__ uint32_t round_clamp_pk(hs16x2 v, int r, int sft)
{
    const hs16x2 t = (v + hspl(r)) >> hspl(sft);
    return hu32(__builtin_elementwise_min(__builtin_elementwise_max(t, hspl(0)), hspl(255)));
}
// bytes k, k + 1 (k = 0 .. 4, a constant) of the 8-byte row {lo, hi} in bytes 0, 1
__device__ __forceinline__ uint32_t row_bytes(uint32_t lo, uint32_t hi, int k) { return k == 0 ? lo : (k == 4 ? hi : __builtin_amdgcn_alignbit(hi, lo, 8 * k)); }
// the 2x2 quad whose upper row is {lo0, hi0}, lower row {lo1, hi1}, at byte offset k
__device__ __forceinline__ uint32_t patch_quad(uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1, int k)
{
    return __builtin_amdgcn_perm(row_bytes(lo1, hi1, k), row_bytes(lo0, hi0, k), 0x05040100u);
}
// (a + b + 1) >> 1 of four bytes at once
__device__ __forceinline__ uint32_t avg2_b4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
__device__ __forceinline__ uint32_t avg4_b4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t m = 0x00ff00ffu;
    const uint32_t e = ((a & m) + (b & m) + (c & m) + (d & m) + 0x00020002u) >> 2;
    const uint32_t o = (((a >> 8) & m) + ((b >> 8) & m) + ((c >> 8) & m) + ((d >> 8) & m) + 0x00020002u) >> 2;
    return (e & m) | ((o & m) << 8);
}

template <class Ctx>
__device__ __forceinline__ unsigned subpel_probes_patch(const Ctx &c, int fpelx, int fpely, int bx, int by, const Quad &a, int qi, int qj, const Psy &psy, unsigned &dirs)
{
    const int lane = hme_lane();
    const DPlane &ref = c.ref[0];
    typedef const __attribute__((address_space(1))) uint8_t *gb_t;
    uint32_t lo[6], hi[6];
    {
        gb_t g = (gb_t) at(ref, bx + fpelx - 2, by + fpely - 2);
        const unsigned off = (unsigned) ((2 * qj) * ref.stride + 2 * qi);
#pragma unroll
        for (int r = 0; r < 6; r++) {
            struct __attribute__((packed)) U2 {
                uint2v_t v;
            };
            const uint2v_t v = ((const __attribute__((address_space(1))) U2 *) (g + off + (unsigned) (r * ref.stride)))->v;
            lo[r] = v[0];
            hi[r] = v[1];
        }
    }
    // the four full-pel neighbours (hme.c:1075): (1, 0), (-1, 0), (0, 1), (0, -1); the lane's own quad sits at rows 2, 3, byte 2
    int v4[4];
    {
        Quad n;
        n.w = patch_quad(lo[2], hi[2], lo[3], hi[3], 3);
        v4[0] = (int) qsse(a, n);
        n.w = patch_quad(lo[2], hi[2], lo[3], hi[3], 1);
        v4[1] = (int) qsse(a, n);
        n.w = patch_quad(lo[3], hi[3], lo[4], hi[4], 2);
        v4[2] = (int) qsse(a, n);
        n.w = patch_quad(lo[1], hi[1], lo[2], hi[2], 2);
        v4[3] = (int) qsse(a, n);
    }
    int r4 = reduceN<4>(v4);
    unsigned quad0 = (unsigned) bcastN<4>(r4, 0), quad1 = (unsigned) bcastN<4>(r4, 1), quad2 = (unsigned) bcastN<4>(r4, 2),
             quad3 = (unsigned) bcastN<4>(r4, 3);
    int pri0 = 0, pri1 = -1, sec0 = -1, sec1 = 0;
    unsigned ms1 = quad1, ms2 = quad3;
    if (quad3 >= quad2) {
        pri1 = 1;
        ms2 = quad2;
    }
    if (quad1 >= quad0) {
        sec0 = 1;
        ms1 = quad0;
    }
    const int s0 = sec0, s1 = pri1; // the half-sample side of every probe: horizontally, vertically
    const bool swapped = ms2 > ms1; // the primary direction is the horizontal one
    if (swapped) {
        int t0 = sec0, t1 = sec1;
        sec0 = pri0, sec1 = pri1;
        pri0 = t0, pri1 = t1;
    }
    dirs = (unsigned) (pri0 + 1) | ((unsigned) (pri1 + 1) << 2) | ((unsigned) (sec0 + 1) << 4) | ((unsigned) (sec1 + 1) << 6);
    // the five rows the vertical taps of both quad rows reach: y0 - 1 .. y0 + 3 (s1 > 0) or y0 - 2 .. y0 + 2; the quad's own rows are T[q0], T[q0 + 1]
    uint32_t tl[5], th[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        tl[k] = s1 > 0 ? lo[k + 1] : lo[k];
        th[k] = s1 > 0 ? hi[k + 1] : hi[k];
    }
    // horizontal taps: for column x the dword x - 1 .. x + 2; the quad's columns are bytes 2, 3 and the half sample lies on the
    // s0 side: dword offsets 1, 2 (s0 > 0) or 0, 1 -- as one wave-uniform shift
    const unsigned sh = s0 > 0 ? 8u : 0u;
    hs16x2 hz[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t d0 = __builtin_amdgcn_alignbit(th[k], tl[k], sh), d1 = __builtin_amdgcn_alignbit(th[k], tl[k], sh + 8u);
        const int h0 = (int) __builtin_amdgcn_udot4(d0, 0x00050500u, 0u, false) - (int) __builtin_amdgcn_udot4(d0, 0x01000001u, 0u, false);
        const int h1 = (int) __builtin_amdgcn_udot4(d1, 0x00050500u, 0u, false) - (int) __builtin_amdgcn_udot4(d1, 0x01000001u, 0u, false);
        hz[k] = hpk(((uint32_t) h0 & 0xffffu) | ((uint32_t) h1 << 16));
    }
    // the quad's columns of the five rows, zero-extended to 16 bits
    hs16x2 e[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        e[k] = hpk(__builtin_amdgcn_perm(0u, tl[k], 0x0c030c02u));
    }
    // samples of the quad's upper / lower pixel row as bytes 0, 2; the rows themselves sit at T[1], T[2] (s1 > 0) or T[2], T[3]
    const uint32_t h_top = round_clamp_pk(s1 > 0 ? hz[1] : hz[2], 4, 3), h_bot = round_clamp_pk(s1 > 0 ? hz[2] : hz[3], 4, 3);
    const uint32_t v_top = round_clamp_pk(hpf_pk(e[0], e[1], e[2], e[3]), 4, 3), v_bot = round_clamp_pk(hpf_pk(e[1], e[2], e[3], e[4]), 4, 3);
    const uint32_t c_top = round_clamp_pk(hpf_pk(hz[0], hz[1], hz[2], hz[3]), 32, 6), c_bot = round_clamp_pk(hpf_pk(hz[1], hz[2], hz[3], hz[4]), 32, 6);
    const uint32_t Fq = patch_quad(lo[2], hi[2], lo[3], hi[3], 2);
    const uint32_t Hq = __builtin_amdgcn_perm(h_bot, h_top, 0x06040200u), Vq = __builtin_amdgcn_perm(v_bot, v_top, 0x06040200u),
                   Cq = __builtin_amdgcn_perm(c_bot, c_top, 0x06040200u);
    // the seven probes (hme.c:1113-1150): 2 pri, pri, 2 sec, sec, 2 diag, diag, pri + diag
    const uint32_t A = swapped ? Hq : Vq, B = swapped ? Vq : Hq;
    uint32_t q[7];
    q[0] = A;
    q[1] = avg2_b4(Fq, A);
    q[2] = B;
    q[3] = avg2_b4(Fq, B);
    q[4] = Cq;
    q[5] = avg4_b4(Fq, Hq, Vq, Cq);
    q[6] = avg2_b4(A, Cq);
    int v8[8];
#pragma unroll
    for (int n = 0; n < 7; n++) {
        Quad qs;
        qs.w = q[n];
        v8[n] = (int) qmetric(a, qs, psy);
    }
    v8[7] = 0;
    int r8 = reduceN<8>(v8);
    unsigned acc = (unsigned) bcastL<8>(r8, lane & 7);
    return metric_return(acc, 16, 16);
}

// mr: lane n (n < 7) holds the normalised metric of probe n
__device__ __forceinline__ unsigned subpel_decide(const CostCtx &cc, int effort, unsigned mr, unsigned dirs, int &sub_x, int &sub_y, int fpelx, int fpely,
                                                  unsigned best, int bw, int bh)
{
    const int lane = hme_lane();
    const unsigned yarea = (unsigned) (bw * bh);
    const int area_ratio = (int) div_nn(8u * 256u, yarea), iarea_ratio = (int) (8 * yarea / 256);
    best = best * (unsigned) area_ratio >> 3;
    int tx[7], ty[7], mtx = 0, mty = 0;
#pragma unroll
    for (int n = 0; n < 7; n++) {
        subpel_probe_offset(dirs, n, tx[n], ty[n]);
        if (lane == n) {
            mtx = tx[n];
            mty = ty[n];
        }
    }
    // lane n finishes probe n: vector cost
    unsigned sc = mr + (unsigned) mv_cost(cc, fpelx * 4 + mtx, fpely * 4 + mty, 0);
    int b0 = 0, b1 = 0;
#pragma unroll
    for (int n = 0; n <= 6; n++) {
        const int t0 = tx[n], t1 = ty[n];
        if (((t0 | t1) & 1) && effort < 8) {
            continue;
        }
        unsigned s = (unsigned) __builtin_amdgcn_readlane((int) sc, n);
        if (best > s) {
            best = s;
            b0 = t0;
            b1 = t1;
        }
    }
    sub_x = b0;
    sub_y = b1;
    return best * (unsigned) iarea_ratio >> 3;
}

// sub-pel search around full-pel vector (fpelx, fpely), both halves in place: the row pipeline's second search
template <class Ctx>
__device__ __forceinline__ unsigned subpixel_me_fast(const Ctx &c, FastLds &S, const CostCtx &cc, int &sub_x, int &sub_y, int fpelx, int fpely, unsigned best,
                                                     int bx, int by, int bw, int bh, const Quad &a, bool act, int qi, int qj, const Psy &psy)
{
    sub_x = sub_y = 0;
    if (best == 0) {
        return best;
    }
    HME_COUNT(S, 13, 1);
    unsigned dirs;
    // (a whole block -- bw, bh are compile-time 16 in the callers' FULL instantiation -- filters a register patch; a clipped one the LDS image)
    const unsigned mr = (bw == 16 && bh == 16) ? subpel_probes_patch(c, fpelx, fpely, bx, by, a, qi, qj, psy, dirs)
                                               : subpel_probes(c, S, fpelx, fpely, bx, by, bw, bh, a, act, qi, qj, psy, dirs);
    return subpel_decide(cc, c.effort, mr, dirs, sub_x, sub_y, fpelx, fpely, best, bw, bh);
}

// what the candidate load round already fetched and the level-0 tail needs again (wave-uniform)
struct NbPre {
    uint32_t l_all, l_flags, t_all, t_flags; // left / top neighbour of the same level: {x, y}, flags
    uint32_t colo;                           // co-located vector of the previous frame
    bool colo_ok;
};

// The four per-frame sums the level-0 blocks feed (hme.c:1825-1832: intra blocks, scene-change votes, eligible blocks, total
// error), carried by the wavefront along its block row and added to the stream's counters ONCE, at the row's end: as up to
// four atomics per block they sat in front of every block's drain-and-publish.
struct RowAcc {
    int intra = 0, ndiff = 0, elig = 0, err = 0;
    // the 8-byte head ({x, y}, flags) this wavefront stored for the previous block of its row: the next block's LEFT neighbour
    // (wave-uniform; read back from registers, not from memory: the store need not have landed)
    unsigned long long left_head = 0;
    bool have_left = false; // (false: the routine reads the left neighbour from memory -- launch per front, or behind a block of the general routine)
    bool failed = false;    // a top / top-left head stayed pending beyond the spin limit: the row gives up (hme_row reports it)
    // Two neighbouring blocks of a row (2 b, 2 b + 1) hang off the SAME parent block, so the parent level's half of the list --
    // the nine parent vectors' average, their inliers, the inliers' average (hme.c:1443-1470: six divisions, a square root, five
    // wave reductions) -- is worked out for the even block and kept for the odd one (par_key: the parent's block index, -1: none)
    int par_key = -1, par_lax = 0, par_lay = 0, par_nin = 0;
    bool par_open = false, par_inl = false; // (par_inl: per lane)
    __device__ __forceinline__ void flush(int *counters)
    {
        if (hme_lane() == 0) {
            if (intra) {
                atomicAdd(&counters[0], intra);
            }
            if (ndiff) {
                atomicAdd(&counters[1], ndiff);
            }
            if (elig) {
                atomicAdd(&counters[2], elig);
            }
            if (err) {
                atomicAdd(&counters[3], err);
            }
        }
        intra = ndiff = elig = err = 0;
    }
};

// neighbour difference of the current block (vector cx,cy not yet stored) -- dsv.c:403
__device__ __forceinline__ void neighbordif2_pre(const NbPre &p, int x, int y, int cx, int cy, int &dx, int &dy)
{
    int lx = cx, ly = cy, tx = cx, ty = cy;
    if (abs(cx) < 2 && abs(cy) < 2) {
        dx = dy = 0;
        return;
    }
    if (x > 0 && p.l_all && !(p.l_flags & (1u << DSV_MV_BIT_SKIP))) {
        lx = (int) (int16_t) (p.l_all & 0xffffu);
        ly = (int) (int16_t) (p.l_all >> 16);
    }
    if (y > 0 && p.t_all && !(p.t_flags & (1u << DSV_MV_BIT_SKIP))) {
        tx = (int) (int16_t) (p.t_all & 0xffffu);
        ty = (int) (int16_t) (p.t_all >> 16);
    }
    dx = abs(lx - cx) + abs(ly - cy);
    dy = abs(tx - cx) + abs(ty - cy);
}

// Sub-block sums of a WHOLE 16 x 16 block in 4:2:0 without the sixteen-entry transposing reduction: a sub-block is a set of lanes
// that differ in fixed lane bits, so its sum is a butterfly over the OTHER bits and every lane ends up with its own sub-block's
// total.  Luma (lane = qi | qj << 3, sub-block = {qi >= 4, qj >= 4} = lane bits 2 and 5): folds over bits 0, 1, 3, 4; the totals of
// sub-blocks 0..3 sit on lanes 0, 4, 32, 36.  Chroma quads (lanes 0..31: plane = bit 4, cqi = lane & 3, cqj = (lane >> 2) & 3,
// sub-block = {cqi >= 2, cqj >= 2} = bits 1 and 3): folds over bits 0 and 2; U's totals on lanes 0, 2, 8, 10, V's on 16, 18, 24, 26.
// (Integer sums: any order gives the reference's value.)  ~12 instructions for the twelve sums the mode decision asks for, against
// twelve selects and a ~46-instruction reduceN<16>.
__device__ __forceinline__ int subblock_sums_luma(int v) { return fold_xor<16>(fold_xor<8>(fold_xor<2>(fold_xor<1>(v)))); }
__device__ __forceinline__ int subblock_sums_chroma(int v) { return fold_xor<4>(fold_xor<1>(v)); }
__device__ __forceinline__ unsigned max4_lanes(int r, int l0, int l1, int l2, int l3)
{
    return max(max((unsigned) __builtin_amdgcn_readlane(r, l0), (unsigned) __builtin_amdgcn_readlane(r, l1)),
               max((unsigned) __builtin_amdgcn_readlane(r, l2), (unsigned) __builtin_amdgcn_readlane(r, l3)));
}

// level-0 tail of the block routine: sub-pel refinement + mode decision (hme.c:1598-1821)
// (sp_done, sp_mr, sp_dirs): the pixel half of the FIRST sub-pel search (around the parent average), from the pre-pass
// (subpel_probes; sp_done = it was run: effort >= 4 and the block at the parent average has a rim of four inside the padded plane)
// CS: chroma format -- 1 = 4:2:0 (a block's chroma is 8x8: one pixel per lane, its 2x2 quads on lanes 0..31),
// 0 = 4:4:4 (16x16 like the luma: every lane owns the quad (qi, qj) of U and of V),
// 2 = 4:2:2 (8 wide, 16 high: 4 x 8 quads a plane -- lanes 0..31 own U's, lanes 32..63 V's)
template <int CS, bool SPLIT, class Ctx>
__device__ __forceinline__ void hme_l0_tail(const Ctx &c_in, int i, int j, FastLds &S, RowAcc &acc, DSV_MV *out, DSV_MV mv, const CostCtx &cc, const Quad &a,
                                  bool act, int qi, int qj, int bx, int by, int bw, int bh, int lax, int lay, int motion_bias, bool good_enough,
                                  unsigned best, unsigned var_src, unsigned avg_src, const Psy &psy, const NbPre &pre, bool sp_done, unsigned sp_mr,
                                  unsigned sp_dirs)
{
    Ctx c = fenced(c_in, 0);
    const int lane = hme_lane();
    const int nxb = c.a.nbh, nyb = c.a.nbv, y_w = 16, y_h = 16;
    DPlane ref0 = c.ref[0];
    const int qw = bw >> 1, qh = bh >> 1;
    int fpelx = mv.u.mv.x, fpely = mv.u.mv.y, sx = 0, sy = 0;
    bool found_sub = false;
    unsigned yarea = (unsigned) (bw * bh);
    if (fpelx == lax && fpely == lay) {
        best += (unsigned) motion_bias;
    }
    unsigned best_fp = best;
    if (c.effort >= 4) {
        // pass 0: around the parent average; pass 1: around the best full-pel vector -- unless pass 0 found a sub-pel offset, the
        // full-pel search ended "good enough", or the centre would be the same (a second search around the SAME full-pel vector
        // repeats the first one operand for operand and ends where it did: no offset, the same score).  ONE copy of the search
        // in the kernel (a loop the compiler may not unroll: the search is ~700 instructions).
        bool searched_lax = false;
        if constexpr (SPLIT) { // (pass 0's pixel half comes from the pre-pass)
            if (sp_done) {
                searched_lax = true;
                if (best_fp != 0) { // (hme.c:1062: a perfect full-pel match is not searched around)
                    best = subpel_decide(cc, c.effort, sp_mr, sp_dirs, sx, sy, lax, lay, best_fp, bw, bh);
                }
                if (sx || sy) {
                    fpelx = lax;
                    fpely = lay;
                    found_sub = true;
                }
            }
        }
#pragma unroll 1
        for (int pass = SPLIT ? 1 : 0; pass < 2; pass++) {
            const int ccx = pass == 0 ? lax : fpelx, ccy = pass == 0 ? lay : fpely;
            const bool same_centre = searched_lax && fpelx == lax && fpely == lay;
            const bool run = (pass == 0 || (!found_sub && !good_enough && !same_centre)) && !invalid_block(ref0, bx + ccx, by + ccy, bw, bh, 4);
            if (!run) {
                continue;
            }
            HME_COUNT(S, 21 + pass, 1);
            if (best_fp != 0) {
                best = subpixel_me_fast(c, S, cc, sx, sy, ccx, ccy, best_fp, bx, by, bw, bh, a, act, qi, qj, psy);
            }
            if (pass == 0) {
                searched_lax = true;
                if (sx || sy) {
                    fpelx = lax;
                    fpely = lay;
                    found_sub = true;
                }
            }
        }
    }
    mv.u.mv.x = (int16_t) (fpelx * 4 + sx);
    mv.u.mv.y = (int16_t) (fpely * 4 + sy);
    unsigned ratio = 32;
    if ((mv.u.mv.x | mv.u.mv.y) & 3) {
        ratio = udiv_fast(best << 5, best_fp + !best_fp);
    }

    HME_MARK(S, 5);
    c = fenced(c_in, 0);
    ref0 = c.ref[0];
    // ---- operands of the mode decision, one load round ----
    constexpr int CSH = CS == 0 ? 0 : 1, CSV = CS == 1 ? 1 : 0; // chroma shifts: horizontal, vertical
    const int cbx = (i * 16) >> CSH, cby = (j * 16) >> CSV;   // 16x16 luma blocks
    const int cbmx = cbx + sarx(fpelx, CSH), cbmy = cby + sarx(fpely, CSV);
    const int cbw = bw >> CSH, cbh = bh >> CSV;
    // 4:2:2: the lane's chroma quad -- plane c2p, quad (c2i, c2j) of its 4 x 8 grid -- of the source / the reference at the vector / at zero
    const int c2p = lane >> 5, c2i = lane & 3, c2j = (lane >> 2) & 7;
    const bool act2 = CS == 2 && c2i < (cbw >> 1) && c2j < (cbh >> 1);
    const int k2 = (c2i >= (cbw >> 2) ? 1 : 0) | (c2j >= (cbh >> 2) ? 2 : 0); // ... and its sub-block
    Quad c2s, c2m, c2z;
    c2s.w = c2m.w = c2z.w = 0;
    const int cxp = lane & 7, cyp = lane >> 3;              // 4:2:0: chroma pixel owned by this lane
    const bool actc = CS == 1 && cxp < cbw && cyp < cbh;
    // 4:2:0: chroma quads for the sub-block metrics: lanes 0..15 U, 16..31 V
    const int cpl = (lane >> 4) & 1, cqi = lane & 3, cqj = (lane >> 2) & 3;
    const bool actq = CS == 1 && lane < 32 && cqi < (cbw >> 1) && cqj < (cbh >> 1);
    Quad r, o, rz, cs, cz, cm;
    int us = 0, vs = 0, um = 0, vm = 0;
    Quad usq, vsq, umq, vmq, uzq, vzq; // 4:4:4: this lane's quad of the source / reference-at-the-vector / reference-at-zero chroma blocks
    usq.w = vsq.w = umq.w = vmq.w = uzq.w = vzq.w = 0;
    cs.w = cz.w = cm.w = 0;
    r = ldq(at(ref0, bx + fpelx, by + fpely), ref0.stride, qi, qj, act);
    o = ldq(at(c.ogr[0], bx + fpelx, by + fpely), c.ogr[0].stride, qi, qj, act);
    // (the zero-motion operands are only looked at by the skip test: hme.c:1686)
    const bool skip_test = (good_enough || (fpelx | fpely | sx | sy) == 0) && c.skip_block_thresh >= 0 && !c.lossless;
    rz.w = 0;
    HME_COUNT(S, 16, skip_test ? 1 : 0);
    HME_COUNT(S, 23, good_enough ? 1 : 0);
    HME_COUNT(S, 27, ((mv.u.mv.x | mv.u.mv.y) & 3) ? 1 : 0);
    if (skip_test) {
        rz = ldq(at(ref0, bx, by), ref0.stride, qi, qj, act);
    }
    if constexpr (CS == 1) {
        us = ldpx(at(c.srcc[0], cbx, cby), cyp * c.srcc[0].stride + cxp, actc);
        vs = ldpx(at(c.srcc[1], cbx, cby), cyp * c.srcc[1].stride + cxp, actc);
        um = ldpx(at(c.refc[0], cbmx, cbmy), cyp * c.refc[0].stride + cxp, actc);
        vm = ldpx(at(c.refc[1], cbmx, cbmy), cyp * c.refc[1].stride + cxp, actc);
        cs = ldq(at(c.srcc[cpl], cbx, cby), c.srcc[cpl].stride, cqi, cqj, actq);
        if (skip_test) {
            cz = ldq(at(c.refc[cpl], cbx, cby), c.refc[cpl].stride, cqi, cqj, actq);
        }
        cm = ldq(at(c.refc[cpl], cbmx, cbmy), c.refc[cpl].stride, cqi, cqj, actq);
    } else if constexpr (CS == 2) {
        c2s = ldq(at(c.srcc[c2p], cbx, cby), c.srcc[c2p].stride, c2i, c2j, act2);
        c2m = ldq(at(c.refc[c2p], cbmx, cbmy), c.refc[c2p].stride, c2i, c2j, act2);
        if (skip_test) {
            c2z = ldq(at(c.refc[c2p], cbx, cby), c.refc[c2p].stride, c2i, c2j, act2);
        }
    } else {
        usq = ldq(at(c.srcc[0], cbx, cby), c.srcc[0].stride, qi, qj, act);
        vsq = ldq(at(c.srcc[1], cbx, cby), c.srcc[1].stride, qi, qj, act);
        umq = ldq(at(c.refc[0], cbmx, cbmy), c.refc[0].stride, qi, qj, act);
        vmq = ldq(at(c.refc[1], cbmx, cbmy), c.refc[1].stride, qi, qj, act);
        if (skip_test) {
            uzq = ldq(at(c.refc[0], cbx, cby), c.refc[0].stride, qi, qj, act);
            vzq = ldq(at(c.refc[1], cbx, cby), c.refc[1].stride, qi, qj, act);
        }
    }
    const int kq = (qi >= (qw >> 1) ? 1 : 0) | (qj >= (qh >> 1) ? 2 : 0);         // luma quadrant of this lane's quad
    const int kc = (cqi >= (cbw >> 2) ? 1 : 0) | (cqj >= (cbh >> 2) ? 2 : 0);      // chroma quadrant of this lane's chroma quad
    const int kp = (cxp >= (cbw >> 1) ? 1 : 0) | (cyp >= (cbh >> 1) ? 2 : 0);      // chroma quadrant of this lane's chroma pixel

    // round 1: block sums
    int v[16];
    {
        int rs, rh, rv;
        quad_grad_partials(r, act, qi, qj, 0, 0, rs, rh, rv);
        v[0] = act ? (int) qmetric(a, o, psy) : 0;
        v[1] = rs;
        v[2] = rh;
        v[3] = rv;
        if constexpr (CS == 1) {
            int ul = __shfl_up(us, 1, 64), uu = __shfl_up(us, 8, 64), vl = __shfl_up(vs, 1, 64), vu = __shfl_up(vs, 8, 64);
            v[4] = us;
            v[5] = vs;
            v[6] = um;
            v[7] = vm;
            v[8] = (actc && cxp > 0) ? abs(us - ul) : 0;
            v[9] = (actc && cyp > 0) ? abs(us - uu) : 0;
            v[10] = (actc && cxp > 0) ? abs(vs - vl) : 0;
            v[11] = (actc && cyp > 0) ? abs(vs - vu) : 0;
        } else if constexpr (CS == 2) { // the 8 x 16 chroma blocks, quad by quad: U's sums on lanes 0..31, V's on lanes 32..63
            int cs_, ch_, cv_;
            quad_grad_partials_w<4>(c2s, act2, c2i, c2j, cs_, ch_, cv_);
            const int ms_ = act2 ? c2m.p1() + c2m.p2() + c2m.p3() + c2m.p4() : 0;
            const bool isu = c2p == 0;
            v[4] = isu ? cs_ : 0;
            v[5] = isu ? 0 : cs_;
            v[6] = isu ? ms_ : 0;
            v[7] = isu ? 0 : ms_;
            v[8] = isu ? ch_ : 0;
            v[9] = isu ? cv_ : 0;
            v[10] = isu ? 0 : ch_;
            v[11] = isu ? 0 : cv_;
        } else { // pixel sums and first-difference sums of the 16x16 chroma blocks, quad by quad (as for the luma)
            quad_grad_partials(usq, act, qi, qj, 0, 0, v[4], v[8], v[9]);
            quad_grad_partials(vsq, act, qi, qj, 0, 0, v[5], v[10], v[11]);
            v[6] = act ? umq.p1() + umq.p2() + umq.p3() + umq.p4() : 0;
            v[7] = act ? vmq.p1() + vmq.p2() + vmq.p3() + vmq.p4() : 0;
        }
        v[12] = v[13] = v[14] = v[15] = 0;
    }
    int R = reduceN<16>(v);
    unsigned ogrerr = metric_return((unsigned) bcastN<16>(R, 0), bw, bh);
    int ref_sum = bcastN<16>(R, 1);
    unsigned ref_sh = (unsigned) bcastN<16>(R, 2), ref_sv = (unsigned) bcastN<16>(R, 3);
    int uavg_src = div_nn(bcastN<16>(R, 4), cbw * cbh), vavg_src = div_nn(bcastN<16>(R, 5), cbw * cbh);
    int uavg_ref = div_nn(bcastN<16>(R, 6), cbw * cbh), vavg_ref = div_nn(bcastN<16>(R, 7), cbw * cbh);
    int utex = (int) max((unsigned) bcastN<16>(R, 8), (unsigned) bcastN<16>(R, 9));
    int vtex = (int) max((unsigned) bcastN<16>(R, 10), (unsigned) bcastN<16>(R, 11));
    unsigned avg_ref = (unsigned) div_nn(ref_sum, bw * bh);

    // round 2: reference deviation + -- for the skip test -- the zero-motion sub-block metrics
    int ref_dev;
    unsigned zsub[3] = {0u, 0u, 0u};
    constexpr bool kWhole420 = CS == 1; // (with bw == bh == 16: a compile-time fact in the whole-block instantiation)
    if (skip_test && kWhole420 && bw == 16 && bh == 16) {
        ref_dev = wave_sum(quad_absdev(r, true, (int) avg_ref)) >> 1;
        const int ys = subblock_sums_luma((int) qmetric(a, rz, psy));
        const int cs_ = subblock_sums_chroma(lane < 32 ? (int) qmetric(cs, cz, psy) : 0);
        zsub[0] = max4_lanes(ys, 0, 4, 32, 36);
        zsub[1] = max4_lanes(cs_, 0, 2, 8, 10);
        zsub[2] = max4_lanes(cs_, 16, 18, 24, 26);
    } else if (skip_test) {
        v[0] = quad_absdev(r, act, (int) avg_ref);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[1 + k] = (act && kq == k) ? (int) qmetric(a, rz, psy) : 0;
            if constexpr (CS == 1) {
                v[5 + k] = (actq && cpl == 0 && kc == k) ? (int) qmetric(cs, cz, psy) : 0;
                v[9 + k] = (actq && cpl == 1 && kc == k) ? (int) qmetric(cs, cz, psy) : 0;
            } else if constexpr (CS == 2) {
                v[5 + k] = (act2 && c2p == 0 && k2 == k) ? (int) qmetric(c2s, c2z, psy) : 0;
                v[9 + k] = (act2 && c2p == 1 && k2 == k) ? (int) qmetric(c2s, c2z, psy) : 0;
            } else {
                v[5 + k] = (act && kq == k) ? (int) qmetric(usq, uzq, psy) : 0;
                v[9 + k] = (act && kq == k) ? (int) qmetric(vsq, vzq, psy) : 0;
            }
        }
        v[13] = v[14] = v[15] = 0;
        R = reduceN<16>(v);
        ref_dev = bcastN<16>(R, 0) >> 1;
#pragma unroll
        for (int z = 0; z < 3; z++) {
            unsigned m0 = (unsigned) bcastN<16>(R, 1 + 4 * z), m1 = (unsigned) bcastN<16>(R, 2 + 4 * z);
            unsigned m2 = (unsigned) bcastN<16>(R, 3 + 4 * z), m3 = (unsigned) bcastN<16>(R, 4 + 4 * z);
            zsub[z] = max(max(m0, m1), max(m2, m3));
        }
    } else {
        ref_dev = wave_sum(quad_absdev(r, act, (int) avg_ref)) >> 1;
    }
    int tex_ref = (int) (max(ref_sh, ref_sv) - (unsigned) ref_dev);
    unsigned var_ref = (unsigned) (ref_dev + max(tex_ref, 0));

    unsigned ogrmad = div_nn(ogrerr + yarea / 2, yarea);
    ogrmad = ogrmad * ratio >> 5;
    unsigned mad = div_nn(best + yarea / 2, yarea);
    int dv = (int) min(ratio, 32u);
    int ipolvar = (int) ((var_src * (unsigned) dv + var_ref * (unsigned) (32 - dv)) >> 5);
    dv = abs((int) var_src - ipolvar);
    if (var_src > 16 * yarea && var_src < 32 * yarea) {
        mv.flags |= 1u << DSV_MV_BIT_MAINTAIN;
    }
    unsigned chroma_ratio = div_nn((unsigned) ((cbw * cbh) << 4), yarea);
    ChromaPsy cpsy = chroma_analysis((int) avg_src, uavg_src, vavg_src);
    unsigned avg_y_dif = (unsigned) abs((int) avg_src - (int) avg_ref);
    unsigned avg_c_dif = (unsigned) AVG2(abs(uavg_src - uavg_ref), abs(vavg_src - vavg_ref));
    // expanded-range votes (hme.c:1790-1821): against the reference block here; against the reference's / the source's mean only if the
    // block ends up intra (one block in 250 of the headline's content) -- worked out there
    int eprmr;
    {
        int cr = 0;
        if (act) {
            cr = (((a.p1() - r.p1()) + 128) | ((a.p2() - r.p2()) + 128) | ((a.p3() - r.p3()) + 128) | ((a.p4() - r.p4()) + 128)) & ~0xff;
        }
        eprmr = __any(cr != 0) ? 1 : 0;
    }
    auto eprm_against_mean = [&](int mean) {
        const int m128 = mean - 128;
        const int cm_ = act ? (((a.p1() - m128) | (a.p2() - m128) | (a.p3() - m128) | (a.p4() - m128)) & ~0xff) : 0;
        return __any(cm_ != 0) ? 1 : 0;
    };
    bool oob;
    {
        int px = i * y_w + sarx(mv.u.mv.x, 2), py = j * y_h + sarx(mv.u.mv.y, 2);
        oob = px < 0 || py < 0 || px >= ((nxb - 1) * y_w) - 1 || py >= ((nyb - 1) * y_h) - 1;
    }
    int neidif;
    {
        int na, nb_;
        neighbordif2_pre(pre, i, j, mv.u.mv.x, mv.u.mv.y, na, nb_);
        neidif = (na + nb_) / 3;
    }
    unsigned skipt = ((unsigned) c.quant * (unsigned) c.quant) >> 19;
    bool skipped = false;
    if (skip_test) {
        unsigned sth = skipt * yarea;
        sth += 4 * var_src;
        sth += yarea * (unsigned) c.skip_block_thresh;
        if (c.quant < (1 << 10)) {
            sth = sth * (unsigned) c.quant >> 10;
        }
        if (avg_y_dif <= 2) {
            sth = max(sth, 3 * (yarea + var_src));
        }
        sth = max(sth, yarea);
        if (good_enough) {
            sth *= 2;
        }
        unsigned cth = chroma_ratio * sth * max(skipt, 1u) >> 5;
        unsigned z0 = zsub[0] * ratio >> 5, z1 = zsub[1] * ratio >> 5, z2 = zsub[2] * ratio >> 5;
        z0 += (unsigned) SQR((int) avg_src - (int) avg_ref) * yarea;
        if (z0 <= sth && z1 <= cth && z2 <= cth) {
            mv.flags |= 1u << DSV_MV_BIT_SKIP;
            mv.u.all = 0;
            mv.err = 0;
            skipped = true;
            HME_COUNT(S, 17, 1);
        }
    }
    int add_err = 0, add_ndiff = 0;
    if (!skipped) {
        if (!oob && !c.lossless) {
            bool y_prereq = avg_y_dif <= 2, c_prereq = !cpsy.greyish && avg_c_dif <= 2;
            if (y_prereq || c_prereq) {
                HME_COUNT(S, 18, 1);
                // round 3: sub-block metrics at the chosen full-pel motion (hme.c:1741)
                unsigned bsub[3];
                if (kWhole420 && bw == 16 && bh == 16) {
                    const int ys = subblock_sums_luma((int) qmetric(a, r, psy));
                    const int cs_ = subblock_sums_chroma(lane < 32 ? (int) qmetric(cs, cm, psy) : 0);
                    bsub[0] = max4_lanes(ys, 0, 4, 32, 36) * ratio >> 5;
                    bsub[1] = max4_lanes(cs_, 0, 2, 8, 10) * ratio >> 5;
                    bsub[2] = max4_lanes(cs_, 16, 18, 24, 26) * ratio >> 5;
                } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    v[k] = (act && kq == k) ? (int) qmetric(a, r, psy) : 0;
                    if constexpr (CS == 1) {
                        v[4 + k] = (actq && cpl == 0 && kc == k) ? (int) qmetric(cs, cm, psy) : 0;
                        v[8 + k] = (actq && cpl == 1 && kc == k) ? (int) qmetric(cs, cm, psy) : 0;
                    } else if constexpr (CS == 2) {
                        v[4 + k] = (act2 && c2p == 0 && k2 == k) ? (int) qmetric(c2s, c2m, psy) : 0;
                        v[8 + k] = (act2 && c2p == 1 && k2 == k) ? (int) qmetric(c2s, c2m, psy) : 0;
                    } else {
                        v[4 + k] = (act && kq == k) ? (int) qmetric(usq, umq, psy) : 0;
                        v[8 + k] = (act && kq == k) ? (int) qmetric(vsq, vmq, psy) : 0;
                    }
                    v[12 + k] = 0;
                }
                R = reduceN<16>(v);
#pragma unroll
                for (int z = 0; z < 3; z++) {
                    unsigned m0 = (unsigned) bcastN<16>(R, 4 * z), m1 = (unsigned) bcastN<16>(R, 4 * z + 1);
                    unsigned m2 = (unsigned) bcastN<16>(R, 4 * z + 2), m3 = (unsigned) bcastN<16>(R, 4 * z + 3);
                    bsub[z] = max(max(m0, m1), max(m2, m3)) * ratio >> 5;
                }
                }
                unsigned xth = skipt * yarea;
                int carea = 4 * cbw * cbh;
                xth += (unsigned) ipolvar;
                xth = (unsigned) max((int) xth - ((int) yarea * neidif * 2), 0);
                xth = xth * (unsigned) c.quant >> 12;
                xth = min(max(xth, 32u), yarea * 4);
                if (y_prereq && bsub[0] < 4 * xth) {
                    mv.flags |= 1u << DSV_MV_BIT_NOXMITY;
                }
                c_prereq = c_prereq && (utex > carea || vtex > carea);
                xth = chroma_ratio * xth >> 4;
                if (c_prereq && bsub[1] < xth && bsub[2] < xth) {
                    mv.flags |= 1u << DSV_MV_BIT_NOXMITC;
                }
            }
            if ((unsigned) dv < var_src / 4) {
                mv.flags |= 1u << DSV_MV_BIT_SIMCMPLX;
            }
        }
        HME_MARK(S, 6);
        c = fenced(c_in, 0);
        ref0 = c.ref[0];
        // ---- test_subblock_intra_y (hme.c:891), all four sub-blocks evaluated together ----
        {
            int rx = mv.u.mv.x, ry = mv.u.mv.y;
            if (c.ref_mvf != nullptr) {
                uint32_t colo = pre.colo;
                if (!pre.colo_ok) { // no coarser level (one-level pyramid): the candidate round did not fetch it
                    colo = (uint32_t) __builtin_amdgcn_readfirstlane((int) *(const uint32_t *) &c.ref_mvf[i + j * nxb]);
                }
                rx = (int) (int16_t) (colo & 0xffffu);
                ry = (int) (int16_t) (colo >> 16);
            }
            int sbw = bw / 2, sbh = bh / 2;
            bool run = !(mv.u.all && neidif < 3 && abs(rx - mv.u.mv.x) < 3 && abs(ry - mv.u.mv.y) < 3) && sbw != 0 && sbh != 0;
            if (run) {
                HME_COUNT(S, 19, 1);
                int ss, sh, sv2;
                quad_grad_partials(a, act, qi, qj, (kq & 1) ? (qw >> 1) : 0, (kq & 2) ? (qh >> 1) : 0, ss, sh, sv2);
                int rsum = act ? r.p1() + r.p2() + r.p3() + r.p4() : 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bool in = act && kq == k;
                    v[4 * k + 0] = in ? ss : 0;
                    v[4 * k + 1] = in ? rsum : 0;
                    v[4 * k + 2] = in ? sh : 0;
                    v[4 * k + 3] = in ? sv2 : 0;
                }
                R = reduceN<16>(v);
                int my_avg_local = div_nn(bcastL<16>(R, 4 * kq + 0), sbw * sbh);
                int my_avg_sub = div_nn(bcastL<16>(R, 4 * kq + 1), sbw * sbh);
                int my_dc = (int) ((unsigned) my_avg_local + (unsigned) avg_src * 3 + 2) >> 2;
                unsigned e_inter = 0, e_sb = 0, e_src = 0;
                if (act) {
                    quad_err_intra(a, r, my_avg_sub, my_dc, (int) ratio, e_inter, e_sb, e_src);
                }
                int w16[16];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bool in = act && kq == k;
                    w16[k] = in ? quad_absdev(a, true, my_avg_local) : 0;
                    w16[4 + 3 * k + 0] = in ? (int) e_inter : 0;
                    w16[4 + 3 * k + 1] = in ? (int) e_sb : 0;
                    w16[4 + 3 * k + 2] = in ? (int) e_src : 0;
                }
                int R2 = reduceN<16>(w16);
                int detail_src = ipolvar, nsub = 0;
                unsigned avg_tot = 0, err_sub = 0, err_src = 0;
                detail_src += sdiv_fast(detail_src, max(neidif, 1));
                for (int k = 0; k < 4; k++) {
                    if (mv.submask & (1 << k)) {
                        continue;
                    }
                    unsigned avg_local = (unsigned) div_nn(bcastN<16>(R, 4 * k + 0), sbw * sbh);
                    unsigned avg_sub = (unsigned) div_nn(bcastN<16>(R, 4 * k + 1), sbw * sbh);
                    unsigned g_sh = (unsigned) bcastN<16>(R, 4 * k + 2), g_sv = (unsigned) bcastN<16>(R, 4 * k + 3);
                    int var = bcastN<16>(R2, k) >> 1;
                    int tex = (int) (max(g_sh, g_sv) - (unsigned) var);
                    unsigned local_detail = (unsigned) (var + max(tex, 0));
                    unsigned dcd = (unsigned) abs((int) avg_local - (int) avg_sub) + 2;
                    if (local_detail > (unsigned) (SQR(dcd) * (unsigned) bw * (unsigned) bh * ratio >> 5)) {
                        continue;
                    }
                    int dc = (int) (avg_local + (unsigned) avg_src * 3 + 2) >> 2;
                    unsigned inter_err = (unsigned) bcastN<16>(R2, 4 + 3 * k + 0) * ratio >> 5;
                    unsigned sub_err = (unsigned) bcastN<16>(R2, 4 + 3 * k + 1), src_err = (unsigned) bcastN<16>(R2, 4 + 3 * k + 2);
                    int lo = AVG2(detail_src, (int) local_detail), hi = detail_src;
                    int lerp = (lo * (32 - c.psyscale) + hi * c.psyscale) >> 5;
                    local_detail = (unsigned) max(lerp, lo);
                    if ((sub_err + local_detail) < inter_err || (src_err + local_detail) < inter_err) {
                        mv.submask |= (uint8_t) (1 << k);
                        err_src += src_err;
                        err_sub += sub_err;
                        avg_tot += sub_err < src_err ? avg_sub : (unsigned) dc;
                        nsub++;
                        detail_src = detail_src * 4 / 5;
                    }
                }
                if (mv.submask) {
                    mv.flags |= 1u << DSV_MV_BIT_INTRA;
                    mv.dc = err_src < err_sub ? (uint16_t) (udiv_fast(avg_tot, (unsigned) nsub) | DSV_SRC_DC_PRED) : 0;
                }
            }
        }
        HME_MARK(S, 7);
        c = fenced(c_in, 0);
        ref0 = c.ref[0];
        // ---- test_subblock_intra_c (hme.c:987) ----
        if (c.effort >= 6) {
            unsigned detail_c = (unsigned) div_nn(ipolvar, bw * bh);
            unsigned thr = (mv.flags & (1u << DSV_MV_BIT_INTRA)) ? detail_c : SQR(detail_c);
            int sbw = cbw / 2, sbh = cbh / 2;
            if (!(sbw == 0 || sbh == 0 || mad <= thr || thr > 64 || (abs((int) mv.u.mv.x) < 4 && abs((int) mv.u.mv.y) < 4))) {
                HME_COUNT(S, 20, 1);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (CS == 1) {
                        bool in = actc && kp == k;
                        v[4 * k + 0] = in ? us : 0;
                        v[4 * k + 1] = in ? vs : 0;
                        v[4 * k + 2] = in ? um : 0;
                        v[4 * k + 3] = in ? vm : 0;
                    } else if constexpr (CS == 2) {
                        const bool in = act2 && k2 == k;
                        const int ssum = c2s.p1() + c2s.p2() + c2s.p3() + c2s.p4(), msum = c2m.p1() + c2m.p2() + c2m.p3() + c2m.p4();
                        v[4 * k + 0] = (in && c2p == 0) ? ssum : 0;
                        v[4 * k + 1] = (in && c2p == 1) ? ssum : 0;
                        v[4 * k + 2] = (in && c2p == 0) ? msum : 0;
                        v[4 * k + 3] = (in && c2p == 1) ? msum : 0;
                    } else {
                        bool in = act && kq == k;
                        v[4 * k + 0] = in ? usq.p1() + usq.p2() + usq.p3() + usq.p4() : 0;
                        v[4 * k + 1] = in ? vsq.p1() + vsq.p2() + vsq.p3() + vsq.p4() : 0;
                        v[4 * k + 2] = in ? umq.p1() + umq.p2() + umq.p3() + umq.p4() : 0;
                        v[4 * k + 3] = in ? vmq.p1() + vmq.p2() + vmq.p3() + vmq.p4() : 0;
                    }
                }
                R = reduceN<16>(v);
                unsigned avg_ramp = avg_src * avg_src >> 8;
                for (int k = 0; k < 4; k++) {
                    if (mv.submask & (1 << k)) {
                        continue;
                    }
                    int a_us = div_nn(bcastN<16>(R, 4 * k + 0), sbw * sbh), a_vs = div_nn(bcastN<16>(R, 4 * k + 1), sbw * sbh);
                    int a_um = div_nn(bcastN<16>(R, 4 * k + 2), sbw * sbh), a_vm = div_nn(bcastN<16>(R, 4 * k + 3), sbw * sbh);
                    unsigned dif = (unsigned) (SQR(a_us - a_um) + SQR(a_vs - a_vm)) * avg_ramp >> 8;
                    if (dif > thr) {
                        mv.submask |= (uint8_t) (1 << k);
                    }
                }
                if (mv.submask) {
                    mv.flags |= 1u << DSV_MV_BIT_INTRA;
                }
            }
        }
        if (!(mv.flags & (1u << DSV_MV_BIT_NOXMITY))) {
            mv.err = (uint16_t) mad;
            add_err = (int) mad;
        }
        add_ndiff = (ogrmad > 11) + (avg_c_dif >= 32);
    }
    int is_intra = 0;
    HME_COUNT(S, 24, (mv.flags & (1u << DSV_MV_BIT_INTRA)) ? 1 : 0);
    if (mv.flags & (1u << DSV_MV_BIT_INTRA)) {
        int merged = eprm_against_mean((mv.dc & DSV_SRC_DC_PRED) ? (int) avg_src : (int) avg_ref);
        if (mv.submask != DSV_MASK_ALL_INTRA) {
            merged |= eprmr;
        }
        mv.flags = (mv.flags & ~(1u << DSV_MV_BIT_EPRM)) | (merged ? (1u << DSV_MV_BIT_EPRM) : 0u);
        is_intra = 1;
        mv.u.mv.x = (int16_t) (fpelx * 4);
        mv.u.mv.y = (int16_t) (fpely * 4);
    } else {
        int merged = eprmr;
        if (mv.submask) { // (never: a sub-block mask sets the intra flag with it; kept as the reference has it, hme.c:1815)
            merged |= eprm_against_mean((int) avg_ref);
        }
        mv.flags = (mv.flags & ~(1u << DSV_MV_BIT_EPRM)) | (merged ? (1u << DSV_MV_BIT_EPRM) : 0u);
    }
    if (mv.flags & ((1u << DSV_MV_BIT_INTRA) | (1u << DSV_MV_BIT_EPRM))) {
        mv.flags &= ~(1u << DSV_MV_BIT_SIMCMPLX);
    }
    HME_MARK(S, 8);
    if (lane == 0) {
        st_mv_final(c, out, mv);
    }
    acc.left_head = (unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) mv.u.all) |
                    ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) mv.flags) << 32);
    acc.have_left = true;
    acc.intra += is_intra;
    acc.ndiff += add_ndiff;
    acc.elig += best > 0 ? 1 : 0;
    acc.err += add_err;
}


// the 3x3 offset tables of the search, entry k in {-1, 0, 1}: arithmetic on a packed constant (a lane-indexed
// constant array would be fetched from memory, one round trip per lookup)
__device__ __forceinline__ int tab9(unsigned packed, int k) { return (int) ((packed >> (2 * k)) & 3u) - 1; }
constexpr unsigned kRectX = 0x22149u, kRectY = 0x28095u; // rect[]: {0,1,-1,0,0,-1,1,-1,1} / {0,0,0,1,-1,-1,-1,1,1} (hme.c:1304)
constexpr unsigned kParX = 0xa161u, kParY = 0x22215u;   // parent offsets / 2: {0,-1,1,0,0,-1,1,1,-1} / {0,0,0,-1,1,-1,1,-1,1} (hme.c:1468)


// ---- parent-level half of the candidate list: the average of the parent vectors' inliers (hme.c:1443-1470, find_inliers :1260) ----
// lanes 16..24 hold the up to nine parent vectors (pvalid).  false: there is none -- the list is then the zero vector alone.
__device__ __forceinline__ bool parent_average(bool pvalid, int pvx, int pvy, int &lax, int &lay, bool &inl, int &nin);
// the same through the row's one-entry cache (RowAcc::par_*): `key` names the parent block
__device__ __forceinline__ bool parent_average_cached(RowAcc &acc, int key, bool pvalid, int pvx, int pvy, int &lax, int &lay, bool &inl, int &nin)
{
    if (acc.par_key != key) {
        acc.par_open = parent_average(pvalid, pvx, pvy, acc.par_lax, acc.par_lay, acc.par_inl, acc.par_nin);
        acc.par_key = key;
    }
    lax = acc.par_lax;
    lay = acc.par_lay;
    inl = acc.par_inl;
    nin = acc.par_nin;
    return acc.par_open;
}
__device__ __forceinline__ bool parent_average(bool pvalid, int pvx, int pvy, int &lax, int &lay, bool &inl, int &nin)
{
    lax = lay = nin = 0;
    inl = false;
    const int npar = __popcll(__ballot(pvalid));
    if (!npar) {
        return false;
    }
    int v2[2] = {pvalid ? pvx : 0, pvalid ? pvy : 0};
    int r = reduceN<2>(v2);
    lax = sdiv_fast(bcastN<2>(r, 0), npar);
    lay = sdiv_fast(bcastN<2>(r, 1), npar);
    int dist = pvalid ? SQR(pvx - lax) + SQR(pvy - lay) : 0;
    int avgd = sdiv_fast(wave_sum(dist), npar);
    int ssd = wave_sum(pvalid ? SQR(dist - avgd) : 0);
    int thresh = avgd + (int) isqrt_u32((unsigned) sdiv_fast(ssd, npar));
    inl = pvalid && dist <= thresh;
    nin = __popcll(__ballot(inl));
    if (nin) {
        int w2[2] = {inl ? pvx : 0, inl ? pvy : 0};
        int r2 = reduceN<2>(w2);
        lax = sdiv_fast(bcastN<2>(r2, 0), nin);
        lay = sdiv_fast(bcastN<2>(r2, 1), nin);
    }
    return true;
}

// first-occurrence de-duplication of the per-lane list entries (hme.c:1166): one round per DISTINCT vector, not per entry (a block
// has ~15 - 25 entries and ~3.5 distinct vectors): the lowest lane that holds an unclassified entry keeps it, every other
// entry with that vector is a duplicate of it.  key: both components, int16 each.  Returns "this lane's entry is a duplicate".
__device__ __forceinline__ bool dedup_lanes(bool exist, int key)
{
    const int lane = hme_lane();
    bool dup = false;
    for (unsigned long long rest = __ballot(exist); rest;) {
        const int m = __ffsll((long long) rest) - 1;
        const int km = __builtin_amdgcn_readlane(key, m);
        const bool same = exist && km == key;
        dup = dup || (same && lane != m);
        rest &= ~__ballot(same);
    }
    return dup;
}

// ---- full-pel refinement (hme.c:1300): each round scores the 3x3 neighbourhood at once --------------------------------------
template <bool L0, int NQ>
__device__ __forceinline__ void refine_fpel(const DPlane &ref, const SrcBlk<NQ> &B, int level, const Psy &psy, const CostCtx &cc, unsigned qthresh, int &dx,
                                            int &dy, unsigned &best, bool &good_enough, FastLds &S)
{
    const int lane = hme_lane();
    const int bx = B.bx, by = B.by, bw = B.bw, bh = B.bh;
    const int step = 1 << level;
    unsigned metr0 = 0xffffffffu, metr1 = 0xffffffffu, metr2 = 0xffffffffu, metr3 = 0xffffffffu;
    bool again = true;
    HME_COUNT(S, 11, 1);
    while (again && !good_enough) {
        again = false;
        HME_COUNT(S, 12, 1);
        VecSet<9> vs;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            vs.dx[t] = dx + tab9(kRectX, t);
            vs.dy[t] = dy + tab9(kRectY, t);
        }
        unsigned raw = score_set<9, NQ>(vs, 9, (unsigned) (dx + 30) <= 60u && (unsigned) (dy + 30) <= 60u, ref, B, level, psy);
        int tx = dx + (lane < 9 ? tab9(kRectX, lane) : 0), ty = dy + (lane < 9 ? tab9(kRectY, lane) : 0);
        bool valid = lane < 9 && !invalid_block(ref, bx + tx, by + ty, bw, bh, 0);
        if (level <= 1) {
            raw = metric_return(raw, bw, bh);
        }
        unsigned full = raw + (unsigned) mv_cost(cc, tx * step * 4, ty * step * 4, level);
        int cdx = dx, cdy = dy;
        {
            // hme.c:1325-1352 walks the centre and its four neighbours in order and leaves at the first one that is "good enough"
            // (level 0, the zero vector, raw score within the threshold) or improves on the best; the axis scores of the positions it
            // passed -- that one included -- stick.  Lane k holds position k: the first exit is a ballot, not a five-way unrolled loop.
            const bool five = lane < 5;
            const bool ge = L0 && five && valid && tx == 0 && ty == 0 && raw <= qthresh;
            const bool imp = five && valid && best > full;
            const unsigned long long gem = __ballot(ge), exm = gem | __ballot(imp);
            const int first = exm ? __ffsll((long long) exm) - 1 : 5;
            const unsigned long long seen = __ballot(valid) & ((2ull << first) - 1ull);
            metr0 = (seen & 2ull) ? (unsigned) __builtin_amdgcn_readlane((int) raw, 1) : metr0;
            metr1 = (seen & 4ull) ? (unsigned) __builtin_amdgcn_readlane((int) raw, 2) : metr1;
            metr2 = (seen & 8ull) ? (unsigned) __builtin_amdgcn_readlane((int) raw, 3) : metr2;
            metr3 = (seen & 16ull) ? (unsigned) __builtin_amdgcn_readlane((int) raw, 4) : metr3;
            if (exm) {
                dx = cdx + tab9(kRectX, first);
                dy = cdy + tab9(kRectY, first);
                if ((gem >> first) & 1ull) {
                    best = (unsigned) __builtin_amdgcn_readlane((int) raw, first);
                    good_enough = true;
                } else {
                    best = (unsigned) __builtin_amdgcn_readlane((int) full, first);
                    again = true;
                }
            }
        }
        if (again || good_enough) {
            continue;
        }
        int sxs = metr0 <= metr1 ? 1 : -1, sys = metr2 <= metr3 ? 1 : -1;
        int kd = sys < 0 ? (sxs < 0 ? 5 : 6) : (sxs < 0 ? 7 : 8); // index of (sxs, sys) in rect[]
        bool vd = __builtin_amdgcn_readlane((int) valid, kd) != 0;
        if (!vd) {
            break;
        }
        unsigned fd = (unsigned) __builtin_amdgcn_readlane((int) full, kd);
        if (best > fd) {
            best = fd;
            dx = cdx + sxs;
            dy = cdy + sys;
            again = true;
        }
    }
}

// the 8-byte heads of a block's same-level neighbours: lanes 3 / 4 / 5 load left / top / top-left (coherent loads: the row
// above may be on another XCD).  Row pipeline: the top / top-left heads may not have been stored yet -- they read kMvPending
// until they are (hme.hip: wait_heads).  Validated HERE, inside the block's first load round: in the usual case (the row above
// is ahead) the hand-off costs no memory round trip of its own.  The left neighbour is the block this wavefront has just
// finished: its head is still in registers (acc.left_head).
__device__ __forceinline__ MvHead load_neighbour_heads(DSV_MV *mvf, DSV_MV *out, int i, int j, int step, int nxb, int *counters, RowAcc &acc, bool &nb_ok)
{
    const int lane = hme_lane();
    const bool need_i = lane != 4, need_j = lane != 3;
    nb_ok = lane >= 3 && lane <= 5 && (!need_i || i > 0) && (!need_j || j > 0);
    const DSV_MV *np = nb_ok ? &mvf[(i - (need_i ? step : 0)) + (j - (need_j ? step : 0)) * nxb] : out;
    typedef const __attribute__((address_space(1))) unsigned long long *gu64p_t;
    unsigned long long head = __hip_atomic_load((gu64p_t) np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        const bool mine = nb_ok && lane != 3;
        unsigned long long t0 = 0;
        for (unsigned spins = 0; __any(mine && head == kMvPending); spins++) {
            __builtin_amdgcn_s_sleep(8);
            head = __hip_atomic_load((gu64p_t) np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((spins & 1023u) == 1023u) {
                const unsigned long long now = wall_clock64();
                int *err = &counters[kHmeErrWord];
                if (t0 == 0) {
                    t0 = now;
                } else if (now - t0 > kHmeSpinLimit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    acc.failed = true;
                    break;
                }
            }
        }
    }
    if (lane == 3) {
        head = acc.left_head;
    }
    MvHead nbv;
    nbv.all = (uint32_t) head;
    nbv.flags = (uint32_t) (head >> 32);
    nbv.x = (int) (int16_t) (nbv.all & 0xffffu);
    nbv.y = (int) (int16_t) (nbv.all >> 16);
    return nbv;
}

// ============================================================================================================================
// Coarser levels (level >= 1): candidate list, best candidate, refinement (hme.c:1373-1597); no sub-pel, no mode decision.
// FULL: the block is a whole 16x16 one (every lane owns a quad): the per-lane activity tests fold away.
// Level 1 reads the source statistics of the pre-pass (c.stats: k_hme_src_stats4_b / _b); the squared-error levels need none.
// ============================================================================================================================
// the lane's share of the source block at (bx, by) of `src` for the metric of `level` (see SrcBlk); o (optional): the same quads of another plane
template <bool FULL, int NQ>
__device__ __forceinline__ SrcBlk<NQ> load_src_blk(const DPlane &src, int bx, int by, int bw, int bh, int level, bool with_other, const DPlane &oth, Quad (&o)[NQ])
{
    const int lane = hme_lane();
    SrcBlk<NQ> B;
    B.bx = bx;
    B.by = by;
    B.bw = bw;
    B.bh = bh;
    B.qi = lane & 7;
    B.qj = lane >> 3;
    const int qw = bw >> 1, qh = bh >> 1;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        const int xq = 8 * (k & 1) + B.qi, yq = 8 * (k >> 1) + B.qj; // the quad's place in the block
        // The psy metric works on whole 2x2 quads and drops an odd last row / column (hme.c:136: loops to h / 2, w / 2); the squared
        // error of the levels above 1 counts every pixel (hme.c:198).  There an odd row / column is a row / column of HALF quads:
        // those lanes take part with the bytes outside the block masked off in both operands.
        B.act[k] = FULL ? true : (xq < qw && yq < qh);
        B.smask[k] = 0xffffffffu;
        if (!FULL && level > 1) {
            B.act[k] = xq < ((bw + 1) >> 1) && yq < ((bh + 1) >> 1);
            B.smask[k] = (((bw & 1) && xq == qw) ? 0x00ff00ffu : 0xffffffffu) & (((bh & 1) && yq == qh) ? 0x0000ffffu : 0xffffffffu);
        }
        B.a[k] = ldq(at(src, bx + 16 * (k & 1), by + 16 * (k >> 1)), src.stride, B.qi, B.qj, B.act[k]);
        B.a[k].w &= B.smask[k];
        if (with_other) {
            o[k] = ldq(at(oth, bx + 16 * (k & 1), by + 16 * (k >> 1)), oth.stride, B.qi, B.qj, B.act[k]);
        }
    }
    return B;
}
template <bool FULL, int NQ> __device__ __forceinline__ SrcBlk<NQ> load_src_blk(const DPlane &src, int bx, int by, int bw, int bh, int level)
{
    Quad unused[NQ];
    return load_src_blk<FULL, NQ>(src, bx, by, bw, bh, level, false, src, unused);
}

template <bool FULL, int NQ, class Ctx>
__device__ __forceinline__ void hme_block_lx_t(const Ctx &c_in, int level, int i, int j, int gx, int gy, FastLds &S, RowAcc &acc)
{
    Ctx c = fenced(c_in, level);
    const int lane = hme_lane();
    const int qi = lane & 7, qj = lane >> 3;
    const int nxb = c.a.nbh, nyb = c.a.nbv, y_w = BlkDim<NQ>::W, y_h = BlkDim<NQ>::H;
    const int step = 1 << level;
    const DPlane src = c.src[level], ogr = c.ogr[level];
    DPlane ref = c.ref[level];
    DSV_MV *mvf = c.mvf[level];
    const DSV_MV *parent = level < c.pyr_levels ? c.mvf[level + 1] : nullptr;
    DSV_MV *out = &mvf[i + j * nxb];
    DSV_MV mv = {};

    const int bx = (i * y_w) >> level, by = (j * y_h) >> level;
    const int bw = FULL ? y_w : min(src.w - bx, y_w), bh = FULL ? y_h : min(src.h - by, y_h);
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) v4i_t *cv4i_t;
    const bool have_pre = level <= 1;
    v4i_t pre_words = {0, 0, 0, 0};
    Quad o_zero[NQ]; // for the "good enough" test far below: same load round (level 1: its outcome comes with the statistics)
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        o_zero[k].w = 0;
    }
    if (have_pre) {
        pre_words = *(cv4i_t) &c.stats[(i >> level) + (j >> level) * ((nxb + step - 1) >> level)];
    }
    const SrcBlk<NQ> B = load_src_blk<FULL, NQ>(src, bx, by, bw, bh, level, !have_pre, ogr, o_zero);
    // ONE load round for every vector the list reads: lanes 3..5 the same-level neighbours, lanes 6..14 the co-located
    // vectors of the previous frame, lanes 16..24 the parent level's.  Lanes without a vector load this block's own entry.
    bool nb_ok = false, pvalid = false, tvalid = false;
    uint32_t ov;
    {
        const DSV_MV *op = out;
        if (parent != nullptr) {
            unsigned parent_mask = ~(((unsigned) step << 1) - 1);
            int pi = (int) ((unsigned) i & parent_mask), pj = (int) ((unsigned) j & parent_mask);
            if (lane >= 16 && lane < 25) {
                int m = lane - 16;
                int x = pi + 2 * tab9(kParX, m) * step, y = pj + 2 * tab9(kParY, m) * step;
                if (x >= 0 && x < nxb && y >= 0 && y < nyb) {
                    op = &parent[x + y * nxb];
                    pvalid = true;
                }
            } else if (lane >= 6 && lane <= 14 && c.ref_mvf != nullptr) {
                int k = lane - 6;
                int rx = i + tab9(kRectX, k) * step, ry = j + tab9(kRectY, k) * step;
                if (rx >= 0 && ry >= 0 && rx < nxb && ry < nyb) {
                    op = &c.ref_mvf[rx + ry * nxb];
                    tvalid = true;
                }
            }
        }
        typedef const __attribute__((address_space(1))) uint32_t *gu32p_t;
        ov = *(gu32p_t) op;
    }
    const MvHead nbv = load_neighbour_heads(mvf, out, i, j, step, nxb, c.counters, acc, nb_ok);
    int pvx = (int) (int16_t) (ov & 0xffffu), pvy = (int) (int16_t) (ov >> 16);
    int motion_bias = y_w * y_h;
    unsigned var_src = 0;
    Psy psy = {2, 1, 0};
    if (level <= 1) {
        var_src = (unsigned) pre_words.y;
        motion_bias = (int) udiv_fast((unsigned) max(pre_words.x, 0), (unsigned) (2 + (abs(gx) + abs(gy))));
        if (var_src <= (unsigned) (8 * bw * bh * c.quant >> 9)) {
            motion_bias = 0;
        }
        psy = psy_of_source(var_src, bw, bh, c.quant);
    }
    // ---- candidate gathering: lane p owns canonical list position p (hme.c:1443-1528) ----
    //  0 zero | 1 parent inlier average | (2 predictor: level 0 only) | 3 left 4 top 5 top-left |
    //  6..14 temporal | 15 global | 16..24 parent inliers
    int lax = 0, lay = 0;
    bool exist = lane == 0;
    int cxv = 0, cyv = 0;
    // dsv_movec_pred (dsv.c:375) at the coarser levels reads entries between the level's grid points, which are never written (zero)
    CostCtx cc;
    cc.px = cc.py = 0;
    if (parent != nullptr) {
        bool inl;
        int nin;
        if (parent_average_cached(acc, (int) ((unsigned) i & ~(((unsigned) step << 1) - 1)), pvalid, pvx, pvy, lax, lay, inl, nin)) {
            // every list entry passes through an int16 store and the qpel->fpel rounding (hme.c:1185-1200)
            if (lane == 1) {
                exist = true;
                cxv = qp2fp((int16_t) (lax * 4));
                cyv = qp2fp((int16_t) (lay * 4));
            } else if (nb_ok) {
                exist = true;
                cxv = qp2fp(nbv.x);
                cyv = qp2fp(nbv.y);
            } else if (tvalid) {
                exist = true;
                cxv = qp2fp(pvx);
                cyv = qp2fp(pvy);
            } else if (lane == 15) {
                exist = true;
                cxv = qp2fp((int16_t) (gx * 4));
                cyv = qp2fp((int16_t) (gy * 4));
            } else if (lane >= 16 && lane < 25 && nin && inl) {
                exist = true;
                cxv = qp2fp((int16_t) (pvx * 4));
                cyv = qp2fp((int16_t) (pvy * 4));
            }
        }
    }
    cxv = (int) (int16_t) ((int) (int16_t) cxv >> level);
    cyv = (int) (int16_t) ((int) (int16_t) cyv >> level);
    // first-occurrence de-duplication (hme.c:1166); a lane = a canonical list position, so "first wins" among equal scores is
    // "lowest lane wins" and the list never has to be compacted
    const int key = (cxv & 0xffff) | (int) ((unsigned) cyv << 16); // both components are int16 by now
    const bool keep = exist && !dedup_lanes(exist, key);
    cc.q = c.quant;
    cc.b2sr = b2sr_of(c);
    // ---- best candidate (hme.c:1530-1557) ----
    int dx, dy;
    unsigned best;
    {
        unsigned raw = score_lanes<NQ>(__ballot(keep), key, ref, B, level, psy);
        const int mx = cxv, my = cyv;
        bool valid = keep && !invalid_block(ref, bx + mx, by + my, bw, bh, 0);
        if (level <= 1) {
            raw = metric_return(raw, bw, bh);
        }
        unsigned sc = raw + (unsigned) mv_cost(cc, mx * step * 4, my * step * 4, level);
        if (mx == lax && my == lay) {
            sc = (unsigned) max((int) sc - (motion_bias >> level), 0);
        }
        if (!valid) {
            sc = 0xffffffffu;
        }
        unsigned mn = wave_min_u(sc);
        unsigned long long hit = __ballot(valid && sc == mn);
        int best_k = (mn != 0xffffffffu && hit) ? (int) __ffsll((long long) hit) - 1 : 0;
        best = mn;
        dx = __builtin_amdgcn_readlane(mx, best_k);
        dy = __builtin_amdgcn_readlane(my, best_k);
    }
    unsigned qthresh = (unsigned) (c.quant * bw * bh >> 11);
    bool good_enough = false;
    {
        unsigned zoscore = (unsigned) pre_words.w;
        if (!have_pre) { // (the psy metric of the squared-error levels' "good enough" test: whole quads only)
            unsigned zp = 0;
            const int qw = bw >> 1, qh = bh >> 1;
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                const bool whole = FULL || (8 * (k & 1) + qi < qw && 8 * (k >> 1) + qj < qh);
                zp += whole ? qmetric(B.a[k], o_zero[k], psy) : 0u;
            }
            zoscore = metric_return(wave_sum(zp), bw, bh);
        }
        if (abs(dx) <= 1 && abs(dy) <= 1) {
            qthresh *= 2;
        }
        if (zoscore < qthresh) {
            best = 0;
            dx = dy = 0;
            good_enough = true;
        }
    }
    if (!good_enough) {
        c = fenced(c_in, level);
        ref = c.ref[level];
        refine_fpel<false, NQ>(ref, B, level, psy, cc, qthresh, dx, dy, best, good_enough, S);
    }
    mv.u.mv.x = (int16_t) (dx * step);
    mv.u.mv.y = (int16_t) (dy * step);
    if (lane == 0) {
        st_mv(out, mv);
    }
    acc.left_head = (unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) mv.u.all) |
                    ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) mv.flags) << 32);
}

template <int NQ, class Ctx> __device__ __forceinline__ void hme_block_lx(const Ctx &c, int level, int i, int j, int gx, int gy, FastLds &S, RowAcc &acc)
{
    constexpr int BW = BlkDim<NQ>::W, BH = BlkDim<NQ>::H;
    const DPlane &src = c.src[level];
    int bx = (i * BW) >> level, by = (j * BH) >> level;
    if (src.w - bx >= BW && src.h - by >= BH) {
        hme_block_lx_t<true, NQ>(c, level, i, j, gx, gy, S, acc);
    } else {
        hme_block_lx_t<false, NQ>(c, level, i, j, gx, gy, S, acc);
    }
}

// ============================================================================================================================
// Level 0, first half: what a block's search can know BEFORE its same-level neighbours are known.
// The row pipeline of the search is serial along a row and down the rows; of a level-0 block's work only the neighbour-derived
// list entries (predictor, left, top, top-left), the vector COSTS (their predictor is the neighbours' median), the refinement
// and what follows it depend on that order.  The rest -- the parent level's half of the list (inlier average, inliers), the
// temporal and global entries, the de-duplicated list's SCORES, and the pixel half of the sub-pel search around the parent
// average (four squared errors, the 34x34 half-pel image, seven probes) -- is a function of the source, the reference, the
// level above and the previous frame's field alone: k_hme_l0_pre_b works it out for every block in any order, at full
// occupancy, right after level 1 has finished, and leaves a 256-byte record per block:
//   dwords  0..41  21 list entries {key = x | y << 16 (int16 each, full-pel), rf} for canonical positions 0, 1, 6 .. 24
//                  (entry e = p < 2 ? p : p - 4); rf = kL0Absent: no such entry (or a duplicate of an earlier one), else the
//                  normalised metric (metric_return: < 2^24) | kL0Invalid if the block at the vector leaves the padded plane
//   dwords 48..54  normalised metrics of the seven sub-pel probes around the parent average
//   dword  56      parent average: lax | lay << 16 (int16 each)
//   dword  57      bit 0: the list is open (a parent level exists and has vectors here: hme.c:1471); bit 1: the sub-pel probes were
//                  run; bit 2: the co-located vector is valid; bits 8..15: the probes' directions (subpel_probes)
//   dword  58      the co-located vector of the previous frame's field ({x, y}), for the mode decision (hme.c:896)
// ============================================================================================================================
constexpr int kL0RecDwords = 64;
constexpr uint32_t kL0Absent = 0xffffffffu, kL0Invalid = 0x40000000u, kL0ScoreMask = 0x00ffffffu;
constexpr unsigned long long kL0PreLanes = 0x1ffffc3ull; // lanes that own a pre-pass entry: 0, 1, 6 .. 24

template <bool FULL, class Ctx> __device__ __forceinline__ void hme_l0_pre_block_t(const Ctx &c, int i, int j, int gx, int gy, FastLds &S, uint32_t *rec)
{
    const int lane = hme_lane();
    const int qi = lane & 7, qj = lane >> 3;
    const int nxb = c.a.nbh, nyb = c.a.nbv;
    const DPlane &src = c.src[0], &ref = c.ref[0];
    const DSV_MV *parent = c.pyr_levels > 0 ? c.mvf[1] : nullptr;
    const int bx = i * 16, by = j * 16;
    const int bw = FULL ? 16 : min(src.w - bx, 16), bh = FULL ? 16 : min(src.h - by, 16);
    const SrcBlk<1> B = load_src_blk<FULL, 1>(src, bx, by, bw, bh, 0);
    const bool act = B.act[0];
    const Quad a = B.a[0];
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) v4i_t *cv4i_t;
    const v4i_t pre_words = *(cv4i_t) &c.stats[i + j * nxb];
    bool pvalid = false, tvalid = false;
    uint32_t ov;
    {
        const DSV_MV *op = &c.mvf[0][i + j * nxb]; // (lanes without a vector: any readable word)
        if (parent != nullptr) {
            const int pi = i & ~1, pj = j & ~1;
            if (lane >= 16 && lane < 25) {
                int m = lane - 16;
                int x = pi + 2 * tab9(kParX, m), y = pj + 2 * tab9(kParY, m);
                if (x >= 0 && x < nxb && y >= 0 && y < nyb) {
                    op = &parent[x + y * nxb];
                    pvalid = true;
                }
            } else if (lane >= 6 && lane <= 14 && c.ref_mvf != nullptr) {
                int k = lane - 6;
                int rx = i + tab9(kRectX, k), ry = j + tab9(kRectY, k);
                if (rx >= 0 && ry >= 0 && rx < nxb && ry < nyb) {
                    op = &c.ref_mvf[rx + ry * nxb];
                    tvalid = true;
                }
            }
        }
        typedef const __attribute__((address_space(1))) uint32_t *gu32p_t;
        ov = *(gu32p_t) op;
    }
    const int pvx = (int) (int16_t) (ov & 0xffffu), pvy = (int) (int16_t) (ov >> 16);
    const Psy psy = psy_of_source((unsigned) pre_words.y, bw, bh, c.quant);
    int lax = 0, lay = 0;
    bool exist = lane == 0, open = false;
    int cxv = 0, cyv = 0;
    if (parent != nullptr) {
        bool inl;
        int nin;
        open = parent_average(pvalid, pvx, pvy, lax, lay, inl, nin);
        if (open) {
            if (lane == 1) {
                exist = true;
                cxv = qp2fp((int16_t) (lax * 4));
                cyv = qp2fp((int16_t) (lay * 4));
            } else if (tvalid) {
                exist = true;
                cxv = qp2fp(pvx);
                cyv = qp2fp(pvy);
            } else if (lane == 15) {
                exist = true;
                cxv = qp2fp((int16_t) (gx * 4));
                cyv = qp2fp((int16_t) (gy * 4));
            } else if (lane >= 16 && lane < 25 && nin && inl) {
                exist = true;
                cxv = qp2fp((int16_t) (pvx * 4));
                cyv = qp2fp((int16_t) (pvy * 4));
            }
        }
    }
    cxv = (int) (int16_t) cxv;
    cyv = (int) (int16_t) cyv;
    const int key = (cxv & 0xffff) | (int) ((unsigned) cyv << 16);
    const bool keep = exist && !dedup_lanes(exist, key);
    const unsigned raw = metric_return(score_lanes<1>(__ballot(keep), key, ref, B, 0, psy), bw, bh);
    uint32_t rf = kL0Absent;
    if (keep) {
        rf = (raw & kL0ScoreMask) | (invalid_block(ref, bx + cxv, by + cyv, bw, bh, 0) ? kL0Invalid : 0u);
    }
    if ((kL0PreLanes >> lane) & 1ull) {
        const int e = lane < 2 ? lane : lane - 4;
        *(uint2 *) (rec + 2 * e) = uint2{(uint32_t) key, rf};
    }
    // the pixel half of the sub-pel search around the parent average (hme.c:1598-1612)
    const bool sp_done = c.effort >= 4 && !invalid_block(ref, bx + lax, by + lay, bw, bh, 4);
    unsigned dirs = 0;
    if (sp_done) {
        const unsigned mr = FULL ? subpel_probes_patch(c, lax, lay, bx, by, a, qi, qj, psy, dirs) : subpel_probes(c, S, lax, lay, bx, by, bw, bh, a, act, qi, qj, psy, dirs);
        if (lane < 7) {
            rec[48 + lane] = mr;
        }
    }
    if (lane == 0) {
        const bool colo_ok = parent != nullptr && c.ref_mvf != nullptr;
        uint32_t colo = 0;
        if (c.ref_mvf != nullptr) {
            colo = *(const uint32_t *) &c.ref_mvf[i + j * nxb];
        }
        *(uint4 *) (rec + 56) = uint4{(uint32_t) (lax & 0xffff) | ((uint32_t) lay << 16),
                                      (open ? 1u : 0u) | (sp_done ? 2u : 0u) | (colo_ok ? 4u : 0u) | (dirs << 8), colo, 0u};
    }
}

template <class Ctx> __device__ __forceinline__ void hme_l0_pre_block(const Ctx &c, int i, int j, int gx, int gy, FastLds &S, uint32_t *rec)
{
    const DPlane &src = c.src[0];
    if (src.w - i * 16 >= 16 && src.h - j * 16 >= 16) {
        hme_l0_pre_block_t<true>(c, i, j, gx, gy, S, rec);
    } else {
        hme_l0_pre_block_t<false>(c, i, j, gx, gy, S, rec);
    }
}

// ============================================================================================================================
// Level 0, second half: the block of the row pipeline.  ONE load round brings the source quads, the pre-pass record and the
// neighbours' heads; the neighbour-derived entries (positions 2..5) are merged into the list by canonical position (a lane =
// a position, so "first wins" among equal scores is "lowest lane wins"); an entry whose vector the pre-pass has already scored
// takes that score, and only a vector nobody has seen costs a load round of its own.
// CS: chroma shift of both axes (mode decision): 1 = 4:2:0, 0 = 4:4:4.
// ============================================================================================================================
// SPLIT: the neighbour-independent half comes from the pre-pass record (k_hme_l0_pre_b); else it is worked out here, in place:
// one kernel does everything, less work in total (no record, one set of loads) but a longer chain per block.  Which is better
// depends on what the GPU is short of: with many pictures per launch and several lockstep groups it is work, with a few it is the chain.
template <bool FULL, int CS, bool SPLIT, class Ctx>
__device__ __forceinline__ void hme_block_l0_t(const Ctx &c_in, int i, int j, int gx, int gy, FastLds &S, RowAcc &acc)
{
    Ctx c = fenced(c_in, 0);
    const int lane = hme_lane();
    const int qi = lane & 7, qj = lane >> 3;
    const int nxb = c.a.nbh, nyb = c.a.nbv;
    const DPlane src = c.src[0];
    DPlane ref = c.ref[0];
    DSV_MV *mvf = c.mvf[0];
    DSV_MV *out = &mvf[i + j * nxb];
    DSV_MV mv = {};
    HME_COUNT(S, 10, 1);
    const int bx = i * 16, by = j * 16;
    const int bw = FULL ? 16 : min(src.w - bx, 16), bh = FULL ? 16 : min(src.h - by, 16);
    const SrcBlk<1> B = load_src_blk<FULL, 1>(src, bx, by, bw, bh, 0);
    const bool act = B.act[0];
    const Quad a = B.a[0];
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) v4i_t *cv4i_t;
    typedef const __attribute__((address_space(1))) uint32_t *gu32p_t;
    typedef const __attribute__((address_space(1))) uint2v_t *gu2p_t;
    const v4i_t pre_words = *(cv4i_t) &c.stats[i + j * nxb];
    const bool pre_lane = ((kL0PreLanes >> lane) & 1ull) != 0;
    // ---- the block's first load round ----
    v4i_t hdr = {0, 0, 0, 0};
    uint2v_t ent = {0u, 0u};
    unsigned sp_mr = 0;
    uint32_t ov = 0;
    bool pvalid = false, tvalid = false;
    const DSV_MV *parent = nullptr;
    if constexpr (SPLIT) {
        const uint32_t *rec = c.l0pre + (size_t) (i + j * nxb) * kL0RecDwords;
        hdr = *(cv4i_t) (rec + 56);
        ent = *(gu2p_t) (rec + 2 * (pre_lane ? (lane < 2 ? lane : lane - 4) : 0));
        sp_mr = *(gu32p_t) (rec + 48 + (lane < 7 ? lane : 0));
    } else {
        // lanes 6..14 fetch the co-located vectors of the previous frame, lanes 16..24 the parent level's (hme.c:1443-1528)
        parent = c.pyr_levels > 0 ? c.mvf[1] : nullptr;
        const DSV_MV *op = out;
        if (parent != nullptr) {
            const int pi = i & ~1, pj = j & ~1;
            if (lane >= 16 && lane < 25) {
                int m = lane - 16;
                int x = pi + 2 * tab9(kParX, m), y = pj + 2 * tab9(kParY, m);
                if (x >= 0 && x < nxb && y >= 0 && y < nyb) {
                    op = &parent[x + y * nxb];
                    pvalid = true;
                }
            } else if (lane >= 6 && lane <= 14 && c.ref_mvf != nullptr) {
                int k = lane - 6;
                int rx = i + tab9(kRectX, k), ry = j + tab9(kRectY, k);
                if (rx >= 0 && ry >= 0 && rx < nxb && ry < nyb) {
                    op = &c.ref_mvf[rx + ry * nxb];
                    tvalid = true;
                }
            }
        }
        ov = *(gu32p_t) op;
    }
    bool nb_ok = false;
    const MvHead nbv = load_neighbour_heads(mvf, out, i, j, 1, nxb, c.counters, acc, nb_ok);

    const unsigned var_src = (unsigned) pre_words.y, avg_src = (unsigned) pre_words.z;
    int motion_bias = (int) udiv_fast((unsigned) max(pre_words.x, 0), (unsigned) (2 + (abs(gx) + abs(gy))));
    if (var_src <= (unsigned) (8 * bw * bh * c.quant >> 9)) {
        motion_bias = 0;
    }
    const Psy psy = psy_of_source(var_src, bw, bh, c.quant);
    HME_MARK(S, 1);
    // dsv_movec_pred (dsv.c:375): the median predictor of the three neighbours just loaded
    CostCtx cc;
    {
        int v0 = __builtin_amdgcn_readlane((int) nbv.all, 3), v1 = __builtin_amdgcn_readlane((int) nbv.all, 4),
            v2 = __builtin_amdgcn_readlane((int) nbv.all, 5);
        v0 = i > 0 ? v0 : 0;
        v1 = j > 0 ? v1 : 0;
        v2 = i > 0 && j > 0 ? v2 : 0;
        cc.px = pred1((int) (int16_t) (v0 & 0xffff), (int) (int16_t) (v1 & 0xffff), (int) (int16_t) (v2 & 0xffff));
        cc.py = pred1(v0 >> 16, v1 >> 16, v2 >> 16);
    }
    cc.q = c.quant;
    cc.b2sr = b2sr_of(c);
    // ---- the list by canonical position (a lane = a position, so "first wins" among equal scores is "lowest lane wins"):
    //  0 zero | 1 parent inlier average | 2 predictor | 3 left 4 top 5 top-left | 6..14 temporal | 15 global | 16..24 parent inliers
    int lax, lay;
    bool open, sp_done = false, colo_ok;
    unsigned sp_dirs = 0;
    uint32_t colo;
    bool exist;
    int key;
    uint32_t rf; // normalised metric of this lane's vector | kL0Invalid, once known
    if constexpr (SPLIT) {
        lax = (int) (int16_t) ((uint32_t) hdr.x & 0xffffu);
        lay = (int) (int16_t) ((uint32_t) hdr.x >> 16);
        open = (hdr.y & 1) != 0;
        sp_done = (hdr.y & 2) != 0;
        colo_ok = (hdr.y & 4) != 0;
        sp_dirs = ((uint32_t) hdr.y >> 8) & 0xffu;
        colo = (uint32_t) hdr.z;
        exist = pre_lane && ent[1] != kL0Absent;
        key = (int) ent[0];
        rf = ent[1];
    } else {
        const int pvx = (int) (int16_t) (ov & 0xffffu), pvy = (int) (int16_t) (ov >> 16);
        colo = (uint32_t) __builtin_amdgcn_readlane((int) ov, 6);
        colo_ok = parent != nullptr && c.ref_mvf != nullptr;
        bool inl = false;
        int nin = 0;
        lax = lay = 0;
        open = parent != nullptr && parent_average_cached(acc, i & ~1, pvalid, pvx, pvy, lax, lay, inl, nin);
        exist = lane == 0;
        int cxv = 0, cyv = 0;
        if (open) { // every list entry passes through an int16 store and the qpel->fpel rounding (hme.c:1185-1200)
            if (lane == 1) {
                exist = true;
                cxv = qp2fp((int16_t) (lax * 4));
                cyv = qp2fp((int16_t) (lay * 4));
            } else if (tvalid) {
                exist = true;
                cxv = qp2fp(pvx);
                cyv = qp2fp(pvy);
            } else if (lane == 15) {
                exist = true;
                cxv = qp2fp((int16_t) (gx * 4));
                cyv = qp2fp((int16_t) (gy * 4));
            } else if (lane >= 16 && lane < 25 && nin && inl) {
                exist = true;
                cxv = qp2fp((int16_t) (pvx * 4));
                cyv = qp2fp((int16_t) (pvy * 4));
            }
        }
        key = ((int) (int16_t) cxv & 0xffff) | (int) ((unsigned) (int) (int16_t) cyv << 16);
        rf = 0;
    }
    if (open && (lane == 2 || nb_ok)) { // the neighbour-derived entries (nb_ok: lanes 3..5 with a neighbour)
        const int vx = lane == 2 ? (int) (int16_t) cc.px : nbv.x, vy = lane == 2 ? (int) (int16_t) cc.py : nbv.y;
        const int cxv = (int) (int16_t) qp2fp(vx), cyv = (int) (int16_t) qp2fp(vy);
        key = (cxv & 0xffff) | (int) ((unsigned) cyv << 16);
        exist = true;
        rf = 0;
    }
    // first occurrence wins (hme.c:1166)
    bool keep;
    unsigned long long want = 0; // lanes whose vector still needs its score
    if constexpr (SPLIT) {
        // an entry whose vector the pre-pass has scored further down the list takes that score
        bool dup = false;
        for (unsigned long long rest = __ballot(exist); rest;) {
            const int m = __ffsll((long long) rest) - 1;
            const int km = __builtin_amdgcn_readlane(key, m);
            const bool same = exist && km == key;
            const unsigned long long sm = __ballot(same);
            dup = dup || (same && lane != m);
            if (m >= 2 && m <= 5) {
                const unsigned long long pm = sm & kL0PreLanes;
                if (pm) {
                    const uint32_t known = (uint32_t) __builtin_amdgcn_readlane((int) rf, __ffsll((long long) pm) - 1);
                    rf = lane == m ? known : rf;
                } else {
                    want |= 1ull << m;
                }
            }
            rest &= ~sm;
        }
        keep = exist && !dup;
    } else {
        keep = exist && !dedup_lanes(exist, key);
        want = __ballot(keep);
    }
    const int mx = (int) (int16_t) (key & 0xffff), my = key >> 16;
    // ONE load round for the vectors nobody has scored yet (SPLIT: usually none)
    if (want) {
        HME_COUNT(S, 14, __popcll(want));
        const unsigned raw = score_lanes<1>(want, key, ref, B, 0, psy);
        if ((want >> lane) & 1ull) {
            rf = (metric_return(raw, bw, bh) & kL0ScoreMask) | (invalid_block(ref, bx + mx, by + my, bw, bh, 0) ? kL0Invalid : 0u);
        }
    }
    HME_MARK(S, 2);
    // ---- best candidate (hme.c:1530-1557) ----
    int dx, dy;
    unsigned best, score_zero;
    {
        const bool valid = keep && !(rf & kL0Invalid);
        const unsigned raw = rf & kL0ScoreMask;
        unsigned sc = raw + (unsigned) mv_cost(cc, mx * 4, my * 4, 0);
        if (mx == lax && my == lay) {
            sc = (unsigned) max((int) sc - motion_bias, 0);
        }
        if (!valid) {
            sc = 0xffffffffu;
        }
        unsigned mn = wave_min_u(sc);
        unsigned long long hit = __ballot(valid && sc == mn);
        int best_k = (mn != 0xffffffffu && hit) ? (int) __ffsll((long long) hit) - 1 : 0;
        best = mn;
        bool z_valid = __builtin_amdgcn_readlane((int) valid, 0) != 0;
        unsigned z_raw = (unsigned) __builtin_amdgcn_readlane((int) raw, 0);
        score_zero = z_valid ? z_raw : 0xffffffffu;
        dx = __builtin_amdgcn_readlane(mx, best_k);
        dy = __builtin_amdgcn_readlane(my, best_k);
    }
    unsigned qthresh = (unsigned) (c.quant * bw * bh >> 11);
    bool good_enough = false;
    {
        const unsigned zoscore = (unsigned) pre_words.w;
        if (abs(dx) <= 1 && abs(dy) <= 1) {
            qthresh *= 2;
        }
        if (zoscore < qthresh) {
            best = score_zero;
            dx = dy = 0;
            good_enough = true;
        }
    }
    HME_MARK(S, 3);
    c = fenced(c_in, 0);
    ref = c.ref[0];
    if (!good_enough) {
        refine_fpel<true, 1>(ref, B, 0, psy, cc, qthresh, dx, dy, best, good_enough, S);
    }
    HME_MARK(S, 4);
    mv.u.mv.x = (int16_t) dx;
    mv.u.mv.y = (int16_t) dy;
    NbPre pre;
    pre.l_all = (uint32_t) __builtin_amdgcn_readlane((int) nbv.all, 3);
    pre.l_flags = (uint32_t) __builtin_amdgcn_readlane((int) nbv.flags, 3);
    pre.t_all = (uint32_t) __builtin_amdgcn_readlane((int) nbv.all, 4);
    pre.t_flags = (uint32_t) __builtin_amdgcn_readlane((int) nbv.flags, 4);
    pre.colo = colo;
    pre.colo_ok = colo_ok;
    hme_l0_tail<CS, SPLIT>(c, i, j, S, acc, out, mv, cc, a, act, qi, qj, bx, by, bw, bh, lax, lay, motion_bias, good_enough, best, var_src, avg_src, psy, pre,
                           sp_done, sp_mr, sp_dirs);
}

template <int CS, bool SPLIT, class Ctx> __device__ __forceinline__ void hme_block_l0(const Ctx &c, int i, int j, int gx, int gy, FastLds &S, RowAcc &acc)
{
    const DPlane &src = c.src[0];
    if (src.w - i * 16 >= 16 && src.h - j * 16 >= 16) {
        hme_block_l0_t<true, CS, SPLIT>(c, i, j, gx, gy, S, acc);
    } else {
        hme_block_l0_t<false, CS, SPLIT>(c, i, j, gx, gy, S, acc);
    }
}
