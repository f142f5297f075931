// hme.hip -- hierarchical block motion estimation and P-frame mode decision on gfx950.
//
// Replaces reference src/hme.c: dsv_hme (:2001), refine_level (:1373), refine_best_fpel_cand
// (:1301), subpixel_ME (:1052) with hpel/qpel/qpsad (:788/:816/:245), the block metrics
// METR_BODY (:136) / SSE_BODY (:198), the mode decision (:1636-1821: skip / no-transmit /
// EPRM / sub-block intra tests :892/:988, calc_EPRM :453, yuv_max_subblock_err :370) and
// global_motion (:1974).  CPU proof of the decomposition: oracle/orc_hme.c.
//
// Mapping to the machine:
//   * one 64-lane wavefront (= one workgroup) per block.  The block's control flow is
//     wave-uniform scalar code; every pixel-touching primitive (psy metric, SSE, block
//     statistics, half-pel interpolation, clip tests) is evaluated cooperatively: a 16x16
//     block is exactly 64 2x2 quads = one quad per lane, partial sums are combined with a
//     6-step cross-lane butterfly (__shfl_xor) and broadcast to all lanes;
//   * a block reads the final vectors of its left / top / top-left neighbours of the same
//     level, so blocks on one anti-diagonal (i/step + j/step = t) are independent: each
//     front is one kernel launch (grid = blocks on the front), launch order gives the
//     inter-front ordering with no in-kernel spinning;
//   * the 68x68 quarter-pel image of the reference is never materialised: the 34x34
//     half-pel image lives in LDS and quarter-pel samples are averaged on the fly.
#include "blockstat.h"
#include "codec.h"
#include "hme.h"
#include "prio.h"

namespace dsv2 {

struct HmeDev {
    AnalysisParams a;
    int effort, lossless, quant, skip_block_thresh, pyr_levels, psyscale;
    int b2sr; // dsv_mv_cost's bits-to-SSE ratio (hme.c:163): one value per frame
    DPlane src[6], ref[6], ogr[6];
    DPlane srcc[2], refc[2];
    DSV_MV *mvf[6];
    const DSV_MV *ref_mvf;
    int *counters; // [0] nintra [1] ndiff [2] eligible [3] total_err [4] gx [5] gy
    DSV_MV *host_mvs;   // optional pinned host mirror of the level-0 field (written straight over PCIe)
    int *host_counters; // optional pinned host copy of counters[0..7]
    int4 *stats[2];     // source statistics of every block of levels 0 and 1 (k_hme_src_stats_b), or null (general routine)
    uint32_t *l0pre;    // level 0's pre-pass records (k_hme_l0_pre_b: 64 dwords per block), or null (general routine)
};

// What the blocks of ONE level need, copied out of the job table once per row as wave-uniform scalars: the table
// sits in global memory, and after every hand-off fence its fields would otherwise be reloaded (vector loads, in
// front of each block).  Same member names as HmeDev; src[] / ref[] / ogr[] answer for this level and for
// level 0, mvf[] for this level and its parent; the chroma planes are picked by select, not by indexing.
// (uni / uni_ptr: dev.h)

struct HmeCtx {
    struct Planes {
        DPlane lvl, zero;
        __device__ __forceinline__ DPlane operator[](int l) const
        {
            bool z = l == 0; // by value, field by field: a reference to either member would force the struct into memory
            return DPlane{z ? zero.data : lvl.data, z ? zero.stride : lvl.stride, z ? zero.w : lvl.w, z ? zero.h : lvl.h};
        }
    };
    struct Pair { // two planes, selected per lane
        DPlane p0, p1;
        __device__ __forceinline__ DPlane operator[](int k) const
        {
            return DPlane{k ? p1.data : p0.data, k ? p1.stride : p0.stride, k ? p1.w : p0.w, k ? p1.h : p0.h};
        }
    };
    struct Fields {
        DSV_MV *cur, *parent;
        int level;
        __device__ __forceinline__ DSV_MV *operator[](int l) const { return l == level ? cur : parent; }
    };
    AnalysisParams a;
    int effort, lossless, quant, skip_block_thresh, pyr_levels, psyscale, b2sr;
    Planes src, ref, ogr;
    Pair srcc, refc;
    Fields mvf;
    const DSV_MV *ref_mvf;
    int *counters;
    DSV_MV *host_mvs;
    const int4 *stats; // this level's source statistics or null
    uint32_t *l0pre;   // level 0's pre-pass records
    const HmeDev *self; // the job record all of the above was read from
};

__device__ __forceinline__ const int4 *src_stats_of(const HmeCtx &c, int) { return c.stats; }
__device__ __forceinline__ const int4 *src_stats_of(const HmeDev &c, int level) { return c.stats[level]; }

// The job record is read through the SCALAR cache (constant address space: the record is written by the host before the launch and
// never during it), so a field costs a fraction of an s_load and no vector register -- and the context can be read AGAIN wherever a
// phase of a block starts (fenced(), below) instead of being carried through the whole persistent loop in scalar registers the
// machine does not have (102 per wavefront; the context and what the compiler derives from it up front came to ~380, of which
// it spilled 280 to vector-register lanes: two v_readlane per use).
typedef const __attribute__((address_space(4))) HmeDev *HmeDevK;
template <class T> __device__ __forceinline__ T *from_k(const __attribute__((address_space(4))) void *p) { return (T *) (unsigned long long) p; }
__device__ __forceinline__ HmeCtx make_ctx(const HmeDev *self, int level)
{
    const HmeDev *p = self;
    asm volatile("" : "+s"(p)); // (the compiler may not merge this read with an earlier one)
    HmeDevK d = (HmeDevK) (unsigned long long) p;
    HmeCtx c;
    c.self = self;
    c.a.width = d->a.width;
    c.a.height = d->a.height;
    c.a.blk_w = d->a.blk_w;
    c.a.blk_h = d->a.blk_h;
    c.a.nbh = d->a.nbh;
    c.a.nbv = d->a.nbv;
    c.a.hshift = d->a.hshift;
    c.a.vshift = d->a.vshift;
    c.a.do_psy = d->a.do_psy;
    c.a.scale = d->a.scale;
    c.effort = d->effort;
    c.lossless = d->lossless;
    c.quant = d->quant;
    c.b2sr = d->b2sr;
    c.skip_block_thresh = d->skip_block_thresh;
    c.pyr_levels = d->pyr_levels;
    c.psyscale = d->psyscale;
    // the source, reference and original-reference pyramids have one geometry per level, and so have the four
    // chroma planes (hme_run_batch checks it): only the data pointers differ, the rest shares scalar registers
    c.src.lvl = DPlane{d->src[level].data, d->src[level].stride, d->src[level].w, d->src[level].h};
    c.ref.lvl = DPlane{d->ref[level].data, c.src.lvl.stride, c.src.lvl.w, c.src.lvl.h};
    c.ogr.lvl = DPlane{d->ogr[level].data, c.src.lvl.stride, c.src.lvl.w, c.src.lvl.h};
    c.src.zero = DPlane{d->src[0].data, d->src[0].stride, d->src[0].w, d->src[0].h};
    c.ref.zero = DPlane{d->ref[0].data, c.src.zero.stride, c.src.zero.w, c.src.zero.h};
    c.ogr.zero = DPlane{d->ogr[0].data, c.src.zero.stride, c.src.zero.w, c.src.zero.h};
    c.srcc.p0 = DPlane{d->srcc[0].data, d->srcc[0].stride, d->srcc[0].w, d->srcc[0].h};
    c.srcc.p1 = DPlane{d->srcc[1].data, c.srcc.p0.stride, c.srcc.p0.w, c.srcc.p0.h};
    c.refc.p0 = DPlane{d->refc[0].data, c.srcc.p0.stride, c.srcc.p0.w, c.srcc.p0.h};
    c.refc.p1 = DPlane{d->refc[1].data, c.srcc.p0.stride, c.srcc.p0.w, c.srcc.p0.h};
    c.mvf.cur = d->mvf[level];
    c.mvf.parent = level < c.pyr_levels ? d->mvf[level + 1] : nullptr;
    c.mvf.level = level;
    c.ref_mvf = d->ref_mvf;
    c.counters = d->counters;
    c.host_mvs = d->host_mvs;
    c.stats = level <= 1 ? (const int4 *) d->stats[level] : nullptr;
    c.l0pre = d->l0pre;
    return c;
}
// the context read afresh: what a phase of a block starts from (the previous phase's copy, and everything derived from it, dies)
__device__ __forceinline__ HmeCtx fenced(const HmeCtx &c, int level) { return make_ctx(c.self, level); }

__device__ __forceinline__ int b2sr_of(const HmeCtx &c) { return c.b2sr; }
__device__ __forceinline__ int b2sr_of(const HmeDev &c) // (the single-call kernels work on the job record itself)
{
    return (256 * (c.quant * c.quant >> 12) * c.a.blk_w * c.a.blk_h) / (c.a.width * c.a.height);
}

struct Psy {
    int err_weight, tex_weight, avg_weight;
};

#define UAVG4(a, b, c, d) ((unsigned) ((a) + (b) + (c) + (d) + 2) >> 2)
#define AVG2(a, b) (((a) + (b) + 1) >> 1)
#define SQR(x) ((x) * (x))
// x * x for |x| < 2^11 as ONE full-rate 24-bit multiply (v_mul_i32_i24; the generic 32-bit v_mul_lo_u32 the compiler
// picks for an int is a quarter-rate instruction, and the metric squares three values per quad and scored vector)
__device__ __forceinline__ int sq24(int x) { return __mul24(x, x); }

__device__ __forceinline__ int sarx(int v, int s) { return v >> s; }

__device__ __forceinline__ unsigned quad_metric(int a1, int a2, int a3, int a4, int b1, int b2, int b3, int b4, const Psy &psy)
{
    int s0 = (int) UAVG4(a1, a2, a3, a4), s1 = (int) UAVG4(b1, b2, b3, b4);
    int se = (int) UAVG4(abs(a1 - b1), abs(a2 - b2), abs(a3 - b3), abs(a4 - b4));
    int ta = (int) UAVG4(abs(a1 - a2), abs(a2 - a3), abs(a3 - a4), abs(a4 - a1));
    int tb = (int) UAVG4(abs(b1 - b2), abs(b2 - b3), abs(b3 - b4), abs(b4 - b1));
    return (unsigned) (sq24(se) << psy.err_weight) + (unsigned) (sq24(ta - tb) << psy.tex_weight) +
           (unsigned) (sq24(s0 - s1) << psy.avg_weight);
}

// ---- wave-cooperative block primitives (all 64 lanes call with identical arguments) ----------

// quad_metric on two packed quads (bytes: top-left, top-right, bottom-left, bottom-right): UAVG4 of four absolute differences is
// (sad + 2) >> 2, a quad's texture the sad against itself rotated by one sample, its sum the sad against zero
struct __attribute__((packed)) U16g { // possibly unaligned 16-bit load
    uint16_t v;
};
struct __attribute__((packed)) U32g { // possibly unaligned 32-bit load
    uint32_t v;
};
__device__ __forceinline__ unsigned quad_metric_pk(uint32_t a, uint32_t b, const Psy &psy)
{
    const uint32_t ra = (a >> 8) | (a << 24), rb = (b >> 8) | (b << 24);
    int se = (int) ((__builtin_amdgcn_sad_u8(a, b, 0u) + 2) >> 2);
    int ta = (int) ((__builtin_amdgcn_sad_u8(a, ra, 0u) + 2) >> 2), tb = (int) ((__builtin_amdgcn_sad_u8(b, rb, 0u) + 2) >> 2);
    int s0 = (int) ((__builtin_amdgcn_sad_u8(a, 0u, 0u) + 2) >> 2), s1 = (int) ((__builtin_amdgcn_sad_u8(b, 0u, 0u) + 2) >> 2);
    return (unsigned) (sq24(se) << psy.err_weight) + (unsigned) (sq24(ta - tb) << psy.tex_weight) + (unsigned) (sq24(s0 - s1) << psy.avg_weight);
}

__device__ __forceinline__ unsigned ws_umetr(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h, const Psy &psy)
{
    int lane = threadIdx.x & 63, qw = w / 2, qh = h / 2;
    unsigned acc = 0;
    const RowSplit split(qw);
    for (int q = lane; q < qw * qh; q += 64) {
        int i, j;
        split(q, i, j);
        const uint8_t *p = a + (ptrdiff_t) (2 * j) * as + 2 * i, *r = b + (ptrdiff_t) (2 * j) * bs + 2 * i;
        // a quad = two 16-bit loads per operand (round 4: eight byte loads and the metric sample by sample until then)
        const uint32_t qa = (uint32_t) ((const U16g *) p)->v | ((uint32_t) ((const U16g *) (p + as))->v << 16);
        const uint32_t qb = (uint32_t) ((const U16g *) r)->v | ((uint32_t) ((const U16g *) (r + bs))->v << 16);
        acc += quad_metric_pk(qa, qb, psy);
    }
    return wave_sum(acc);
}

__device__ __forceinline__ unsigned metric_return(unsigned acc, int w, int h)
{
    const unsigned num = isqrt_u32(acc) * (unsigned) w * (unsigned) h, d = (unsigned) AVG2(w, h);
    return (d & (d - 1u)) == 0u ? num >> (31 - __clz((int) d)) : num / d; // (16 for every whole block: a shift, not a division)
}

__device__ __forceinline__ unsigned ws_metr(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h, const Psy &psy)
{
    if (w == 0 || h == 0) {
        return 0x7fffffffu;
    }
    return metric_return(ws_umetr(a, as, b, bs, w, h, psy), w, h);
}

__device__ __forceinline__ unsigned ws_sse(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h)
{
    if (w == 0 || h == 0) {
        return 0x7fffffffu;
    }
    int lane = threadIdx.x & 63;
    unsigned acc = 0;
    if ((w & 3) == 0) { // four pixels a lane and step: sum (a - b)^2 = a.a + b.b - 2 a.b as three byte dot products
        const int w4 = w >> 2;
        const RowSplit split(w4);
        for (int idx = lane; idx < w4 * h; idx += 64) {
            int x, y;
            split(idx, x, y);
            const uint32_t va = ((const U32g *) (a + (ptrdiff_t) y * as + 4 * x))->v, vb = ((const U32g *) (b + (ptrdiff_t) y * bs + 4 * x))->v;
            acc += __builtin_amdgcn_udot4(va, va, 0u, false) + __builtin_amdgcn_udot4(vb, vb, 0u, false) - 2u * __builtin_amdgcn_udot4(va, vb, 0u, false);
        }
        return wave_sum(acc);
    }
    const RowSplit split(w);
    for (int idx = lane; idx < w * h; idx += 64) {
        int x, y;
        split(idx, x, y);
        int d = (int) a[(ptrdiff_t) y * as + x] - (int) b[(ptrdiff_t) y * bs + x];
        acc += (unsigned) (d * d);
    }
    return wave_sum(acc);
}

__device__ __forceinline__ unsigned ws_hier_metr(int level, const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h,
                                                 const Psy &psy)
{
    return level > 1 ? ws_sse(a, as, b, bs, w, h) : ws_metr(a, as, b, bs, w, h, psy);
}

__device__ __forceinline__ const uint8_t *at(const DPlane &p, int x, int y) { return p.data + (ptrdiff_t) y * p.stride + x; }

__device__ __forceinline__ bool invalid_block(const DPlane &f, int bx, int by, int bw, int bh, int pad)
{
    return (bx - pad) < -kBorder || (by - pad) < -kBorder || (bx + bw + pad) >= (f.w + kBorder) || (by + bh + pad) >= (f.h + kBorder);
}

// ---- same-level motion field hand-off ----------------------------------------------------------
// A block reads the vectors of its left / top / top-left neighbours of the SAME level.  In the
// row-pipelined kernel those were stored by other workgroups of the same launch, possibly on another
// XCD whose L2 is not coherent with ours: the first 8 bytes of a DSV_MV ({x,y}, flags -- all a
// neighbour ever looks at) are therefore stored and loaded as ONE agent-scope 8-byte access
// (write-through store, L1-bypassing load); a head reads kMvPending until that one store has written it (wait_heads
// below).  In the launch-per-front kernels the same accessors are merely redundant.
struct MvHead {
    int x, y;
    uint32_t all, flags;
};

__device__ __forceinline__ MvHead ld_mv_head(const DSV_MV *m)
{
    unsigned long long v = __hip_atomic_load((const unsigned long long *) m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    MvHead h;
    h.all = (uint32_t) v;
    h.flags = (uint32_t) (v >> 32);
    h.x = (int) (int16_t) (h.all & 0xffffu);
    h.y = (int) (int16_t) (h.all >> 16);
    return h;
}

// the same for a wave-uniform address: the record comes back as scalars
__device__ __forceinline__ MvHead ld_mv_head_u(const DSV_MV *m)
{
    unsigned long long v = __hip_atomic_load((const unsigned long long *) m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    MvHead h;
    h.all = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) v);
    h.flags = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (v >> 32));
    h.x = (int) (int16_t) (h.all & 0xffffu);
    h.y = (int) (int16_t) (h.all >> 16);
    return h;
}

__device__ __forceinline__ void st_mv(DSV_MV *out, const DSV_MV &mv)
{
    unsigned long long head = (unsigned long long) (uint32_t) mv.u.all | ((unsigned long long) mv.flags << 32);
    ((unsigned long long *) out)[1] = (unsigned long long) mv.err | ((unsigned long long) mv.dc << 16) | ((unsigned long long) mv.submask << 32);
    __hip_atomic_store((unsigned long long *) out, head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// level-0 result: device field + (when asked for) the host's pinned copy, so no read-back copy is needed
template <class Ctx> __device__ __forceinline__ void st_mv_final(const Ctx &c, DSV_MV *out, const DSV_MV &mv)
{
    st_mv(out, mv);
    if (c.host_mvs) {
        unsigned long long *h = (unsigned long long *) (c.host_mvs + (out - c.mvf[0]));
        h[0] = (unsigned long long) (uint32_t) mv.u.all | ((unsigned long long) mv.flags << 32);
        h[1] = (unsigned long long) mv.err | ((unsigned long long) mv.dc << 16) | ((unsigned long long) mv.submask << 32);
    }
}

// ---- motion vector cost ---------------------------------------------------------------------
__device__ __forceinline__ int pred1(int left, int top, int topleft)
{
    int dif = left + top - topleft;
    return abs(dif - left) < abs(dif - top) ? left : top;
}

__device__ __forceinline__ void movec_pred(const DSV_MV *v, int nbh, int x, int y, int &px, int &py)
{
    int vx0 = 0, vx1 = 0, vx2 = 0, vy0 = 0, vy1 = 0, vy2 = 0;
    if (x > 0) {
        MvHead m = ld_mv_head_u(&v[y * nbh + x - 1]);
        vx0 = m.x;
        vy0 = m.y;
    }
    if (y > 0) {
        MvHead m = ld_mv_head_u(&v[(y - 1) * nbh + x]);
        vx1 = m.x;
        vy1 = m.y;
    }
    if (x > 0 && y > 0) {
        MvHead m = ld_mv_head_u(&v[(y - 1) * nbh + x - 1]);
        vx2 = m.x;
        vy2 = m.y;
    }
    px = pred1(vx0, vx1, vx2);
    py = pred1(vy0, vy1, vy2);
}

__device__ __forceinline__ int seg_bits(int v)
{
    v = abs(v) + 1;
    return (31 - __clz(v)) * 2 + 2;
}

struct CostCtx { // the per-block constant part of mv_cost (hme.c:354, dsv.c:357)
    int px, py, b2sr, q;
};

__device__ __forceinline__ int mv_cost(const CostCtx &c, int mx, int my, int level)
{
    int bits = seg_bits(mx - c.px) + seg_bits(my - c.py);
    bool sqr = level > 1;
    bits += bits * c.b2sr >> 7;
    if (sqr) {
        bits *= bits;
    }
    int cost = min(bits, 1 << 19);
    if (sqr) {
        return (int) ((unsigned) cost * (unsigned) (c.q * c.q >> 12)) >> 10;
    }
    return 3 * cost * c.q >> 12;
}

// neighbour difference of the current block (vector cx,cy not yet stored) -- dsv.c:403
__device__ __forceinline__ void neighbordif2_cur(const DSV_MV *v, int nbh, int x, int y, int cx, int cy, int &dx, int &dy)
{
    int lx = cx, ly = cy, tx = cx, ty = cy;
    if (abs(cx) < 2 && abs(cy) < 2) {
        dx = dy = 0;
        return;
    }
    if (x > 0) {
        MvHead m = ld_mv_head_u(&v[x - 1 + y * nbh]);
        if (m.all && !(m.flags & (1u << DSV_MV_BIT_SKIP))) {
            lx = m.x;
            ly = m.y;
        }
    }
    if (y > 0) {
        MvHead m = ld_mv_head_u(&v[x + (y - 1) * nbh]);
        if (m.all && !(m.flags & (1u << DSV_MV_BIT_SKIP))) {
            tx = m.x;
            ty = m.y;
        }
    }
    dx = abs(lx - cx) + abs(ly - cy);
    dy = abs(tx - cx) + abs(ty - cy);
}

__device__ __forceinline__ int qp2fp(int v) { return (v + 2) >> 2; } // DSV_SAR_R(v, 2)

struct Vec2 {
    int x, y;
};

__device__ int find_inliers(const Vec2 *list, Vec2 *out, int n, int &ax, int &ay) // hme.c:1260
{
    int dist[16], avgd = 0, ssd = 0, nin = 0, sx = 0, sy = 0;
    if (n == 0) {
        return 0;
    }
    for (int i = 0; i < n; i++) {
        dist[i] = SQR(list[i].x - ax) + SQR(list[i].y - ay);
        avgd += dist[i];
    }
    avgd /= n;
    for (int i = 0; i < n; i++) {
        ssd += SQR(dist[i] - avgd);
    }
    int thresh = avgd + (int) isqrt_u32((unsigned) (ssd / n));
    for (int i = 0; i < n; i++) {
        if (dist[i] <= thresh) {
            sx += list[i].x;
            sy += list[i].y;
            out[nin++] = list[i];
        }
    }
    if (nin == 0) {
        return 0;
    }
    ax = sx / nin;
    ay = sy / nin;
    return nin;
}

// ---- half / quarter-pel refinement ---------------------------------------------------------------
#define HPF_ME(a, b, c, d) ((5 * ((b) + (c))) - ((a) + (d)))
__device__ __forceinline__ uint8_t clamp_u8(int v) { return (uint8_t) (v > 255 ? 255 : (v < 0 ? 0 : v)); }

struct SubpelLds {
    alignas(4) uint8_t win[20 * 20]; // reference window rows/cols -1..18 around the 17x17 area
    uint8_t h[34 * 34];   // half-pel image
};

// the 20x20 reference window of a sub-pel search as 100 row dwords: lane L holds dwords L and L + 64
struct HpelWin {
    uint32_t d0, d1;
};
struct __attribute__((packed)) U32u { // possibly unaligned 32-bit load (one global_load_dword)
    uint32_t v;
};

// r = full-pel sample at window position (1, 1); both loads are issued together with the caller's other loads
__device__ __forceinline__ HpelWin load_hpel_window(const uint8_t *r, int rs)
{
    typedef const __attribute__((address_space(1))) uint8_t *gb_t;
    typedef const __attribute__((address_space(1))) U32u *gu32_t;
    const int lane = threadIdx.x & 63;
    gb_t g = (gb_t) r - rs - 1;
    const int k0 = lane, k1 = lane + 64 < 100 ? lane + 64 : 0;
    HpelWin w;
    w.d0 = ((gu32_t) (g + (k0 / 5) * rs + (k0 % 5) * 4))->v;
    w.d1 = ((gu32_t) (g + (k1 / 5) * rs + (k1 % 5) * 4))->v;
    return w;
}

// cooperative construction of the 34x34 half-pel image (hme.c:787) from the loaded window
__device__ __forceinline__ void build_hpel(SubpelLds &s, const HpelWin &w)
{
    int lane = threadIdx.x & 63;
    uint32_t *win32 = (uint32_t *) s.win;
    win32[lane] = w.d0;
    if (lane + 64 < 100) {
        win32[lane + 64] = w.d1;
    }
    __syncthreads();
    for (int idx = lane; idx < 289; idx += 64) {
        int i = idx % 17, j = idx / 17;
        const uint8_t *p = &s.win[(j + 1) * 20 + (i + 1)];
        int hz[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint8_t *q = p + (k - 1) * 20;
            hz[k] = HPF_ME(q[-1], q[0], q[1], q[2]);
        }
        int c = HPF_ME(hz[0], hz[1], hz[2], hz[3]);
        uint8_t *o = &s.h[(2 * j) * 34 + 2 * i];
        o[0] = p[0];
        o[1] = clamp_u8((hz[1] + 4) >> 3);
        o[34] = clamp_u8((HPF_ME(p[-20], p[0], p[20], p[40]) + 4) >> 3);
        o[35] = clamp_u8((c + 32) >> 6);
    }
    __syncthreads();
}

struct __attribute__((packed)) U32l { // possibly unaligned 32-bit LDS read (one ds_read_b32)
    uint32_t v;
};

// The 34x34 half-pel image (hme.c:787) from a 20x20 area that sits in LDS with row pitch PITCH (w = its first sample;
// PITCH = 20: the search's own copy, else the staged search window, filtered in place).  Per point one dword per window row
// (the four taps of a row filter are one unaligned LDS dword) and the 5,5,-1,-1 filter as two dot products.
template <int PITCH> __device__ __forceinline__ void build_hpel_at(SubpelLds &s, const uint8_t *w)
{
    int lane = threadIdx.x & 63;
    for (int idx = lane; idx < 289; idx += 64) {
        int i = idx % 17, j = idx / 17;
        const uint8_t *p = &w[(j + 1) * PITCH + (i + 1)];
        uint32_t D[4]; // row k - 1: samples p[-1], p[0], p[1], p[2] of that row in bytes 0..3
        int hz[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            D[k] = ((const U32l *) (p + (k - 1) * PITCH - 1))->v;
            hz[k] = (int) __builtin_amdgcn_udot4(D[k], 0x00050500u, 0u, false) - (int) __builtin_amdgcn_udot4(D[k], 0x01000001u, 0u, false);
        }
        int c = HPF_ME(hz[0], hz[1], hz[2], hz[3]);
        // the centre column (byte 1 of every row) for the vertical filter
        int b0 = (int) ((D[0] >> 8) & 0xffu), b1 = (int) ((D[1] >> 8) & 0xffu), b2 = (int) ((D[2] >> 8) & 0xffu), b3 = (int) ((D[3] >> 8) & 0xffu);
        int vz = HPF_ME(b0, b1, b2, b3);
        uint32_t o01 = clamp_u8((hz[1] + 4) >> 3), o10 = clamp_u8((vz + 4) >> 3), o11 = clamp_u8((c + 32) >> 6);
        uint8_t *o = &s.h[(2 * j) * 34 + 2 * i]; // even offset in an array that starts on a 4-byte boundary: aligned 16-bit stores
        *(uint16_t *) o = (uint16_t) ((uint32_t) b1 | (o01 << 8));
        *(uint16_t *) (o + 34) = (uint16_t) (o10 | (o11 << 8));
    }
    __syncthreads();
}

// one source quad's four quarter-pel samples (qi, qj spacing: X, X + 4 / Y, Y + 4 in quarter-pel units = two half-pel
// samples apart) at quarter-pel phase `ph`, as a packed Quad: two or four unaligned dwords of the half-pel image and
// byte-lane arithmetic instead of up to 16 byte reads (hme.c:815-835: AVG2 / AVG4 of the neighbouring half-pel samples)
__device__ __forceinline__ uint32_t qquad_ph(const uint8_t *h, int X, int Y, int ph)
{
    const uint8_t *p = h + (Y >> 1) * 34 + (X >> 1);
    const uint32_t m = 0x00ff00ffu;
    uint32_t r0 = ((const U32l *) p)->v, r2 = ((const U32l *) (p + 68))->v;
    uint32_t a, b; // the samples of the quad's upper / lower row in bytes 0 and 2
    if (ph == 0) {
        a = r0;
        b = r2;
    } else if (ph == 1) {
        a = ((r0 & m) + ((r0 >> 8) & m) + 0x00010001u) >> 1;
        b = ((r2 & m) + ((r2 >> 8) & m) + 0x00010001u) >> 1;
    } else {
        uint32_t r1 = ((const U32l *) (p + 34))->v, r3 = ((const U32l *) (p + 102))->v;
        if (ph == 2) {
            a = ((r0 & m) + (r1 & m) + 0x00010001u) >> 1;
            b = ((r2 & m) + (r3 & m) + 0x00010001u) >> 1;
        } else {
            a = ((r0 & m) + ((r0 >> 8) & m) + (r1 & m) + ((r1 >> 8) & m) + 0x00020002u) >> 2;
            b = ((r2 & m) + ((r2 >> 8) & m) + (r3 & m) + ((r3 >> 8) & m) + 0x00020002u) >> 2;
        }
    }
    return __builtin_amdgcn_perm(b, a, 0x06040200u); // (a.b0, a.b2, b.b0, b.b2)
}

__device__ __forceinline__ int qsample_ph(const uint8_t *h, int X, int Y, int phase)
{
    const uint8_t *p = h + (Y >> 1) * 34 + (X >> 1);
    if (phase == 0) {
        return p[0];
    } else if (phase == 1) {
        return AVG2(p[0], p[1]);
    } else if (phase == 2) {
        return AVG2(p[0], p[34]);
    }
    return (p[0] + p[1] + p[34] + p[35] + 2) >> 2;
}

__device__ __forceinline__ int qsample(const uint8_t *h, int X, int Y) // hme.c:815
{
    const uint8_t *p = h + (Y >> 1) * 34 + (X >> 1);
    switch ((X & 1) | ((Y & 1) << 1)) {
        case 0: return p[0];
        case 1: return AVG2(p[0], p[1]);
        case 2: return AVG2(p[0], p[34]);
        default: return (p[0] + p[1] + p[34] + p[35] + 2) >> 2;
    }
}

// ---- staging for the general block routine (round 4) ---------------------------------------------------------------------
// The routine below is the reference's block loop with wave-cooperative primitives that walk a block pixel by pixel: on
// global memory that is a memory round trip per 64 pixels (16 for one squared error of a 32 x 32 block, some 250 per block).
// Every operand block is therefore copied into LDS first -- all its dwords in ONE round trip -- and the same primitives run
// on the copies.  Blocks are at most 32 x 32 (dsv_encoder.c:1203-1211; checked by hme_run_batch).
struct GenLds {
    alignas(4) uint8_t src[32 * 32];   // the source block, pitch 32
    alignas(4) uint8_t ref[34 * 36];   // a reference block, or one with a one-pixel rim (sub-pel search), pitch 36
    alignas(4) uint8_t aux[32 * 32];   // the sub-pel search's source window; the reference block at zero motion (skip test)
    alignas(4) uint8_t cs[2][32 * 32]; // chroma: source,
    alignas(4) uint8_t cr[2][32 * 32]; //         reference at the vector,
    alignas(4) uint8_t cz[2][32 * 32]; //         reference at zero motion
    // the candidate list and the parent vectors (wave-uniform values: every lane stores the same ones); as arrays of a lane's
    // own they were indexed at run time and lived in scratch memory -- the library's only scratch
    Vec2 cands[40], lc[16], inl[16];
};

template <int PITCH> __device__ __forceinline__ void stage_block(uint8_t *dst, const uint8_t *g, int gs, int w, int h)
{
    typedef const __attribute__((address_space(1))) uint8_t *gb_t;
    typedef const __attribute__((address_space(1))) U32u *gu32_t;
    const int lane = threadIdx.x & 63;
    const int nd = w >> 2, tail = w & 3, per_row = nd + (tail ? 1 : 0), total = per_row * h; // <= 9 x 34 dwords: five a lane
    const unsigned inv = (65536u + (unsigned) per_row - 1u) / (unsigned) per_row;             // idx / per_row for idx < 4096
    __syncthreads(); // the previous tenant's readers are done
    uint32_t v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int idx = lane + 64 * k;
        v[k] = 0;
        if (idx < total) {
            const int r = (int) (((unsigned) idx * inv) >> 16), cc = idx - r * per_row;
            gb_t q = (gb_t) g + (ptrdiff_t) r * gs + 4 * cc;
            if (cc < nd) {
                v[k] = ((gu32_t) q)->v;
            } else { // a row's last one to three pixels: nothing is read beyond the block
                v[k] = q[0];
                if (tail > 1) {
                    v[k] |= (uint32_t) q[1] << 8;
                }
                if (tail > 2) {
                    v[k] |= (uint32_t) q[2] << 16;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int idx = lane + 64 * k;
        if (idx < total) {
            const int r = (int) (((unsigned) idx * inv) >> 16), cc = idx - r * per_row;
            *(uint32_t *) (dst + r * PITCH + 4 * cc) = v[k];
        }
    }
    __syncthreads();
}

// one quad per lane: the 16x16 source window against the block sampled at quarter-pel offset (tx, ty) (hme.c:244)
__device__ __forceinline__ unsigned ws_qpsad(const uint8_t *a, int as, const uint8_t *h, int tx, int ty, const Psy &psy)
{
    int lane = threadIdx.x & 63;
    int i = lane & 7, j = lane >> 3;
    const uint8_t *p = a + (ptrdiff_t) (2 * j) * as + 2 * i;
    int X = 4 + 8 * i + tx, Y = 4 + 8 * j + ty;
    unsigned acc = quad_metric(p[0], p[1], p[as], p[as + 1], qsample(h, X, Y), qsample(h, X + 4, Y), qsample(h, X, Y + 4),
                               qsample(h, X + 4, Y + 4), psy);
    return metric_return(wave_sum(acc), 16, 16);
}

__device__ __forceinline__ unsigned subpixel_me(const HmeDev &c, SubpelLds &lds, GenLds &G, const CostCtx &cc, int &sub_x, int &sub_y, int fpelx, int fpely,
                                unsigned best, int bx, int by, int bw, int bh, const Psy &psy) // hme.c:1051
{
    const DPlane &src = c.src[0], &ref = c.ref[0];
    sub_x = sub_y = 0;
    if (best == 0) {
        return best;
    }
    unsigned yarea = (unsigned) (bw * bh);
    unsigned quad[4];
    const int dxs[4] = {1, -1, 0, 0}, dys[4] = {0, 0, 1, -1};
    // the four neighbours of the full-pel position out of ONE staged copy of the block with a one-pixel rim (the caller has
    // checked a rim of four: invalid_block(.., 4)); the source block sits in G.src
    stage_block<36>(G.ref, at(ref, bx + fpelx - 1, by + fpely - 1), ref.stride, bw + 2, bh + 2);
#pragma unroll
    for (int n = 0; n < 4; n++) {
        quad[n] = ws_sse(G.src, 32, G.ref + (1 + dys[n]) * 36 + (1 + dxs[n]), 36, bw, bh);
    }
    int area_ratio = (int) (8 * 256 / yarea), iarea_ratio = (int) (8 * yarea / 256);
    best = best * (unsigned) area_ratio >> 3;
    int xx = bx + ((bw >> 1) - 8), yy = by + ((bh >> 1) - 8);
    // (the 16 x 16 source window is centred on the block and reaches beyond a clipped one: staged from the plane, not from G.src)
    stage_block<32>(G.aux, at(src, xx, yy), src.stride, 16, 16);
    const uint8_t *srcw = G.aux;
    const int srcw_stride = 32;
    build_hpel(lds, load_hpel_window(at(ref, xx + fpelx - 1, yy + fpely - 1), ref.stride));

    int pri0 = 0, pri1 = -1, sec0 = -1, sec1 = 0;
    unsigned ms1 = quad[1], ms2 = quad[3];
    if (quad[3] >= quad[2]) {
        pri1 = 1;
        ms2 = quad[2];
    }
    if (quad[1] >= quad[0]) {
        sec0 = 1;
        ms1 = quad[0];
    }
    if (ms2 > ms1) {
        int t0 = sec0, t1 = sec1;
        sec0 = pri0, sec1 = pri1;
        pri0 = t0, pri1 = t1;
    }
    int diag0 = pri0 + sec0, diag1 = pri1 + sec1;
    int bestv0 = 0, bestv1 = 0;
    for (int n = 0; n <= 6; n++) {
        int t0, t1;
        if (n == 6) {
            t0 = pri0 + diag0;
            t1 = pri1 + diag1;
        } else {
            int v0 = (n >> 1) == 0 ? pri0 : ((n >> 1) == 1 ? sec0 : diag0);
            int v1 = (n >> 1) == 0 ? pri1 : ((n >> 1) == 1 ? sec1 : diag1);
            int hp = !(n & 1);
            t0 = v0 * (1 << hp);
            t1 = v1 * (1 << hp);
        }
        if (((t0 | t1) & 1) && c.effort < 8) {
            continue;
        }
        unsigned score = ws_qpsad(srcw, srcw_stride, lds.h, t0, t1, psy);
        score += (unsigned) mv_cost(cc, fpelx * 4 + t0, fpely * 4 + t1, 0);
        if (best > score) {
            best = score;
            bestv0 = t0;
            bestv1 = t1;
        }
    }
    sub_x = bestv0;
    sub_y = bestv1;
    __syncthreads(); // the LDS image may be rebuilt by a second call
    return best * (unsigned) iarea_ratio >> 3;
}

// ---- mode decision helpers -----------------------------------------------------------------------
// (operands: staged copies -- luma source / reference, and the two chroma planes' source / reference blocks, pitch 32 unless said)
__device__ __forceinline__ void ws_yuv_max_subblock_err(unsigned out[3], const uint8_t *ys, int yss, const uint8_t *yr, int yrs, int bw, int bh,
                                        const uint8_t (*cs)[32 * 32], const uint8_t (*cr)[32 * 32], int cbw, int cbh, const Psy &psy) // hme.c:369
{
#pragma unroll
    for (int z = 0; z < 3; z++) { // (unrolled: out[z] indexed at run time would put the caller's array in scratch memory)
        const uint8_t *sp, *rp;
        int ss, rs, w, h;
        if (z == 0) {
            sp = ys;
            ss = yss;
            rp = yr;
            rs = yrs;
            w = bw / 2;
            h = bh / 2;
        } else {
            sp = cs[z - 1];
            ss = 32;
            rp = cr[z - 1];
            rs = 32;
            w = cbw / 2;
            h = cbh / 2;
        }
        unsigned mx = 0;
        for (int k = 0; k < 4; k++) {
            int f = (k & 1) ? w : 0, g = (k & 2) ? h : 0;
            unsigned e = ws_umetr(sp + f + g * ss, ss, rp + f + g * rs, rs, w, h, psy);
            mx = max(mx, e);
        }
        out[z] = mx;
    }
}

__device__ __forceinline__ void ws_calc_eprm(const uint8_t *src, int ss, const uint8_t *mvr, int rs, int avg_src, int avg_ref, int w, int h, int &eprmi,
                             int &eprmd, int &eprmr) // hme.c:452
{
    int lane = threadIdx.x & 63;
    int ci = 0, cd = 0, cr = 0;
    avg_src -= 128;
    avg_ref -= 128;
    const RowSplit split(w);
    for (int idx = lane; idx < w * h; idx += 64) {
        int x, y;
        split(idx, x, y);
        int s = src[(ptrdiff_t) y * ss + x];
        cr |= ((s - (int) mvr[(ptrdiff_t) y * rs + x]) + 128) & ~0xff;
        ci |= (s - avg_ref) & ~0xff;
        cd |= (s - avg_src) & ~0xff;
    }
    eprmi = __any(ci != 0) ? 1 : 0;
    eprmd = __any(cd != 0) ? 1 : 0;
    eprmr = __any(cr != 0) ? 1 : 0;
}

__device__ __forceinline__ void ws_err_intra(const uint8_t *a, int as, const uint8_t *b, int bs, int avg_sb, int avg_src, int w, int h,
                             unsigned &intra_err, unsigned &intrasrc_err, unsigned &inter_err, const Psy &psy, int ratio) // hme.c:839
{
    int lane = threadIdx.x & 63, qw = w / 2, qh = h / 2;
    unsigned isb = 0, isrc = 0, inter = 0;
    const RowSplit split(qw);
    for (int q = lane; q < qw * qh; q += 64) {
        int i, j;
        split(q, i, j);
        const uint8_t *p = a + (ptrdiff_t) (2 * j) * as + 2 * i, *r = b + (ptrdiff_t) (2 * j) * bs + 2 * i;
        int a1 = p[0], a2 = p[1], a3 = p[as], a4 = p[as + 1];
        int b1 = r[0], b2 = r[1], b3 = r[bs], b4 = r[bs + 1];
        int s0 = (int) UAVG4(a1, a2, a3, a4), s1 = (int) UAVG4(b1, b2, b3, b4);
        int ae = (int) UAVG4(abs(a1 - b1), abs(a2 - b2), abs(a3 - b3), abs(a4 - b4));
        int ta = (int) UAVG4(abs(a1 - a2), abs(a2 - a3), abs(a3 - a4), abs(a4 - a1));
        int tb = (int) UAVG4(abs(b1 - b2), abs(b2 - b3), abs(b3 - b4), abs(b4 - b1));
        inter += (unsigned) (sq24(ae) * ratio >> (5 - psy.err_weight));
        inter += (unsigned) (sq24(ta - tb) << psy.tex_weight);
        inter += (unsigned) (sq24(s0 - s1) << psy.avg_weight);
        ae = (int) UAVG4(abs(a1 - avg_sb), abs(a2 - avg_sb), abs(a3 - avg_sb), abs(a4 - avg_sb));
        isb += (unsigned) (sq24(ae) << psy.err_weight);
        isb += (unsigned) (sq24(ta) << psy.tex_weight);
        isb += (unsigned) (sq24(s0 - avg_sb) << (psy.avg_weight + 1));
        ae = (int) UAVG4(abs(a1 - avg_src), abs(a2 - avg_src), abs(a3 - avg_src), abs(a4 - avg_src));
        isrc += (unsigned) (sq24(ae) << psy.err_weight);
        isrc += (unsigned) (sq24(ta) << psy.tex_weight);
        isrc += (unsigned) (sq24(s0 - avg_src) << (psy.avg_weight + 1));
    }
    intra_err = wave_sum(isb);
    intrasrc_err = wave_sum(isrc);
    inter_err = wave_sum(inter) * (unsigned) ratio >> 5;
}

__device__ __forceinline__ int ws_plane_avg(const DPlane &p, int x, int y, int w, int h) { return ws_block_avg(at(p, x, y), p.stride, w, h); }

__device__ __forceinline__ void test_subblock_intra_y(const HmeDev &c, const DSV_MV *refmv, DSV_MV &mv, const uint8_t *srcd, int ss, const uint8_t *refd,
                                      int rs, int detail_src, int avg_src, int neidif, unsigned ratio, int bw, int bh) // hme.c:891
{
    int sbw = bw / 2, sbh = bh / 2, nsub = 0;
    unsigned avg_tot = 0, err_sub = 0, err_src = 0;
    Psy psy = {0, 1, 2};
    int rx = refmv ? refmv->u.mv.x : mv.u.mv.x, ry = refmv ? refmv->u.mv.y : mv.u.mv.y;
    if (mv.u.all && neidif < 3 && abs(rx - mv.u.mv.x) < 3 && abs(ry - mv.u.mv.y) < 3) {
        return;
    }
    if (sbw == 0 || sbh == 0) {
        return;
    }
    detail_src += detail_src / max(neidif, 1);
    for (int k = 0; k < 4; k++) {
        int f = (k & 1) ? sbw : 0, g = (k & 2) ? sbh : 0;
        const uint8_t *sd = srcd + f + (ptrdiff_t) g * ss, *md = refd + f + (ptrdiff_t) g * rs;
        if (mv.submask & (1 << k)) {
            continue;
        }
        unsigned avg_local;
        unsigned avg_sub = (unsigned) ws_block_avg(md, rs, sbw, sbh);
        unsigned local_detail = (unsigned) ws_block_detail(sd, ss, sbw, sbh, avg_local);
        unsigned dcd = (unsigned) abs((int) avg_local - (int) avg_sub) + 2;
        if (local_detail > (unsigned) (SQR(dcd) * (unsigned) bw * (unsigned) bh * ratio >> 5)) {
            continue;
        }
        int dc = (int) (avg_local + (unsigned) avg_src * 3 + 2) >> 2;
        unsigned sub_err, src_err, inter_err;
        ws_err_intra(sd, ss, md, rs, (int) avg_sub, dc, sbw, sbh, sub_err, src_err, inter_err, psy, (int) ratio);
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
        mv.dc = err_src < err_sub ? (uint16_t) ((avg_tot / (unsigned) nsub) | DSV_SRC_DC_PRED) : 0;
    }
}

__device__ __forceinline__ void test_subblock_intra_c(const HmeDev &c, DSV_MV &mv, unsigned mad, unsigned detail_src, unsigned avg_src,
                                      const uint8_t (*cs)[32 * 32], const uint8_t (*cr)[32 * 32], int cbw, int cbh) // hme.c:987
{
    int sbw = cbw / 2, sbh = cbh / 2;
    if (c.effort < 6) {
        return;
    }
    unsigned thr = (mv.flags & (1u << DSV_MV_BIT_INTRA)) ? detail_src : SQR(detail_src);
    if (sbw == 0 || sbh == 0 || mad <= thr || thr > 64 || (abs((int) mv.u.mv.x) < 4 && abs((int) mv.u.mv.y) < 4)) {
        return;
    }
    unsigned avg_ramp = avg_src * avg_src >> 8;
    for (int k = 0; k < 4; k++) {
        int f = (k & 1) ? sbw : 0, g = (k & 2) ? sbh : 0;
        if (mv.submask & (1 << k)) {
            continue;
        }
        int us = ws_block_avg(cs[0] + f + g * 32, 32, sbw, sbh);
        int vs = ws_block_avg(cs[1] + f + g * 32, 32, sbw, sbh);
        int um = ws_block_avg(cr[0] + f + g * 32, 32, sbw, sbh);
        int vm = ws_block_avg(cr[1] + f + g * 32, 32, sbw, sbh);
        unsigned dif = (unsigned) (SQR(us - um) + SQR(vs - vm)) * avg_ramp >> 8;
        if (dif > thr) {
            mv.submask |= (uint8_t) (1 << k);
        }
    }
    if (mv.submask) {
        mv.flags |= 1u << DSV_MV_BIT_INTRA;
    }
}

// ---- one block (wave-uniform control flow) --------------------------------------------------------
#define MAXC 40

__device__ void hme_block(const HmeDev &c, int level, int i, int j, int gx, int gy, int *hist, SubpelLds &lds)
{
    const int rectx[9] = {0, 1, -1, 0, 0, -1, 1, -1, 1};
    const int recty[9] = {0, 0, 0, 1, -1, -1, -1, 1, 1};
    int lane = threadIdx.x & 63;
    int nxb = c.a.nbh, nyb = c.a.nbv, y_w = c.a.blk_w, y_h = c.a.blk_h;
    int step = 1 << level;
    const DPlane &src = c.src[level], &ref = c.ref[level], &ogr = c.ogr[level];
    DSV_MV *mvf = c.mvf[level];
    const DSV_MV *parent = level < c.pyr_levels ? c.mvf[level + 1] : nullptr;
    DSV_MV *out = &mvf[i + j * nxb];
    DSV_MV mv = {};
    __shared__ GenLds G;
    Vec2 *cands = G.cands;
    int n = 0;

    int bx = (i * y_w) >> level, by = (j * y_h) >> level;
    if (bx >= src.w || by >= src.h) {
        if (lane == 0) {
            st_mv(out, mv);
        }
        return;
    }
    int bw = min(src.w - bx, y_w), bh = min(src.h - by, y_h);
    // the source block, staged once (every primitive below reads it out of LDS); reference blocks are staged per evaluation
    stage_block<32>(G.src, at(src, bx, by), src.stride, bw, bh);
    const uint8_t *sblk = G.src;
    const int sblk_s = 32;
    cands[n++] = Vec2{0, 0};
    int motion_bias = y_w * y_h;
    unsigned var_src = 0, avg_src = 0;
    Psy psy = {2, 1, 0};
    int lax = 0, lay = 0;
    if (level <= 1) {
        var_src = (unsigned) ws_block_detail(sblk, sblk_s, bw, bh, avg_src);
        int tvar = (int) (var_src + SQR(var_src >> 10));
        tvar = (8 * tvar * c.quant >> 9) / (bw * bh);
        if (tvar) {
            int hvar = (int) ws_hist_var(sblk, sblk_s, bw, bh, hist);
            int qtex = ws_quant_tex(sblk, sblk_s, bw, bh);
            int npeaks = ws_peaks(sblk, sblk_s, bw, bh, (int) avg_src, hist);
            motion_bias += tvar * (hvar - qtex) * npeaks;
        }
        motion_bias = max(motion_bias, 0) / (2 + (abs(gx) + abs(gy)));
        if (var_src <= (unsigned) (8 * bw * bh * c.quant >> 9)) {
            psy = Psy{2, 1, 2};
            motion_bias = 0;
        } else {
            psy = Psy{1, 2, 1};
        }
        if (var_src > (unsigned) (24 * bw * bh)) {
            psy.avg_weight = 0;
        }
    }
    if (parent != nullptr) {
        const int pt[18] = {0, 0, -2, 0, 2, 0, 0, -2, 0, 2, -2, -2, 2, 2, 2, -2, -2, 2};
        unsigned parent_mask = ~(((unsigned) step << 1) - 1);
        int pi = (int) ((unsigned) i & parent_mask), pj = (int) ((unsigned) j & parent_mask);
        int sumx = 0, sumy = 0, npar = 0;
        Vec2 *lc = G.lc, *inl = G.inl;
        for (int m = 0; m < 9; m++) {
            int x = pi + pt[2 * m] * step, y = pj + pt[2 * m + 1] * step;
            if (x >= 0 && x < nxb && y >= 0 && y < nyb) {
                const DSV_MV *pm = &parent[x + y * nxb];
                int vx = pm->u.mv.x, vy = pm->u.mv.y;
                sumx += vx;
                sumy += vy;
                lc[npar++] = Vec2{vx, vy};
            }
        }
        if (npar) {
            lax = sumx / npar;
            lay = sumy / npar;
            int nl = find_inliers(lc, inl, npar, lax, lay);
            cands[n++] = Vec2{lax, lay};
            if (level == 0) {
                int px, py;
                movec_pred(mvf, nxb, i, j, px, py);
                cands[n++] = Vec2{qp2fp(px), qp2fp(py)};
            }
            if (i > 0) {
                MvHead m = ld_mv_head(&mvf[(i - step) + j * nxb]);
                cands[n++] = Vec2{qp2fp(m.x), qp2fp(m.y)};
            }
            if (j > 0) {
                MvHead m = ld_mv_head(&mvf[i + (j - step) * nxb]);
                cands[n++] = Vec2{qp2fp(m.x), qp2fp(m.y)};
            }
            if (i > 0 && j > 0) {
                MvHead m = ld_mv_head(&mvf[(i - step) + (j - step) * nxb]);
                cands[n++] = Vec2{qp2fp(m.x), qp2fp(m.y)};
            }
            if (c.ref_mvf != nullptr) {
                for (int k = 0; k < 9; k++) {
                    int rx = i + rectx[k] * step, ry = j + recty[k] * step;
                    if (rx < 0 || ry < 0 || rx >= nxb || ry >= nyb) {
                        continue;
                    }
                    const DSV_MV *m = &c.ref_mvf[rx + ry * nxb];
                    cands[n++] = Vec2{qp2fp(m->u.mv.x), qp2fp(m->u.mv.y)};
                }
            }
            cands[n++] = Vec2{gx, gy};
            for (int m = 0; m < nl; m++) {
                cands[n++] = inl[m];
            }
        }
    }
    for (int k = 0; k < n; k++) {
        cands[k].x = (int) (int16_t) ((int) (int16_t) cands[k].x >> level);
        cands[k].y = (int) (int16_t) ((int) (int16_t) cands[k].y >> level);
    }
    {
        int newn = 1;
        for (int k = 1; k < n; k++) {
            int m;
            for (m = 0; m < newn; m++) {
                if (cands[k].x == cands[m].x && cands[k].y == cands[m].y) {
                    break;
                }
            }
            if (m == newn) {
                cands[newn++] = cands[k];
            }
        }
        n = newn;
    }
    CostCtx cc;
    movec_pred(mvf, nxb, i, j, cc.px, cc.py);
    cc.q = c.quant;
    cc.b2sr = b2sr_of(c);

    int best_k = 0, dx, dy;
    unsigned best_score = 0xffffffffu, score_zero = 0xffffffffu, score;
    for (int k = 0; k < n; k++) {
        dx = cands[k].x;
        dy = cands[k].y;
        if (invalid_block(ref, bx + dx, by + dy, bw, bh, 0)) {
            continue;
        }
        stage_block<36>(G.ref, at(ref, bx + dx, by + dy), ref.stride, bw, bh);
        score = ws_hier_metr(level, sblk, sblk_s, G.ref, 36, bw, bh, psy);
        if (dx == 0 && dy == 0) {
            score_zero = score;
        }
        score += (unsigned) mv_cost(cc, dx * step * 4, dy * step * 4, level);
        if (dx == lax && dy == lay) {
            score = (unsigned) max((int) score - (motion_bias >> level), 0);
        }
        if (best_score > score) {
            best_score = score;
            best_k = k;
        }
    }
    dx = cands[best_k].x;
    dy = cands[best_k].y;
    unsigned best = best_score;
    unsigned qthresh = (unsigned) (c.quant * bw * bh >> 11);
    bool good_enough = false;
    {
        stage_block<36>(G.ref, at(ogr, bx, by), ogr.stride, bw, bh);
        unsigned zoscore = ws_metr(sblk, sblk_s, G.ref, 36, bw, bh, psy);
        if (abs(dx) <= 1 && abs(dy) <= 1) {
            qthresh *= 2;
        }
        if (zoscore < qthresh) {
            best = level == 0 ? score_zero : 0;
            dx = dy = 0;
            good_enough = true;
        }
    }
    if (!good_enough) { // refine_best_fpel_cand, hme.c:1300
        unsigned metr[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        bool again = true;
        while (again && !good_enough) {
            int tvx, tvy;
            again = false;
            // the positions of one round -- the centre, its four neighbours, then one diagonal -- lie inside the block at the
            // centre with a one-pixel rim: where all four neighbours are valid positions (so the rim is inside the padded
            // plane) it is staged ONCE and every position of the round read out of it
            const bool rim = !invalid_block(ref, bx + dx - 1, by + dy, bw, bh, 0) && !invalid_block(ref, bx + dx + 1, by + dy, bw, bh, 0) &&
                             !invalid_block(ref, bx + dx, by + dy - 1, bw, bh, 0) && !invalid_block(ref, bx + dx, by + dy + 1, bw, bh, 0);
            const int rim_x = dx, rim_y = dy;
            if (rim) {
                stage_block<36>(G.ref, at(ref, bx + dx - 1, by + dy - 1), ref.stride, bw + 2, bh + 2);
            }
            for (int k = 0; k < 5; k++) {
                tvx = dx + rectx[k];
                tvy = dy + recty[k];
                if (invalid_block(ref, bx + tvx, by + tvy, bw, bh, 0)) {
                    continue;
                }
                if (rim) {
                    score = ws_hier_metr(level, sblk, sblk_s, G.ref + (1 + tvy - rim_y) * 36 + (1 + tvx - rim_x), 36, bw, bh, psy);
                } else {
                    stage_block<36>(G.ref, at(ref, bx + tvx, by + tvy), ref.stride, bw, bh);
                    score = ws_hier_metr(level, sblk, sblk_s, G.ref, 36, bw, bh, psy);
                }
                // (selects, not metr[k - 1]: an array indexed at run time lives in scratch memory)
                metr[0] = k == 1 ? score : metr[0];
                metr[1] = k == 2 ? score : metr[1];
                metr[2] = k == 3 ? score : metr[2];
                metr[3] = k == 4 ? score : metr[3];
                if (level == 0 && !tvx && !tvy && score <= qthresh) {
                    dx = tvx;
                    dy = tvy;
                    best = score;
                    good_enough = true;
                    break;
                }
                score += (unsigned) mv_cost(cc, tvx * step * 4, tvy * step * 4, level);
                if (best > score) {
                    best = score;
                    dx = tvx;
                    dy = tvy;
                    again = true;
                    break;
                }
            }
            if (again || good_enough) {
                continue;
            }
            tvx = dx + (metr[0] <= metr[1] ? 1 : -1);
            tvy = dy + (metr[2] <= metr[3] ? 1 : -1);
            if (invalid_block(ref, bx + tvx, by + tvy, bw, bh, 0)) {
                break;
            }
            if (rim && dx == rim_x && dy == rim_y) {
                score = ws_hier_metr(level, sblk, sblk_s, G.ref + (1 + tvy - rim_y) * 36 + (1 + tvx - rim_x), 36, bw, bh, psy);
            } else {
                stage_block<36>(G.ref, at(ref, bx + tvx, by + tvy), ref.stride, bw, bh);
                score = ws_hier_metr(level, sblk, sblk_s, G.ref, 36, bw, bh, psy);
            }
            score += (unsigned) mv_cost(cc, tvx * step * 4, tvy * step * 4, level);
            if (best > score) {
                best = score;
                dx = tvx;
                dy = tvy;
                again = true;
            }
        }
    }
    mv.u.mv.x = (int16_t) (dx * step);
    mv.u.mv.y = (int16_t) (dy * step);
    if (level != 0) {
        if (lane == 0) {
            st_mv(out, mv);
        }
        return;
    }

    // ---- sub-pel refinement + mode decision (hme.c:1598-1821) ----
    int fpelx = mv.u.mv.x, fpely = mv.u.mv.y, sx = 0, sy = 0;
    bool found_sub = false;
    unsigned yarea = (unsigned) (bw * bh);
    const DPlane &ref0 = c.ref[0];
    if (fpelx == lax && fpely == lay) {
        best += (unsigned) motion_bias;
    }
    unsigned best_fp = best;
    if (c.effort >= 4) {
        if (!invalid_block(ref0, bx + lax, by + lay, bw, bh, 4)) {
            best = subpixel_me(c, lds, G, cc, sx, sy, lax, lay, best_fp, bx, by, bw, bh, psy);
            if (sx || sy) {
                fpelx = lax;
                fpely = lay;
                found_sub = true;
            }
        }
        if (!found_sub && !good_enough && !invalid_block(ref0, bx + fpelx, by + fpely, bw, bh, 4)) {
            best = subpixel_me(c, lds, G, cc, sx, sy, fpelx, fpely, best_fp, bx, by, bw, bh, psy);
        }
    }
    mv.u.mv.x = (int16_t) (fpelx * 4 + sx);
    mv.u.mv.y = (int16_t) (fpely * 4 + sy);

    unsigned ratio = 32;
    if ((mv.u.mv.x | mv.u.mv.y) & 3) {
        ratio = (best << 5) / (best_fp + !best_fp);
    }
    // the original-reference block at the vector, then -- for everything that follows -- the reference block there
    stage_block<36>(G.ref, at(c.ogr[0], bx + fpelx, by + fpely), c.ogr[0].stride, bw, bh);
    unsigned ogrerr = ws_metr(sblk, sblk_s, G.ref, 36, bw, bh, psy);
    stage_block<36>(G.ref, at(ref0, bx + fpelx, by + fpely), ref0.stride, bw, bh);
    const uint8_t *refd = G.ref;
    const int refd_s = 36;
    unsigned ogrmad = (ogrerr + yarea / 2) / yarea;
    ogrmad = ogrmad * ratio >> 5;
    unsigned mad = (best + yarea / 2) / yarea;
    unsigned avg_ref;
    unsigned var_ref = (unsigned) ws_block_detail(refd, refd_s, bw, bh, avg_ref);
    int dv = (int) min(ratio, 32u);
    int ipolvar = (int) ((var_src * (unsigned) dv + var_ref * (unsigned) (32 - dv)) >> 5);
    dv = abs((int) var_src - ipolvar);
    if (var_src > 16 * yarea && var_src < 32 * yarea) {
        mv.flags |= 1u << DSV_MV_BIT_MAINTAIN;
    }
    int hs = c.a.hshift, vs = c.a.vshift;
    int cbx = i * (y_w >> hs), cby = j * (y_h >> vs);
    int cbmx = cbx + sarx(fpelx, hs), cbmy = cby + sarx(fpely, vs);
    int cbw = bw >> hs, cbh = bh >> vs;
    unsigned chroma_ratio = (unsigned) ((cbw * cbh) << 4) / yarea;
    // the chroma blocks: source and reference at the vector (staged copies, pitch 32)
    for (int z = 0; z < 2; z++) {
        stage_block<32>(G.cs[z], at(c.srcc[z], cbx, cby), c.srcc[z].stride, cbw, cbh);
        stage_block<32>(G.cr[z], at(c.refc[z], cbmx, cbmy), c.refc[z].stride, cbw, cbh);
    }
    int uavg_src = ws_block_avg(G.cs[0], 32, cbw, cbh), vavg_src = ws_block_avg(G.cs[1], 32, cbw, cbh);
    int uavg_ref = ws_block_avg(G.cr[0], 32, cbw, cbh), vavg_ref = ws_block_avg(G.cr[1], 32, cbw, cbh);
    ChromaPsy cpsy = chroma_analysis((int) avg_src, uavg_src, vavg_src);
    unsigned avg_y_dif = (unsigned) abs((int) avg_src - (int) avg_ref);
    unsigned avg_c_dif = (unsigned) AVG2(abs(uavg_src - uavg_ref), abs(vavg_src - vavg_ref));
    int eprmi, eprmd, eprmr;
    ws_calc_eprm(sblk, sblk_s, refd, refd_s, (int) avg_src, (int) avg_ref, bw, bh, eprmi, eprmd, eprmr);
    bool oob;
    {
        int px = i * y_w + sarx(mv.u.mv.x, 2), py = j * y_h + sarx(mv.u.mv.y, 2);
        oob = px < 0 || py < 0 || px >= ((nxb - 1) * y_w) - 1 || py >= ((nyb - 1) * y_h) - 1;
    }
    int neidif;
    {
        int a, b;
        neighbordif2_cur(mvf, nxb, i, j, mv.u.mv.x, mv.u.mv.y, a, b);
        neidif = (a + b) / 3;
    }
    unsigned skipt = ((unsigned) c.quant * (unsigned) c.quant) >> 19;
    bool skipped = false;
    if ((good_enough || mv.u.all == 0) && c.skip_block_thresh >= 0 && !c.lossless) {
        unsigned sth = skipt * yarea, zsub[3];
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
        // (against the reference at ZERO motion: luma into G.aux, chroma into G.cz)
        stage_block<32>(G.aux, at(ref0, bx, by), ref0.stride, bw, bh);
        for (int z = 0; z < 2; z++) {
            stage_block<32>(G.cz[z], at(c.refc[z], cbx, cby), c.refc[z].stride, cbw, cbh);
        }
        ws_yuv_max_subblock_err(zsub, sblk, sblk_s, G.aux, 32, bw, bh, G.cs, G.cz, cbw, cbh, psy);
        unsigned cth = chroma_ratio * sth * max(skipt, 1u) >> 5;
        zsub[0] = zsub[0] * ratio >> 5;
        zsub[1] = zsub[1] * ratio >> 5;
        zsub[2] = zsub[2] * ratio >> 5;
        zsub[0] += (unsigned) SQR((int) avg_src - (int) avg_ref) * yarea;
        if (zsub[0] <= sth && zsub[1] <= cth && zsub[2] <= cth) {
            mv.flags |= 1u << DSV_MV_BIT_SKIP;
            mv.u.all = 0;
            mv.err = 0;
            skipped = true;
        }
    }
    int add_err = 0, add_ndiff = 0;
    if (!skipped) {
        if (!oob && !c.lossless) {
            bool y_prereq = avg_y_dif <= 2, c_prereq = !cpsy.greyish && avg_c_dif <= 2;
            if (y_prereq || c_prereq) {
                unsigned bsub[3], xth = skipt * yarea;
                int carea = 4 * cbw * cbh;
                ws_yuv_max_subblock_err(bsub, sblk, sblk_s, refd, refd_s, bw, bh, G.cs, G.cr, cbw, cbh, psy);
                xth += (unsigned) ipolvar;
                xth = (unsigned) max((int) xth - ((int) yarea * neidif * 2), 0);
                xth = xth * (unsigned) c.quant >> 12;
                xth = min(max(xth, 32u), yarea * 4);
                bsub[0] = bsub[0] * ratio >> 5;
                bsub[1] = bsub[1] * ratio >> 5;
                bsub[2] = bsub[2] * ratio >> 5;
                if (y_prereq && bsub[0] < 4 * xth) {
                    mv.flags |= 1u << DSV_MV_BIT_NOXMITY;
                }
                int utex = (int) ws_block_tex(G.cs[0], 32, cbw, cbh);
                int vtex = (int) ws_block_tex(G.cs[1], 32, cbw, cbh);
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
        const DSV_MV *refmv = c.ref_mvf ? &c.ref_mvf[i + j * nxb] : nullptr;
        test_subblock_intra_y(c, refmv, mv, sblk, sblk_s, refd, refd_s, ipolvar, (int) avg_src, neidif, ratio, bw, bh);
        test_subblock_intra_c(c, mv, mad, (unsigned) (ipolvar / (bw * bh)), avg_src, G.cs, G.cr, cbw, cbh);
        if (!(mv.flags & (1u << DSV_MV_BIT_NOXMITY))) {
            mv.err = (uint16_t) mad;
            add_err = (int) mad;
        }
        add_ndiff = (ogrmad > 11) + (avg_c_dif >= 32);
    }
    int is_intra = 0;
    if (mv.flags & (1u << DSV_MV_BIT_INTRA)) {
        int merged = (mv.dc & DSV_SRC_DC_PRED) ? eprmd : eprmi;
        if (mv.submask != DSV_MASK_ALL_INTRA) {
            merged |= eprmr;
        }
        mv.flags = (mv.flags & ~(1u << DSV_MV_BIT_EPRM)) | (merged ? (1u << DSV_MV_BIT_EPRM) : 0u);
        is_intra = 1;
        mv.u.mv.x = (int16_t) (fpelx * 4);
        mv.u.mv.y = (int16_t) (fpely * 4);
    } else {
        int merged = eprmr;
        if (mv.submask) {
            merged |= eprmi;
        }
        mv.flags = (mv.flags & ~(1u << DSV_MV_BIT_EPRM)) | (merged ? (1u << DSV_MV_BIT_EPRM) : 0u);
    }
    if (mv.flags & ((1u << DSV_MV_BIT_INTRA) | (1u << DSV_MV_BIT_EPRM))) {
        mv.flags &= ~(1u << DSV_MV_BIT_SIMCMPLX);
    }
    if (lane == 0) {
        st_mv_final(c, out, mv);
        if (is_intra) {
            atomicAdd(&c.counters[0], 1);
        }
        if (add_ndiff) {
            atomicAdd(&c.counters[1], add_ndiff);
        }
        if (best > 0) {
            atomicAdd(&c.counters[2], 1);
        }
        if (add_err) {
            atomicAdd(&c.counters[3], add_err);
        }
    }
}

// (shared with hme_fast.h: the hand-off through the vector heads, see wait_heads below)
constexpr int kHmeErrWord = 7;                              // counters[7]: a row gave up waiting (the host turns it into a fatal error)
constexpr unsigned long long kHmeSpinLimit = 400000000ull;  // 100 MHz ticks = 4 s
constexpr unsigned long long kMvPending = ~0ull;            // a head the search has not stored yet (k_hme_clear_b)

#include "hme_fast.h"
#include "hme_fast32.h"

// DSV2_HME_FAST=0 forces the general per-block routine at every level (A/B checks, and the parity tests' second opinion)
static int g_hme_fast = getenv("DSV2_HME_FAST") ? atoi(getenv("DSV2_HME_FAST")) : 1;

// ---- row-pipelined level: ONE launch per pyramid level for all streams --------------------------
// One wavefront walks one block row of one stream left to right (which row: see take_row below -- tickets run row-major
// ACROSS streams, so by the time row j of any stream is taken its row j-1 is well under way and few resident wavefronts
// sit spinning).  Block (bi, bj) needs (bi-1, bj) -- the same wavefront, earlier -- and (bi, bj-1), (bi-1, bj-1) of the
// row above, so a row only ever waits for heads of the row above it: a slow block delays its own neighbourhood, not a
// whole anti-diagonal of every stream as a launch per front would.  Every spin is bounded by the wall clock and reports
// through counters[7] and the host's pinned counter block.
constexpr int kHmeTicket = 8 /* 8 .. 13: one ticket counter per pyramid level */, kHmeProgress = 16;
constexpr int kHmeExhausted = 14; // level 0: ticket partitions whose last row has been handed out (counter block of stream 0)
constexpr int kHmeHostTail = 12;  // ... and the word of the pinned host counter block that says "all of them" (hme.h)

// Which row a workgroup works on is NOT its index in the grid: every wavefront takes the next TICKET of its launch from a
// counter (rows in row-major order across the launch's streams).  A row only ever waits for the row above it, whose
// ticket is lower, i.e. was taken by a wavefront that is already running: the chain of waits ends at a row 0, which
// waits for nothing -- progress is guaranteed whatever order the hardware dispatches the workgroups in (HIP promises
// none), as long as a workgroup, once started, keeps running (no preemption of a launch's resident wavefronts).
__device__ __forceinline__ int take_ticket(int *counter)
{
    int t = 0;
    if (hme_lane() == 0) {
        t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __builtin_amdgcn_readfirstlane(t);
}
// XCD-aware tickets.  The rows of one picture read overlapping windows of the same three planes (a row's window reaches 16
// lines into the rows above and below), and every XCD has its own L2: handed out across the whole chip, the ~20 rows of a
// picture that are in flight at a time sat on eight XCDs and each fetched the shared lines for itself (FETCH_SIZE 1.36 x
// the algorithmic bytes, every such line an HBM-latency miss in a kernel that spends half its life waiting).  So the launch's
// streams are dealt to the XCDs (stream s -> partition s mod 8), every partition has its own ticket counter (kept in the
// counter block of stream `partition`), and a wavefront takes the next row of ITS XCD's partition (HW_REG_XCC_ID); when that is
// exhausted it takes from the next partition, so no row is left behind and the tail balances.  Within a partition tickets
// still run row-major across its streams, and a row's predecessor still holds a lower ticket of the same counter: the
// progress argument above is unchanged.
struct RowTicket {
    int stream, row;
};
__device__ __forceinline__ int xcc_id()
{
    return (int) __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; // hwreg(HW_REG_XCC_ID, 0, 4)
}
__device__ __forceinline__ RowTicket take_row(const HmeDev *tab, int level, int nstreams, int nrows, int parts)
{
    int x = parts > 1 ? xcc_id() % parts : 0;
    for (int k = 0; k < parts; k++) {
        const int ns = (nstreams - x + parts - 1) / parts; // streams x, x + parts, ...
        const int t = take_ticket(&tab[x].counters[kHmeTicket + level]);
        if (t < ns * nrows) {
            if (level == 0 && t == ns * nrows - 1) {
                // the LAST row of this partition has just been handed out; when that is true of every partition the launch
                // has no work left to give -- from here on its wavefronts only drain -- and the host is told (pinned word 12
                // of the first stream's counter block), so that the next lockstep group's search may start filling the
                // slots this one frees (encoder.cpp: the search token is passed on at that point, not at the launch's end)
                int done = 0;
                if (hme_lane() == 0) {
                    done = __hip_atomic_fetch_add(&tab[0].counters[kHmeExhausted], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (done + 1 == parts && tab[0].host_counters) {
                        __hip_atomic_store(&tab[0].host_counters[kHmeHostTail], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            const int row = t / ns;
            return RowTicket{x + parts * (t - row * ns), row};
        }
        x = x + 1 == parts ? 0 : x + 1;
    }
    return RowTicket{-1, -1};
}
// ---- hand-off through the vector heads themselves --------------------------------------------------------------------------
// The grid points of every level's field start a search as kMvPending (k_hme_clear_b); a block's 8-byte head is written once,
// as ONE agent-scope atomic store.  A consumer reads the heads it needs -- its top and top-left neighbours -- as agent-scope
// atomic loads and retries while either still reads pending: every location is validated on its own, so the hand-off needs
// no ordering BETWEEN locations and the producer does not drain its store and publish after every block.  No real head equals
// the pattern: its flag word would need all 32 bits set.  (The fast block routines validate the heads inside their own first
// load round -- load_neighbour_heads, hme_fast.h; blocks of the general routine wait here.)
__device__ __forceinline__ bool wait_heads(const DSV_MV *top, const DSV_MV *top_left, int *err)
{
    const int lane = hme_lane();
    const unsigned long long *p = (const unsigned long long *) (lane == 1 ? top_left : top);
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; spins++) {
        unsigned long long v = lane < 2 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        if (!__any(v == kMvPending)) {
            return true;
        }
        __builtin_amdgcn_s_sleep(8);
        if ((spins & 1023u) == 1023u) {
            unsigned long long now = wall_clock64();
            if (t0 == 0) {
                t0 = now;
            } else if (now - t0 > kHmeSpinLimit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

constexpr int kHmeRowsDone = 6;

// The last row of a stream to finish closes the level for that stream: global_motion (hme.c:1973) of
// the level for the next one -- or, after level 0, the host's copy of the counters -- and the reset
// of the hand-off words.  Everything it reads was stored write-through or by atomics and drained
// before the owner's arrival was counted.
__device__ __forceinline__ void hme_level_epilogue(const HmeDev &c, int level, int nbx, int nby)
{
    const int lane = hme_lane();
    if (level != 0) {
        int step = 1 << level, ax = 0, ay = 0;
        for (int idx = lane; idx < nbx * nby; idx += 64) {
            int i = (idx % nbx) * step, j = (idx / nbx) * step;
            MvHead m = ld_mv_head(&c.mvf[level][i + j * c.a.nbh]);
            ax += m.x;
            ay += m.y;
        }
        ax = wave_sum(ax);
        ay = wave_sum(ay);
        if (lane == 0) {
            int nblk = nbx * nby;
            c.counters[4] = nblk ? ax * 2 / nblk : 0;
            c.counters[5] = nblk ? ay * 2 / nblk : 0;
        }
    } else if (c.host_counters && lane < 8) {
        c.host_counters[lane] = __hip_atomic_load(&c.counters[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int r = lane; r < c.a.nbv; r += 64) {
        c.counters[kHmeProgress + r] = 0;
    }
    if (lane == 0) {
        c.counters[kHmeRowsDone] = 0;
    }
}

// KIND: which block routine walks the row -- the general one (any geometry: blocks up to 32 x 32, any chroma format, odd
// clipped sizes), the fast one of the coarser levels, or the fast one of level 0 (CS: chroma shift, 1 = 4:2:0, 0 = 4:4:4)
enum {
    ROW_GENERAL = 0,
    ROW_FAST_LX = 1,
    ROW_FAST_L0 = 2,
    ROW_FAST_LX32 = 3,  // the coarser levels' routine for 32 x 32 blocks
    ROW_FAST_L0_32 = 4, // level 0 for 32 x 32 blocks, 4:2:0
    ROW_FAST_LX32W = 5, // the same two for 32 x 16 blocks (hme_fast32.h with two quadrants)
    ROW_FAST_L0_32W = 6
};
template <int KIND, int CS = 1, bool SPLIT = false> __device__ __forceinline__ void hme_row(const HmeDev &c, int bj, int level_rt, int nbx, int nby, FastLds &S)
{
    const int level = (KIND == ROW_FAST_L0 || KIND == ROW_FAST_L0_32 || KIND == ROW_FAST_L0_32W) ? 0 : level_rt;
    int gx = uni(c.counters[4]), gy = uni(c.counters[5]);
    int j = bj << level;
    const HmeCtx x = make_ctx(uni_ptr(&c), level);
#ifdef DSV2_HME_PROF
    if (hme_lane() == 0) {
        S.prof_on = level == 0;
        for (int k = 0; k < 32; k++) {
            S.prof_acc[k] = 0;
        }
        S.prof_t = __builtin_amdgcn_s_memtime();
    }
#endif
    RowAcc acc; // this row's share of the frame's counters (flushed behind the loop) and the left neighbour's head
    const int step_ = 1 << level;
    for (int bi = 0; bi < nbx; bi++) {
        if constexpr (KIND == ROW_GENERAL) {
            const DSV_MV *top_ = x.mvf.cur + (bi << level) + (j - step_) * x.a.nbh;
            if (bj > 0 && !wait_heads(top_, bi > 0 ? top_ - step_ : top_, &c.counters[kHmeErrWord])) {
                acc.failed = true;
            }
        }
        HME_MARK(S, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // no neighbour load may move above the poll
        __syncthreads();                                       // LDS scratch of the previous block is dead
        int i = bi << level;
        if (!acc.failed) {
            if constexpr (KIND == ROW_FAST_L0) {
                hme_block_l0<CS, SPLIT>(x, i, j, gx, gy, S, acc);
            } else if constexpr (KIND == ROW_FAST_L0_32) {
                hme_block_l0_32<4>(x, i, j, gx, gy, S, acc);
            } else if constexpr (KIND == ROW_FAST_L0_32W) {
                hme_block_l0_32<2>(x, i, j, gx, gy, S, acc);
            } else if constexpr (KIND == ROW_FAST_LX) {
                hme_block_lx<1>(x, level, i, j, gx, gy, S, acc);
            } else if constexpr (KIND == ROW_FAST_LX32) {
                hme_block_lx<4>(x, level, i, j, gx, gy, S, acc);
            } else if constexpr (KIND == ROW_FAST_LX32W) {
                hme_block_lx<2>(x, level, i, j, gx, gy, S, acc);
            } else {
                hme_block(c, level, i, j, gx, gy, S.hist, S.sp);
                // the general routine reads its LEFT neighbour back from memory (the fast ones carry it in registers): this
                // wavefront's own store has to have landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (acc.failed) {
            // a neighbour's head never arrived (bounded spin), or another row gave up: tell the host directly -- the level's
            // epilogue, which normally delivers the counters, will not run because this row does not arrive
            if (c.host_counters && hme_lane() == 0) {
                __hip_atomic_store(&c.host_counters[kHmeErrWord], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        HME_MARK(S, 9);
    }
#ifdef DSV2_HME_PROF
    if (hme_lane() == 0 && S.prof_on) {
        for (int k = 0; k < 32; k++) {
            atomicAdd(&g_hme_prof[k], S.prof_acc[k]);
        }
    }
#endif
    // this row's counter sums, drained before the row counts as arrived (the last row to arrive hands the counters on)
    acc.flush(c.counters);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // arrival of this row; its vectors and counter updates were drained above
    int done = 0;
    if (hme_lane() == 0) {
        // (release: this row's heads and counter sums; acquire: the last row to arrive reads every row's heads in the epilogue)
        done = __hip_atomic_fetch_add(&c.counters[kHmeRowsDone], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    }
    done = __builtin_amdgcn_readfirstlane(done);
    if (done == nby - 1) {
        hme_level_epilogue(c, level, nbx, nby);
    }
}

// PERSISTENT kernels: `P` workgroups (default 3 072 = three per SIMD) that each walk row after row -- the tickets of take_row hand
// rows to whoever asks.  The cap on the search's wavefronts per SIMD is the LAUNCH, not a padded register allocation
// (amdgpu_waves_per_eu(W, W) caps by padding every wavefront's registers until a (W + 1)th does not fit, and the other
// lockstep groups' streaming kernels -- whose bandwidth is their bytes in flight, i.e. their resident wavefronts -- then live
// in what is left).  A row still only waits for a lower ticket, which a RUNNING worker holds (a worker takes its next ticket
// after finishing its row), so the progress argument is unchanged.
// The launch's arguments, read again from the kernel-argument segment for every row: six values that would otherwise sit in
// scalar registers (or their spill lanes) through the whole persistent loop, with the reciprocals take_row derives from them.
struct HmeRowsArgs { // (the HME_ROWS_P kernels' parameter list, in its order: the kernel-argument segment is read AS this struct)
    const HmeDev *tab;
    int level, nbx, parts, nstreams, nrows;
};
static_assert(offsetof(HmeRowsArgs, tab) == 0 && offsetof(HmeRowsArgs, level) == 8 && offsetof(HmeRowsArgs, nbx) == 12 && offsetof(HmeRowsArgs, parts) == 16 &&
                  offsetof(HmeRowsArgs, nstreams) == 20 && offsetof(HmeRowsArgs, nrows) == 24,
              "HmeRowsArgs: not the layout of (const HmeDev *, int, int, int, int, int) in the kernel-argument segment");
__device__ __forceinline__ HmeRowsArgs hme_rows_args()
{
    typedef const __attribute__((address_space(4))) HmeRowsArgs *ArgsK;
    ArgsK p = (ArgsK) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return HmeRowsArgs{p->tab, p->level, p->nbx, p->parts, p->nstreams, p->nrows};
}
#ifdef DSV2_HME_WMAX
#define HME_WMAX(W) DSV2_HME_WMAX
#else
#define HME_WMAX(W) W
#endif
#define HME_ROWS_P(NAME, WAVES, LEVEL_EXPR, ...)                                                                         \
    __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, HME_WMAX(WAVES)))) void NAME(                       \
        const HmeDev *__restrict__, int, int, int, int, int)                                                             \
    {                                                                                                                    \
        __shared__ FastLds S;                                                                                            \
        DSV2_CENSUS_SCOPE();                                                                                             \
        for (;;) {                                                                                                       \
            const HmeRowsArgs a_ = hme_rows_args();                                                                      \
            const HmeDev *tab = a_.tab;                                                                                  \
            const int level = a_.level, nbx = a_.nbx, parts = a_.parts, nstreams = a_.nstreams, nrows = a_.nrows;        \
            (void) level;                                                                                                \
            const RowTicket t_ = take_row(tab, LEVEL_EXPR, nstreams, nrows, parts);                                      \
            if (t_.row < 0) {                                                                                            \
                return;                                                                                                  \
            }                                                                                                            \
            hme_row<__VA_ARGS__>(tab[t_.stream], t_.row, LEVEL_EXPR, nbx, nrows, S);                                     \
            __syncthreads();                                                                                             \
        }                                                                                                                \
    }
#ifndef DSV2_HME32_L0_WAVES
#define DSV2_HME32_L0_WAVES 3
#define DSV2_HME32_LX_WAVES 4
#endif
HME_ROWS_P(k_hme_rows_l0, 4, 0, ROW_FAST_L0, 1, false)
HME_ROWS_P(k_hme_rows_l0_444, 4, 0, ROW_FAST_L0, 0, false)
HME_ROWS_P(k_hme_rows_l0_422, 4, 0, ROW_FAST_L0, 2, false)
HME_ROWS_P(k_hme_rows_l0s, 4, 0, ROW_FAST_L0, 1, true) // ... with the neighbour-independent half from the pre-pass (k_hme_l0_pre_b)
HME_ROWS_P(k_hme_rows_l0s_444, 4, 0, ROW_FAST_L0, 0, true)
HME_ROWS_P(k_hme_rows_lx, 4, level, ROW_FAST_LX)
// (the 32 x 32 forms: 143 and 123 registers -- three and four wavefronts per SIMD; at two each, as until late in round 5, the
// 2160p leg ran 5 % slower: 1 850 against 1 940 frames/s at 64 streams)
HME_ROWS_P(k_hme_rows_l0_32, DSV2_HME32_L0_WAVES, 0, ROW_FAST_L0_32)
HME_ROWS_P(k_hme_rows_lx32, DSV2_HME32_LX_WAVES, level, ROW_FAST_LX32)
// (32 x 16 blocks -- 1920 x 800, 2560 x 1080: two quadrants a block; 97 and 75 registers)
HME_ROWS_P(k_hme_rows_l0_32w, 4, 0, ROW_FAST_L0_32W)
HME_ROWS_P(k_hme_rows_lx32w, 4, level, ROW_FAST_LX32W)
HME_ROWS_P(k_hme_rows_general, 2, level, ROW_GENERAL)
static int g_hme_persist = getenv("DSV2_HME_PERSIST") ? atoi(getenv("DSV2_HME_PERSIST")) : 3072;
// DSV2_HME_SPLIT: level 0 with its neighbour-independent half in an unordered pre-pass (1) or in place (0); default: by launch size
// (-1: split below kSplitMaxRows block rows per launch -- few pictures: the chain per block is what the frame waits for -- and in place
// above -- many pictures: the chip is short of work slots, and the pre-pass costs a quarter more instructions in total)
static int g_hme_split = getenv("DSV2_HME_SPLIT") ? atoi(getenv("DSV2_HME_SPLIT")) : -1;
constexpr int kSplitMaxRows = 2048;

// ---- level 0's unordered pre-pass (hme_fast.h: hme_l0_pre_block) --------------------------------------------------------------
// grid = (ceil(blocks / per_wg), streams); one wavefront works through per_wg blocks.  Runs after the level-1 launch (it reads
// that level's finished field and its global motion) and before the level-0 launch.
__global__ __launch_bounds__(64) void k_hme_l0_pre_b(const HmeDev *__restrict__ tab, int nbx, int nby, int per_wg)
{
    DSV2_CENSUS_SCOPE();
    __shared__ FastLds S;
    const HmeDev &c = tab[blockIdx.y];
    const HmeCtx x = make_ctx(uni_ptr(&c), 0);
    const int gx = uni(c.counters[4]), gy = uni(c.counters[5]);
    const int b_end = min(nbx * nby, ((int) blockIdx.x + 1) * per_wg);
    for (int b = (int) blockIdx.x * per_wg; b < b_end; b++) {
        const int j = b / nbx, i = b - j * nbx;
        hme_l0_pre_block(x, i, j, gx, gy, S, x.l0pre + (size_t) b * kL0RecDwords);
        __syncthreads();
    }
}


// make_ctx() keeps ONE geometry per level for the source / reference / original-reference luma planes and one for
// the four chroma planes; frames made by dframe_alloc() always satisfy this, anything else takes the general routine
static bool same_geom(const DPlane &a, const DPlane &b) { return a.stride == b.stride && a.w == b.w && a.h == b.h; }
static bool uniform_geometry(const HmeFrames &f, int pyr_levels)
{
    for (int l = 0; l <= pyr_levels; l++) {
        if (!same_geom(f.src[l], f.ref[l]) || !same_geom(f.src[l], f.ogr[l])) {
            return false;
        }
    }
    return same_geom(f.srcc[0], f.srcc[1]) && same_geom(f.srcc[0], f.refc[0]) && same_geom(f.srcc[0], f.refc[1]);
}

// host mirror of fast_path_ok() over a whole level
static bool level_all_fast(const AnalysisParams &a, const DPlane &src, int level)
{
    // the chroma planes only enter at level 0 (mode decision): 4:2:0 and 4:4:4 have a block routine there, the coarser
    // levels take any format
    const bool c420 = a.hshift == 1 && a.vshift == 1, c444 = a.hshift == 0 && a.vshift == 0, c422 = a.hshift == 1 && a.vshift == 0;
    const int bs = a.blk_w, bsh = a.blk_h; // 16 x 16, 32 x 32 or 32 x 16
    const bool wide = bs == 32 && bsh == 16;
    if ((bs != 16 && bs != 32) || (bsh != bs && !wide) || (level == 0 && !c420 && !c444 && !(c422 && bs == 16))) {
        return false;
    }
    if (bs == 32 && level == 0 && !c420) { // (32-wide blocks at level 0: the 4:2:0 routine of hme_fast32.h)
        return false;
    }
    int step = 1 << level;
    int nbx = (a.nbh + step - 1) / step, nby = (a.nbv + step - 1) / step;
    int lx = ((nbx - 1) * step * bs) >> level, ly = ((nby - 1) * step * bsh) >> level; // origin of the last block column / row
    if (lx >= src.w || ly >= src.h) {
        return false;
    }
    int bw = src.w - lx < bs ? src.w - lx : bs, bh = src.h - ly < bsh ? src.h - ly : bsh;
    if (level == 0 && bs == 32) {
        // (a clipped block's sub-blocks must not straddle a quadrant seam; a 32 x 16 block has no seam across its height, and its
        // chroma sub-blocks are whole quads down to a height of 8)
        return (bw & 15) == 0 && (bh & (wide ? 7 : 15)) == 0;
    }
    return level == 0 ? ((bw & 7) == 0 && (bh & 7) == 0) : (level > 1 || (!(bw & 1) && !(bh & 1)));
}

// level < 0: blockIdx.z enumerates the levels (row pipeline: one clear for the whole search)
__global__ __launch_bounds__(256) void k_hme_clear_b(const HmeDev *__restrict__ tab, int level, int nwords, int clear_counters)
{
    DSV2_CENSUS_SCOPE();
    const HmeDev &c = tab[blockIdx.y];
    bool first = level >= 0 || blockIdx.z == 0;
    if (level < 0) {
        level = (int) blockIdx.z;
    }
    uint32_t *p = (uint32_t *) c.mvf[level];
    const int nbh = c.a.nbh, gmask = (1 << level) - 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nwords; i += gridDim.x * 256) {
        // a block of this level's grid: its head (words 0, 1 of the record) reads "pending" until the search stores it
        const int e = i >> 2, bx = e % nbh, by = e / nbh;
        p[i] = ((i & 2) == 0 && ((bx | by) & gmask) == 0) ? 0xffffffffu : 0u;
    }
    if (clear_counters && first && blockIdx.x == 0 && threadIdx.x < 16) {
        c.counters[threadIdx.x] = 0;
    }
    if (first && blockIdx.x == 0) { // row progress words of the level about to run
        for (int r = threadIdx.x; r < c.a.nbv; r += 256) {
            c.counters[kHmeProgress + r] = 0;
        }
    }
}

// ---- block statistics for the host controller (see BlockStatsJob, hme.h) --------------------------------------------
__device__ __forceinline__ int bs_seg_bits(int v) // dsv.c:344
{
    v = abs(v) + 1;
    return (31 - __clz(v)) * 2 + 2;
}

// grid = (ceil(nblocks / 256), streams); one thread per block
__global__ __launch_bounds__(256) void k_block_stats_b(const BlockStatsJob *__restrict__ tab, int nbh, int nbv)
{
    DSV2_CENSUS_SCOPE();
    const BlockStatsJob &J = tab[blockIdx.y];
    const int nblk = nbh * nbv;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int v[BS_USED] = {};
    if (idx < nblk) {
        const int j = idx / nbh, i = idx - j * nbh;
        const DSV_MV *mv = &J.mvs[idx];
        const uint4 rec = *(const uint4 *) mv; // {x | y << 16, flags, err | dc << 16, submask ...}
        if (J.host_mvs) {
            *(uint4 *) &J.host_mvs[idx] = rec;
        }
        const uint32_t all = rec.x, flags = rec.y;
        const int mx = (int) (int16_t) (all & 0xffffu), my = (int) (int16_t) (all >> 16);
        const int err = (int) (rec.z & 0xffffu), submask = (int) (rec.w & 0xffu);
        const bool skip = (flags >> DSV_MV_BIT_SKIP) & 1, intra = (flags >> DSV_MV_BIT_INTRA) & 1;
        uint2 l = {0, 0}, t = {0, 0}, tl = {0, 0};
        if (i > 0) {
            l = *(const uint2 *) (mv - 1);
        }
        if (j > 0) {
            t = *(const uint2 *) (mv - nbh);
        }
        if (i > 0 && j > 0) {
            tl = *(const uint2 *) (mv - nbh - 1);
        }
        // avg_motion (dsv_encoder.c:129)
        if (skip) {
            v[BS_STAT] = 1;
        } else {
            v[BS_AX] = mx;
            v[BS_AY] = my;
            int ndx = 0, ndy = 0; // dsv_neighbordif2 (dsv.c:403)
            if (!(abs(mx) < 2 && abs(my) < 2)) {
                int lx = mx, ly = my, tx = mx, ty = my;
                if (i > 0 && l.x && !((l.y >> DSV_MV_BIT_SKIP) & 1)) {
                    lx = (int) (int16_t) (l.x & 0xffffu);
                    ly = (int) (int16_t) (l.x >> 16);
                }
                if (j > 0 && t.x && !((t.y >> DSV_MV_BIT_SKIP) & 1)) {
                    tx = (int) (int16_t) (t.x & 0xffffu);
                    ty = (int) (int16_t) (t.x >> 16);
                }
                ndx = abs(lx - mx) + abs(ly - my);
                ndy = abs(tx - mx) + abs(ty - my);
            }
            if (ndx > 4 || ndy > 4) {
                v[BS_CHAOS] = 1;
            } else {
                v[BS_STAT] = 1;
            }
        }
        // scene_complexity (dsv_encoder.c:188); dsv_mv_cost with the median predictor of the raw field (dsv.c:357, :375)
        int cost = 0;
        if (!skip) {
            const int px = pred1((int) (int16_t) (l.x & 0xffffu), (int) (int16_t) (t.x & 0xffffu), (int) (int16_t) (tl.x & 0xffffu));
            const int py = pred1((int) (int16_t) (l.x >> 16), (int) (int16_t) (t.x >> 16), (int) (int16_t) (tl.x >> 16));
            int bits = bs_seg_bits(mx - px) + bs_seg_bits(my - py);
            bits += bits * J.b2sr >> 7;
            cost = bits;
        }
        if (J.rc_mode == DSV_RATE_CONTROL_ABR) {
            const int avg_err = (int) ((unsigned) J.counters[3] / (unsigned) nblk);
            int c = skip ? 0 : cost + err - avg_err;
            if (intra) {
                c += submask == DSV_MASK_ALL_INTRA ? 16 : 4;
            }
            v[BS_COMPLEXITY] = c;
        } else if (J.rc_mode == DSV_RATE_CONTROL_CRF) {
            int c = skip ? -100 : cost;
            if (intra) {
                c += submask == DSV_MASK_ALL_INTRA ? 100 : 40;
            }
            v[BS_COMPLEXITY] = c;
        }
        // the running intra map and the counts of scene_change_detection's second test (dsv_encoder.c:617-640)
        const int m = (J.map_in ? J.map_in[idx] : 0) | (intra ? 1 : 0);
        J.map_out[idx] = (uint8_t) m;
        int nintra = 0, skipn = 0;
        if (m) {
            const bool maintain = (flags >> DSV_MV_BIT_MAINTAIN) & 1;
            if (skip || all == 0) {
                nintra += maintain ? 3 : 1;
                skipn += maintain ? 2 : 1;
            } else if (((flags >> DSV_MV_BIT_NOXMITY) & 1) && maintain) {
                nintra++;
            }
        }
        nintra += m;
        v[BS_NINTRA] = nintra;
        v[BS_SKIPN] = skipn;
        // gather_stats, P frame (dsv_encoder.c:1004-1012)
        const int stable = intra ? 0 : (skip ? 1 : 0);
        if (!skip) {
            v[BS_MODE] = intra ? 1 : -1;
            v[BS_EPRM] = ((flags >> DSV_MV_BIT_EPRM) & 1) ? 1 : -1;
        }
        v[BS_STABLE] = stable ? 1 : -1;
        // DSV_STATS block counts (dsv_encoder.c:1505)
        v[BS_ST_EPRM] = (flags >> DSV_MV_BIT_EPRM) & 1;
        if (skip) {
            v[BS_ST_SKIP] = 1;
        } else if (intra) {
            const int dc = (int) (rec.z >> 16);
            v[BS_ST_MBI] = 1;
            v[BS_ST_MBDC] = (dc & DSV_SRC_DC_PRED) ? 1 : 0;
            if (submask != DSV_MASK_ALL_INTRA) {
                v[BS_ST_MBSUB] = 1;
                v[BS_ST_SUB0] = submask & 1;
                v[BS_ST_SUB1] = (submask >> 1) & 1;
                v[BS_ST_SUB2] = (submask >> 2) & 1;
                v[BS_ST_SUB3] = (submask >> 3) & 1;
            }
        } else {
            v[BS_ST_MBP] = 1;
            v[(mx & 1) ? BS_ST_QPX : ((mx & 3) ? BS_ST_HPX : BS_ST_FPX)] = 1;
            v[(my & 1) ? BS_ST_QPY : ((my & 3) ? BS_ST_HPY : BS_ST_FPY)] = 1;
        }
    }
#pragma unroll
    for (int k = 0; k < BS_USED; k++) {
        const int r = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0 && r != 0) {
            atomicAdd(&J.out[k], r);
        }
    }
}

void block_stats_batch(hipStream_t s, const BlockStatsJob *d_jobs, int n, int nbh, int nbv)
{
    if (n > 0) {
        DSV2_LAUNCH(k_block_stats_b, dim3((nbh * nbv + 255) / 256, n), dim3(256), 0, s, d_jobs, nbh, nbv);
    }
}

int hme_estimate(hipStream_t s, CodecDev &dv, const PicSet &cur, const PicSet &ref, const HmeParams &hp)
{
    HmeFrames f;
    f.src[0] = cur.src.p[0];
    f.ref[0] = ref.recon.p[0];
    f.ogr[0] = ref.src.p[0];
    for (int l = 0; l < hp.pyr_levels; l++) {
        f.src[l + 1] = cur.src_pyr[l].p[0];
        f.ref[l + 1] = ref.recon_pyr[l].p[0];
        f.ogr[l + 1] = ref.src_pyr[l].p[0];
    }
    for (int k = 0; k < 2; k++) {
        f.srcc[k] = cur.src.p[k + 1];
        f.refc[k] = ref.recon.p[k + 1];
    }
    for (int l = 0; l <= hp.pyr_levels; l++) {
        f.mvf[l] = dv.d_mvf[l];
    }
    f.ref_mvf = ref.has_final_mvs ? ref.d_final_mvs : nullptr;
    f.counters = dv.d_counters;
    return hme_run(s, f, hp);
}

// ---- source statistics ahead of the search (see source_analysis in hme_fast.h) -----------------------------------------
// grid = (ceil(blocks of level 0 + blocks of level 1, per_wg), streams); one wavefront works through per_wg blocks.
__global__ __launch_bounds__(64) void k_hme_src_stats_b(const HmeDev *__restrict__ tab, int nb0x, int nb0y, int nb1x, int nb1y, int fb0x, int fb0y,
                                                        int fb1x, int fb1y, int per_wg)
{
    DSV2_CENSUS_SCOPE();
    // [0, fbx) x [0, fby) of a level is left to k_hme_src_stats4_b; what is enumerated here is the strip of block rows below it,
    // then the strip of block columns to its right (fb = 0: every block)
    __shared__ int hist[16];
    const HmeDev &c = tab[blockIdx.y];
    const int lane = threadIdx.x & 63, qi = lane & 7, qj = lane >> 3;
    const int n0 = nb0x * nb0y - fb0x * fb0y, total = n0 + nb1x * nb1y - fb1x * fb1y;
    const int quant = uni(c.quant);
    int4 *const out0 = uni_ptr(c.stats[0]), *const out1 = uni_ptr(c.stats[1]);
    const DPlane src0 = uni(c.src[0]), src1 = uni(c.src[1]), ogr0 = uni(c.ogr[0]), ogr1 = uni(c.ogr[1]);
    const int b_end = min(total, ((int) blockIdx.x + 1) * per_wg);
    for (int b = (int) blockIdx.x * per_wg; b < b_end; b++) {
        const bool l1 = b >= n0;
        const int e = l1 ? b - n0 : b, nbx = l1 ? nb1x : nb0x, nby = l1 ? nb1y : nb0y, fbx = l1 ? fb1x : fb0x, fby = l1 ? fb1y : fb0y;
        const int below = (nby - fby) * nbx;
        int bi, bj;
        if (e < below) {
            bj = fby + e / nbx;
            bi = e % nbx;
        } else {
            bj = (e - below) / (nbx - fbx);
            bi = fbx + (e - below) % (nbx - fbx);
        }
        const int idx = bi + bj * nbx;
        const DPlane src = l1 ? src1 : src0;
        const int bx = bi * 16, by = bj * 16;
        const int bw = min(src.w - bx, 16), bh = min(src.h - by, 16);
        if (bw <= 0 || bh <= 0) {
            continue;
        }
        const int qw = bw >> 1, qh = bh >> 1;
        const bool act = qi < qw && qj < qh;
        Quad a = ldq(at(src, bx, by), src.stride, qi, qj, act);
        const DPlane ogr = l1 ? ogr1 : ogr0;
        const Quad o = ldq(at(ogr, bx, by), ogr.stride, qi, qj, act);
        SrcStats st = source_analysis(a, act, qi, qj, qw, bw, bh, quant, hist);
        st.zoscore = metric_return(wave_sum(act ? qmetric(a, o, psy_of_source(st.var_src, bw, bh, quant)) : 0u), bw, bh);
        if (lane == 0) {
            (l1 ? out1 : out0)[idx] = int4{st.bias_raw, (int) st.var_src, (int) st.avg_src, (int) st.zoscore};
        }
    }
}

// ---- the same statistics for 32 x 32 blocks (2160p and up), one block per wavefront ------------------------------------------------
// The general routine's own formulation (hme_block: wave-cooperative loops over a copy of the block in LDS) -- it runs once per block
// and in no order, so its cost is not the search's.
// (blk_h: 32, or 16 for the 32 x 16 blocks of wide pictures)
__global__ __launch_bounds__(64) void k_hme_src_stats32_b(const HmeDev *__restrict__ tab, int nb0x, int nb0y, int nb1x, int nb1y, int per_wg, int blk_h)
{
    DSV2_CENSUS_SCOPE();
    __shared__ int hist[16];
    __shared__ alignas(4) uint8_t sblk[32 * 32], oblk[32 * 36];
    const HmeDev &c = tab[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int n0 = nb0x * nb0y, total = n0 + nb1x * nb1y;
    const int quant = uni(c.quant);
    int4 *const out0 = uni_ptr(c.stats[0]), *const out1 = uni_ptr(c.stats[1]);
    const DPlane src0 = uni(c.src[0]), src1 = uni(c.src[1]), ogr0 = uni(c.ogr[0]), ogr1 = uni(c.ogr[1]);
    const int b_end = min(total, ((int) blockIdx.x + 1) * per_wg);
    for (int b = (int) blockIdx.x * per_wg; b < b_end; b++) {
        const bool l1 = b >= n0;
        const int e = l1 ? b - n0 : b, nbx = l1 ? nb1x : nb0x;
        const int bj = e / nbx, bi = e - bj * nbx;
        const DPlane src = l1 ? src1 : src0, ogr = l1 ? ogr1 : ogr0;
        const int bx = bi * 32, by = bj * blk_h;
        const int bw = min(src.w - bx, 32), bh = min(src.h - by, blk_h);
        if (bw <= 0 || bh <= 0) {
            continue;
        }
        stage_block<32>(sblk, at(src, bx, by), src.stride, bw, bh);
        unsigned avg_src = 0;
        const unsigned var_src = (unsigned) ws_block_detail(sblk, 32, bw, bh, avg_src);
        int tvar = (int) (var_src + SQR(var_src >> 10));
        tvar = (8 * tvar * quant >> 9) / (bw * bh);
        int bias_raw = 32 * blk_h; // hme.c:1444
        if (tvar) { // hme.c:1405-1417
            const int hvar = (int) ws_hist_var(sblk, 32, bw, bh, hist);
            const int qtex = ws_quant_tex(sblk, 32, bw, bh);
            const int npeaks = ws_peaks(sblk, 32, bw, bh, (int) avg_src, hist);
            bias_raw += tvar * (hvar - qtex) * npeaks;
        }
        stage_block<36>(oblk, at(ogr, bx, by), ogr.stride, bw, bh);
        const unsigned zoscore = ws_metr(sblk, 32, oblk, 36, bw, bh, psy_of_source(var_src, bw, bh, quant));
        if (lane == 0) {
            (l1 ? out1 : out0)[e] = int4{bias_raw, (int) var_src, (int) avg_src, (int) zoscore};
        }
        __syncthreads();
    }
}

// ---- the same statistics, FOUR whole 16x16 blocks per wavefront -------------------------------------------------------------
// source_analysis() is written as the block routine needs it: one block per wavefront, a 2x2 quad per lane, full-width
// reductions -- 330 vector instructions per block, more than the search saves by not running it.  Every quantity in it is an
// integer sum over the block's pixels, pixel pairs or quads, so the order of evaluation is free: here 16 lanes share a block
// (a DPP row), lane r holds pixel row r as four dwords, sums go through v_sad_u8 / v_dot4_u32_u8 on whole dwords, row
// totals through four DPP steps.  Same results (the batched encoder's parity tests run both forms), a third of the instructions.
// Clipped blocks of the last column / row keep the one-block form (k_hme_src_stats_b over the edge strips).
template <int CTRL> __device__ __forceinline__ int dppmov(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int row16_sum(int v) // total over the 16 lanes of a DPP row, in every lane of it
{
    v += dppmov<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dppmov<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dppmov<0x141>(v); // row_half_mirror
    v += dppmov<0x140>(v); // row_mirror
    return v;
}
__device__ __forceinline__ int row16_max(int v)
{
    v = max(v, dppmov<0xB1>(v));
    v = max(v, dppmov<0x4E>(v));
    v = max(v, dppmov<0x141>(v));
    v = max(v, dppmov<0x140>(v));
    return v;
}

typedef unsigned uint4v_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_hme_src_stats4_b(const HmeDev *__restrict__ tab, int nb0x, int nb1x, int fb0x, int fb0y, int fb1x, int fb1y)
{
    DSV2_CENSUS_SCOPE();
    __shared__ int hist[2][4][16];
    const HmeDev &c = tab[blockIdx.y];
    const int lane = threadIdx.x & 63, blk = lane >> 4, r = lane & 15;
    const int gpr0 = (fb0x + 3) >> 2, gpr1 = (fb1x + 3) >> 2, G0 = gpr0 * fb0y;
    const bool l1 = (int) blockIdx.x >= G0;
    const int g = l1 ? (int) blockIdx.x - G0 : (int) blockIdx.x, gpr = l1 ? gpr1 : gpr0, fbx = l1 ? fb1x : fb0x;
    const int bj = g / gpr, bi_raw = (g % gpr) * 4 + blk;
    const bool valid = bi_raw < fbx;
    const int bi = valid ? bi_raw : fbx - 1;
    const DPlane src = uni(c.src[l1 ? 1 : 0]), ogr = uni(c.ogr[l1 ? 1 : 0]);
    const int quant = uni(c.quant);
    typedef const __attribute__((address_space(1))) uint4v_t *gu4_t;
    const uint4v_t A = *(gu4_t) (src.data + (ptrdiff_t) (bj * 16 + r) * src.stride + bi * 16);
    const uint4v_t O = *(gu4_t) (ogr.data + (ptrdiff_t) (bj * 16 + r) * ogr.stride + bi * 16);
    const uint32_t a[4] = {A[0], A[1], A[2], A[3]}, o[4] = {O[0], O[1], O[2], O[3]};
    // the row shifted by one pixel (the last pixel stands for itself: a difference of zero) and the row below (the last
    // row stands for itself)
    const uint32_t s[4] = {__builtin_amdgcn_alignbit(a[1], a[0], 8), __builtin_amdgcn_alignbit(a[2], a[1], 8), __builtin_amdgcn_alignbit(a[3], a[2], 8),
                           (a[3] >> 8) | (a[3] & 0xff000000u)};
    uint32_t d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d[k] = (uint32_t) __shfl_down((int) a[k], 1, 16);
    }
    // block sum and first-difference sums (quad_grad_partials over every quad), mean, mean absolute deviation
    unsigned ps = 0, ph = 0, pv = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ps = __builtin_amdgcn_sad_u8(a[k], 0u, ps);
        ph = __builtin_amdgcn_sad_u8(a[k], s[k], ph);
        pv = __builtin_amdgcn_sad_u8(a[k], d[k], pv);
    }
    const int sum = row16_sum((int) ps);
    const unsigned sh = (unsigned) row16_sum((int) ph), sv = (unsigned) row16_sum((int) pv);
    const int bw = 16, bh = 16;
    const int mean = div_nn(sum, bw * bh);
    unsigned pd = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        pd = __builtin_amdgcn_sad_u8(a[k], rep4(mean), pd);
    }
    const int var = row16_sum((int) pd) >> 1;
    const int tex = (int) (max(sh, sv) - (unsigned) var);
    const unsigned var_src = (unsigned) (var + max(tex, 0));
    int tvar = (int) (var_src + SQR(var_src >> 10));
    tvar = div_nn(8 * tvar * quant >> 9, bw * bh);
    // src_hist_var: 16-bin histogram of the pixels scaled by 8 / mean
    unsigned avg = (unsigned) div_nn(sum, bw * bh);
    if (avg == 0) {
        avg = 1;
    }
    const unsigned q16 = udiv_fast(8u << 16, avg);
    hist[0][blk][r] = 0;
    hist[1][blk][r] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            atomicAdd(&hist[0][blk][min((int) (((a[k] >> (8 * b)) & 0xffu) * q16 >> 16), 15)], 1);
        }
    }
    // the cell's quads: an even / odd lane pair holds two pixel rows = eight quads; the even lane takes the left four
    // (its own dwords 0, 1 on top of the partner's), the odd lane the right four (the partner's dwords 2, 3 on top of its own)
    const bool odd = (r & 1) != 0;
    uint32_t tq[2], bq[2], to[2], bo[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t ra = (uint32_t) dppmov<0xB1>((int) (odd ? a[k] : a[2 + k])), ro = (uint32_t) dppmov<0xB1>((int) (odd ? o[k] : o[2 + k]));
        tq[k] = odd ? ra : a[k];
        bq[k] = odd ? a[2 + k] : ra;
        to[k] = odd ? ro : o[k];
        bo[k] = odd ? o[2 + k] : ro;
    }
    const Psy psy = psy_of_source(var_src, bw, bh, quant);
    // src_peaks (histogram of the quads' means, same scale) and the metric against the previous source picture, quad by quad
    const int q16p = (int) udiv_fast(8u << 16, (unsigned) (mean ? mean : 1));
    unsigned zacc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t sel = (k & 1) ? 0x07060302u : 0x05040100u;
        Quad qa, qo;
        qa.w = __builtin_amdgcn_perm(bq[k >> 1], tq[k >> 1], sel);
        qo.w = __builtin_amdgcn_perm(bo[k >> 1], to[k >> 1], sel);
        const int ds = (int) ((sad4(qa.w, 0u) + 2) >> 2);
        atomicAdd(&hist[1][blk][min(ds * q16p >> 16, 15)], 1);
        zacc += qmetric(qa, qo, psy);
    }
    const unsigned zoscore = metric_return((unsigned) row16_sum((int) zacc), bw, bh);
    __syncthreads();
    int hvar;
    {
        const unsigned dd = (unsigned) hist[0][blk][r] - (unsigned) (bw * bh) / 16;
        const unsigned hv = (unsigned) row16_sum((int) (dd * dd));
        hvar = (int) div_nn(hv * 16 * 16, 16u * (unsigned) (bw * bh * bw * bh));
    }
    // src_quant_tex: squared first differences of the pixels' high nibbles, sum (a - b)^2 = a.a + b.b - 2 a.b on whole dwords
    int qtex;
    {
        unsigned aa = 0, ss = 0, as = 0, dd = 0, ad = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t na = (a[k] >> 4) & 0x0f0f0f0fu, ns = (s[k] >> 4) & 0x0f0f0f0fu, nd = (d[k] >> 4) & 0x0f0f0f0fu;
            aa = __builtin_amdgcn_udot4(na, na, aa, false);
            ss = __builtin_amdgcn_udot4(ns, ns, ss, false);
            as = __builtin_amdgcn_udot4(na, ns, as, false);
            dd = __builtin_amdgcn_udot4(nd, nd, dd, false);
            ad = __builtin_amdgcn_udot4(na, nd, ad, false);
        }
        const unsigned qsh = (unsigned) row16_sum((int) (aa + ss - 2u * as)), qsv = (unsigned) row16_sum((int) (aa + dd - 2u * ad));
        qtex = (int) div_nn(isqrt_u32(max(qsh, qsv)), (unsigned) ((bw + bh + 1) >> 1));
    }
    int npeaks;
    {
        const int cnt = hist[1][blk][r];
        const int total = row16_sum(cnt);
        const int maxv = row16_max(cnt) >> 2;
        const int left = __shfl_up(cnt, 1, 16), right = __shfl_down(cnt, 1, 16);
        int pk = 1;
        if (r > 0) {
            pk &= cnt > left;
        }
        if (r < 15) {
            pk &= cnt > right;
        }
        pk &= (cnt > maxv) || (cnt > total / 16);
        npeaks = row16_sum(pk);
    }
    int bias_raw = 16 * 16;
    if (tvar) {
        bias_raw += tvar * (hvar - qtex) * npeaks;
    }
    if (valid && r == 0) {
        uni_ptr(c.stats[l1 ? 1 : 0])[bi + bj * (l1 ? nb1x : nb0x)] = int4{bias_raw, (int) var_src, mean, (int) zoscore};
    }
}

static void fill_hme_dev(HmeDev &c, const HmeFrames &f, const HmeParams &hp)
{
    c.a = hp.a;
    c.effort = hp.effort;
    c.lossless = hp.lossless;
    c.quant = hp.quant;
    c.skip_block_thresh = hp.skip_block_thresh;
    c.pyr_levels = hp.pyr_levels;
    c.psyscale = spatial_psy_factor(hp.a.blk_w, hp.a.blk_h, hp.a.nbh, hp.a.nbv, -1);
    c.b2sr = (256 * (hp.quant * hp.quant >> 12) * hp.a.blk_w * hp.a.blk_h) / (hp.a.width * hp.a.height);
    for (int l = 0; l <= hp.pyr_levels; l++) {
        c.src[l] = f.src[l];
        c.ref[l] = f.ref[l];
        c.ogr[l] = f.ogr[l];
        c.mvf[l] = f.mvf[l];
    }
    for (int k = 0; k < 2; k++) {
        c.srcc[k] = f.srcc[k];
        c.refc[k] = f.refc[k];
    }
    c.ref_mvf = f.ref_mvf;
    c.counters = f.counters;
    c.host_mvs = f.host_mvs;
    c.host_counters = f.host_counters;
    c.stats[0] = c.stats[1] = nullptr; // (hme_run_batch sets them when it runs the source pre-pass)
    c.l0pre = (uint32_t *) f.l0_pre;
}

size_t hme_table_bytes(int n) { return (size_t) n * sizeof(HmeDev); }

// n independent streams of identical geometry in lockstep: every level is ONE launch for all of them
int hme_run_batch(hipStream_t s, const HmeFrames *f, const HmeParams *hp, int n, void *h_table, void *d_table, StageProf *prof, int level_hi,
                  int level_lo, int phases)
{
    if (n <= 0) {
        return 0;
    }
    const HmeParams &g = hp[0];
    if (g.a.blk_w > 32 || g.a.blk_h > 32) { // (the general block routine stages its operand blocks in LDS: GenLds)
        fatal("motion search: blocks larger than 32 x 32 (dsv_encoder.c:1203-1211 makes 16 or 32)", __FILE__, __LINE__);
    }
    // a call that starts the search: job table, clears, source pre-pass
    const bool from_top = (phases & HME_PREPARE) && (level_hi < 0 || level_hi >= g.pyr_levels);
    if (level_hi < 0 || level_hi > g.pyr_levels) {
        level_hi = g.pyr_levels;
    }
    HmeDev *ht = (HmeDev *) h_table;
    if (from_top) {
        for (int k = 0; k < n; k++) {
            fill_hme_dev(ht[k], f[k], hp[k]);
        }
    }
    const HmeDev *tab = (const HmeDev *) d_table;
    int nlaunch = 0;
    // The fast block routines: 16 x 16 blocks, one geometry per pyramid level (make_ctx), the source statistics and the level-0
    // records of the pre-passes (memory for both handed in by the caller).  Anything else takes the general routine.
    // (A function of the jobs alone: a call that only runs the levels, behind a separate HME_PREPARE call, comes to the same answer.)
    const bool b16 = g.a.blk_w == 16 && g.a.blk_h == 16, b32 = g.a.blk_w == 32 && g.a.blk_h == 32, b32w = g.a.blk_w == 32 && g.a.blk_h == 16;
    bool fast = g_hme_fast != 0 && (b16 || b32 || b32w);
    for (int k = 0; k < n; k++) {
        fast = fast && uniform_geometry(f[k], g.pyr_levels) && f[k].src_stats != nullptr && f[k].l0_pre != nullptr;
    }
    auto fast_level = [&](int level) { return fast && level_all_fast(g.a, f[0].src[level], level); };
    const bool c422 = g.a.hshift == 1 && g.a.vshift == 0;
    const bool split = b16 && !c422 && fast_level(0) && (g_hme_split >= 0 ? g_hme_split != 0 : n * g.a.nbv <= kSplitMaxRows);
    if (from_top) {
        const int nb0x = g.a.nbh, nb0y = g.a.nbv, nb1x = g.pyr_levels >= 1 ? (g.a.nbh + 1) / 2 : 0, nb1y = g.pyr_levels >= 1 ? (g.a.nbv + 1) / 2 : 0;
        if (fast) {
            for (int k = 0; k < n; k++) {
                ht[k].stats[0] = (int4 *) f[k].src_stats;
                ht[k].stats[1] = nb1x ? (int4 *) f[k].src_stats + (size_t) nb0x * nb0y : nullptr;
            }
        }
        HIPCHK(hipMemcpyAsync(d_table, h_table, (size_t) n * sizeof(HmeDev), hipMemcpyHostToDevice, s));
        if (fast && (b32 || b32w)) {
            const int per_wg = 2, total = nb0x * nb0y + nb1x * nb1y;
            DSV2_LAUNCH(k_hme_src_stats32_b, dim3((total + per_wg - 1) / per_wg, n), dim3(64), 0, s, tab, nb0x, nb0y, nb1x, nb1y, per_wg, g.a.blk_h);
        }
        if (fast && b16) {
            // whole blocks four to a wavefront, clipped blocks of the last block row / column one to a wavefront
            bool four = true;
            for (int k = 0; k < n && four; k++) {
                for (int l = 0; l <= (nb1x ? 1 : 0); l++) {
                    four = four && (((uintptr_t) f[k].src[l].data | (uintptr_t) f[k].ogr[l].data | (uintptr_t) f[k].src[l].stride | (uintptr_t) f[k].ogr[l].stride) & 15) == 0;
                }
            }
            int fb0x = 0, fb0y = 0, fb1x = 0, fb1y = 0;
            if (four) {
                fb0x = std::min(f[0].src[0].w / 16, nb0x);
                fb0y = std::min(f[0].src[0].h / 16, nb0y);
                if (nb1x) {
                    fb1x = std::min(f[0].src[1].w / 16, nb1x);
                    fb1y = std::min(f[0].src[1].h / 16, nb1y);
                }
                const int groups = ((fb0x + 3) / 4) * fb0y + ((fb1x + 3) / 4) * fb1y;
                if (groups > 0) {
                    DSV2_LAUNCH(k_hme_src_stats4_b, dim3(groups, n), dim3(64), 0, s, tab, nb0x, nb1x, fb0x, fb0y, fb1x, fb1y);
                }
            }
            const int per_wg = 4, total = nb0x * nb0y - fb0x * fb0y + nb1x * nb1y - fb1x * fb1y;
            if (total > 0) {
                DSV2_LAUNCH(k_hme_src_stats_b, dim3((total + per_wg - 1) / per_wg, n), dim3(64), 0, s, tab, nb0x, nb0y, nb1x, nb1y, fb0x, fb0y, fb1x, fb1y, per_wg);
            }
        }
        // one clear for all levels; each level's last row then re-arms the hand-off words itself
        const int nwords = g.a.nbh * g.a.nbv * (int) (sizeof(DSV_MV) / 4);
        DSV2_LAUNCH(k_hme_clear_b, dim3((nwords + 2047) / 2048, n, g.pyr_levels + 1), dim3(256), 0, s, tab, -1, nwords, 1);
    }
    const int parts = n < 8 ? n : 8; // ticket partitions (take_row): 8 XCDs on MI355X
    for (int level = level_hi; (phases & HME_LEVELS) && level >= level_lo; level--) {
        const int step = 1 << level;
        const int nbx = (g.a.nbh + step - 1) / step, nby = (g.a.nbv + step - 1) / step;
        // (three workers per SIMD for the kernels whose wavefronts are small enough to share a SIMD with the other groups' kernels;
        // the general routine holds 224 registers -- two per SIMD is all that fits, a third set of workers would only queue)
        const bool two_per_simd = !fast_level(level);
        const int persist = g_hme_persist > 0 ? g_hme_persist : 3072;
        const int workers = std::min(two_per_simd ? std::min(persist, 2048) : persist, n * nby);
        if (level == 0 && g.pyr_levels == 0 && split) { // (no level above: the pre-pass has nothing to wait for)
            DSV2_LAUNCH(k_hme_l0_pre_b, dim3((nbx * nby + 1) / 2, n), dim3(64), 0, s, tab, nbx, nby, 2);
        }
        if (prof && level == 0) {
            prof->begin(s, ST_HME_L0);
        }
        if (level == 0 && (b32 || b32w) && fast_level(0)) {
            DSV2_LAUNCH(b32 ? k_hme_rows_l0_32 : k_hme_rows_l0_32w, dim3(workers), dim3(64), 0, s, tab, level, nbx, parts, n, nby);
        } else if (level == 0 && fast_level(0)) {
            auto pk = c422 ? k_hme_rows_l0_422 : g.a.hshift == 0 ? (split ? k_hme_rows_l0s_444 : k_hme_rows_l0_444) : (split ? k_hme_rows_l0s : k_hme_rows_l0);
            DSV2_LAUNCH(pk, dim3(workers), dim3(64), 0, s, tab, level, nbx, parts, n, nby);
        } else if (level > 0 && fast_level(level)) {
            DSV2_LAUNCH(b32 ? k_hme_rows_lx32 : b32w ? k_hme_rows_lx32w : k_hme_rows_lx, dim3(workers), dim3(64), 0, s, tab, level, nbx, parts, n, nby);
        } else {
            DSV2_LAUNCH(k_hme_rows_general, dim3(workers), dim3(64), 0, s, tab, level, nbx, parts, n, nby);
        }
        if (prof && level == 0) {
            prof->end(s, ST_HME_L0, n, 1);
        }
        nlaunch++;
        // what of level 0 does not depend on the order of its blocks (hme_fast.h: hme_l0_pre_block): needs level 1's field and its
        // global motion, nothing else -- so it belongs to the call that runs level 1 (a caller that serialises level-0 launches --
        // the encoder's search token -- may run the coarser levels, and this with them, outside that)
        if (level == 1 && split) {
            const int per_wg = 2;
            DSV2_LAUNCH(k_hme_l0_pre_b, dim3((g.a.nbh * g.a.nbv + per_wg - 1) / per_wg, n), dim3(64), 0, s, tab, g.a.nbh, g.a.nbv, per_wg);
        }
    }
    HIPCHK(hipGetLastError());
    return nlaunch;
}

// one search (the per-stage seam, dsv_hme): the batch of one, so that the stage tests exercise the kernels the encoder runs
int hme_run(hipStream_t s, const HmeFrames &f_in, const HmeParams &hp)
{
    HmeFrames f = f_in;
    void *h_table = nullptr, *d_table = nullptr, *stats = nullptr, *l0pre = nullptr;
    HIPCHK(hipHostMalloc(&h_table, hme_table_bytes(1), hipHostMallocDefault));
    HIPCHK(hipMalloc(&d_table, hme_table_bytes(1)));
    if (f.src_stats == nullptr) {
        HIPCHK(hipMalloc(&stats, hme_src_stats_bytes(hp.a.nbh, hp.a.nbv)));
        f.src_stats = stats;
    }
    if (f.l0_pre == nullptr) {
        HIPCHK(hipMalloc(&l0pre, hme_l0_pre_bytes(hp.a.nbh, hp.a.nbv)));
        f.l0_pre = l0pre;
    }
    const int nlaunch = hme_run_batch(s, &f, &hp, 1, h_table, d_table);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(d_table));
    HIPCHK(hipHostFree(h_table));
    if (stats) {
        HIPCHK(hipFree(stats));
    }
    if (l0pre) {
        HIPCHK(hipFree(l0pre));
    }
    return nlaunch;
}

} // namespace dsv2

#ifdef DSV2_HME_PROF
// debugging build only: cumulative phase clocks of the level-0 search (see HME_MARK)
extern "C" void dsv2hip_debug_hme_prof(unsigned long long out[32])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(dsv2::g_hme_prof), 32 * sizeof(unsigned long long)));
}
#endif
