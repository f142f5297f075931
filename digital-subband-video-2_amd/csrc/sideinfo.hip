// sideinfo.hip -- the per-block side information of a P picture, coded on the GPU.
//
// What the reference's controller does with the motion field after the search (dsv_encoder.c:693-952: encode_stable_blocks,
// encode_motion): it finalises every vector (a skipped block transmits the zero vector, an intra block a full-pel one),
// derives the block's DSV_IS_* flag byte for the filters and the quantiser, and writes six small sub-streams -- three
// zero-run-length coded bit planes (stable / mode / EPRM) and three plain code streams (x and y vector differences against
// the median predictor, the intra sub-block masks and DCs).  The only thing sequential about it is where a code lands in
// its sub-stream (the sum of the lengths before it) and, for the run-length planes, the zero run a block inherits from the
// blocks before it.  Everything else is a function of a block's own record and its left / top / top-left neighbours',
// whose FINAL values are themselves functions of their own raw records.
//
// One workgroup per stream: every thread walks a contiguous run of blocks twice -- pass A adds up code lengths (and the
// run-length state at both ends of its run), a serial scan over the 256 threads turns those into bit offsets, pass B
// recomputes the codes and ORs them into LDS images of the six sub-streams, which leave as coalesced words for the host's
// pinned memory together with their byte lengths.  The host only concatenates them into the packet (encoder.cpp phase H1).
// A frame the images cannot hold (or a code the 15-pair fast form of the exp-Golomb code cannot express) raises a flag and
// is coded by the host from the same field: same bytes either way.  The finalised field and the flag bytes are written to
// HBM for the kernels that follow, so neither travels through the host any more.
#include "sideinfo.h"
#include "prio.h"

#include "dev.h"

namespace dsv2 {

namespace {

constexpr int kThreads = 256;
// sub-stream images (bytes); a frame that needs more is coded by the host
constexpr int kCapRle = 2048, kCapMv = 16384, kCapSbim = 8192;
constexpr int kImgOff[SIDE_SUBS + 1] = {0, kCapRle, 2 * kCapRle, 2 * kCapRle + kCapMv, 2 * kCapRle + 2 * kCapMv, 2 * kCapRle + 2 * kCapMv + kCapSbim,
                                        3 * kCapRle + 2 * kCapMv + kCapSbim};
static_assert(kImgOff[SIDE_SUBS] == SIDE_IMG_BYTES, "sideinfo.h: SIDE_IMG_BYTES");

__device__ __forceinline__ int sar_i(int v, int s) { return v >> s; } // arithmetic on int (dsv.h DSV_SAR)
__device__ __forceinline__ int sar_r(int v, int s) { return (v + (1 << (s - 1))) >> s; }

// bit i of a 16-bit value moves to bit 2i
__device__ __forceinline__ unsigned spread16(unsigned x)
{
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// interleaved exp-Golomb (bs.c:186): code and length; nb > 15 pairs is left to the host (len = -1)
struct Code {
    unsigned bits;
    int len;
};
__device__ __forceinline__ Code ueg(unsigned v)
{
    v++;
    const int nb = 31 - __clz((int) v);
    Code c;
    c.bits = (spread16(v & ((1u << nb) - 1u)) << 1) | 1u;
    c.len = nb <= 15 ? 2 * nb + 1 : -1;
    return c;
}
__device__ __forceinline__ Code seg(int v) // bs.c:210: magnitude, then the sign of a non-zero value
{
    const unsigned a = (unsigned) abs(v);
    Code c = ueg(a);
    if (a && c.len > 0) {
        c.bits = (c.bits << 1) | (v < 0 ? 1u : 0u);
        c.len++;
    }
    return c;
}

// MSB-first: bit position p of the stream is bit (31 - p % 32) of word p / 32 (the words are byte-swapped on the way out)
__device__ __forceinline__ void put(uint32_t *img, unsigned pos, unsigned bits, int len)
{
    if (len <= 0) {
        return;
    }
    const unsigned w = pos >> 5, sh = pos & 31u;
    const unsigned long long x = (unsigned long long) bits << (64 - len - (int) sh); // len <= 32, sh <= 31
    atomicOr(&img[w], (uint32_t) (x >> 32));
    if ((uint32_t) x) {
        atomicOr(&img[w + 1], (uint32_t) x);
    }
}

__device__ __forceinline__ int pred1(int left, int top, int topleft) // dsv.c:324
{
    const int dif = left + top - topleft;
    return abs(dif - left) < abs(dif - top) ? left : top;
}

// the vector a block finally transmits, from its raw record: {x | y << 16}
__device__ __forceinline__ uint32_t final_all(uint32_t all, uint32_t flags)
{
    if ((flags >> DSV_MV_BIT_SKIP) & 1u) {
        return 0u;
    }
    if ((flags >> DSV_MV_BIT_INTRA) & 1u) {
        const int x = sar_i((int) (int16_t) (all & 0xffffu), 2) * 4, y = sar_i((int) (int16_t) (all >> 16), 2) * 4;
        return ((uint32_t) x & 0xffffu) | ((uint32_t) y << 16);
    }
    return all;
}
__device__ __forceinline__ int vx(uint32_t all) { return (int) (int16_t) (all & 0xffffu); }
__device__ __forceinline__ int vy(uint32_t all) { return (int) (int16_t) (all >> 16); }

// state of one zero-run-length plane inside a thread's run of blocks
struct Rle {
    int ones = 0, lead = 0, run = 0, rest_bits = 0; // ones seen; zeros before the first; zeros since the last; code bits after the first one
    bool bad = false;
};

struct Walk {
    // per-thread results of pass A
    Rle r[3];         // stable, mode, eprm
    int mv_bits[2] = {0, 0}, sbim_bits = 0;
    bool bad = false;
};

// everything one block contributes; EMIT: write the codes (pass B), else only count (pass A)
// codes: false = positions are not valid (the host will code the frame): form the field and the flag bytes, write no code
template <bool EMIT>
__device__ __forceinline__ void block(const SideJob &J, int nbh, int idx, Walk &W, uint32_t *img, unsigned (&at)[SIDE_SUBS], int (&carry)[3], bool codes)
{
    const int j = idx / nbh, i = idx - j * nbh;
    const DSV_MV *mv = &J.raw[idx];
    const uint4 rec = *(const uint4 *) mv;
    const uint32_t flags = rec.y;
    const bool skip = (flags >> DSV_MV_BIT_SKIP) & 1u, intra = (flags >> DSV_MV_BIT_INTRA) & 1u, eprm = (flags >> DSV_MV_BIT_EPRM) & 1u;
    const uint32_t fin = final_all(rec.x, flags);
    const bool stable = !intra && skip;
    unsigned bd = (intra ? DSV_IS_INTRA : 0u) | (stable ? DSV_IS_SKIP : 0u) | (((flags >> DSV_MV_BIT_SIMCMPLX) & 1u) ? DSV_IS_SIMCMPLX : 0u) |
                  (eprm ? DSV_IS_EPRM : 0u);
    auto rle_bit = [&](int k, int bit) {
        Rle &R = W.r[k];
        if (!bit) {
            R.run++;
            if (EMIT) {
                carry[k]++;
            }
            return;
        }
        if (EMIT) {
            const int sub = k == 0 ? SIDE_STABLE : (k == 1 ? SIDE_MODE : SIDE_EPRM);
            const Code c = ueg((unsigned) carry[k]);
            if (codes) {
                put(img + kImgOff[sub] / 4, at[sub], c.bits, c.len);
            }
            at[sub] += (unsigned) c.len;
            carry[k] = 0;
        } else {
            if (R.ones == 0) {
                R.lead = R.run;
            } else {
                const Code c = ueg((unsigned) R.run);
                R.bad = R.bad || c.len < 0;
                R.rest_bits += c.len;
            }
        }
        R.ones++;
        R.run = 0;
    };
    rle_bit(0, J.inv_stable ? !stable : stable);
    if (skip) {
        bd |= DSV_IS_STABLE;
    } else {
        // the median predictor over the FINAL left / top / top-left vectors (dsv.c:375)
        uint32_t l = 0, t = 0, tl = 0, lf = 0, tf = 0;
        if (i > 0) {
            const uint2 h = *(const uint2 *) (mv - 1);
            l = final_all(h.x, h.y);
            lf = h.y;
        }
        if (j > 0) {
            const uint2 h = *(const uint2 *) (mv - nbh);
            t = final_all(h.x, h.y);
            tf = h.y;
        }
        if (i > 0 && j > 0) {
            const uint2 h = *(const uint2 *) (mv - nbh - 1);
            tl = final_all(h.x, h.y);
        }
        int px = pred1(vx(l), vx(t), vx(tl)), py = pred1(vy(l), vy(t), vy(tl));
        int cvx = vx(fin), cvy = vy(fin);
        if (intra) {
            px = sar_r(px, 2);
            py = sar_r(py, 2);
            cvx = sar_i(vx(rec.x), 2);
            cvy = sar_i(vy(rec.x), 2);
            const unsigned submask = rec.w & 0xffu, dc = (rec.z >> 16) & 0xffffu;
            unsigned bits = submask == DSV_MASK_ALL_INTRA ? 1u : (submask & 0xfu);
            int len = submask == DSV_MASK_ALL_INTRA ? 1 : 5;
            if (dc & DSV_SRC_DC_PRED) {
                bits = (bits << 9) | 0x100u | (dc & 0xffu);
                len += 9;
            } else {
                bits <<= 1;
                len += 1;
            }
            if (EMIT) {
                if (codes) {
                    put(img + kImgOff[SIDE_SBIM] / 4, at[SIDE_SBIM], bits, len);
                }
                at[SIDE_SBIM] += (unsigned) len;
            } else {
                W.sbim_bits += len;
            }
        }
        const Code cx = seg(cvx - px), cy = seg(cvy - py);
        if (EMIT) {
            if (codes) {
                put(img + kImgOff[SIDE_MVX] / 4, at[SIDE_MVX], cx.bits, cx.len);
                put(img + kImgOff[SIDE_MVY] / 4, at[SIDE_MVY], cy.bits, cy.len);
            }
            at[SIDE_MVX] += (unsigned) cx.len;
            at[SIDE_MVY] += (unsigned) cy.len;
        } else {
            W.bad = W.bad || cx.len < 0 || cy.len < 0;
            W.mv_bits[0] += cx.len;
            W.mv_bits[1] += cy.len;
        }
        // dsv_neighbordif (dsv.c:403, :430) on the final field
        int nd = 0;
        {
            const int cx_ = vx(fin), cy_ = vy(fin);
            if (!(abs(cx_) < 2 && abs(cy_) < 2)) {
                int lx = cx_, ly = cy_, tx = cx_, ty = cy_;
                if (i > 0 && l && !((lf >> DSV_MV_BIT_SKIP) & 1u)) {
                    lx = vx(l);
                    ly = vy(l);
                }
                if (j > 0 && t && !((tf >> DSV_MV_BIT_SKIP) & 1u)) {
                    tx = vx(t);
                    ty = vy(t);
                }
                nd = (abs(lx - cx_) + abs(ly - cy_) + abs(tx - cx_) + abs(ty - cy_)) / 3;
            }
        }
        if (nd > 8) {
            bd |= DSV_IS_STABLE;
        }
        rle_bit(1, J.inv_mode ? !intra : intra);
        rle_bit(2, J.inv_eprm ? !eprm : eprm);
    }
    if (EMIT) {
        uint4 o = rec;
        o.x = fin;
        *(uint4 *) &J.final_mvs[idx] = o;
        J.bd[idx] = (uint8_t) bd;
    }
}

} // namespace

// grid = streams; dynamic LDS: none (static images).  The six images plus the per-thread scan arrays are ~70 KB of static LDS:
// this kernel is written for the 160 KB of a gfx950 CU (a 64 KB part would need the scan arrays folded into the image).
static_assert(SIDE_IMG_BYTES + 25 * kThreads * 4 < 150 * 1024, "k_side_info: static LDS beyond a gfx950 CU");
__global__ __launch_bounds__(kThreads) void k_side_info(const SideJob *__restrict__ tab, int nbh, int nbv)
{
    DSV2_KERNEL_PRIO();
    __shared__ uint32_t img[SIDE_IMG_BYTES / 4 + 2];
    __shared__ int s_ones[3][kThreads], s_lead[3][kThreads], s_trail[3][kThreads], s_rest[3][kThreads];
    __shared__ int s_bits[3][kThreads];        // mvx, mvy, sbim bits per thread
    __shared__ unsigned s_off[SIDE_SUBS][kThreads]; // bit offset of every thread in every sub-stream
    __shared__ int s_carry[3][kThreads];       // zero run a thread inherits in each run-length plane
    __shared__ int s_bytes[SIDE_SUBS], s_bad;
    const SideJob &J = tab[blockIdx.x];
    const int nblk = nbh * nbv, tid = threadIdx.x;
    const int bpt = (nblk + kThreads - 1) / kThreads;
    const int b0 = min(nblk, tid * bpt), b1 = min(nblk, b0 + bpt);
    for (int w = tid; w < SIDE_IMG_BYTES / 4 + 2; w += kThreads) {
        img[w] = 0;
    }
    if (tid == 0) {
        s_bad = nblk > 8 * (kCapRle - 8) ? 1 : 0;
    }
    // ---- pass A: lengths ----
    Walk W;
    unsigned at[SIDE_SUBS] = {0, 0, 0, 0, 0, 0};
    int carry[3] = {0, 0, 0};
    for (int idx = b0; idx < b1; idx++) {
        block<false>(J, nbh, idx, W, img, at, carry, false);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        s_ones[k][tid] = W.r[k].ones;
        s_lead[k][tid] = W.r[k].lead;
        s_trail[k][tid] = W.r[k].run;
        s_rest[k][tid] = W.r[k].rest_bits;
    }
    s_bits[0][tid] = W.mv_bits[0];
    s_bits[1][tid] = W.mv_bits[1];
    s_bits[2][tid] = W.sbim_bits;
    __syncthreads();
    if (W.bad || W.r[0].bad || W.r[1].bad || W.r[2].bad) {
        atomicOr(&s_bad, 1);
    }
    // ---- serial scans over the threads: six short ones, side by side ----
    if (tid < 3) { // run-length plane tid: the run every thread inherits, the bit offsets, the closing code
        const int sub = tid == 0 ? SIDE_STABLE : (tid == 1 ? SIDE_MODE : SIDE_EPRM);
        int c = 0;
        unsigned off = 0;
        bool bad = false;
        for (int t = 0; t < kThreads; t++) {
            s_carry[tid][t] = c;
            s_off[sub][t] = off;
            if (s_ones[tid][t] > 0) {
                const Code f = ueg((unsigned) (c + s_lead[tid][t]));
                bad = bad || f.len < 0;
                off += (unsigned) (f.len + s_rest[tid][t]);
                c = s_trail[tid][t];
            } else {
                c += s_trail[tid][t];
            }
        }
        const Code last = ueg((unsigned) c); // RleWriter::finish (bs.c:322)
        bad = bad || last.len < 0 || (off + 32) / 8 + 8 > (unsigned) kCapRle;
        if (!bad) {
            put(img + kImgOff[sub] / 4, off, last.bits, last.len);
        }
        s_bytes[sub] = (int) ((off + (unsigned) last.len + 7) >> 3);
        if (bad) {
            atomicOr(&s_bad, 1);
        }
    } else if (tid < 6) {
        const int k = tid - 3, sub = k == 0 ? SIDE_MVX : (k == 1 ? SIDE_MVY : SIDE_SBIM);
        unsigned off = 0;
        for (int t = 0; t < kThreads; t++) {
            s_off[sub][t] = off;
            off += (unsigned) s_bits[k][t];
        }
        s_bytes[sub] = (int) ((off + 7) >> 3);
        if ((off + 40) / 8 + 8 > (unsigned) (k == 2 ? kCapSbim : kCapMv)) {
            atomicOr(&s_bad, 1);
        }
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    // ---- pass B: codes into the images; the finalised field and the flag bytes into HBM (whatever becomes of the codes) ----
    {
        Walk W2;
#pragma unroll
        for (int s = 0; s < SIDE_SUBS; s++) {
            at[s] = bad ? 0u : s_off[s][tid];
        }
        carry[0] = s_carry[0][tid];
        carry[1] = s_carry[1][tid];
        carry[2] = s_carry[2][tid];
        for (int idx = b0; idx < b1; idx++) {
            block<true>(J, nbh, idx, W2, img, at, carry, !bad);
        }
    }
    __syncthreads();
    // ---- out: whole words, byte order restored; lengths and the flag ----
    if (!bad) {
        for (int s = 0; s < SIDE_SUBS; s++) {
            const int nw = (s_bytes[s] + 3) >> 2;
            uint32_t *dst = (uint32_t *) (J.out + kImgOff[s]);
            for (int w = tid; w < nw; w += kThreads) {
                dst[w] = __builtin_bswap32(img[kImgOff[s] / 4 + w]);
            }
        }
    }
    if (tid < SIDE_SUBS) {
        J.info[1 + tid] = bad ? 0 : s_bytes[tid];
    }
    if (tid == 0) {
        J.info[0] = bad ? 1 : 0;
    }
}

void side_info_batch(hipStream_t s, const SideJob *d_jobs, int n, int nbh, int nbv)
{
    if (n > 0) {
        DSV2_LAUNCH(k_side_info, dim3(n), dim3(kThreads), 0, s, d_jobs, nbh, nbv);
    }
}

int side_image_offset(int sub) { return kImgOff[sub]; }

} // namespace dsv2
