// entropy_gpu.hip -- the symbol-stream half of hzcc_enc (reference src/hzcc.c:230-447, :586-613) and the bit codes
// of src/bs.c (UEG :132, SEG :175, NEG :206, adaptive Rice :237) as kernels: the picture packet's three plane
// sections are assembled on the GPU and only their bytes cross PCIe (50 KB per 1080p P picture instead of 1.4 MB of
// symbols; 345 KB instead of 10.5 MB for an intra picture), and the host's per-symbol loop disappears.
//
// What is sequential in the reference is ONE integer per plane: the adaptive Rice state `vk` (bs.c:237-251:
// k = vk >> damp codes the value, then vk moves up by one if the unary part was non-empty, else down by one,
// floored at 0).  Everything else -- run lengths, code words, code lengths -- is a pure function of a symbol, its
// predecessor's position and the vk it meets.  The state is made parallel exactly, with no speculation:
//
//   1 k_ent_planes   per stream: where each plane's symbols start in the compacted (position, value) list, and how many of
//                    each plane's symbols are coded without the state (the LL prefix)
//   2 k_ent_tables   per chunk of 1024 symbols: the chunk's TRANSFER FUNCTION vk_in -> vk_out as a 256-entry table,
//                    by walking all 256 start states through the chunk at once (two states per lane in packed 16-bit
//                    arithmetic against one threshold per symbol, staged in LDS) UNTIL all trajectories have joined two
//                    neighbouring values (m, m + 1) -- some 250 symbols: the states above every threshold march down
//                    together.  Leaves each symbol's threshold as a byte (T >> 3) for the two walks below.
//   2b k_ent_pair    the rest of the chunk is the walk of that ONE pair: a lane per chunk, 64 chunks a wavefront
//                    (round 4: until then one wavefront per chunk walked it with 63 lanes idle)
//   3 k_ent_chain    per plane: vk at the start of every chunk by following the tables (one lookup per chunk -- and the
//                    pair's result where the chunk's trajectories had joined --, the tables staged 32 chunks at a time in LDS)
//   4 k_ent_walk     a lane per chunk again: the walk from the chunk's now known start state, the state every symbol meets
//                    written over its threshold byte
//   4b k_ent_bits    per chunk, a lane per symbol: the states into Rice parameters and the chunk's total code length
//   5 k_ent_layout   per stream: exclusive scan of the chunk lengths, byte layout of the three plane sections
//   6 k_ent_zero / k_ent_emit   every symbol ORs its code words into an LDS image of its chunk at its bit offset; the
//                    image leaves as whole words (only a chunk's first and last word are ORed into the zeroed output)
//   7 k_ent_out      the finished bytes and their sizes to pinned host memory
//
//
// A state beyond 255 (k >= 32 at damp 3, never seen on real pictures) or a plane that outgrows its buffer raises a
// flag; the host then codes that picture from the symbol list as before (entropy.cpp) -- same bytes either way.
#include "entropy_gpu.h"
#include "prio.h"

namespace dsv2 {

namespace {

constexpr int kStates = 256;

// info words (device and, mirrored, pinned host)
enum { EI_FLAGS = 0, EI_PSTART = 1 /* 1..3 */, EI_N = 4, EI_PBYTES = 5 /* 5..7 */, EI_TOTAL = 8, EI_NCH = 9 /* 9..11 */, EI_SYMBIT = 12 /* 12..14 */,
       EI_NSKIP = 16 /* 16..18: device only */, EI_WORDS = 32 };

__device__ __forceinline__ uint64_t spread32(uint32_t x) // bit i -> bit 2i
{
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// interleaved exp-Golomb (bs.c:132): for every bit of v + 1 below its leading one a 0 followed by that bit, then a 1
__device__ __forceinline__ void ueg_code(uint32_t v, uint64_t &code, int &len)
{
    uint32_t x = v + 1u;
    int nb = 31 - __clz((int) x);
    uint32_t low = nb ? (x & (0xffffffffu >> (32 - nb))) : 0u;
    code = (spread32(low) << 1) | 1ull;
    len = 2 * nb + 1;
}

__device__ __forceinline__ int ueg_len(uint32_t v) { return 2 * (31 - __clz((int) (v + 1u))) + 1; }

__device__ __forceinline__ int seg_of(const EntGeom &g, int c, uint32_t p)
{
    int seg = 0;
#pragma unroll
    for (int k = 1; k < 10; k++) {
        seg += p >= (uint32_t) g.base[c][k];
    }
    return seg;
}

__device__ __forceinline__ uint32_t rice_u(int32_t v) { return ((uint32_t) (2 * v) ^ (v < 0 ? ~0u : 0u)) - 1u; }
__device__ __forceinline__ int bitlen(uint32_t u) { return u ? 32 - __clz((int) u) : 0; }

// the stream's bit `bitpos` is bit 7 - (bitpos & 7) of byte bitpos >> 3: ORs the `len` low bits of `code`, MSB first
__device__ __forceinline__ void put_code(uint32_t *out32, uint32_t bitpos, uint64_t code, int len)
{
    uint64_t c64 = code << (64 - len);
    uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    uint64_t hi = c64 >> sh;
    uint32_t w0 = (uint32_t) (hi >> 32), w1 = (uint32_t) hi, w2 = sh ? (uint32_t) ((c64 << (64 - sh)) >> 32) : 0u;
    if (w0) {
        atomicOr(&out32[w], __builtin_bswap32(w0));
    }
    if (w1) {
        atomicOr(&out32[w + 1], __builtin_bswap32(w1));
    }
    if (w2) {
        atomicOr(&out32[w + 2], __builtin_bswap32(w2));
    }
}

// the same into a chunk's LDS image of its words (bit 0 of the image = bit 0 of output word `w_base`)
__device__ __forceinline__ void put_code_lds(uint32_t *img, uint32_t bitpos, uint64_t code, int len)
{
    uint64_t c64 = code << (64 - len);
    uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    uint64_t hi = c64 >> sh;
    uint32_t w0 = (uint32_t) (hi >> 32), w1 = (uint32_t) hi, w2 = sh ? (uint32_t) ((c64 << (64 - sh)) >> 32) : 0u;
    if (w0) {
        atomicOr(&img[w], __builtin_bswap32(w0));
    }
    if (w1) {
        atomicOr(&img[w + 1], __builtin_bswap32(w1));
    }
    if (w2) {
        atomicOr(&img[w + 2], __builtin_bswap32(w2));
    }
}

__device__ __forceinline__ int wave_incl_scan_u(unsigned v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned t = (unsigned) __shfl_up((int) v, d, 64);
        if (lane >= d) {
            v += t;
        }
    }
    return (int) v;
}

// ---- 1 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ent_planes(const EntJob *__restrict__ tab, EntGeom g)
{
    DSV2_KERNEL_PRIO();
    const EntJob &J = tab[blockIdx.x];
    const int lane = threadIdx.x;
    const int Nall = *J.total;
    const bool over = Nall > J.list_cap; // more symbols than the lists hold: nothing of this picture is coded here (flag 16)
    const int N = over ? 0 : Nall;
    int res = 0;
    if (lane >= 1 && lane <= 5) { // 1, 2: first symbol whose position lies in plane `lane`; 3..5: first of plane lane - 3 behind its LL region
        uint32_t key = lane <= 2 ? (uint32_t) g.qv_off[lane] : (uint32_t) (g.qv_off[lane - 3] + g.base[lane - 3][1]);
        int lo = 0, hi = N;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (J.pos[mid] < key) {
                lo = mid + 1;
            } else {
                hi = mid;
            }
        }
        res = lo;
    }
    int p1 = __shfl(res, 1, 64), p2 = __shfl(res, 2, 64);
    if (lane == 0) {
        int *I = J.info;
        I[EI_FLAGS] = over ? ENT_LIST_OVERFLOW : 0;
        I[EI_PSTART + 0] = 0;
        I[EI_PSTART + 1] = p1;
        I[EI_PSTART + 2] = p2;
        I[EI_N] = N;
        I[EI_NCH + 0] = (p1 + kEntChunk - 1) / kEntChunk;
        I[EI_NCH + 1] = (p2 - p1 + kEntChunk - 1) / kEntChunk;
        I[EI_NCH + 2] = (N - p2 + kEntChunk - 1) / kEntChunk;
    }
    if (lane >= 3 && lane <= 5) { // symbols of the plane that are coded without the adaptive state: a prefix of its list
        const int c = lane - 3, first = c == 0 ? 0 : (c == 1 ? p1 : p2);
        J.info[EI_NSKIP + c] = res - first;
    }
}

struct PlaneSpan {
    int first, end, nch, cbase; // symbols [first, end) of the plane, its chunk count and first chunk slot
};
__device__ __forceinline__ PlaneSpan plane_span(const int *I, int c)
{
    PlaneSpan s;
    s.first = I[EI_PSTART + c];
    s.end = c == 2 ? I[EI_N] : I[EI_PSTART + c + 1];
    s.nch = I[EI_NCH + c];
    s.cbase = (c > 0 ? I[EI_NCH] : 0) + (c > 1 ? I[EI_NCH + 1] : 0);
    return s;
}

// ---- 2 ----------------------------------------------------------------------------------------------------
// A Rice-coded symbol moves the state up iff bitlen(u) > (vk >> damp), i.e. iff vk < T with T = bitlen(u) << damp: one
// threshold per symbol, staged in LDS as a splat pair of 16-bit values.  Every lane walks TWO start states through the
// chunk in packed 16-bit arithmetic (128 lanes = 256 states; five v_pk instructions a symbol for the pair).  The
// symbols coded without the state (LL region, seg 0) are a prefix of the plane's list (positions ascend): the walk
// simply starts behind them.
typedef short pk16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pk16 pk_step(pk16 vk, uint32_t tm1_splat) // tm1_splat = (T - 1) in both halves
{
    pk16 x = __builtin_bit_cast(pk16, tm1_splat) - vk;   // negative: vk >= T, the state moves down
    pk16 m = x >> (pk16){15, 15};                         // -1 down, 0 up
    uint32_t step = (__builtin_bit_cast(uint32_t, m) << 1) | 0x00010001u; // -1 / +1 per half (the bit the low half's sign
                                                                          // shifts into the high half is the one OR'd anyway)
    vk = vk + __builtin_bit_cast(pk16, step);
    return __builtin_elementwise_max(vk, (pk16){0, 0});
}

__device__ __forceinline__ uint32_t rice_threshold(int32_t v, int seg) // T of a symbol of subband index seg >= 1
{
    return (uint32_t) bitlen(rice_u(v)) << (3 + (seg - 1) / 3);
}

constexpr uint32_t kNotJoined = 0xffffffffu;

template <bool kLanes> // kLanes: stop where the trajectories have joined (k_ent_pair walks on), leave the threshold bytes behind
__global__ __launch_bounds__(kStates / 2) void k_ent_tables(const EntJob *__restrict__ tab, EntGeom g)
{
    DSV2_KERNEL_PRIO();
    __shared__ __attribute__((aligned(16))) uint32_t sT[kEntChunk];
    __shared__ int s_nskip;
    __shared__ int s_rng[2][2];
    __shared__ int s_pair[2]; // phase 2's result: NOT s_rng, which the other wavefront may still be reading for its last look
    const EntJob &J = tab[blockIdx.y];
    const int c = blockIdx.z;
    const PlaneSpan ps = plane_span(J.info, c);
    const uint32_t off = (uint32_t) g.qv_off[c];
    const int tid = threadIdx.x;
    for (int lc = blockIdx.x; lc < ps.nch; lc += gridDim.x) {
        // (the plane's span comes from memory: told to the compiler as wave-uniform, or the loop control below goes vector)
        const int first = __builtin_amdgcn_readfirstlane(ps.first + lc * kEntChunk);
        const int cnt = __builtin_amdgcn_readfirstlane(min(kEntChunk, ps.end - first));
        __syncthreads();
        if (tid == 0) {
            s_nskip = 0;
        }
        __syncthreads();
        int nskip_mine = 0;
#pragma unroll
        for (int j = 0; j < kEntChunk / (kStates / 2); j++) {
            int s = j * (kStates / 2) + tid;
            uint32_t t = 0, tc = 0;
            if (s < cnt) {
                int seg = seg_of(g, c, J.pos[first + s] - off);
                if (seg > 0) {
                    t = rice_threshold(J.val[first + s], seg); // T in 0 .. 1024, a multiple of 8
                    tc = t >> 3;
                    t = t - 1u;
                    t = (t & 0xffffu) | (t << 16);
                } else {
                    nskip_mine++;
                }
            }
            sT[s] = t;
            if (kLanes) { // (chunk-major, whole chunks: the walks read on behind a plane's last symbol)
                J.ksym[(size_t) (ps.cbase + lc) * kEntChunk + s] = (uint8_t) tc;
            }
        }
        if (nskip_mine) {
            atomicAdd(&s_nskip, nskip_mine);
        }
        __syncthreads();
        const int nskip = __builtin_amdgcn_readfirstlane(s_nskip); // uniform: scalar loop control
        pk16 vk = (pk16){(short) (2 * tid), (short) (2 * tid + 1)};
        // Phase 1: all 256 states in vector registers.  States above every threshold march down together and the ones that
        // meet merge for good (same-parity states never cross), so after a few hundred symbols all 256 trajectories sit on at
        // most two neighbouring values; from there (phase 2) ONE wavefront walks that pair and every start state takes the
        // result of the value it had joined.  The range of the states is looked at every 64 symbols.  (Walking the pair in
        // scalar registers instead trades each vector instruction for 1.3 scalar ones and lengthens the dependent chain:
        // measured 3 % slower end to end -- the scalar port is as loaded as the vector one in this workload.)
        const int qend = (cnt + 3) >> 2;
        int q = nskip >> 2;
        bool joined = false;
        int m = 0;
        while (q < qend) {
            const int qstop = min(qend, q + 16);
            for (; q < qstop; q++) {
                const uint4 t4 = *(const uint4 *) &sT[4 * q]; // same address in every lane: an LDS broadcast
                if (4 * q >= nskip && 4 * q + 4 <= cnt) {
                    vk = pk_step(pk_step(pk_step(pk_step(vk, t4.x), t4.y), t4.z), t4.w);
                } else { // the group holding the end of the LL prefix or the end of the chunk
                    const uint32_t tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (4 * q + b >= nskip && 4 * q + b < cnt) {
                            vk = pk_step(vk, tt[b]);
                        }
                    }
                }
            }
            if (q < qend && 4 * q - nskip >= 128) {
                int lo = min((int) vk.x, (int) vk.y), hi = max((int) vk.x, (int) vk.y);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    lo = min(lo, __shfl_xor(lo, o, 64));
                    hi = max(hi, __shfl_xor(hi, o, 64));
                }
                __syncthreads(); // (the previous look's readers are done)
                if ((tid & 63) == 0) {
                    s_rng[tid >> 6][0] = lo;
                    s_rng[tid >> 6][1] = hi;
                }
                __syncthreads();
                // (made scalar explicitly: a loop exit the compiler takes for divergent would put phase 2 in vector registers)
                lo = __builtin_amdgcn_readfirstlane(min(s_rng[0][0], s_rng[1][0]));
                hi = __builtin_amdgcn_readfirstlane(max(s_rng[0][1], s_rng[1][1]));
                if (hi - lo <= 1) {
                    joined = true;
                    m = lo;
                    break;
                }
            }
        }
        if (kLanes) {
            if (tid == 0) {
                J.chunk_join[ps.cbase + lc] = make_uint2(joined ? (uint32_t) m | ((uint32_t) (4 * q) << 16) : kNotJoined, 0u);
            }
        } else if (joined) {
            if (tid < 64) { // the pair (m, m + 1), the same five instructions a symbol, ONE wavefront
                pk16 x = (pk16){(short) m, (short) (m + 1)};
                for (; q < qend; q++) {
                    const uint4 t4 = *(const uint4 *) &sT[4 * q];
                    if (4 * q >= nskip && 4 * q + 4 <= cnt) {
                        x = pk_step(pk_step(pk_step(pk_step(x, t4.x), t4.y), t4.z), t4.w);
                    } else {
                        const uint32_t tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            if (4 * q + b >= nskip && 4 * q + b < cnt) {
                                x = pk_step(x, tt[b]);
                            }
                        }
                    }
                }
                if (tid == 0) {
                    s_pair[0] = x.x;
                    s_pair[1] = x.y;
                }
            }
            __syncthreads();
            const int x0 = s_pair[0], x1 = s_pair[1];
            vk = (pk16){(short) ((int) vk.x == m ? x0 : x1), (short) ((int) vk.y == m ? x0 : x1)};
        }
        *(uint32_t *) &J.tables[(size_t) (ps.cbase + lc) * kStates + 2 * tid] = __builtin_bit_cast(uint32_t, vk);
    }
}

// ---- 2b ---------------------------------------------------------------------------------------------------
// The pair (m, m + 1) a chunk's trajectories had joined at symbol q0, walked to the chunk's end: a lane per chunk.  Lane l's
// thresholds are 1 KB apart from lane l + 1's: a load of 16 bytes a lane brings 16 symbols, fetched one group ahead.
// (A plane's last chunk is walked over its zero padding: its transfer function is never used.)
__device__ __forceinline__ uint32_t tm1_splat(uint32_t w, int b) // byte b of w = T >> 3: (T - 1) in both halves
{
    const uint32_t t = (((w >> (8 * b)) & 0xffu) << 3) - 1u;
    return (t & 0xffffu) * 0x10001u;
}

__global__ __launch_bounds__(64) void k_ent_pair(const EntJob *__restrict__ tab)
{
    DSV2_KERNEL_PRIO();
    const EntJob &J = tab[blockIdx.y];
    const int c = blockIdx.z;
    const PlaneSpan ps = plane_span(J.info, c);
    for (int lc = blockIdx.x * 64 + threadIdx.x; lc - (int) threadIdx.x < ps.nch; lc += gridDim.x * 64) {
        const bool live = lc < ps.nch;
        const uint32_t jn = live ? J.chunk_join[ps.cbase + lc].x : kNotJoined;
        const bool joined = jn != kNotJoined;
        const int m = (int) (jn & 0xffffu), q0 = joined ? (int) (jn >> 16) : kEntChunk; // (q0: a multiple of 4)
        int g0 = q0 >> 4;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            g0 = min(g0, __shfl_xor(g0, o, 64));
        }
        g0 = __builtin_amdgcn_readfirstlane(g0);
        if (g0 >= kEntChunk / 16) {
            continue;
        }
        const uint4 *src = (const uint4 *) (J.ksym + (size_t) (ps.cbase + (live ? lc : 0)) * kEntChunk);
        pk16 x = (pk16){(short) m, (short) (m + 1)};
        uint4 cur = src[g0];
        for (int gq = g0; gq < kEntChunk / 16; gq++) {
            const uint4 nxt = src[min(gq + 1, kEntChunk / 16 - 1)];
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const pk16 y = pk_step(x, tm1_splat(w[k >> 2], k & 3));
                x = 16 * gq + k >= q0 ? y : x;
            }
            cur = nxt;
        }
        if (live && joined) {
            J.chunk_join[ps.cbase + lc].y = (uint32_t) (uint16_t) x.x | ((uint32_t) (uint16_t) x.y << 16);
        }
    }
}

// ---- 3 ----------------------------------------------------------------------------------------------------
// One wavefront per plane (grid = (3, streams)).  A link of the chain is one table lookup, but each waits for the one
// before it: straight from global memory that is a memory round trip per chunk (~1.4 us under load, 250 chunks a luma
// plane).  So the wavefront stages the tables of 32 chunks at a time in LDS (one coalesced round trip for 16 KB) and lane
// 0 follows the chain through them there.
constexpr int kChainBatch = 32;
__global__ __launch_bounds__(64) void k_ent_chain(const EntJob *__restrict__ tab, int lanes)
{
    DSV2_KERNEL_PRIO();
    __shared__ __attribute__((aligned(16))) uint16_t st[kChainBatch * kStates];
    __shared__ uint2 sj[kChainBatch];
    const EntJob &J = tab[blockIdx.y];
    const int c = blockIdx.x;
    const PlaneSpan ps = plane_span(J.info, c);
    const int lane = threadIdx.x;
    int vk = 0;
    bool ovf = false;
    for (int base = 0; base < ps.nch; base += kChainBatch) {
        const int nb = min(kChainBatch, ps.nch - base);
        const uint4 *src = (const uint4 *) (J.tables + (size_t) (ps.cbase + base) * kStates); // 512 bytes a chunk: 16-byte aligned
        uint4 *dst = (uint4 *) st;
        for (int i = lane; i < nb * (kStates * 2 / 16); i += 64) {
            dst[i] = src[i];
        }
        if (lane < nb) {
            sj[lane] = lanes ? J.chunk_join[ps.cbase + base + lane] : make_uint2(kNotJoined, 0u);
        }
        __syncthreads();
        if (lane == 0) {
            for (int k = 0; k < nb; k++) {
                J.chunk_vk[ps.cbase + base + k] = (uint16_t) vk;
                vk = st[k * kStates + vk];
                const uint2 jn = sj[k];
                if (jn.x != kNotJoined) { // the table holds where the start state stood when all had joined m, m + 1: k_ent_pair took those on
                    vk = vk == (int) (jn.x & 0xffffu) ? (int) (jn.y & 0xffffu) : (int) (jn.y >> 16);
                }
                if (vk >= kStates) {
                    ovf = true;
                    vk = kStates - 1;
                }
            }
        }
        __syncthreads();
    }
    if (lane == 0 && ovf) {
        atomicOr(&J.info[EI_FLAGS], 1);
    }
}

// ---- 4 (round 4) ------------------------------------------------------------------------------------------
// The walk from the chunk's known start state, a lane per chunk: the state each symbol meets goes back over its threshold byte
// as vk >> 3 (all a Rice parameter needs: k = vk >> (3 + damp); vk <= 1024).  Symbols of the LL prefix (a plane's first nskip)
// are stepped over.
__global__ __launch_bounds__(64) void k_ent_walk(const EntJob *__restrict__ tab)
{
    DSV2_KERNEL_PRIO();
    const EntJob &J = tab[blockIdx.y];
    const int c = blockIdx.z;
    const PlaneSpan ps = plane_span(J.info, c);
    const int nskip = J.info[EI_NSKIP + c];
    for (int lc = blockIdx.x * 64 + threadIdx.x; lc - (int) threadIdx.x < ps.nch; lc += gridDim.x * 64) {
        const bool live = lc < ps.nch;
        const int s0 = live ? min(max(nskip - lc * kEntChunk, 0), kEntChunk) : kEntChunk; // first symbol of the chunk that moves the state
        // groups of 16 symbols: the wavefront starts at g0, and from g1 on every live lane steps every symbol
        int g0 = live ? s0 >> 4 : kEntChunk / 16, g1 = live ? (s0 + 15) >> 4 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            g0 = min(g0, __shfl_xor(g0, o, 64));
            g1 = max(g1, __shfl_xor(g1, o, 64));
        }
        g0 = __builtin_amdgcn_readfirstlane(g0);
        g1 = __builtin_amdgcn_readfirstlane(g1);
        uint4 *buf = (uint4 *) (J.ksym + (size_t) (ps.cbase + (live ? lc : 0)) * kEntChunk);
        int vk = live ? (int) J.chunk_vk[ps.cbase + lc] : 0;
        if (g0 >= kEntChunk / 16) {
            continue;
        }
        uint4 cur = buf[g0];
        for (int gq = g0; gq < kEntChunk / 16; gq++) {
            const uint4 nxt = buf[min(gq + 1, kEntChunk / 16 - 1)];
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
            uint32_t o4[4];
            if (gq < g1) { // some lane's prefix ends in here
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int T = (int) (((w[k >> 2] >> (8 * (k & 3))) & 0xffu) << 3);
                    const int met = vk;
                    const int nv = vk < T ? vk + 1 : max(vk - 1, 0);
                    vk = 16 * gq + k >= s0 ? nv : vk;
                    o4[k >> 2] = (k & 3) ? (o4[k >> 2] | ((uint32_t) (met >> 3) << (8 * (k & 3)))) : (uint32_t) (met >> 3);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int T = (int) (((w[k >> 2] >> (8 * (k & 3))) & 0xffu) << 3);
                    const int met = vk;
                    vk = vk < T ? vk + 1 : max(vk - 1, 0);
                    o4[k >> 2] = (k & 3) ? (o4[k >> 2] | ((uint32_t) (met >> 3) << (8 * (k & 3)))) : (uint32_t) (met >> 3);
                }
            }
            if (live) {
                buf[gq] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
            cur = nxt;
        }
    }
}

// ---- 4b ---------------------------------------------------------------------------------------------------
// per chunk, a lane per symbol (four a thread): the Rice parameter of every symbol from the state it met, the chunk's code length
__global__ __launch_bounds__(256) void k_ent_bits(const EntJob *__restrict__ tab, EntGeom g)
{
    DSV2_KERNEL_PRIO();
    __shared__ unsigned long long wsum[4];
    const EntJob &J = tab[blockIdx.y];
    const int c = blockIdx.z;
    const PlaneSpan ps = plane_span(J.info, c);
    const uint32_t off = (uint32_t) g.qv_off[c];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int lc = blockIdx.x; lc < ps.nch; lc += gridDim.x) {
        const int first = ps.first + lc * kEntChunk, cnt = min(kEntChunk, ps.end - first);
        const int s0 = threadIdx.x * 4;
        unsigned long long bits = 0;
        bool bad = false;
        if (s0 < cnt) {
            const uint32_t st4 = *(const uint32_t *) (J.ksym + (size_t) (ps.cbase + lc) * kEntChunk + s0);
            const int i0 = first + s0;
            uint32_t prev_end = i0 > ps.first ? J.pos[i0 - 1] - off + 1u : 0u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (s0 + j < cnt) {
                    const uint32_t p = J.pos[i0 + j] - off;
                    const int32_t v = J.val[i0 + j];
                    const int seg = seg_of(g, c, p);
                    bits += (unsigned) ueg_len(p - prev_end);
                    prev_end = p + 1u;
                    if (seg == 0) {
                        const uint32_t a = (uint32_t) (v < 0 ? -v : v);
                        bits += (unsigned) ueg_len(a - 1u) + 1u;
                    } else {
                        const uint32_t u = rice_u(v);
                        int kk = (int) ((st4 >> (8 * j)) & 0xffu) >> ((seg - 1) / 3);
                        if (kk >= 32) { // a state no real picture reaches (a long run of 16-bit values): the host codes it
                            bad = true;
                            kk = 31;
                        }
                        bits += (unsigned long long) (u >> kk) + (unsigned) (kk + 1);
                    }
                }
            }
        }
        if (bad) {
            atomicOr(&J.info[EI_FLAGS], 1);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            bits += (unsigned long long) __shfl_xor((long long) bits, o, 64);
        }
        __syncthreads();
        if (lane == 0) {
            wsum[wv] = bits;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            bits = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (bits >= (1ull << 24)) { // 16 Kbit per symbol on average: not a picture (and keeps the 32-bit scans below exact)
                atomicOr(&J.info[EI_FLAGS], 4);
                bits = 0;
            }
            J.chunk_bits[ps.cbase + lc] = (uint32_t) bits;
        }
    }
}

// ---- 5 ----------------------------------------------------------------------------------------------------
// byte layout of one plane section (hzcc.c:586-613): 32-bit length | SEG(DC) | pad | 24-bit count | codes | pad | 0x55
__global__ __launch_bounds__(64) void k_ent_layout(const EntJob *__restrict__ tab)
{
    DSV2_KERNEL_PRIO();
    const EntJob &J = tab[blockIdx.x];
    const int lane = threadIdx.x;
    int *I = J.info;
    unsigned long long byte_at = 0; // start of the current plane section
    bool too_big = false;
    for (int c = 0; c < 3; c++) {
        const PlaneSpan ps = plane_span(I, c);
        int32_t ll = J.ll[c];
        uint32_t a = (uint32_t) (ll < 0 ? -ll : ll);
        int lseg = ueg_len(a) + (a ? 1 : 0);
        unsigned long long sym_byte = byte_at + 4 + (unsigned) ((lseg + 7) >> 3) + 3;
        unsigned long long carry = sym_byte * 8;
        for (int base = 0; base < ps.nch; base += 64) {
            int lc = base + lane;
            unsigned v = lc < ps.nch ? J.chunk_bits[ps.cbase + lc] : 0u;
            // (64-bit running offset, 32-bit partial sums: a chunk holds < 2^24 bits, see k_ent_ks)
            unsigned inc = (unsigned) wave_incl_scan_u(v, lane);
            unsigned long long o = carry + inc - v;
            if (lc < ps.nch) {
                if (o >= (1ull << 32)) {
                    too_big = true;
                    o = 0;
                }
                J.chunk_off[ps.cbase + lc] = (uint32_t) o;
            }
            carry += (unsigned long long) __shfl((int) inc, 63, 64);
        }
        unsigned long long end_byte = (carry + 7) >> 3; // the 0x55 goes here
        if (lane == 0) {
            I[EI_SYMBIT + c] = (int) (unsigned) (sym_byte & 0xffffffffull);
            I[EI_PBYTES + c] = (int) (unsigned) ((end_byte + 1 - byte_at) & 0xffffffffull);
        }
        byte_at = end_byte + 1;
    }
    if (byte_at + 16 > J.out_cap || too_big) {
        too_big = true;
    }
    if (lane == 0) {
        I[EI_TOTAL] = too_big ? 0 : (int) byte_at;
        if (too_big) {
            atomicOr(&I[EI_FLAGS], 2);
        }
    }
}

// ---- 6 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ent_zero(const EntJob *__restrict__ tab)
{
    DSV2_KERNEL_PRIO();
    const EntJob &J = tab[blockIdx.y];
    const unsigned words = ((unsigned) J.info[EI_TOTAL] + 16u + 3u) >> 2;
    uint32_t *o = (uint32_t *) J.out;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) {
        o[i] = 0;
    }
}

// A chunk's code words are ORed together in LDS (a chunk of 1024 symbols is ~1 KB of output) and leave as plain
// coalesced word stores; only the chunk's first and last word, which it may share with its neighbours or with the
// section's fixed fields, are ORed into the zeroed output.  A chunk too long for the image (> 127 bits a symbol on
// average) ORs every code word into global memory as before.
constexpr int kEmitWords = 4096;

__global__ __launch_bounds__(256) void k_ent_emit(const EntJob *__restrict__ tab, EntGeom g, unsigned img_words, int lanes)
{
    DSV2_KERNEL_PRIO();
    __shared__ unsigned wsum[4];
    __shared__ uint32_t img[kEmitWords + 2];
    const EntJob &J = tab[blockIdx.y];
    const int c = blockIdx.z;
    const int *I = J.info;
    if (I[EI_FLAGS] & ENT_FALLBACK_MASK) { // the host codes this picture itself
        return;
    }
    const PlaneSpan ps = plane_span(I, c);
    const uint32_t off = (uint32_t) g.qv_off[c];
    uint32_t *out32 = (uint32_t *) J.out;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) { // the section's fixed fields
        unsigned sec_bytes = (unsigned) I[EI_PBYTES + c];
        unsigned sym_byte = (unsigned) I[EI_SYMBIT + c];
        int32_t ll = J.ll[c];
        uint32_t a = (uint32_t) (ll < 0 ? -ll : ll);
        uint64_t code;
        int len;
        ueg_code(a, code, len);
        if (a) {
            code = (code << 1) | (uint64_t) (ll < 0);
            len++;
        }
        // section start = symbol start - 3 (count) - SEG bytes - 4 (length)
        unsigned sec_start = sym_byte - 3u - (unsigned) ((len + 7) >> 3) - 4u;
        put_code(out32, sec_start * 8u, (uint64_t) (sec_bytes - 4u), 32);
        put_code(out32, (sec_start + 4u) * 8u, code, len);
        put_code(out32, (sym_byte - 3u) * 8u, (uint64_t) (unsigned) (ps.end - ps.first), 24);
        put_code(out32, (sec_start + sec_bytes - 1u) * 8u, 0x55ull, 8);
    }
    for (int lc = blockIdx.x; lc < ps.nch; lc += gridDim.x) {
        const int first = ps.first + lc * kEntChunk, cnt = min(kEntChunk, ps.end - first);
        // four consecutive symbols per thread
        uint64_t rc[4], vc[4];
        int rl[4], vl[4];
        unsigned lead[4], tot = 0;
        const int s0 = threadIdx.x * 4;
        uint32_t prev_end = 0;
        if (s0 < cnt) {
            int i0 = first + s0;
            prev_end = i0 > ps.first ? J.pos[i0 - 1] - off + 1u : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            rl[j] = vl[j] = 0;
            lead[j] = 0;
            rc[j] = vc[j] = 0;
            if (s0 + j < cnt) {
                uint32_t p = J.pos[first + s0 + j] - off;
                int32_t v = J.val[first + s0 + j];
                int seg = seg_of(g, c, p);
                ueg_code(p - prev_end, rc[j], rl[j]);
                prev_end = p + 1u;
                if (seg == 0) { // NEG (bs.c:206)
                    uint32_t a = (uint32_t) (v < 0 ? -v : v);
                    ueg_code(a - 1u, vc[j], vl[j]);
                    vc[j] = (vc[j] << 1) | (uint64_t) (v < 0);
                    vl[j]++;
                } else { // adaptive Rice (bs.c:237) with the parameter found by k_ent_ks
                    uint32_t u = rice_u(v);
                    // (the parameter k_ent_ks left, or the state >> 3 k_ent_walk left in the chunk-major bytes)
                    int kk = lanes ? min((int) J.ksym[(size_t) (ps.cbase + lc) * kEntChunk + s0 + j] >> ((seg - 1) / 3), 31) : (int) J.ksym[first + s0 + j];
                    lead[j] = kk < 32 ? u >> kk : 0u;
                    vc[j] = (1ull << kk) | (kk < 32 ? (uint64_t) (u & (uint32_t) ((1ull << kk) - 1ull)) : (uint64_t) u);
                    vl[j] = kk + 1;
                }
                tot += (unsigned) rl[j] + lead[j] + (unsigned) vl[j];
            }
        }
        __syncthreads();
        unsigned inc = (unsigned) wave_incl_scan_u(tot, lane);
        if (lane == 63) {
            wsum[wv] = inc;
        }
        __syncthreads();
        unsigned o = inc - tot;
        for (int k = 0; k < wv; k++) {
            o += wsum[k];
        }
        const uint32_t b0 = J.chunk_off[ps.cbase + lc], blen = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const uint32_t w_base = b0 >> 5, nw = ((b0 + blen + 31u) >> 5) - w_base;
        uint32_t bit = b0 + o;
        if (nw <= img_words) {
            for (uint32_t i = threadIdx.x; i < nw + 2u; i += 256u) {
                img[i] = 0;
            }
            __syncthreads();
            bit -= w_base << 5;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (s0 + j < cnt) {
                    put_code_lds(img, bit, rc[j], rl[j]);
                    bit += (uint32_t) rl[j] + lead[j];
                    put_code_lds(img, bit, vc[j], vl[j]);
                    bit += (uint32_t) vl[j];
                }
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < nw; i += 256u) {
                const uint32_t w = img[i];
                if (i == 0 || i == nw - 1u) {
                    if (w) {
                        atomicOr(&out32[w_base + i], w);
                    }
                } else {
                    out32[w_base + i] = w;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (s0 + j < cnt) {
                    put_code(out32, bit, rc[j], rl[j]);
                    bit += (uint32_t) rl[j] + lead[j];
                    put_code(out32, bit, vc[j], vl[j]);
                    bit += (uint32_t) vl[j];
                }
            }
        }
    }
}

// ---- 7 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ent_out(const EntJob *__restrict__ tab)
{
    DSV2_KERNEL_PRIO();
    const EntJob &J = tab[blockIdx.y];
    const int *I = J.info;
    unsigned total = (unsigned) I[EI_TOTAL];
    int flags = I[EI_FLAGS];
    if (total > J.host_cap) {
        flags |= 8; // finished on the device, but larger than the pinned mirror: the host fetches it with a copy
    }
    if (blockIdx.x == 0 && threadIdx.x < 16) {
        J.host_info[threadIdx.x] = threadIdx.x == EI_FLAGS ? flags : I[threadIdx.x];
    }
    if (flags & 15) {
        return;
    }
    const uint4 *src = (const uint4 *) J.out;
    uint4 *dst = (uint4 *) J.host_out;
    const unsigned n16 = (total + 15u) >> 4;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
        dst[i] = src[i];
    }
}

} // namespace

EntGeom ent_geom(const size_t qv_off[4], const ScanGeom scan[3])
{
    EntGeom g;
    for (int c = 0; c < 4; c++) {
        g.qv_off[c] = (int) qv_off[c];
    }
    for (int c = 0; c < 3; c++) {
        for (int k = 0; k < 11; k++) {
            g.base[c][k] = scan[c].base[k];
        }
    }
    return g;
}

void EntBuffers::ensure(size_t nsym_cap, uint32_t out_bytes, uint32_t host_bytes)
{
    if (tables && nsym_cap <= this->nsym_cap) {
        return;
    }
    if (tables) { // grown: the lists of this stream were enlarged (a picture had more symbols than they held)
        release();
    }
    this->nsym_cap = nsym_cap;
    size_t nch = (nsym_cap + kEntChunk - 1) / kEntChunk + 3;
    HIPCHK(dev_alloc((void **) &tables, nch * kStates * sizeof(uint16_t)));
    HIPCHK(dev_alloc((void **) &chunk_vk, nch * sizeof(uint16_t)));
    HIPCHK(dev_alloc((void **) &chunk_bits, nch * sizeof(uint32_t)));
    HIPCHK(dev_alloc((void **) &chunk_off, nch * sizeof(uint32_t)));
    HIPCHK(dev_alloc((void **) &chunk_join, nch * sizeof(uint2)));
    HIPCHK(dev_alloc((void **) &ksym, nch * kEntChunk)); // (chunk-major: whole chunks)
    out_cap = (out_bytes + 15u) & ~15u;
    HIPCHK(dev_alloc((void **) &out, out_cap + 64));
    HIPCHK(dev_alloc((void **) &info, EI_WORDS * sizeof(int)));
    host_cap = (host_bytes + 15u) & ~15u;
    HIPCHK(hipHostMalloc((void **) &host_out, host_cap, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &host_info, 16 * sizeof(int), hipHostMallocDefault));
}

void EntBuffers::release()
{
    if (!tables) {
        return;
    }
    dev_release(tables);
    dev_release(chunk_vk);
    dev_release(chunk_bits);
    dev_release(chunk_off);
    dev_release(chunk_join);
    dev_release(ksym);
    dev_release(out);
    dev_release(info);
    HIPCHK(hipHostFree(host_out));
    HIPCHK(hipHostFree(host_info));
    tables = nullptr;
}

EntJob EntBuffers::job(const uint32_t *pos, const int32_t *val, const int *total, const int32_t *ll, size_t list_cap) const
{
    EntJob j;
    j.pos = pos;
    j.val = val;
    j.total = total;
    j.list_cap = (int) (list_cap < nsym_cap ? list_cap : nsym_cap);
    j.ll = ll;
    j.tables = tables;
    j.chunk_vk = chunk_vk;
    j.chunk_bits = chunk_bits;
    j.chunk_off = chunk_off;
    j.chunk_join = (uint2 *) chunk_join;
    j.ksym = ksym;
    j.out = out;
    j.out_cap = out_cap;
    j.info = info;
    j.host_out = host_out;
    j.host_info = host_info;
    j.host_cap = host_cap;
    return j;
}

void entropy_gpu_jobs(hipStream_t s, const EntJob *d_jobs, int n, const EntGeom &g, int chunk_slots)
{
    if (n <= 0) {
        return;
    }
    const int slots = chunk_slots < 1 ? 1 : chunk_slots;
    // the single-trajectory walks run a lane per chunk (k_ent_pair, k_ent_walk)
    constexpr int lanes = 1;
    const int wslots = (slots + 63) / 64 < 1 ? 1 : (slots + 63) / 64; // workgroups of 64 chunks per (stream, plane)
    DSV2_LAUNCH(k_ent_planes, dim3(n), dim3(64), 0, s, d_jobs, g);
    DSV2_LAUNCH(k_ent_tables<true>, dim3(slots, n, 3), dim3(kStates / 2), 0, s, d_jobs, g);
    DSV2_LAUNCH(k_ent_pair, dim3(wslots, n, 3), dim3(64), 0, s, d_jobs);
    DSV2_LAUNCH(k_ent_chain, dim3(3, n), dim3(64), 0, s, d_jobs, 1);
    DSV2_LAUNCH(k_ent_walk, dim3(wslots, n, 3), dim3(64), 0, s, d_jobs);
    DSV2_LAUNCH(k_ent_bits, dim3(slots, n, 3), dim3(256), 0, s, d_jobs, g);
    DSV2_LAUNCH(k_ent_layout, dim3(n), dim3(64), 0, s, d_jobs);
    DSV2_LAUNCH(k_ent_zero, dim3(16, n), dim3(256), 0, s, d_jobs);
    // (tests shrink the image to send chunks down the global-memory path: DSV2_ENT_EMIT_WORDS)
    static const unsigned img_words = getenv("DSV2_ENT_EMIT_WORDS") ? (unsigned) min(atoi(getenv("DSV2_ENT_EMIT_WORDS")), kEmitWords) : (unsigned) kEmitWords;
    DSV2_LAUNCH(k_ent_emit, dim3(slots, n, 3), dim3(256), 0, s, d_jobs, g, img_words, lanes);
    DSV2_LAUNCH(k_ent_out, dim3(8, n), dim3(256), 0, s, d_jobs);
    HIPCHK(hipGetLastError());
}

} // namespace dsv2
