// dec_parse_dev.hip -- the decoder's entropy PARSE of a plane section on the device (reference hzcc.c:451-585, bs.c:151-290;
// host form: entropy.cpp entropy_decode_plane, which this restates code for code).
//
// The bit position of symbol k depends on every symbol before it, so a plane section is one dependency chain: one WAVEFRONT per
// section, every lane computing the same thing (no divergence, no cross-lane traffic), lane 0 storing the (position, value) pairs.
// The section's bytes are read through a 2 KB window in LDS that the wavefront refills with two 16-byte loads per lane when the
// parse leaves it; a code is parsed from a 64-bit big-endian view of that window, exactly as the host's BitReader does from its
// private copy of the packet.  What makes this worth having is not speed per symbol (~ten times a host core's time) but where the time
// goes: a P picture's sections are parsed beside the other lockstep groups' kernels instead of on one of sixteen host cores
// (DESIGN 5.9).  Intra pictures (1 in a GOP, ten times the symbols) stay on the host.
//
// UNTRUSTED INPUT.  Every read is bounded exactly as on the host: no code is parsed from a bit position at or beyond `limit_bits`
// (the packet's length + 8 bytes; the staged packet is followed by zero bytes and by at least one window of readable memory), a
// parse call advances by less than 128 bits, the symbol loop runs at most `runs` times (24 bits in the header, and clamped by the
// host to the plane's coefficient count: a run advances the scan position by at least one), and a pair is only stored at
// n < cap.  A damaged section ends like the host's: ok = 0, what was parsed so far is dropped (seg counts zero) and the plane's
// residual is zeroed by a conditional fill (k_zero_linear_if).
#include "dec_parse_dev.h"

#include "prio.h"

namespace dsv2 {
namespace {

constexpr int kWinBytes = 2048; // the LDS window: two 16-byte loads per lane

struct DevReader {
    const uint8_t *pkt;    // staged packet (device memory)
    uint32_t *win;         // LDS: kWinBytes / 4 dwords
    unsigned base;         // byte offset of win[0] in the packet (multiple of 16)
    unsigned pos, limit;   // bits
    bool overrun;
    // The bits from `pos` on, kept in registers: `hi` holds the next 64, `lo` the `nlo` bits behind them (left-aligned; its end sits
    // on a dword boundary of the packet, so it is topped up with whole aligned dwords out of the LDS window).  A code is parsed from
    // `hi` alone -- one LDS read per 32 bits consumed instead of three per code.
    uint64_t hi, lo;
    int nlo;
    unsigned next_dw;      // byte offset (in the packet) of the next dword to append

    __device__ __forceinline__ void refill(unsigned byte)
    {
        base = byte & ~15u;
        const int lane = (int) threadIdx.x;
        const uint4 *src = (const uint4 *) (pkt + base);
        __builtin_amdgcn_wave_barrier();
        const uint4 a = src[lane], b = src[64 + lane];
        ((uint4 *) win)[lane] = a;
        ((uint4 *) win)[64 + lane] = b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ uint32_t dword_at(unsigned byte) // big-endian value of the aligned dword at packet offset `byte`
    {
        if (byte < base || byte + 4 > base + kWinBytes) {
            refill(byte);
        }
        // (every lane reads the same word; taking lane 0's copy tells the compiler it is wave-uniform, and with it everything the
        // parse derives from it: the whole symbol loop runs on the scalar unit)
        return __builtin_bswap32((uint32_t) __builtin_amdgcn_readfirstlane((int) win[(byte - base) >> 2]));
    }
    __device__ __forceinline__ void top_up()
    {
        if (nlo <= 32) {
            lo |= (uint64_t) dword_at(next_dw) << (32 - nlo);
            nlo += 32;
            next_dw += 4;
        }
    }
    // (re)start the register view at bit position `p`
    __device__ __forceinline__ void start_at(unsigned p)
    {
        pos = p;
        const unsigned b0 = (p >> 3) & ~3u, r = p - (b0 << 3); // r: 0 .. 31
        const uint64_t w01 = ((uint64_t) dword_at(b0) << 32) | dword_at(b0 + 4);
        const uint64_t w23 = ((uint64_t) dword_at(b0 + 8) << 32) | dword_at(b0 + 12);
        hi = r ? (w01 << r) | (w23 >> (64 - r)) : w01;
        lo = r ? w23 << r : w23;
        nlo = 64 - (int) r;
        next_dw = b0 + 16;
    }
    __device__ __forceinline__ void consume(unsigned n) // n <= 32
    {
        if (n) {
            hi = (hi << n) | (lo >> (64 - n));
            lo <<= n;
            nlo -= (int) n;
            pos += n;
            top_up();
        }
    }
    __device__ __forceinline__ void skip(unsigned n) // any n < 128
    {
        while (n > 32) {
            consume(32);
            n -= 32;
        }
        consume(n);
    }
    // the next 64 bits, left-aligned (BitReader::window gives 57+)
    __device__ __forceinline__ uint64_t window() const { return hi; }
    __device__ __forceinline__ bool past_end()
    {
        if (pos >= limit) {
            overrun = true;
            return true;
        }
        return false;
    }
    __device__ __forceinline__ void align() { consume((8u - (pos & 7u)) & 7u); }
    __device__ __forceinline__ unsigned byte_pos() const { return pos >> 3; }
    __device__ __forceinline__ unsigned get_bit()
    {
        if (past_end()) {
            return 1;
        }
        const unsigned b = (unsigned) (window() >> 63);
        consume(1);
        return b;
    }
    __device__ __forceinline__ unsigned get_bits(unsigned n) // n <= 32 (the wide path of BitReader::get_bits)
    {
        if (past_end()) {
            return 0;
        }
        const unsigned out = n ? (unsigned) (window() >> (64 - n)) : 0u;
        consume(n);
        return out;
    }
    static __device__ __forceinline__ unsigned compress16(unsigned x) // bit 2i -> bit i
    {
        x &= 0x55555555u;
        x = (x | (x >> 1)) & 0x33333333u;
        x = (x | (x >> 2)) & 0x0f0f0f0fu;
        x = (x | (x >> 4)) & 0x00ff00ffu;
        x = (x | (x >> 8)) & 0x0000ffffu;
        return x;
    }
    __device__ __forceinline__ unsigned get_ueg()
    {
        if (past_end()) {
            return 0;
        }
        {
            const uint64_t w = window();
            const uint64_t starts = w & 0xaaaaaaaaaaaaaa00ull;
            if (starts) {
                const int lz = __builtin_clzll(starts);
                const int nb = lz >> 1;
                if (nb <= 16) {
                    const unsigned low = nb ? compress16((unsigned) (w >> (64 - lz))) : 0u;
                    skip((unsigned) lz + 1);
                    return ((1u << nb) | low) - 1;
                }
            }
        }
        unsigned v = 1;
        while (!get_bit()) {
            v = (v << 1) | get_bit();
        }
        return v - 1;
    }
    __device__ __forceinline__ int get_neg()
    {
        const int v = (int) get_ueg() + 1;
        if (v && get_bit()) {
            return -v;
        }
        return v;
    }
    __device__ __forceinline__ int get_nrice(int &rk, int damp)
    {
        int k = rk >> damp;
        unsigned qq = 0;
        if (past_end()) {
            return 0;
        }
        if (k > 31) {
            k = 31;
        }
        const uint64_t w = window() & 0xffffffffffffff00ull;
        if (w) {
            qq = (unsigned) __builtin_clzll(w);
            skip(qq + 1);
        } else {
            while (!get_bit()) {
                qq++;
            }
        }
        if (qq) {
            rk++;
        } else if (rk > 0) {
            rk--;
        }
        const unsigned u = ((qq << k) | get_bits((unsigned) k)) + 1;
        return (int) ((u >> 1) ^ (0u - (u & 1)));
    }
};

// ---- lane-parallel rounds ---------------------------------------------------------------------------------------------------------
// One step of the symbol loop is [value code][next run code].  Where a step starts depends on every step before it -- but what a
// step WOULD decode if it started at bit P + i can be worked out for i = 0 .. 63 at once, a lane each, given the coding in force
// (LL region: signed exp-Golomb; a detail level: Rice with parameter k).  A round does that, then walks the true chain through
// the lanes' answers with scalar code -- lane 0 is a true start, its length names the next true start, ... about five steps per
// round -- about 40 instructions a symbol instead of 250.  The walk checks per step what the guess assumed: the subband's coding
// and k (k = vk >> damp moves at most every 8th symbol), and leaves the round when they change.  Everything that needs the
// reference's exact end-of-data behaviour -- the last symbol, positions within 256 bits of the section's end or the packet's
// limit, codes longer than a lane's window, anything a lane could not decode by the fast rules -- is handed to the exact
// serial step (DevReader), restarted at that bit.  The fast rules decode a well-formed code to the same value the serial reader
// does; only complete codes inside the lane's 64 bits count as decoded.
struct LaneStep {
    int v;          // value
    unsigned run;   // the run behind it
    unsigned pack;  // bits of both codes (0 .. 127) | quotient != 0 (moves vk) << 7 | decoded << 8
};
constexpr unsigned kStepValid = 1u << 8, kStepQnz = 1u << 7;

__device__ __forceinline__ uint64_t lane_window(const uint32_t *win, unsigned base, unsigned bitpos)
{
    const unsigned byte = bitpos >> 3;
    const unsigned d = (byte - base) >> 2;
    const unsigned r = bitpos - ((base + 4 * d) << 3);
    const uint32_t w0 = __builtin_bswap32(win[d]), w1 = __builtin_bswap32(win[d + 1]), w2 = __builtin_bswap32(win[d + 2]);
    const uint64_t hi = ((uint64_t) w0 << 32) | w1;
    return r ? (hi << r) | ((uint64_t) w2 >> (32 - r)) : hi;
}

// exp-Golomb at the top of w with `avail` valid bits: value and length, or invalid (longer than 16 pairs / not complete inside avail)
__device__ __forceinline__ bool lane_ueg(uint64_t w, unsigned avail, unsigned &val, unsigned &len)
{
    const uint64_t starts = w & 0xaaaaaaaaaaaaaaaaull;
    const int lz = starts ? __builtin_clzll(starts) : 64;
    const int nb = lz >> 1;
    const unsigned low = nb ? DevReader::compress16((unsigned) (w >> ((64 - lz) & 63))) : 0u;
    val = ((1u << (nb & 31)) | low) - 1;
    len = (unsigned) lz + 1;
    return starts != 0 && nb <= 16 && len <= avail;
}

// the run code behind a value code of lenv bits
__device__ __forceinline__ LaneStep lane_finish(uint64_t w, int v, unsigned lenv, bool ok1, bool qnz)
{
    LaneStep o;
    unsigned l2;
    const bool ok2 = lane_ueg(w << (lenv & 63), 64 - (lenv & 63), o.run, l2);
    o.v = v;
    o.pack = ((lenv + l2) & 127u) | (qnz ? kStepQnz : 0u) | ((ok1 && ok2 && lenv + l2 <= 64) ? kStepValid : 0u);
    return o;
}

// adaptive Rice with parameter k (bs.c:237): unary quotient, a 1, k remainder bits
__device__ __forceinline__ LaneStep lane_rice(uint64_t w, int k)
{
    const uint64_t wm = w & 0xffffffffffffff00ull;
    const unsigned qq = wm ? (unsigned) __builtin_clzll(wm) : 64u;
    const unsigned lenv = qq + 1 + (unsigned) k;
    const unsigned rem = k ? (unsigned) ((w << ((qq + 1) & 63)) >> ((64 - k) & 63)) : 0u;
    const unsigned u = ((qq << k) | rem) + 1;
    return lane_finish(w, (int) ((u >> 1) ^ (0u - (u & 1))), lenv, wm != 0 && lenv <= 40 && k >= 0, qq != 0);
}

// signed exp-Golomb of |v| - 1, then the sign (bs.c:206; |v| >= 1: the sign bit is always there)
__device__ __forceinline__ LaneStep lane_neg(uint64_t w)
{
    unsigned uv, l;
    const bool ok1 = lane_ueg(w, 56, uv, l);
    const int mag = (int) uv + 1;
    const unsigned sign = (unsigned) (w >> ((63 - l) & 63)) & 1u;
    return lane_finish(w, sign ? -mag : mag, l + 1, ok1 && l + 1 <= 40, false);
}

// DSV2_DEC_PARSE_STATS=1: [0] rounds, [1] symbols out of rounds, [2] symbols of the serial step; =2: [3] luma sections, [4..7] shader clocks (lanes, walk, commit, all)
__device__ unsigned long long g_parse_stats[8];

__global__ __launch_bounds__(64) void k_dec_parse(const DecParseJob *__restrict__ tab, DecScanBases luma, DecScanBases chroma, int lane_rounds, int stats)
{
    unsigned st_rounds = 0, st_lane = 0, st_exact = 0;
    unsigned long long cy_lane = 0, cy_walk = 0, cy_commit = 0, cy_t0 = 0, cy_all0 = __builtin_readcyclecounter();
    DSV2_CENSUS_SCOPE();
    // one wavefront, one dependency chain, and a whole step waits for the longest of them: it issues ahead of the streaming kernels'
    // wavefronts it shares a SIMD with (DSV2_DEC_PARSE_PRIO=0: default priority)
    if (stats >= 0) {
        __builtin_amdgcn_s_setprio(3);
    }
    __shared__ uint32_t win[kWinBytes / 4 + 4];
    const DecParseJob J = tab[blockIdx.x];
    const DecScanBases &g = J.chroma ? chroma : luma;
    DevReader br;
    br.pkt = J.pkt;
    br.win = win;
    br.base = 0xffffffffu - kWinBytes; // (no window yet: the first read refills)
    br.limit = J.limit_bits;
    br.overrun = false;
    br.start_at(J.data_bitpos);
    int runs = J.runs;
    int vk = 0, n = 0, ok = 1;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0; // symbols in {LL region, level 0, 1, 2}
    uint32_t cur = 0;
    // Scan positions only grow, so the subband a symbol falls in is tracked, not searched: `seg` and the first position behind it
    // (the host walks base[] from 0 for every symbol; here that walk was ten dependent scalar loads a symbol -- most of a
    // microsecond).  The bases sit in LDS for the rare step to the next subband.
    __shared__ uint32_t sbase[12];
    if (threadIdx.x < 11) {
        sbase[threadIdx.x] = (uint32_t) g.base[threadIdx.x];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t total = (uint32_t) __builtin_amdgcn_readfirstlane((int) sbase[10]);
    int seg = 0;
    uint32_t seg_end = (uint32_t) __builtin_amdgcn_readfirstlane((int) sbase[1]);
    bool truncated = false;
    uint32_t run = runs > 0 ? br.get_ueg() : 0;
    // bits below which a round may run: every code a lane looks at (64 lanes + 64 bits) ends well before the section's end and the limit
    const unsigned end_bits = J.end_byte < (1u << 28) ? J.end_byte * 8 : 0xfffffff8u;
    const unsigned safe_end = (end_bits < J.limit_bits ? end_bits : J.limit_bits);
    for (;;) {
        // ---- lane-parallel round(s) from br.pos, while the exact step is not needed ----
        if (lane_rounds && runs >= 2 && br.pos + 256 < safe_end) {
            unsigned P = br.pos;
            bool left = false; // the serial step has to take over at P
            while (!left && runs >= 2 && P + 256 < safe_end) {
                // (the walk's state is wave-uniform -- it only ever depends on lane-selected values -- but the compiler does not see
                // that through the loop nest and would keep it in vector registers under exec masks, ninety instructions a step:
                // it is told, once per round)
                vk = __builtin_amdgcn_readfirstlane(vk);
                cur = (uint32_t) __builtin_amdgcn_readfirstlane((int) cur);
                run = (uint32_t) __builtin_amdgcn_readfirstlane((int) run);
                runs = __builtin_amdgcn_readfirstlane(runs);
                n = __builtin_amdgcn_readfirstlane(n);
                seg = __builtin_amdgcn_readfirstlane(seg);
                seg_end = (uint32_t) __builtin_amdgcn_readfirstlane((int) seg_end);
                P = (unsigned) __builtin_amdgcn_readfirstlane((int) P);
                if (run >= total - cur) {
                    left = true; // (the run walks off the plane: the serial loop ends it)
                    break;
                }
                st_rounds++;
                uint32_t p = cur + run;
                while (p >= seg_end) {
                    seg++;
                    seg_end = (uint32_t) __builtin_amdgcn_readfirstlane((int) sbase[seg + 1]);
                }
                const bool rice = seg != 0;
                const int damp = 3 + (seg - 1) / 3;
                // the Rice parameter in force, and the one vk is closer to: vk moves by one a symbol and k = vk >> damp flips back
                // and forth while vk sits at a multiple of 2^damp (five rounds in six ended there with one guess) -- the lanes
                // answer for both.  vk in [loA, hiA] means kA, in [loB, hiB] kB.
                int kA = rice ? vk >> damp : 0;
                kA = kA > 31 ? 31 : kA;
                const int loA = kA << damp, hiA = kA >= 31 ? 0x7fffffff : loA + (1 << damp) - 1;
                const bool up = rice && (hiA - vk) < (vk - loA) && kA < 31;
                const int kB = rice ? (up ? kA + 1 : kA - 1) : -1; // (-1: no second guess)
                const int loB = kB >= 0 ? kB << damp : 1, hiB = kB >= 0 ? (kB >= 31 ? 0x7fffffff : loB + (1 << damp) - 1) : 0;
                if (stats >= 2) {
                    cy_t0 = __builtin_readcyclecounter();
                }
                if ((P >> 3) < br.base || (P >> 3) + 40 > br.base + kWinBytes) {
                    br.refill(P >> 3);
                }
                const uint64_t w = lane_window(win, br.base, P + threadIdx.x);
                LaneStep sA, sB;
                if (rice) {
                    sA = lane_rice(w, kA);
                    sB = lane_rice(w, kB);
                } else {
                    sA = lane_neg(w);
                    sB = sA;
                }
                // ---- the true chain through the lanes' answers, in two phases ----
                // Phase 1, scalar, as little as a step allows: ONE lane read (the two guesses' lengths / flags, packed), the guess that
                // vk picks, a bit set in two 64-bit masks -- which lanes START a step, which of them take the second guess -- vk moved,
                // the next start lane.  (Measured before this form: with the step's bookkeeping in vector registers the walk was 55 %
                // of a section's clocks, 400 a step.)  It stops at an undecodable lane, when neither guess holds, after 64 bits, or at the
                // step count the section / the list allows.
                const unsigned pk2 = sA.pack | (sB.pack << 16);
                if (stats >= 2) {
                    const unsigned long long t = __builtin_readcyclecounter();
                    cy_lane += t - cy_t0 + (unsigned long long) (__builtin_amdgcn_readfirstlane((int) pk2) & 0); // (after the lanes' answers exist)
                    cy_t0 = t;
                }
                const int max_steps = min(min(runs - 1, J.cap - n), 64);
                const int vk0 = vk;
                unsigned j = 0;
                int T = 0;
                unsigned long long starts = 0, selB = 0;
                bool bad_lane = false;
                while (T < max_steps && j < 64) {
                    unsigned sel = 0;
                    if (rice && (vk < loA || vk > hiA)) {
                        if (vk < loB || vk > hiB) {
                            break; // neither guess: a new round
                        }
                        sel = 1;
                    }
                    const unsigned pk = ((unsigned) __builtin_amdgcn_readlane((int) pk2, (int) j) >> (16 * sel)) & 0x1ffu;
                    if (!(pk & kStepValid)) {
                        bad_lane = true;
                        break;
                    }
                    starts |= 1ull << j;
                    selB |= (unsigned long long) sel << j;
                    if (rice) {
                        vk += (pk & kStepQnz) ? 1 : (vk > 0 ? -1 : 0);
                    }
                    j += pk & 127u;
                    T++;
                }
                if (stats >= 2) {
                    const unsigned long long t = __builtin_readcyclecounter();
                    cy_walk += t - cy_t0;
                    cy_t0 = t;
                }
                // Phase 2, in the start lanes themselves: a start lane holds its step's value and run; its scan position is p plus the
                // sum of (run + 1) over the start lanes before it (a DPP scan: no LDS), its place in the list its rank among them; the
                // first step that leaves the subband (or the plane) ends the round before it.
                int Tc = T; // steps committed
                if (T > 0) {
                    const int lane = (int) threadIdx.x;
                    const bool isS = (starts >> lane) & 1ull, useB = (selB >> lane) & 1ull;
                    const int v = useB ? sB.v : sA.v;
                    const unsigned nr = useB ? sB.run : sA.run;
                    const unsigned inc = isS ? min(nr, total) + 1u : 0u; // (a run that reaches the plane's end is as good as any larger one: the sums stay inside 32 bits)
                    int x = (int) inc;
                    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); // row_shr:1 .. 8: inclusive scan inside each row of 16
                    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
                    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
                    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
                    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
                    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
                    const unsigned pt = p + ((unsigned) x - inc);
                    const int rank = (int) __builtin_amdgcn_mbcnt_hi((unsigned) (starts >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) starts, 0u));
                    unsigned long long committed = starts;
                    const unsigned long long over = __ballot(isS && pt >= seg_end);
                    if (over) { // (never lane 0: step 0's position was checked above)
                        const int jo = __ffsll((long long) over) - 1;
                        committed = starts & ((1ull << jo) - 1ull);
                        Tc = __popcll(committed);
                        // the round ended early: the bit position of the step that left, and vk as it was before it (re-walked: once per subband)
                        j = (unsigned) jo;
                        vk = vk0;
                        if (rice) {
                            for (unsigned long long m = committed; m; m &= m - 1) {
                                const int jl = __ffsll((long long) m) - 1;
                                const unsigned pkl = ((unsigned) __builtin_amdgcn_readlane((int) pk2, jl) >> (16 * (int) ((selB >> jl) & 1ull))) & 0x1ffu;
                                vk += (pkl & kStepQnz) ? 1 : (vk > 0 ? -1 : 0);
                            }
                        }
                        bad_lane = false;
                    }
                    if (isS && rank < Tc) {
                        J.pos[n + rank] = pt;
                        J.val[n + rank] = v;
                    }
                    const int lastl = 63 - __builtin_clzll(committed);
                    cur = (uint32_t) __builtin_amdgcn_readlane((int) pt, lastl) + 1;
                    run = (uint32_t) __builtin_amdgcn_readlane((int) nr, lastl);
                    n += Tc;
                    runs -= Tc;
                }
                left = bad_lane || runs < 2 || n >= J.cap;
                c0 += seg == 0 ? Tc : 0;
                c1 += seg >= 1 && seg <= 3 ? Tc : 0;
                c2 += seg >= 4 && seg <= 6 ? Tc : 0;
                c3 += seg >= 7 ? Tc : 0;
                st_lane += (unsigned) Tc;
                P += j;
                if (stats >= 2) {
                    cy_commit += __builtin_readcyclecounter() - cy_t0 + (unsigned long long) (__builtin_amdgcn_readfirstlane((int) cur) & 0);
                }
            }
            br.start_at(P);
        }
        // ---- one exact step (entropy.cpp entropy_decode_plane, code for code) ----
        if (runs-- <= 0) {
            break;
        }
        const uint64_t p = (uint64_t) cur + run;
        if (p >= total) {
            break;
        }
        while (p >= seg_end) { // (p < total = base[10]: ends at seg <= 9)
            seg++;
            seg_end = (uint32_t) __builtin_amdgcn_readfirstlane((int) sbase[seg + 1]);
        }
        st_exact++;
        const int v = seg == 0 ? br.get_neg() : br.get_nrice(vk, 3 + (seg - 1) / 3);
        if (runs > 0) {
            run = br.get_ueg();
            if (br.byte_pos() >= J.end_byte) {
                truncated = true;
            }
        } else if (br.byte_pos() >= J.end_byte) {
            truncated = true;
        }
        if (truncated || n >= J.cap) { // (n < cap always holds for cap = min(runs, total): belt and braces on untrusted input)
            truncated = true;
            break;
        }
        if (threadIdx.x == 0) {
            J.pos[n] = (uint32_t) p;
            J.val[n] = v;
        }
        n++;
        c0 += seg == 0;
        c1 += seg >= 1 && seg <= 3;
        c2 += seg >= 4 && seg <= 6;
        c3 += seg >= 7;
        cur = (uint32_t) p + 1;
    }
    if (!truncated) {
        br.align();
        if (br.get_bits(8) != 0x55) {
            ok = 0;
        }
    } else {
        ok = 0;
    }
    if (threadIdx.x == 0) {
        // a damaged plane: "decoding error in plane", its residual stays zero (dsv_decoder.c:516-523) -- no symbol is placed and the
        // conditional fill behind the inverse transform zeroes the plane
        J.seg_out[0] = ok ? c0 : 0;
        J.seg_out[1] = ok ? c1 : 0;
        J.seg_out[2] = ok ? c2 : 0;
        J.seg_out[3] = ok ? c3 : 0;
        *J.fail = ok ? 0 : 1;
        if (stats > 0) {
            atomicAdd(&g_parse_stats[0], (unsigned long long) st_rounds);
            atomicAdd(&g_parse_stats[1], (unsigned long long) st_lane);
            atomicAdd(&g_parse_stats[2], (unsigned long long) st_exact);
            if (stats >= 2 && J.chroma == 0) { // shader clocks of a luma section: the lanes' answers, the scalar walk, the commit, everything
                atomicAdd(&g_parse_stats[4], cy_lane);
                atomicAdd(&g_parse_stats[5], cy_walk);
                atomicAdd(&g_parse_stats[6], cy_commit);
                atomicAdd(&g_parse_stats[7], __builtin_readcyclecounter() - cy_all0);
                atomicAdd(&g_parse_stats[3], 1ull);
            }
        }
    }
}

// zero fill of `bytes` (multiple of 16) at dst for the jobs whose flag is set
__global__ __launch_bounds__(256) void k_zero_linear_if(const CopyJob *__restrict__ tab, const int *__restrict__ flags)
{
    if (flags[blockIdx.y] == 0) {
        return;
    }
    const CopyJob &j = tab[blockIdx.y];
    uint4 *dp = (uint4 *) j.dst;
    const size_t n = j.bytes >> 4;
    const uint4 z = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        dp[i] = z;
    }
}

} // namespace

void dec_parse_planes(hipStream_t s, const DecParseJob *d_jobs, int n, const DecScanBases &luma, const DecScanBases &chroma)
{
    if (n > 0) {
        // DSV2_DEC_LANE_ROUNDS=0: the serial step only (A/B, and the cross-check of the lane-parallel rounds)
        static const int lane_rounds = getenv("DSV2_DEC_LANE_ROUNDS") ? atoi(getenv("DSV2_DEC_LANE_ROUNDS")) : 1;
        static const int stats = getenv("DSV2_DEC_PARSE_PRIO") && atoi(getenv("DSV2_DEC_PARSE_PRIO")) == 0 ? -1 : (getenv("DSV2_DEC_PARSE_STATS") ? atoi(getenv("DSV2_DEC_PARSE_STATS")) : 0);
        DSV2_LAUNCH(k_dec_parse, dim3((unsigned) n), dim3(64), 0, s, d_jobs, luma, chroma, lane_rounds, stats);
        if (stats > 0) {
            static struct AtExit {
                ~AtExit()
                {
                    unsigned long long h[8];
                    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_parse_stats), sizeof(h)) == hipSuccess) {
                        fprintf(stderr, "[dec parse] lane-parallel rounds %llu, symbols out of them %llu (%.1f a round), serial steps %llu\n", h[0], h[1],
                                h[0] ? (double) h[1] / (double) h[0] : 0.0, h[2]);
                        if (h[3]) { // DSV2_DEC_PARSE_STATS=2
                            fprintf(stderr, "[dec parse] shader clocks of a luma section (mean of %llu): all %.0f k; in rounds: the lanes' answers %.0f k, the walk %.0f k, "
                                            "the commit %.0f k\n", h[3], h[7] / 1e3 / h[3], h[4] / 1e3 / h[3], h[5] / 1e3 / h[3], h[6] / 1e3 / h[3]);
                        }
                    }
                }
            } at_exit;
        }
    }
}

void zero_linear_if_batch(hipStream_t s, const CopyJob *d_jobs, const int *d_flags, int n, size_t max_bytes)
{
    if (n <= 0) {
        return;
    }
    const size_t vecs = max_bytes >> 4;
    const int gx = (int) ((vecs + 256 * 8 - 1) / (256 * 8));
    DSV2_LAUNCH(k_zero_linear_if, dim3(gx < 1 ? 1 : gx, n), dim3(256), 0, s, d_jobs, d_flags);
}

} // namespace dsv2
