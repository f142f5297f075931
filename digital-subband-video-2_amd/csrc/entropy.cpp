// entropy.cpp -- host-side serial entropy back end: the bit codes of reference src/bs.c
// (UEG :132, SEG :175, NEG :206, adaptive Rice :237, zero-bit RLE :284-330) and the
// symbol-stream half of src/hzcc.c (run lengths carried across subbands :242, one adaptive
// Rice state per plane :247, 24-bit symbol count :251/:445, end-of-plane byte 0x55 :604).
// It consumes the scan-ordered nonzero symbols produced by quant.hip; it is not a kernel
// because every code length depends on the adaptive state left by the previous symbol.
#include <string.h>

#include "quant.h"

namespace dsv2 {

void BitWriter::put_bits(unsigned n, unsigned v)
{
    if (wide) { // n <= 32: OR the field into the big-endian 64-bit window that starts at the current byte
        unsigned sh = pos & 7;
        uint64_t x = (uint64_t) (n < 32 ? v & ((1u << n) - 1) : v) << (64 - n - sh);
        uint64_t cur;
        memcpy(&cur, start + (pos >> 3), 8);
        cur |= __builtin_bswap64(x);
        memcpy(start + (pos >> 3), &cur, 8);
        pos += n;
        return;
    }
    while (n > 0) {
        unsigned room = 8 - (pos & 7);
        unsigned take = n < room ? n : room;
        unsigned chunk = (v >> (n - take)) & ((1u << take) - 1);
        start[pos >> 3] |= (uint8_t) (chunk << (room - take));
        pos += take;
        n -= take;
    }
}

// bit i of a 16-bit value moves to bit 2i
static inline unsigned spread16(unsigned x)
{
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// interleaved exp-Golomb (bs.c:132): for each bit of v+1 below its leading one a 0 followed by that bit, then a 1.
// The pairs are built with a bit spread and written with one or two put_bits calls instead of bit by bit.
void BitWriter::put_ueg(unsigned v)
{
    v++;
    int nb = 31 - __builtin_clz(v);
    unsigned low = v & ((1u << nb) - 1);
    if (nb <= 15) {
        put_bits((unsigned) (2 * nb + 1), (spread16(low) << 1) | 1u);
        return;
    }
    int hi = nb - 15; // the upper `hi` pairs first, then the lower 15 pairs and the terminator
    put_bits((unsigned) (2 * hi), spread16(low >> 15));
    put_bits(31, (spread16(low & 0x7fffu) << 1) | 1u);
}

void BitWriter::put_seg(int v)
{
    int s = v < 0;
    unsigned a = (unsigned) (s ? -v : v);
    put_ueg(a);
    if (a) {
        put_bit(s);
    }
}

void BitWriter::put_neg(int v)
{
    int s = v < 0;
    unsigned a = (unsigned) (s ? -v : v);
    put_ueg(a - 1);
    if (a) {
        put_bit(s);
    }
}

void BitWriter::put_nrice(int v, int *rk, int damp)
{
    unsigned u = ((unsigned) (2 * v) ^ (v < 0 ? ~0u : 0u)) - 1;
    unsigned k = (unsigned) (*rk >> damp), qq = u >> k;
    if (qq) {
        (*rk)++;
    } else if (*rk > 0) {
        (*rk)--;
    }
    pos += qq; // qq zero bits
    put_bits(k + 1, (1u << k) | (u & ((1u << k) - 1))); // the terminating 1 and the k remainder bits
}

void BitWriter::concat(const uint8_t *data, int len)
{
    if (len > 0) {
        memcpy(start + (pos >> 3), data, (size_t) len);
        pos += (unsigned) len * 8;
    }
}

// bit 2i of a 32-bit value moves to bit i (inverse of spread16)
static inline unsigned compress16(unsigned x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}

unsigned BitReader::get_bits(unsigned n)
{
    if (past_end()) {
        return 0;
    }
    if (wide && n <= 32) {
        unsigned out = n ? (unsigned) (window() >> (64 - n)) : 0u;
        pos += n;
        return out;
    }
    unsigned out = 0;
    while (n > 0) {
        if (past_end()) {
            return out << n;
        }
        unsigned room = 8 - (pos & 7);
        unsigned take = n < room ? n : room;
        unsigned chunk = (start[pos >> 3] >> (room - take)) & ((1u << take) - 1);
        out = (out << take) | chunk;
        pos += take;
        n -= take;
    }
    return out;
}

unsigned BitReader::get_ueg()
{
    if (past_end()) {
        return 0;
    }
    if (wide) {
        // pairs (0, b) ... closed by a 1 in a pair-start position: find that 1 among the even offsets
        uint64_t w = window();
        uint64_t starts = w & 0xaaaaaaaaaaaaaa00ull; // pair starts within the 56 bits that are certainly valid
        if (starts) {
            int lz = __builtin_clzll(starts); // = 2 * nb
            int nb = lz >> 1;
            if (nb <= 16) {
                unsigned low = nb ? compress16((unsigned) (w >> (64 - lz))) : 0u; // value bits sit at the odd offsets
                pos += (unsigned) lz + 1;
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

int BitReader::get_seg()
{
    int v = (int) get_ueg();
    if (v && get_bit()) {
        return -v;
    }
    return v;
}

int BitReader::get_neg()
{
    int v = (int) get_ueg() + 1;
    if (v && get_bit()) {
        return -v;
    }
    return v;
}

int BitReader::get_nrice(int *rk, int damp)
{
    int k = *rk >> damp;
    unsigned qq = 0;
    if (past_end()) {
        return 0;
    }
    if (k > 31) {
        k = 31; // (only reachable on damaged input: keeps the shifts below defined)
    }
    if (wide) {
        uint64_t w = window() & 0xffffffffffffff00ull;
        if (w) { // the unary part ends inside the window
            qq = (unsigned) __builtin_clzll(w);
            pos += qq + 1;
            goto have_q;
        }
    }
    while (!get_bit()) {
        qq++;
    }
have_q:
    if (qq) {
        (*rk)++;
    } else if (*rk > 0) {
        (*rk)--;
    }
    unsigned u = ((qq << k) | get_bits((unsigned) k)) + 1;
    return (int) ((u >> 1) ^ (0u - (u & 1)));
}

void RleWriter::put(int b)
{
    if (b) {
        bw.put_ueg((unsigned) nz);
        nz = 0;
    } else {
        nz++;
    }
}

int RleWriter::finish()
{
    bw.put_ueg((unsigned) nz);
    nz = 0;
    bw.align();
    return (int) bw.byte_pos();
}

int RleReader::get()
{
    if (nz == 0) {
        nz = (int) br.get_ueg();
        return nz == 0;
    }
    nz--;
    return nz == 0;
}

// MSB-first bit accumulator over a zero-filled buffer with >= 8 bytes of slack: same bytes as BitWriter, without a
// memory read-modify-write per code and without a data-dependent branch per code.  The pending bits sit left-aligned
// in `acc`; every put stores the whole 8-byte window at the write pointer and advances by the completed bytes.
struct AccWriter {
    uint8_t *p;
    uint64_t acc = 0;
    unsigned n = 0; // pending bits (< 8 between calls)
    explicit AccWriter(uint8_t *at) : p(at) {}
    inline void put(unsigned len, uint64_t code) // 1 <= len <= 56, code < 2^len
    {
        acc |= code << (64 - n - len);
        n += len;
        uint64_t be = __builtin_bswap64(acc);
        memcpy(p, &be, 8);
        p += n >> 3;
        acc <<= n & ~7u;
        n &= 7;
    }
    inline void zeros(unsigned q)
    {
        while (q > 56) {
            put(56, 0);
            q -= 56;
        }
        if (q) {
            put(q, 0);
        }
    }
    inline uint8_t *finish() { return p + (n ? 1 : 0); } // the partial byte is already in place, zero padded
};

// codes of the small values, which are nearly all of them: entry v = (code << 8) | length
struct UegTable {
    uint32_t e[256];
    UegTable()
    {
        for (unsigned v = 0; v < 256; v++) {
            unsigned x = v + 1;
            int nb = 31 - __builtin_clz(x);
            unsigned low = x & ((1u << nb) - 1);
            e[v] = (((spread16(low) << 1) | 1u) << 8) | (unsigned) (2 * nb + 1);
        }
    }
};
static const UegTable g_ueg;

// interleaved exp-Golomb code of v (bs.c:132) as (bits, length); length <= 63
static inline void ueg_code(unsigned v, uint64_t &code, unsigned &len)
{
    if (__builtin_expect(v < 256, 1)) {
        uint32_t e = g_ueg.e[v];
        code = e >> 8;
        len = e & 0xffu;
        return;
    }
    v++;
    int nb = 31 - __builtin_clz(v);
    unsigned low = v & ((1u << nb) - 1);
    len = (unsigned) (2 * nb + 1);
    if (__builtin_expect(nb <= 15, 1)) {
        code = ((uint64_t) spread16(low) << 1) | 1u;
    } else {
        code = ((uint64_t) spread16(low >> 15) << 31) | ((uint64_t) spread16(low & 0x7fffu) << 1) | 1u;
    }
}

void entropy_encode_plane(BitWriter &bw, int32_t LL, const uint32_t *pos, const int32_t *val, int n, const ScanGeom &g)
{
    bw.align();
    unsigned plane_start = bw.pos;
    bw.pos += 32; // byte length, patched below
    bw.put_seg(LL);
    bw.align();
    unsigned count_at = bw.pos;
    bw.pos += 24;
    bw.align();

    int vk = 0, seg = 0;
    uint32_t prev_end = 0; // scan position following the previous nonzero
    if (bw.wide) {
        // the symbol loop proper: codes are gathered in a register and leave it 32 bits at a time (the generic writer
        // read-modify-writes memory twice per symbol); starts and ends on a byte boundary like the code below
        AccWriter aw(bw.start + (bw.pos >> 3));
        uint32_t seg_end = (uint32_t) g.base[1];
        int damp = 3;
        for (int i = 0; i < n; i++) {
            uint32_t p = pos[i];
            if (__builtin_expect(p >= seg_end, 0)) {
                while (p >= (uint32_t) g.base[seg + 1]) {
                    seg++;
                }
                seg_end = (uint32_t) g.base[seg + 1];
                damp = 3 + (seg - 1) / 3;
            }
            uint64_t rc, vc;
            unsigned rl, vl;
            ueg_code(p - prev_end, rc, rl);
            prev_end = p + 1;
            int v = val[i];
            unsigned lead = 0; // zero bits in front of the value code
            if (seg == 0) { // NEG (bs.c:206)
                unsigned a = (unsigned) (v < 0 ? -v : v);
                ueg_code(a - 1, vc, vl);
                if (a) {
                    vc = (vc << 1) | (unsigned) (v < 0);
                    vl++;
                }
            } else { // adaptive Rice (bs.c:237)
                unsigned u = ((unsigned) (2 * v) ^ (v < 0 ? ~0u : 0u)) - 1;
                unsigned k = (unsigned) (vk >> damp);
                lead = k < 32 ? u >> k : 0;
                vk += lead ? 1 : (vk > 0 ? -1 : 0);
                vc = (1ull << k) | (k < 32 ? (u & ((1u << k) - 1)) : u);
                vl = k + 1;
            }
            if (__builtin_expect(rl + lead + vl <= 56, 1)) {
                aw.put(rl + lead + vl, (rc << (lead + vl)) | vc); // run code, `lead` zeros, value code: one store
            } else {
                if (rl > 56) {
                    aw.put(rl - 32, rc >> 32);
                    aw.put(32, rc & 0xffffffffu);
                } else {
                    aw.put(rl, rc);
                }
                aw.zeros(lead);
                if (vl > 56) {
                    aw.put(vl - 32, vc >> 32);
                    aw.put(32, vc & 0xffffffffu);
                } else {
                    aw.put(vl, vc);
                }
            }
        }
        bw.pos = (unsigned) (aw.finish() - bw.start) * 8;
    } else {
        for (int i = 0; i < n; i++) {
            uint32_t p = pos[i];
            while (p >= (uint32_t) g.base[seg + 1]) {
                seg++;
            }
            bw.put_ueg(p - prev_end);
            if (seg == 0) {
                bw.put_neg(val[i]);
            } else {
                bw.put_nrice(val[i], &vk, 3 + (seg - 1) / 3);
            }
            prev_end = p + 1;
        }
    }
    bw.align();
    unsigned after = bw.pos;
    bw.pos = count_at;
    bw.put_bits(24, (unsigned) n);
    bw.pos = after;
    bw.put_bits(8, 0x55);
    bw.align();
    unsigned plane_end = bw.pos;
    bw.pos = plane_start;
    bw.put_bits(32, (plane_end - plane_start) / 8 - 4);
    bw.pos = plane_end;
}

int entropy_decode_plane(BitReader &br, int32_t *LL, uint32_t *pos, int32_t *val, int seg_count[4], const ScanGeom &g)
{
    seg_count[0] = seg_count[1] = seg_count[2] = seg_count[3] = 0;
    br.align();
    unsigned plen = br.get_bits(32);
    br.align();
    if (!(plen > 0 && plen < (unsigned) g.w * g.h * sizeof(int32_t) * 2)) {
        return -1; // "plane length was strange" (hzcc.c:645): nothing is stored
    }
    unsigned start = br.byte_pos();
    unsigned limit = start + plen;
    *LL = br.get_seg();
    br.align();
    int runs = (int) br.get_bits(24);
    br.align();

    int vk = 0, n = 0;
    int ok = 1;
    // position of the next nonzero = current position + run
    uint32_t cur = 0;
    uint32_t total = (uint32_t) g.base[10];
    bool truncated = false;
    uint32_t run = runs > 0 ? br.get_ueg() : 0; // every run is parsed once: the look-ahead below becomes the next run
    while (runs-- > 0) {
        uint64_t p = (uint64_t) cur + run;
        if (p >= total) {
            break; // the run walks off the plane: nothing further is placed (hzcc.c:481-581 with run never reaching 0)
        }
        int seg = 0;
        while (p >= (uint32_t) g.base[seg + 1]) {
            seg++;
        }
        int v = seg == 0 ? br.get_neg() : br.get_nrice(&vk, 3 + (seg - 1) / 3);
        // the reference reads the next run before its overrun check (hzcc.c:525-529), so a symbol whose
        // successor's run crosses the plane-length limit is dropped together with everything after it
        if (runs > 0) {
            run = br.get_ueg();
            if (br.byte_pos() >= limit) {
                truncated = true;
            }
        } else if (br.byte_pos() >= limit) {
            truncated = true;
        }
        if (truncated) {
            break;
        }
        pos[n] = (uint32_t) p;
        val[n] = v;
        n++;
        seg_count[seg == 0 ? 0 : 1 + (seg - 1) / 3]++;
        cur = (uint32_t) p + 1;
    }
    if (!truncated) {
        br.align();
        if (br.get_bits(8) != 0x55) {
            ok = 0; // "bad eop" (hzcc.c:636)
        }
    } else {
        ok = 0;
    }
    br.seek(((uint64_t) start + plen) * 8);
    return ok;
}

} // namespace dsv2
