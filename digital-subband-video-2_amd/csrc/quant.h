// quant.h -- interfaces of the quantiser kernels (quant.hip) and the host entropy coder (entropy.cpp).
#pragma once

#include "dev.h"

namespace dsv2 {

// everything the per-coefficient quantiser rule depends on (hzcc.c:235-439)
struct QuantCfg {
    int w, h; // coefficient plane size
    int plane, isP, lossless, do_psy;
    int hshift, vshift;
    int blk_w, blk_h, nbh, nbv;
    const uint8_t *bd; // device: per-block flag bytes
    const DSV_MV *mvs; // device: motion field (P frames only)
};

// scan order of one plane: segment 0 = LL region, then level-major / subband-major (hzcc.c:264-342)
struct ScanGeom {
    int w, h;
    int off[10], sw[10], sh[10], base[11];
};
void make_scan(ScanGeom *g, int w, int h);
int spatial_psy_factor(int blk_w, int blk_h, int nbh, int nbv, int sub);

// quantise + dequantise `coefs` in place, dense quantised values to qv[scan position]
void quant_plane(hipStream_t s, DCoefs coefs, int32_t *qv, const QuantCfg &cfg, int q);
// table form: n PlaneJob records (coefs, qv, bd, mvs and the step sizes filled by quant_steps) sharing `cfg`
void quant_steps(PlaneJob *job, const QuantCfg &cfg, int q);
void quant_jobs(hipStream_t s, const PlaneJob *d_jobs, int n, const QuantCfg &cfg);

struct CompactJob {
    const int32_t *qv;
    int n;
    int *tile_count, *tile_base, *total;
    uint32_t *pos;
    int32_t *val;
    // optional pinned host mirror of the first host_cap symbols (spares the read-back copy)
    uint32_t *host_pos;
    int32_t *host_val;
    int host_cap;
    int list_cap; // symbols pos / val hold: the scatter drops what lies beyond (the count in *total still says how many there were)
};

// ordered stream compaction of nonzero entries of a dense int32 array
struct Compactor {
    size_t cap = 0;      // dense values a compaction may span (tile counters)
    size_t list_cap = 0; // symbols the lists hold (<= cap): sized by need since round 4, see CodecDev::init
    void ensure_lists(size_t n, size_t symbols); // (ensure(n) = ensure_lists(n, n))
    void grow_lists(size_t symbols);             // new, larger lists (contents dropped)
    int *tile_count = nullptr, *tile_base = nullptr, *d_total = nullptr;
    uint32_t *d_pos = nullptr;
    int32_t *d_val = nullptr;
    int *h_total = nullptr; // pinned
    void ensure(size_t n);
    void release();
    // after the stream reaches this point *h_total holds the count and d_pos/d_val the symbols
    void run(hipStream_t s, const int32_t *qv, size_t n);
    CompactJob job(const int32_t *qv, size_t n); // this compactor's buffers as a table entry
};
// njobs compactions of n values each in one set of launches; each job's count lands in *job.total
void compact_jobs(hipStream_t s, const CompactJob *d_jobs, int njobs, size_t n, bool counted = false);

// decoder: scatter + dequantise symbols sorted by scan position into a ZEROED coefficient plane;
// seg_count = {LL, l0, l1, l2}; LL = the separately transmitted DC, stored to coefs[0]
void dequant_plane(hipStream_t s, DCoefs coefs, const uint32_t *d_pos, const int32_t *d_val, const int seg_count[4], int32_t LL,
                   const QuantCfg &cfg, int q);
struct DequantJob {
    int32_t *coefs;
    const uint32_t *pos;
    const int32_t *val;
    int seg[4];
    const uint8_t *bd;
    int32_t LL;
    int qll, qp[3][3];
};
void dequant_steps(DequantJob *job, const QuantCfg &cfg, int q);
// table form: n jobs sharing `cfg`; max_seg[k] >= every job's seg[k]
void dequant_jobs(hipStream_t s, const DequantJob *d_jobs, int n, const int max_seg[4], const QuantCfg &cfg);

// ---- host entropy coder (entropy.cpp): bs.c codes + the serial part of hzcc.c ----
struct BitWriter { // MSB-first, buffer must be zero-filled (bs.c:143)
    uint8_t *start;
    unsigned pos;
    bool wide = false; // the buffer has >= 8 bytes of slack behind every write: put_bits may use 64-bit stores
    void align() { pos = (pos + 7) & ~7u; }
    unsigned byte_pos() const { return pos >> 3; }
    void put_bit(int v)
    {
        if (v) {
            start[pos >> 3] |= (uint8_t) (0x80u >> (pos & 7));
        }
        pos++;
    }
    void put_bits(unsigned n, unsigned v);
    void put_ueg(unsigned v);
    void put_seg(int v);
    void put_neg(int v);
    void put_nrice(int v, int *rk, int damp);
    void concat(const uint8_t *data, int len);
};

struct BitReader {
    const uint8_t *start;
    unsigned pos;
    bool wide = false; // >= 8 readable bytes behind every position that is read: codes are parsed from a 64-bit window
    // Untrusted input: no code is parsed from a position at or beyond `limit` (bits) -- such reads return a value that
    // ends every parsing loop and raise `overrun`.  A parse call advances by < 128 bits, so with the buffer readable
    // (zero-filled) for 32 bytes past limit/8 no read leaves it.  The default (no limit) is for trusted buffers.
    unsigned limit = 0xffffffffu;
    bool overrun = false;
    bool past_end()
    {
        if (pos >= limit) {
            overrun = true;
            return true;
        }
        return false;
    }
    // move to an absolute bit position taken from the stream (sub-stream and plane lengths): clamped to the limit
    void seek(uint64_t bitpos)
    {
        if (bitpos > limit) {
            overrun = true;
            bitpos = limit;
        }
        pos = (unsigned) bitpos;
    }
    // the next 57+ bits, left-aligned (wide mode only)
    uint64_t window() const
    {
        uint64_t w;
        __builtin_memcpy(&w, start + (pos >> 3), 8);
        return __builtin_bswap64(w) << (pos & 7);
    }
    void align() { pos = (pos + 7) & ~7u; }
    unsigned byte_pos() const { return pos >> 3; }
    unsigned get_bit()
    {
        if (past_end()) {
            return 1; // closes every unary / exp-Golomb loop
        }
        unsigned b = (start[pos >> 3] >> (7 - (pos & 7))) & 1;
        pos++;
        return b;
    }
    unsigned get_bits(unsigned n);
    unsigned get_ueg();
    int get_seg();
    int get_neg();
    int get_nrice(int *rk, int damp);
};

// zero-bit run-length coder (bs.c:284-330)
struct RleWriter {
    BitWriter bw;
    int nz = 0;
    void put(int b);
    int finish(); // returns byte length
};
struct RleReader {
    BitReader br;
    int nz = 0;
    int get();
};

// one plane of symbols -> bitstream (dsv_encode_plane minus the quantiser, hzcc.c:586-613)
void entropy_encode_plane(BitWriter &bw, int32_t LL, const uint32_t *pos, const int32_t *val, int n, const ScanGeom &g);
// bitstream -> symbols (hzcc.c:617-649, :451-583 minus dequantisation). Returns 1 on success, 0 on a
// damaged plane (LL still valid), -1 when the plane length field is implausible;
// pos/val must hold g.base[10] entries; seg_count = symbols in {LL, l0, l1, l2}.
int entropy_decode_plane(BitReader &br, int32_t *LL, uint32_t *pos, int32_t *val, int seg_count[4], const ScanGeom &g);

} // namespace dsv2
