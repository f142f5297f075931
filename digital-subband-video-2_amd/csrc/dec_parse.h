// dec_parse.h -- the decoder's parsing of an UNTRUSTED packet, free of any device call: packet header, metadata, the per-block
// side information of a picture (dsv_decoder.c:21-235) on caller-owned buffers.  decoder.cpp runs it in phase A of a step;
// tests/parser_fuzz.cpp compiles it (with entropy.cpp) under AddressSanitizer and feeds it damaged packets on the CPU.
#pragma once

#include <stdlib.h>

#include <vector>

#include "quant.h"

namespace dsv2 {
namespace decparse {

// what the side-information readers fill: one flag byte and (P pictures) one vector record per block
struct SideBufs {
    std::vector<DSV_MV> mvs;
    std::vector<uint8_t> blockdata;
    int nbh = 0, nbv = 0;
};

inline int sar(int v, int s) { return v < 0 ? ~(~v >> s) : v >> s; }
inline int sar_r(int v, int s) { return sar(v + (1 << (s - 1)), s); }

inline int mv_pred1(int left, int top, int topleft)
{
    int dif = left + top - topleft;
    return abs(dif - left) < abs(dif - top) ? left : top;
}

inline void movec_pred(const DSV_MV *v, int nbh, int x, int y, int *px, int *py) // dsv.c:375
{
    int vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0};
    if (x > 0) {
        vx[0] = v[y * nbh + x - 1].u.mv.x;
        vy[0] = v[y * nbh + x - 1].u.mv.y;
    }
    if (y > 0) {
        vx[1] = v[(y - 1) * nbh + x].u.mv.x;
        vy[1] = v[(y - 1) * nbh + x].u.mv.y;
    }
    if (x > 0 && y > 0) {
        vx[2] = v[(y - 1) * nbh + x - 1].u.mv.x;
        vy[2] = v[(y - 1) * nbh + x - 1].u.mv.y;
    }
    *px = mv_pred1(vx[0], vx[1], vx[2]);
    *py = mv_pred1(vy[0], vy[1], vy[2]);
}

inline int neighbordif(const DSV_MV *v, int nbh, int x, int y) // dsv.c:404-447
{
    const DSV_MV *c = &v[x + y * nbh];
    int cx = c->u.mv.x, cy = c->u.mv.y, lx = cx, ly = cy, tx = cx, ty = cy;
    if (abs(cx) < 2 && abs(cy) < 2) {
        return 0;
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
    return (abs(lx - cx) + abs(ly - cy) + abs(tx - cx) + abs(ty - cy)) / 3;
}

enum { ST_STABLE = 0, ST_MAINTAIN, ST_RINGING, ST_MODE, ST_EPRM, ST_MAX };

inline int read_packet_hdr(BitReader &br) // dsv_decoder.c:21
{
    unsigned c0 = br.get_bits(8), c1 = br.get_bits(8), c2 = br.get_bits(8), c3 = br.get_bits(8);
    if (c0 != 'D' || c1 != 'S' || c2 != 'V' || c3 != '2') {
        return -1;
    }
    br.get_bits(8); /* minor version */
    int type = (int) br.get_bits(8);
    br.get_bits(32);
    br.get_bits(32);
    return type;
}

inline void read_meta(DSV_DECODER *d, BitReader &br) // dsv_decoder.c:51
{
    DSV_META *m = &d->vidmeta;
    m->width = (int) br.get_ueg();
    m->height = (int) br.get_ueg();
    m->subsamp = (int) br.get_ueg();
    m->fps_num = (int) br.get_ueg();
    m->fps_den = (int) br.get_ueg();
    m->aspect_num = (int) br.get_ueg();
    m->aspect_den = (int) br.get_ueg();
    m->inter_sharpen = (int) br.get_ueg();
    m->reserved = br.get_bit() ? (int) br.get_bits(15) : 0;
}

// a byte-aligned, length-prefixed sub-stream: returns a reader positioned on it and skips it
inline BitReader take_sub(BitReader &br, const uint8_t *base)
{
    br.align();
    unsigned len = br.get_ueg();
    br.align();
    if (br.pos > br.limit) {
        br.seek(br.pos);
    }
    BitReader sub{base + br.byte_pos(), 0};
    sub.wide = br.wide;
    sub.limit = br.limit - br.pos; // a sub-stream may be read up to the end of the packet, as in the reference
    br.seek((uint64_t) br.pos + (uint64_t) len * 8);
    return sub;
}

inline void read_stability(SideBufs *im, BitReader &br, const uint8_t *base, int isP, const int *stats) // dsv_decoder.c:176
{
    RleReader r;
    r.br = take_sub(br, base);
    int shift = isP ? 2 : 0; /* DSV_SKIP_BIT : DSV_STABLE_BIT */
    for (size_t i = 0; i < im->blockdata.size(); i++) {
        int bit = r.get();
        if (stats[ST_STABLE]) {
            bit = !bit;
        }
        im->blockdata[i] = (uint8_t) (bit << shift);
    }
}

inline void read_intra_meta(SideBufs *im, BitReader &br, const uint8_t *base, const int *stats) // dsv_decoder.c:201
{
    RleReader rr, rm;
    rr.br = take_sub(br, base);
    rm.br = take_sub(br, base);
    for (size_t i = 0; i < im->blockdata.size(); i++) {
        int bitr = rr.get(), bitm = rm.get();
        if (stats[ST_RINGING]) {
            bitr = !bitr;
        }
        if (stats[ST_MAINTAIN]) {
            bitm = !bitm;
        }
        im->blockdata[i] |= (uint8_t) ((bitm << 1) | (bitr << 3));
    }
}

inline void read_motion(SideBufs *im, BitReader &br, const uint8_t *base, const int *stats) // dsv_decoder.c:81
{
    const SideBufs &dv = *im;
    br.align();
    RleReader mode, eprm;
    mode.br = take_sub(br, base);
    BitReader mvx = take_sub(br, base), mvy = take_sub(br, base), sbim = take_sub(br, base);
    eprm.br = take_sub(br, base);
    DSV_MV *mvs = im->mvs.data();
    for (int j = 0; j < dv.nbv; j++) {
        for (int i = 0; i < dv.nbh; i++) {
            int idx = i + j * dv.nbh;
            DSV_MV *mv = &mvs[idx];
            if (im->blockdata[idx] & DSV_IS_SKIP) {
                mv->flags |= 1u << DSV_MV_BIT_SKIP;
                mv->u.all = 0;
                im->blockdata[idx] |= DSV_IS_STABLE;
                continue;
            }
            int m = mode.get(), e = eprm.get();
            if (stats[ST_MODE]) {
                m = !m;
            }
            if (stats[ST_EPRM]) {
                e = !e;
            }
            mv->flags = (m ? (1u << DSV_MV_BIT_INTRA) : 0u) | (e ? (1u << DSV_MV_BIT_EPRM) : 0u);
            im->blockdata[idx] &= (uint8_t) ~DSV_IS_STABLE;
            im->blockdata[idx] |= (uint8_t) (e << 5);
            int px, py;
            movec_pred(mvs, dv.nbh, i, j, &px, &py);
            if (m) {
                px = sar_r(px, 2);
                py = sar_r(py, 2);
            }
            mv->u.mv.x = (int16_t) (mvx.get_seg() + px);
            mv->u.mv.y = (int16_t) (mvy.get_seg() + py);
            if (m) {
                mv->u.mv.x = (int16_t) (mv->u.mv.x * 4);
                mv->u.mv.y = (int16_t) (mv->u.mv.y * 4);
                mv->submask = sbim.get_bit() ? DSV_MASK_ALL_INTRA : (uint8_t) sbim.get_bits(4);
                mv->dc = sbim.get_bit() ? (uint16_t) (sbim.get_bits(8) | DSV_SRC_DC_PRED) : 0;
                im->blockdata[idx] |= DSV_IS_INTRA;
            }
            if (neighbordif(mvs, dv.nbh, i, j) > 8) {
                im->blockdata[idx] |= DSV_IS_STABLE;
            }
        }
    }
}


// ---- a whole picture packet, in two steps (the decoder creates / checks its device instance in between) ---------------------
constexpr int kParsePicture = -100; // parse_head: a picture packet whose body is to be parsed (anything else: a DSV_DEC_* code)

struct PictureHead {
    int has_ref = 0, is_ref = 0, blk_w = 16, blk_h = 16;
    DSV_FNUM fno = 0;
};

// packet header, metadata packets, the first fields of a picture packet; the stream's metadata is untrusted: only geometries
// the device pipeline can allocate and run are accepted (the reference would pass anything on to calloc)
inline int parse_head(BitReader &br, DSV_DECODER *d, PictureHead &hd)
{
    int type = read_packet_hdr(br);
    if (type == -1) {
        return DSV_DEC_ERROR;
    }
    if (!(type & DSV_PT_PIC)) {
        if (type == DSV_PT_META) {
            read_meta(d, br);
            d->got_metadata = 1;
            return DSV_DEC_GOT_META;
        }
        return type == DSV_PT_EOS ? DSV_DEC_EOS : DSV_DEC_ERROR;
    }
    if (!d->got_metadata) {
        return DSV_DEC_OK; /* picture before any metadata: skipped (dsv_decoder.c:436) */
    }
    const DSV_META *meta = &d->vidmeta;
    hd.has_ref = type & 1;
    hd.is_ref = (type & 0x6) == 0x6;
    br.align();
    hd.fno = br.get_bits(32);
    br.align();
    unsigned ew = br.get_ueg(), eh = br.get_ueg(); // log2 of the block size - 4: 0 or 1 (checked before it becomes a shift count)
    if (ew > 1 || eh > 1 || br.overrun) {
        return DSV_DEC_ERROR;
    }
    hd.blk_w = 16 << ew;
    hd.blk_h = 16 << eh;
    const int ss = meta->subsamp;
    const bool known = ss == DSV_SUBSAMP_444 || ss == DSV_SUBSAMP_422 || ss == DSV_SUBSAMP_420 || ss == DSV_SUBSAMP_411 ||
                       ss == DSV_SUBSAMP_410 || ss == DSV_SUBSAMP_UYVY;
    if (!known || meta->width < 16 || meta->height < 16 || meta->width > 16384 || meta->height > 16384 || (meta->width & 1) ||
        (meta->height & 1)) {
        return DSV_DEC_ERROR;
    }
    return kParsePicture;
}

// a plane section whose symbols are NOT parsed here (the device does it: dec_parse_dev.hip): what the section's head says
struct PlaneHead {
    uint32_t data_bitpos = 0, end_byte = 0; // first bit of the first run; section start + plane length
    int runs = 0;                           // symbol count of the header (24 bits)
};

struct PictureBody {
    PlaneHead head[3];
    int do_filter = 0, quant = 0, lossless = 0;
    int ok[3] = {0, 0, 0};
    int seg[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int32_t LL[3] = {0, 0, 0};
    size_t sym_first[3] = {0, 0, 0}, nsym = 0;
};

// the rest of a picture packet: statistics bits, per-block side information, the three planes' symbols (one after the other
// in pos / val, which are grown to the planes' scan lengths)
// heads_only: the planes' symbols stay unparsed -- out.head[c] says where they are; ok[c] is -1 for a section whose length field
// is implausible (exactly the test, and the reader position afterwards, of entropy_decode_plane) and 1 otherwise: whether the
// section is damaged further in is found by whoever parses it
inline void parse_body(BitReader &br, const uint8_t *pkt, int has_ref, int nbh, int nbv, const ScanGeom scan[3], SideBufs &side,
                       std::vector<uint32_t> &pos, std::vector<int32_t> &val, PictureBody &out, bool heads_only = false)
{
    const size_t nb = (size_t) nbh * nbv;
    br.align();
    int stats[ST_MAX] = {0, 0, 0, 0, 0};
    stats[ST_STABLE] = (int) br.get_bit();
    if (!has_ref) {
        stats[ST_MAINTAIN] = (int) br.get_bit();
        stats[ST_RINGING] = (int) br.get_bit();
    } else {
        stats[ST_MODE] = (int) br.get_bit();
        stats[ST_EPRM] = (int) br.get_bit();
    }
    out.do_filter = (int) br.get_bit();
    out.quant = (int) br.get_bits(DSV_MAX_QP_BITS);
    out.lossless = out.quant == 1;
    if (br.get_bit()) {
        br.get_bits(15);
    }
    br.align();
    side.nbh = nbh;
    side.nbv = nbv;
    side.blockdata.assign(nb, 0);
    read_stability(&side, br, pkt, has_ref, stats);
    if (has_ref) {
        side.mvs.assign(nb, DSV_MV{});
        read_motion(&side, br, pkt, stats);
    } else {
        read_intra_meta(&side, br, pkt, stats);
    }
    br.align();
    if (heads_only) {
        for (int c = 0; c < 3; c++) {
            out.LL[c] = 0;
            out.ok[c] = -1;
            out.seg[c][0] = out.seg[c][1] = out.seg[c][2] = out.seg[c][3] = 0;
            br.align();
            const unsigned plen = br.get_bits(32);
            br.align();
            if (!(plen > 0 && plen < (unsigned) scan[c].w * scan[c].h * sizeof(int32_t) * 2)) {
                continue; // "plane length was strange" (hzcc.c:645): the next section is looked for right behind the length field
            }
            const unsigned start = br.byte_pos();
            out.LL[c] = br.get_seg();
            br.align();
            out.head[c].runs = (int) br.get_bits(24);
            br.align();
            out.head[c].data_bitpos = br.pos;
            out.head[c].end_byte = start + plen;
            out.ok[c] = 1;
            br.seek(((uint64_t) start + plen) * 8);
        }
        out.nsym = 0;
        return;
    }
    size_t cap = (size_t) scan[0].base[10] + (size_t) scan[1].base[10] + (size_t) scan[2].base[10];
    if (pos.size() < cap) {
        pos.resize(cap);
        val.resize(cap);
    }
    size_t at = 0;
    for (int c = 0; c < 3; c++) {
        out.sym_first[c] = at;
        out.LL[c] = 0;
        out.ok[c] = entropy_decode_plane(br, &out.LL[c], pos.data() + at, val.data() + at, out.seg[c], scan[c]);
        if (out.ok[c] <= 0) { /* "decoding error in plane": the residual plane stays zero (dsv_decoder.c:516-523) */
            out.seg[c][0] = out.seg[c][1] = out.seg[c][2] = out.seg[c][3] = 0;
        }
        at += (size_t) (out.seg[c][0] + out.seg[c][1] + out.seg[c][2] + out.seg[c][3]);
    }
    out.nsym = at;
}

} // namespace decparse
} // namespace dsv2
