// decoder.cpp -- host frame controller of the decoder (C ABI section 3 of include/dsv2_hip.h).
//
// Restates the serial parsing of reference src/dsv_decoder.c (packet header :22, metadata :52,
// stability blocks :177, intra metadata :202, motion data :82, picture packet :394) and drives
// the device pipeline: symbol scatter + dequantisation -> inverse SBT -> intra filter, or
// motion-compensated prediction + reconstruction + in-loop filters -> border extension of
// reference pictures.  Entropy *parsing* is serial adaptive-state work and stays on the host.
#include <string.h>

#include <vector>

#include "codec.h"

using namespace dsv2;

namespace {

struct DecImpl {
    CodecDev dev;
    bool ready = false;
    int cur = 0;
    bool have_ref = false;
    std::vector<DSV_MV> mvs;
    std::vector<uint8_t> blockdata;
    std::vector<uint32_t> pos;
    std::vector<int32_t> val;
};

inline int sar(int v, int s) { return v < 0 ? ~(~v >> s) : v >> s; }
inline int sar_r(int v, int s) { return sar(v + (1 << (s - 1)), s); }

int mv_pred1(int left, int top, int topleft)
{
    int dif = left + top - topleft;
    return abs(dif - left) < abs(dif - top) ? left : top;
}

void movec_pred(const DSV_MV *v, int nbh, int x, int y, int *px, int *py) // dsv.c:375
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

int neighbordif(const DSV_MV *v, int nbh, int x, int y) // dsv.c:404-447
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

int read_packet_hdr(BitReader &br) // dsv_decoder.c:21
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

void read_meta(DSV_DECODER *d, BitReader &br) // dsv_decoder.c:51
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
BitReader take_sub(BitReader &br, const uint8_t *base)
{
    br.align();
    unsigned len = br.get_ueg();
    br.align();
    BitReader sub{base + br.byte_pos(), 0};
    br.pos += len * 8;
    return sub;
}

void read_stability(DecImpl *im, BitReader &br, const uint8_t *base, int isP, const int *stats) // dsv_decoder.c:176
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

void read_intra_meta(DecImpl *im, BitReader &br, const uint8_t *base, const int *stats) // dsv_decoder.c:201
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

void read_motion(DecImpl *im, BitReader &br, const uint8_t *base, const int *stats) // dsv_decoder.c:81
{
    const CodecDev &dv = im->dev;
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

} // namespace

extern "C" {

void dsv_dec_free(DSV_DECODER *d)
{
    if (d->ref) {
        DecImpl *im = (DecImpl *) d->ref;
        if (im->ready) {
            im->dev.destroy();
        }
        delete im;
        d->ref = NULL;
    }
}

DSV_META *dsv_get_metadata(DSV_DECODER *d)
{
    DSV_META *m = (DSV_META *) dsv_alloc(sizeof(DSV_META));
    memcpy(m, &d->vidmeta, sizeof(DSV_META));
    return m;
}

int dsv_dec(DSV_DECODER *d, DSV_BUF *buffer, DSV_FRAME **out, DSV_FNUM *fn) // dsv_decoder.c:393
{
    *fn = (DSV_FNUM) -1;
    BitReader br{buffer->data, 0};
    int type = read_packet_hdr(br);
    if (type == -1) {
        dsv_buf_free(buffer);
        return DSV_DEC_ERROR;
    }
    if (!(type & DSV_PT_PIC)) {
        int ret = DSV_DEC_ERROR;
        if (type == DSV_PT_META) {
            read_meta(d, br);
            d->got_metadata = 1;
            ret = DSV_DEC_GOT_META;
        } else if (type == DSV_PT_EOS) {
            ret = DSV_DEC_EOS;
        }
        dsv_buf_free(buffer);
        return ret;
    }
    if (!d->got_metadata) {
        dsv_buf_free(buffer);
        return DSV_DEC_OK; /* picture before any metadata: skipped (dsv_decoder.c:436) */
    }
    const DSV_META *meta = &d->vidmeta;
    int has_ref = type & 1, is_ref = (type & 0x6) == 0x6;

    br.align();
    DSV_FNUM fno = br.get_bits(32);
    br.align();
    int blk_w = 16 << br.get_ueg(), blk_h = 16 << br.get_ueg();
    if (blk_w < 16 || blk_h < 16 || blk_w > 32 || blk_h > 32) {
        dsv_buf_free(buffer);
        return DSV_DEC_ERROR;
    }
    bind_device();
    DecImpl *im = (DecImpl *) d->ref;
    if (!im) {
        im = new DecImpl();
        d->ref = im;
    }
    if (im->ready && (im->dev.w != meta->width || im->dev.h != meta->height || im->dev.format != meta->subsamp ||
                      im->dev.blk_w != blk_w || im->dev.blk_h != blk_h)) {
        im->dev.destroy(); // stream parameters changed: start over
        im->ready = false;
        im->have_ref = false;
    }
    if (!im->ready) {
        im->dev.init(meta->subsamp, meta->width, meta->height, blk_w, blk_h, 0, false);
        im->ready = true;
    }
    CodecDev &dv = im->dev;
    size_t nb = dv.nblocks();

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
    int do_filter = (int) br.get_bit();
    int quant = (int) br.get_bits(DSV_MAX_QP_BITS);
    int lossless = quant == 1;
    if (br.get_bit()) {
        br.get_bits(15);
    }
    br.align();

    im->blockdata.assign(nb, 0);
    read_stability(im, br, buffer->data, has_ref, stats);
    if (has_ref) {
        im->mvs.assign(nb, DSV_MV{});
        read_motion(im, br, buffer->data, stats);
    } else {
        read_intra_meta(im, br, buffer->data, stats);
    }
    br.align();

    // ---- device side ----
    PicSet &cur = dv.pics[im->cur];
    PicSet &ref = dv.pics[im->cur ^ 1];
    DFrame &resid = dv.pred; // the decoder's residual picture
    HIPCHK(hipMemsetAsync(resid.alloc, 0, resid.bytes, dv.stream)); // a fresh zeroed frame per picture (dsv_decoder.c:506)
    HIPCHK(hipMemcpyAsync(dv.d_blockdata, im->blockdata.data(), nb, hipMemcpyHostToDevice, dv.stream));
    if (has_ref) {
        HIPCHK(hipMemcpyAsync(cur.d_final_mvs, im->mvs.data(), nb * sizeof(DSV_MV), hipMemcpyHostToDevice, dv.stream));
    }
    BlockMap bm{dv.d_blockdata, dv.nbh, dv.nbv};
    MCParams mc = dv.mc_params((int) (fno % 2), lossless);
    for (int c = 0; c < 3; c++) {
        const ScanGeom &g = dv.scan[c];
        im->pos.resize((size_t) g.base[10]);
        im->val.resize((size_t) g.base[10]);
        int seg_count[4];
        int32_t LL = 0;
        int ok = entropy_decode_plane(br, &LL, im->pos.data(), im->val.data(), seg_count, g);
        if (ok <= 0) {
            continue; /* "decoding error in plane": the residual plane stays zero (dsv_decoder.c:516-523) */
        }
        int nsym = seg_count[0] + seg_count[1] + seg_count[2] + seg_count[3];
        size_t ncoef = (size_t) dv.cw[c] * dv.ch[c];
        DCoefs co{dv.coefs[c], dv.cw[c], dv.ch[c]};
        HIPCHK(hipStreamSynchronize(dv.stream)); // the symbol staging buffers are reused per plane
        HIPCHK(hipMemsetAsync(co.data, 0, ncoef * sizeof(int32_t), dv.stream));
        if (nsym) {
            dv.ensure_dev_syms((size_t) nsym);
            HIPCHK(hipMemcpyAsync(dv.d_sym_pos, im->pos.data(), (size_t) nsym * sizeof(uint32_t), hipMemcpyHostToDevice, dv.stream));
            HIPCHK(hipMemcpyAsync(dv.d_sym_val, im->val.data(), (size_t) nsym * sizeof(int32_t), hipMemcpyHostToDevice, dv.stream));
            dequant_plane(dv.stream, co, dv.d_sym_pos, dv.d_sym_val, seg_count, dv.quant_cfg(c, has_ref, lossless, 0, nullptr), quant);
        }
        HIPCHK(hipMemcpyAsync(co.data, &LL, sizeof(int32_t), hipMemcpyHostToDevice, dv.stream)); /* dst->data[0] = LL */
        HIPCHK(hipStreamSynchronize(dv.stream));
        sbt_inverse(dv.stream, resid.p[c], co, dv.scratch, quant, c, has_ref, lossless, bm);
        if (!has_ref && c == 0 && do_filter) {
            intra_filter_luma(dv.stream, dv.d_blockdata, mc, quant, resid.p[0]);
        }
    }
    *fn = fno;
    if (has_ref) {
        if (!im->have_ref) {
            HIPCHK(hipStreamSynchronize(dv.stream));
            return DSV_DEC_ERROR; /* reference frame not found (dsv_decoder.c:535) */
        }
        HIPCHK(hipMemsetAsync(cur.recon.alloc, 0, cur.recon.bytes, dv.stream));
        mc_add_pred(dv.stream, cur.d_final_mvs, mc, quant, resid, cur.recon, ref.recon, do_filter, meta->inter_sharpen);
    } else {
        copy_frame_pixels(dv.stream, cur.recon, resid);
        extend_frame(dv.stream, cur.recon, false);
    }
    if (is_ref) {
        extend_frame(dv.stream, cur.recon, false);
    }
    DSV_FRAME *of = dsv_mk_frame(meta->subsamp, meta->width, meta->height, 1);
    if (is_ref || !has_ref) {
        dframe_download_full(&cur.recon, of, dv.stream);
    } else {
        dframe_download(&cur.recon, of, dv.stream);
    }
    HIPCHK(hipStreamSynchronize(dv.stream));
    if (is_ref) {
        im->cur ^= 1;
        im->have_ref = true;
    }
    dsv_buf_free(buffer);
    *out = of;
    return DSV_DEC_OK;
}

} // extern "C"
