// api_seam.cpp -- the hot-path seam of the C ABI (include/dsv2_hip.h sections 5 and 6).
//
// Section 5 functions have the reference's internal signatures (dsv_internal.h:112-147):
// operands arrive in host memory, are uploaded to HBM, processed by the HIP kernels and
// downloaded again, so a parity test can call the reference and this library with the
// very same arguments.  They serialise on one process-wide seam context.
// Section 6 (dsv2hip_planeset_*) keeps operands resident in HBM for measurement.
#include <string.h>

#include <mutex>

#include <vector>

#include "dev.h"
#include "bmc.h"
#include "hme.h"
#include "quant.h"

using namespace dsv2;

namespace {

struct SeamCtx {
    std::mutex mu;
    hipStream_t stream = nullptr;
    SbtScratch scratch;
    // cached device objects, re-created when the geometry changes
    DFrame frame[3];
    bool frame_ok[3] = {false, false, false};
    int32_t *coefs = nullptr;
    size_t coefs_elems = 0;
    uint8_t *blockdata = nullptr;
    size_t blockdata_bytes = 0;
    DSV_MV *mvs = nullptr;
    size_t mvs_elems = 0;
    int32_t *qv = nullptr;
    size_t qv_elems = 0;
    Compactor comp;
    uint32_t *sym_pos = nullptr;
    int32_t *sym_val = nullptr;
    size_t sym_elems = 0;

    void init()
    {
        bind_device();
        if (!stream) {
            HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
    }
    DFrame *get_frame(int slot, int format, int w, int h)
    {
        DFrame *f = &frame[slot];
        if (frame_ok[slot] && (f->format != format || f->w != w || f->h != h)) {
            dframe_free(f);
            frame_ok[slot] = false;
        }
        if (!frame_ok[slot]) {
            dframe_alloc(f, format, w, h);
            frame_ok[slot] = true;
        }
        return f;
    }
    int32_t *get_coefs(size_t n)
    {
        if (n > coefs_elems) {
            if (coefs) {
                HIPCHK(hipFree(coefs));
            }
            HIPCHK(hipMalloc((void **) &coefs, n * sizeof(int32_t)));
            coefs_elems = n;
        }
        return coefs;
    }
    const DSV_MV *put_mvs(const DSV_MV *host, size_t n)
    {
        if (!host || n == 0) {
            return nullptr;
        }
        if (n > mvs_elems) {
            if (mvs) {
                HIPCHK(hipFree(mvs));
            }
            HIPCHK(hipMalloc((void **) &mvs, n * sizeof(DSV_MV)));
            mvs_elems = n;
        }
        HIPCHK(hipMemcpyAsync(mvs, host, n * sizeof(DSV_MV), hipMemcpyHostToDevice, stream));
        return mvs;
    }
    int32_t *get_qv(size_t n)
    {
        if (n > qv_elems) {
            if (qv) {
                HIPCHK(hipFree(qv));
            }
            HIPCHK(hipMalloc((void **) &qv, n * sizeof(int32_t)));
            qv_elems = n;
        }
        return qv;
    }
    void get_syms(size_t n)
    {
        if (n > sym_elems) {
            if (sym_pos) {
                HIPCHK(hipFree(sym_pos));
                HIPCHK(hipFree(sym_val));
            }
            HIPCHK(hipMalloc((void **) &sym_pos, n * sizeof(uint32_t)));
            HIPCHK(hipMalloc((void **) &sym_val, n * sizeof(int32_t)));
            sym_elems = n;
        }
    }
    const uint8_t *put_blockdata(const uint8_t *host, size_t n)
    {
        if (!host || n == 0) {
            return nullptr;
        }
        if (n > blockdata_bytes) {
            if (blockdata) {
                HIPCHK(hipFree(blockdata));
            }
            HIPCHK(hipMalloc((void **) &blockdata, n));
            blockdata_bytes = n;
        }
        HIPCHK(hipMemcpyAsync(blockdata, host, n, hipMemcpyHostToDevice, stream));
        return blockdata;
    }
};

SeamCtx g_seam;

// A DPlane for a lone host DSV_PLANE: device storage with the bordered layout, filled from
// whatever host memory surrounds the plane when the host plane itself is bordered
// (stride >= w + 64 is taken as "bordered": the frame layouts of frame.c:88,130).
struct PlaneStage {
    uint8_t *dev = nullptr;
    size_t bytes = 0;
    DPlane p;
} g_pstage;

void stage_plane_in(SeamCtx &c, const DSV_PLANE *hp, bool copy_in)
{
    int stride = (hp->w + 2 * kBorder + 15) & ~15;
    size_t bytes = (size_t) stride * (hp->h + 2 * kBorder);
    if (bytes + 4096 > g_pstage.bytes) {
        if (g_pstage.dev) {
            HIPCHK(hipFree(g_pstage.dev));
        }
        HIPCHK(hipMalloc((void **) &g_pstage.dev, bytes + 4096));
        g_pstage.bytes = bytes + 4096;
    }
    g_pstage.p.data = g_pstage.dev + (size_t) stride * kBorder + kBorder;
    g_pstage.p.stride = stride;
    g_pstage.p.w = hp->w;
    g_pstage.p.h = hp->h;
    if (!copy_in) {
        return;
    }
    bool bordered = hp->stride >= hp->w + 2 * kBorder;
    if (bordered) {
        // bring the border along: the transform of an odd-width chroma plane reads one column of it
        HIPCHK(hipMemcpy2DAsync(g_pstage.dev, stride, hp->data - (size_t) hp->stride * kBorder - kBorder, hp->stride,
                                hp->w + 2 * kBorder, hp->h + 2 * kBorder, hipMemcpyHostToDevice, c.stream));
    } else {
        HIPCHK(hipMemsetAsync(g_pstage.dev, 0, bytes, c.stream));
        HIPCHK(hipMemcpy2DAsync(g_pstage.p.data, stride, hp->data, hp->stride, hp->w, hp->h, hipMemcpyHostToDevice,
                                c.stream));
    }
}

} // namespace

extern "C" {

int dsv2hip_device_ok(void)
{
    return device_status();
}

const char *dsv2hip_version(void) { return "dsv2hip 0.1 (DSV2 v2.8 bitstream, encoder v14 / decoder v2 semantics, gfx950)"; }

int dsv2hip_set_device(int ordinal)
{
    ensure_device();
    if (hipSetDevice(ordinal) != hipSuccess) {
        return -1;
    }
    set_default_device(ordinal);
    return 0;
}

void dsv_fwd_sbt(DSV_PLANE *src, DSV_COEFS *dst, DSV_FMETA *fm)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DSV_PARAMS *p = fm->params;
    size_t n = (size_t) dst->width * dst->height;
    stage_plane_in(c, src, true);
    DCoefs dc{c.get_coefs(n), dst->width, dst->height};
    // (rows of the coefficient plane below the picture are taken as zero: they are for every
    // plane the codec creates -- the reference leaves the caller's values there, sbt.c:805)
    BlockMap bm{c.put_blockdata(fm->blockdata, (size_t) p->nblocks_h * p->nblocks_v), p->nblocks_h, p->nblocks_v};
    sbt_forward(c.stream, g_pstage.p, dc, c.scratch, fm->cur_plane, fm->isP, p->lossless, bm);
    HIPCHK(hipMemcpyAsync(dst->data, dc.data, n * sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
}

void dsv_inv_sbt(DSV_PLANE *dst, DSV_COEFS *src, int q, DSV_FMETA *fm)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DSV_PARAMS *p = fm->params;
    size_t n = (size_t) src->width * src->height;
    stage_plane_in(c, dst, false);
    DCoefs dc{c.get_coefs(n), src->width, src->height};
    HIPCHK(hipMemcpyAsync(dc.data, src->data, n * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
    BlockMap bm{c.put_blockdata(fm->blockdata, (size_t) p->nblocks_h * p->nblocks_v), p->nblocks_h, p->nblocks_v};
    sbt_inverse(c.stream, g_pstage.p, dc, c.scratch, q, fm->cur_plane, fm->isP, p->lossless, bm);
    HIPCHK(hipMemcpy2DAsync(dst->data, dst->stride, g_pstage.p.data, g_pstage.p.stride, dst->w, dst->h,
                            hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
}

static QuantCfg make_quant_cfg(SeamCtx &c, const DSV_COEFS *co, const DSV_FMETA *fm)
{
    DSV_PARAMS *p = fm->params;
    size_t nb = (size_t) p->nblocks_h * p->nblocks_v;
    QuantCfg cfg;
    cfg.w = co->width;
    cfg.h = co->height;
    cfg.plane = fm->cur_plane;
    cfg.isP = fm->isP;
    cfg.lossless = p->lossless;
    cfg.do_psy = p->do_psy;
    cfg.hshift = DSV_FORMAT_H_SHIFT(p->vidmeta->subsamp);
    cfg.vshift = DSV_FORMAT_V_SHIFT(p->vidmeta->subsamp);
    cfg.blk_w = p->blk_w;
    cfg.blk_h = p->blk_h;
    cfg.nbh = p->nblocks_h;
    cfg.nbv = p->nblocks_v;
    cfg.bd = c.put_blockdata(fm->blockdata, nb);
    cfg.mvs = fm->isP ? c.put_mvs(fm->mvs, nb) : nullptr; // I frames never dereference it (hzcc.c:367-372)
    return cfg;
}

void dsv_encode_plane(DSV_BS *bs, DSV_COEFS *src, int q, DSV_FMETA *fm)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    size_t n = (size_t) src->width * src->height;
    ScanGeom g;
    make_scan(&g, src->width, src->height);
    DCoefs dc{c.get_coefs(n), src->width, src->height};
    HIPCHK(hipMemcpyAsync(dc.data, src->data, n * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
    QuantCfg cfg = make_quant_cfg(c, src, fm);
    int32_t *qv = c.get_qv((size_t) g.base[10]);
    quant_plane(c.stream, dc, qv, cfg, q);
    c.comp.run(c.stream, qv, (size_t) g.base[10]);
    HIPCHK(hipMemcpyAsync(src->data, dc.data, n * sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    int nsym = *c.comp.h_total;
    std::vector<uint32_t> pos((size_t) nsym);
    std::vector<int32_t> val((size_t) nsym);
    if (nsym) {
        HIPCHK(hipMemcpy(pos.data(), c.comp.d_pos, (size_t) nsym * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(val.data(), c.comp.d_val, (size_t) nsym * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    BitWriter bw{bs->start, bs->pos};
    entropy_encode_plane(bw, src->data[0], pos.data(), val.data(), nsym, g);
    bs->pos = bw.pos;
}

int dsv_decode_plane(DSV_BS *bs, DSV_COEFS *dst, int q, DSV_FMETA *fm)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    size_t n = (size_t) dst->width * dst->height;
    ScanGeom g;
    make_scan(&g, dst->width, dst->height);
    std::vector<uint32_t> pos((size_t) g.base[10]);
    std::vector<int32_t> val((size_t) g.base[10]);
    int seg_count[4];
    int32_t LL = 0;
    BitReader br{bs->start, bs->pos};
    int ok = entropy_decode_plane(br, &LL, pos.data(), val.data(), seg_count, g);
    bs->pos = br.pos;
    int nsym = seg_count[0] + seg_count[1] + seg_count[2] + seg_count[3];
    DCoefs dc{c.get_coefs(n), dst->width, dst->height};
    HIPCHK(hipMemcpyAsync(dc.data, dst->data, n * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
    QuantCfg cfg = make_quant_cfg(c, dst, fm);
    c.get_syms((size_t) (nsym > 0 ? nsym : 1));
    if (nsym) {
        HIPCHK(hipMemcpyAsync(c.sym_pos, pos.data(), (size_t) nsym * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipMemcpyAsync(c.sym_val, val.data(), (size_t) nsym * sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
        dequant_plane(c.stream, dc, c.sym_pos, c.sym_val, seg_count, dst->data[0], cfg, q); // DC handled below
    }
    HIPCHK(hipMemcpyAsync(dst->data, dc.data, n * sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (ok >= 0) {
        dst->data[0] = LL; // hzcc.c:633
    }
    return ok > 0;
}

static MCParams make_mc_params(const DSV_PARAMS *p)
{
    MCParams m;
    m.blk_w = p->blk_w;
    m.blk_h = p->blk_h;
    m.nbh = p->nblocks_h;
    m.nbv = p->nblocks_v;
    m.hshift = DSV_FORMAT_H_SHIFT(p->vidmeta->subsamp);
    m.vshift = DSV_FORMAT_V_SHIFT(p->vidmeta->subsamp);
    m.temporal_mc = p->temporal_mc;
    m.lossless = p->lossless;
    return m;
}

void dsv_sub_pred(DSV_MV *mv, DSV_PARAMS *p, DSV_FRAME *pred, DSV_FRAME *resd, DSV_FRAME *ref)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DFrame *dpred = c.get_frame(0, pred->format, pred->width, pred->height);
    DFrame *dres = c.get_frame(1, resd->format, resd->width, resd->height);
    DFrame *dref = c.get_frame(2, ref->format, ref->width, ref->height);
    dframe_upload_full(dpred, pred, c.stream);
    dframe_upload_full(dres, resd, c.stream);
    dframe_upload_full(dref, ref, c.stream);
    const DSV_MV *dmv = c.put_mvs(mv, (size_t) p->nblocks_h * p->nblocks_v);
    mc_sub_pred(c.stream, dmv, make_mc_params(p), *dpred, *dres, *dref);
    dframe_download_full(dpred, pred, c.stream);
    dframe_download_full(dres, resd, c.stream);
    HIPCHK(hipStreamSynchronize(c.stream));
}

void dsv_add_res(DSV_MV *mv, DSV_FMETA *fm, int q, DSV_FRAME *resd, DSV_FRAME *pred, int do_filter)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DSV_PARAMS *p = fm->params;
    DFrame *dpred = c.get_frame(0, pred->format, pred->width, pred->height);
    DFrame *dres = c.get_frame(1, resd->format, resd->width, resd->height);
    dframe_upload_full(dpred, pred, c.stream);
    dframe_upload_full(dres, resd, c.stream);
    const DSV_MV *dmv = c.put_mvs(mv, (size_t) p->nblocks_h * p->nblocks_v);
    mc_add_res(c.stream, dmv, make_mc_params(p), q, *dres, *dpred, do_filter, p->vidmeta->inter_sharpen);
    dframe_download_full(dres, resd, c.stream);
    HIPCHK(hipStreamSynchronize(c.stream));
}

void dsv_add_pred(DSV_MV *mv, DSV_FMETA *fm, int q, DSV_FRAME *resd, DSV_FRAME *out, DSV_FRAME *ref, int do_filter)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DSV_PARAMS *p = fm->params;
    DFrame *dout = c.get_frame(0, out->format, out->width, out->height);
    DFrame *dres = c.get_frame(1, resd->format, resd->width, resd->height);
    DFrame *dref = c.get_frame(2, ref->format, ref->width, ref->height);
    dframe_upload_full(dout, out, c.stream);
    dframe_upload_full(dres, resd, c.stream);
    dframe_upload_full(dref, ref, c.stream);
    const DSV_MV *dmv = c.put_mvs(mv, (size_t) p->nblocks_h * p->nblocks_v);
    mc_add_pred(c.stream, dmv, make_mc_params(p), q, *dres, *dout, *dref, do_filter, p->vidmeta->inter_sharpen);
    dframe_download_full(dout, out, c.stream);
    HIPCHK(hipStreamSynchronize(c.stream));
}

void dsv_intra_filter(int q, DSV_PARAMS *p, DSV_FMETA *fm, int cpl, DSV_PLANE *dp, int do_filter)
{
    if (p->lossless || cpl != 0 || !do_filter) {
        return; // bmc.c:396-404
    }
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    stage_plane_in(c, dp, true);
    const uint8_t *dbd = c.put_blockdata(fm->blockdata, (size_t) p->nblocks_h * p->nblocks_v);
    intra_filter_luma(c.stream, dbd, make_mc_params(p), q, g_pstage.p);
    HIPCHK(hipMemcpy2DAsync(dp->data, dp->stride, g_pstage.p.data, g_pstage.p.stride, dp->w, dp->h, hipMemcpyDeviceToHost,
                            c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
}

void dsv_post_process(DSV_PLANE *dp)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    stage_plane_in(c, dp, true);
    post_process_plane(c.stream, g_pstage.p);
    HIPCHK(hipMemcpy2DAsync(dp->data, dp->stride, g_pstage.p.data, g_pstage.p.stride, dp->w, dp->h, hipMemcpyDeviceToHost,
                            c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
}

DSV_MV *dsv_intra_analysis(DSV_FRAME *src, DSV_PARAMS *p)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    size_t nb = (size_t) p->nblocks_h * p->nblocks_v;
    DFrame *d = c.get_frame(0, src->format, src->width, src->height);
    dframe_upload(d, src, c.stream);
    AnalysisParams ap;
    ap.width = src->width;
    ap.height = src->height;
    ap.blk_w = p->blk_w;
    ap.blk_h = p->blk_h;
    ap.nbh = p->nblocks_h;
    ap.nbv = p->nblocks_v;
    ap.hshift = DSV_FORMAT_H_SHIFT(p->vidmeta->subsamp);
    ap.vshift = DSV_FORMAT_V_SHIFT(p->vidmeta->subsamp);
    ap.do_psy = p->do_psy;
    ap.scale = 2 * spatial_psy_factor(p->blk_w, p->blk_h, p->nblocks_h, p->nblocks_v, -1);
    DSV_MV *host = (DSV_MV *) dsv_alloc((int) (nb * sizeof(DSV_MV)));
    c.put_mvs(host, nb); // sizes the device field (contents are overwritten by the kernel)
    intra_analysis(c.stream, *d, ap, c.mvs);
    HIPCHK(hipMemcpyAsync(host, c.mvs, nb * sizeof(DSV_MV), hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return host;
}

int dsv_hme(DSV_HME *hme, int *scene_change_blocks, int *avg_err) // hme.c:2001
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DSV_PARAMS *p = hme->params;
    int levels = hme->enc->pyramid_levels;
    size_t nb = (size_t) p->nblocks_h * p->nblocks_v;
    // upload every pyramid level exactly as handed over (bordered host frames)
    std::vector<DFrame> dsrc((size_t) levels + 1), dref((size_t) levels + 1), dogr((size_t) levels + 1);
    HmeFrames f;
    for (int l = 0; l <= levels; l++) {
        DSV_FRAME *hs = hme->src[l], *hr = hme->ref[l], *ho = hme->ogr[l];
        dframe_alloc(&dsrc[(size_t) l], hs->format, hs->width, hs->height);
        dframe_alloc(&dref[(size_t) l], hr->format, hr->width, hr->height);
        dframe_alloc(&dogr[(size_t) l], ho->format, ho->width, ho->height);
        dframe_upload_full(&dsrc[(size_t) l], hs, c.stream);
        dframe_upload_full(&dref[(size_t) l], hr, c.stream);
        dframe_upload_full(&dogr[(size_t) l], ho, c.stream);
        f.src[l] = dsrc[(size_t) l].p[0];
        f.ref[l] = dref[(size_t) l].p[0];
        f.ogr[l] = dogr[(size_t) l].p[0];
        HIPCHK(hipMalloc((void **) &f.mvf[l], nb * sizeof(DSV_MV)));
    }
    for (int k = 0; k < 2; k++) {
        f.srcc[k] = dsrc[0].p[k + 1];
        f.refc[k] = dref[0].p[k + 1];
    }
    f.ref_mvf = c.put_mvs(hme->ref_mvf, nb);
    int *d_counters;
    HIPCHK(hipMalloc((void **) &d_counters, hme_counter_words(p->nblocks_v) * sizeof(int)));
    f.counters = d_counters;
    HmeParams hp;
    hp.a.width = p->vidmeta->width;
    hp.a.height = p->vidmeta->height;
    hp.a.blk_w = p->blk_w;
    hp.a.blk_h = p->blk_h;
    hp.a.nbh = p->nblocks_h;
    hp.a.nbv = p->nblocks_v;
    hp.a.hshift = DSV_FORMAT_H_SHIFT(p->vidmeta->subsamp);
    hp.a.vshift = DSV_FORMAT_V_SHIFT(p->vidmeta->subsamp);
    hp.a.do_psy = p->do_psy;
    hp.a.scale = 0;
    hp.effort = p->effort;
    hp.lossless = p->lossless;
    hp.quant = hme->quant;
    hp.skip_block_thresh = hme->enc->skip_block_thresh;
    hp.pyr_levels = levels;
    hme_run(c.stream, f, hp);
    int counters[16];
    HIPCHK(hipMemcpyAsync(counters, d_counters, sizeof(counters), hipMemcpyDeviceToHost, c.stream));
    for (int l = 0; l <= levels; l++) {
        hme->mvf[l] = (DSV_MV *) dsv_alloc((int) (nb * sizeof(DSV_MV)));
        HIPCHK(hipMemcpyAsync(hme->mvf[l], f.mvf[l], nb * sizeof(DSV_MV), hipMemcpyDeviceToHost, c.stream));
    }
    HIPCHK(hipStreamSynchronize(c.stream));
    for (int l = 0; l <= levels; l++) {
        dframe_free(&dsrc[(size_t) l]);
        dframe_free(&dref[(size_t) l]);
        dframe_free(&dogr[(size_t) l]);
        HIPCHK(hipFree(f.mvf[l]));
    }
    HIPCHK(hipFree(d_counters));
    if (counters[7]) {
        fatal("motion estimation row pipeline timed out", __FILE__, __LINE__);
    }
    *scene_change_blocks = counters[1] * 100 / (counters[2] ? counters[2] : 1); // hme.c:1825-1832
    *avg_err = (int) ((unsigned) counters[3] / (unsigned) nb);
    return counters[0] * 100 / (int) nb;
}

DSV_FRAME *dsv_extend_frame(DSV_FRAME *frame)
{
    if (!frame->border) {
        return frame;
    }
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DFrame *d = c.get_frame(0, frame->format, frame->width, frame->height);
    dframe_upload(d, frame, c.stream);
    extend_frame(c.stream, *d, false);
    dframe_download_full(d, frame, c.stream);
    HIPCHK(hipStreamSynchronize(c.stream));
    return frame;
}

DSV_FRAME *dsv_extend_frame_luma(DSV_FRAME *frame)
{
    if (!frame->border) {
        return frame;
    }
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DFrame *d = c.get_frame(0, frame->format, frame->width, frame->height);
    // only the luma plane travels: upload its pixels, extend, download plane 0 with its border
    const DSV_PLANE *hp = &frame->planes[0];
    HIPCHK(hipMemcpy2DAsync(d->p[0].data, d->p[0].stride, hp->data, hp->stride, hp->w, hp->h, hipMemcpyHostToDevice,
                            c.stream));
    extend_plane(c.stream, d->p[0]);
    HIPCHK(hipMemcpy2DAsync(hp->data - (size_t) hp->stride * kBorder - kBorder, hp->stride, d->alloc + d->plane_off[0],
                            d->p[0].stride, hp->w + 2 * kBorder, hp->h + 2 * kBorder, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return frame;
}

void dsv_frame_copy(DSV_FRAME *dst, DSV_FRAME *src)
{
    // pixel copy on the host (frame.c:186-203: src width bytes for each of dst's rows) ...
    for (int c = 0; c < 3; c++) {
        const DSV_PLANE *sp = &src->planes[c];
        DSV_PLANE *dp = &dst->planes[c];
        for (int y = 0; y < dp->h; y++) {
            memcpy(dp->data + (size_t) y * dp->stride, sp->data + (size_t) y * sp->stride, (size_t) sp->w);
        }
    }
    // ... and the border synthesis on the GPU
    if (dst->border) {
        dsv_extend_frame(dst);
    }
}

void dsv_ds2x_frame_luma(DSV_FRAME *dst, DSV_FRAME *src)
{
    SeamCtx &c = g_seam;
    std::lock_guard<std::mutex> lk(c.mu);
    c.init();
    DFrame *s = c.get_frame(0, src->format, src->width, src->height);
    DFrame *d = c.get_frame(1, dst->format, dst->width, dst->height);
    const DSV_PLANE *sp = &src->planes[0];
    DSV_PLANE *dp = &dst->planes[0];
    if (sp->stride >= sp->w + 2 * kBorder) { // bordered source: odd sizes read one row/column of border
        HIPCHK(hipMemcpy2DAsync(s->alloc + s->plane_off[0], s->p[0].stride, sp->data - (size_t) sp->stride * kBorder - kBorder,
                                sp->stride, sp->w + 2 * kBorder, sp->h + 2 * kBorder, hipMemcpyHostToDevice, c.stream));
    } else {
        HIPCHK(hipMemcpy2DAsync(s->p[0].data, s->p[0].stride, sp->data, sp->stride, sp->w, sp->h, hipMemcpyHostToDevice,
                                c.stream));
    }
    ds2x_luma(c.stream, s->p[0], d->p[0]);
    HIPCHK(hipMemcpy2DAsync(dp->data, dp->stride, d->p[0].data, d->p[0].stride, dp->w, dp->h, hipMemcpyDeviceToHost,
                            c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
}

/* ---- section 6: device-resident plane set ---- */

struct dsv2hip_planeset {
    hipStream_t stream;
    DFrame pic;
    int32_t *coefs[3];
    int cw[3], ch[3];
    SbtScratch scratch;
    uint8_t *bd;
    int nbh, nbv;
    hipEvent_t e0, e1;
};

dsv2hip_planeset *dsv2hip_planeset_create(int format, int width, int height)
{
    ensure_device();
    dsv2hip_planeset *ps = new dsv2hip_planeset();
    HIPCHK(hipStreamCreateWithFlags(&ps->stream, hipStreamNonBlocking));
    dframe_alloc(&ps->pic, format, width, height);
    coef_dims(format, width, height, ps->cw, ps->ch);
    for (int c = 0; c < 3; c++) {
        size_t n = (size_t) ps->cw[c] * ps->ch[c];
        HIPCHK(hipMalloc((void **) &ps->coefs[c], n * sizeof(int32_t)));
        dev_zero(ps->coefs[c], n * sizeof(int32_t));
    }
    ps->scratch.ensure((size_t) ps->cw[0] * ps->ch[0]);
    ps->bd = nullptr;
    ps->nbh = ps->nbv = 0;
    HIPCHK(hipEventCreate(&ps->e0));
    HIPCHK(hipEventCreate(&ps->e1));
    return ps;
}

void dsv2hip_planeset_destroy(dsv2hip_planeset *ps)
{
    if (!ps) {
        return;
    }
    HIPCHK(hipStreamSynchronize(ps->stream));
    dframe_free(&ps->pic);
    for (int c = 0; c < 3; c++) {
        HIPCHK(hipFree(ps->coefs[c]));
    }
    ps->scratch.release();
    if (ps->bd) {
        HIPCHK(hipFree(ps->bd));
    }
    HIPCHK(hipEventDestroy(ps->e0));
    HIPCHK(hipEventDestroy(ps->e1));
    HIPCHK(hipStreamDestroy(ps->stream));
    delete ps;
}

int dsv2hip_planeset_upload(dsv2hip_planeset *ps, const DSV_FRAME *frame)
{
    dframe_upload(&ps->pic, frame, ps->stream);
    HIPCHK(hipStreamSynchronize(ps->stream));
    return 0;
}

int dsv2hip_planeset_download(dsv2hip_planeset *ps, DSV_FRAME *frame)
{
    dframe_download(&ps->pic, frame, ps->stream);
    HIPCHK(hipStreamSynchronize(ps->stream));
    return 0;
}

int dsv2hip_planeset_set_blockdata(dsv2hip_planeset *ps, const uint8_t *blockdata, int nblocks_h, int nblocks_v)
{
    size_t n = (size_t) nblocks_h * nblocks_v;
    if (ps->bd) {
        HIPCHK(hipFree(ps->bd));
    }
    HIPCHK(hipMalloc((void **) &ps->bd, n));
    HIPCHK(hipMemcpy(ps->bd, blockdata, n, hipMemcpyHostToDevice));
    ps->nbh = nblocks_h;
    ps->nbv = nblocks_v;
    return 0;
}

int dsv2hip_planeset_fwd_sbt(dsv2hip_planeset *ps, int c, int isP, int lossless)
{
    sbt_forward(ps->stream, ps->pic.p[c], DCoefs{ps->coefs[c], ps->cw[c], ps->ch[c]}, ps->scratch, c, isP, lossless,
                BlockMap{ps->bd, ps->nbh, ps->nbv});
    return 0;
}

int dsv2hip_planeset_inv_sbt(dsv2hip_planeset *ps, int c, int q, int isP, int lossless)
{
    sbt_inverse(ps->stream, ps->pic.p[c], DCoefs{ps->coefs[c], ps->cw[c], ps->ch[c]}, ps->scratch, q, c, isP, lossless,
                BlockMap{ps->bd, ps->nbh, ps->nbv});
    return 0;
}

int dsv2hip_planeset_get_coefs(dsv2hip_planeset *ps, int c, DSV_SBC *out)
{
    HIPCHK(hipMemcpyAsync(out, ps->coefs[c], (size_t) ps->cw[c] * ps->ch[c] * sizeof(int32_t), hipMemcpyDeviceToHost,
                          ps->stream));
    HIPCHK(hipStreamSynchronize(ps->stream));
    return 0;
}

int dsv2hip_planeset_set_coefs(dsv2hip_planeset *ps, int c, const DSV_SBC *in)
{
    HIPCHK(hipMemcpyAsync(ps->coefs[c], in, (size_t) ps->cw[c] * ps->ch[c] * sizeof(int32_t), hipMemcpyHostToDevice,
                          ps->stream));
    HIPCHK(hipStreamSynchronize(ps->stream));
    return 0;
}

int dsv2hip_planeset_sync(dsv2hip_planeset *ps)
{
    HIPCHK(hipStreamSynchronize(ps->stream));
    return 0;
}

float dsv2hip_planeset_time_sbt(dsv2hip_planeset *ps, int c, int isP, int lossless, int inverse, int q, int iters)
{
    if (iters <= 0) {
        return -1.0f;
    }
    HIPCHK(hipEventRecord(ps->e0, ps->stream));
    for (int i = 0; i < iters; i++) {
        if (inverse) {
            dsv2hip_planeset_inv_sbt(ps, c, q, isP, lossless);
        } else {
            dsv2hip_planeset_fwd_sbt(ps, c, isP, lossless);
        }
    }
    HIPCHK(hipEventRecord(ps->e1, ps->stream));
    HIPCHK(hipEventSynchronize(ps->e1));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, ps->e0, ps->e1));
    return ms / (float) iters;
}

} // extern "C"
