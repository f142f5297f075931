// codec_dev.cpp -- allocation of the device-resident state of a codec instance (codec.h).
//
// HBM footprint of one 1080p 4:2:0 encoder instance: 2 picture sets (source, source pyramid,
// reconstruction, reconstruction pyramid: ~9 MB each), prediction 3.5 MB, coefficient planes
// 12.4 MB, dense quantised array 12.4 MB, transform scratch 3 x 8.3 MB, symbol buffers 25 MB:
// ~95 MB, so thousands of instances fit in 288 GB -- concurrency is bounded by CUs, not memory.
#include "codec.h"

#include <stdlib.h>

#include <algorithm>

#include <atomic>
#include <mutex>

namespace dsv2 {

static std::atomic<int> g_prof_on{0};
static std::mutex g_prof_mu;
static double g_prof_ms[ST_COUNT];
static long long g_prof_launches[ST_COUNT], g_prof_units[ST_COUNT];
static long long g_prof_frames;

bool prof_enabled() { return g_prof_on.load() != 0; }

void StageProf::destroy()
{
    if (!created) {
        return;
    }
    for (int i = 0; i < ST_COUNT; i++) {
        HIPCHK(hipEventDestroy(ev[i][0]));
        HIPCHK(hipEventDestroy(ev[i][1]));
    }
    created = false;
}

void StageProf::begin(hipStream_t s, int st)
{
    if (!prof_enabled()) {
        return;
    }
    if (!created) { // events exist only once profiling has been asked for
        for (int i = 0; i < ST_COUNT; i++) {
            HIPCHK(hipEventCreate(&ev[i][0]));
            HIPCHK(hipEventCreate(&ev[i][1]));
            used[i] = false;
            launches[i] = units[i] = 0;
        }
        created = true;
    }
    if (!used[st]) {
        HIPCHK(hipEventRecord(ev[st][0], s));
    }
    mark = t_launch_count;
}

void StageProf::end(hipStream_t s, int st, int nunits, int nlaunch)
{
    if (!prof_enabled() || !created) {
        return;
    }
    HIPCHK(hipEventRecord(ev[st][1], s)); // the last end() of a step closes the stage's span
    used[st] = true;
    launches[st] += nlaunch >= 0 ? nlaunch : t_launch_count - mark;
    units[st] += nunits;
}

void StageProf::collect()
{
    if (!created) {
        return;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < ST_COUNT; i++) {
        if (used[i]) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev[i][0], ev[i][1]));
            g_prof_ms[i] += ms;
            g_prof_launches[i] += launches[i];
            g_prof_units[i] += units[i];
            used[i] = false;
            launches[i] = units[i] = 0;
        }
    }
    g_prof_frames++;
}

extern "C" void dsv2hip_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on;
    for (int i = 0; i < ST_COUNT; i++) {
        g_prof_ms[i] = 0;
        g_prof_launches[i] = 0;
        g_prof_units[i] = 0;
    }
    g_prof_frames = 0;
}

// stream-frames each stage processed since profiling was switched on (same order as dsv2hip_prof_read)
extern "C" int dsv2hip_prof_read_units(long long *units)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < ST_COUNT; i++) {
        units[i] = g_prof_units[i];
    }
    return ST_COUNT;
}

extern "C" int dsv2hip_prof_read(double *ms, long long *launches, long long *frames)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < ST_COUNT; i++) {
        ms[i] = g_prof_ms[i];
        launches[i] = g_prof_launches[i];
    }
    *frames = g_prof_frames;
    return ST_COUNT;
}

void block_geometry(int w, int h, int ovx, int ovy, int *blk_w, int *blk_h, int *nbh, int *nbv) // dsv_encoder.c:1203-1222
{
    int bw = w > 1280 ? DSV_MAX_BLOCK_SIZE : DSV_MIN_BLOCK_SIZE;
    int bh = h > 1280 ? DSV_MAX_BLOCK_SIZE : DSV_MIN_BLOCK_SIZE;
    int d = w > h ? w - h : h - w;
    if (d < (w < h ? w : h)) { // mostly square picture: square blocks
        bw = bh = bw < bh ? bw : bh;
    }
    if (ovx >= 0) {
        bw = 16 << ovx;
        bw = bw < 16 ? 16 : (bw > 32 ? 32 : bw);
    }
    if (ovy >= 0) {
        bh = 16 << ovy;
        bh = bh < 16 ? 16 : (bh > 32 ? 32 : bh);
    }
    *blk_w = bw;
    *blk_h = bh;
    *nbh = (w + bw - 1) / bw;
    *nbv = (h + bh - 1) / bh;
}

// Section buffer of the GPU entropy coder: 4 bits per coefficient (1.5 MB at 1080p: a 370 Mbit/s stream at 30 pictures a second),
// at least 1 MB; 4 MB or that, whichever is more, for streams whose lists are sized for the worst case (lossless).  A picture
// that does not fit raises the coder's "no room" flag and is coded on the host.
static uint32_t ent_out_bytes(size_t list_symbols, size_t ncoef)
{
    const size_t half = (ncoef / 2 + 65535) & ~(size_t) 65535;
    return (uint32_t) std::max<size_t>(list_symbols >= ncoef ? (4u << 20) : (1u << 20), half);
}

void CodecDev::init(int format_, int w_, int h_, int blk_w_, int blk_h_, int pyr_levels_, bool encoder, size_t list_symbols)
{
    ensure_device();
    format = format_;
    w = w_;
    h = h_;
    blk_w = blk_w_;
    blk_h = blk_h_;
    nbh = (w + blk_w - 1) / blk_w;
    nbv = (h + blk_h - 1) / blk_h;
    pyr_levels = pyr_levels_;
    alive = true;
    coef_dims(format, w, h, cw, ch);
    size_t nb = nblocks();
    // encoder: everything this function (and the entropy buffers) allocates on the device comes out of ONE block, sized here
    // from the same formulas the allocations use, plus slack; what does not fit falls back to an allocation of its own
    constexpr bool use_arena = true;
    if (encoder && use_arena) {
        // (the same sizes the allocations below ask for; 256 bytes of alignment for each of the ~100 pieces)
        size_t est = 0;
        est += 2 * (2 * dframe_bytes(format, w, h) + nb * sizeof(DSV_MV));
        for (int l = 0; l < pyr_levels; l++) {
            int lw = (w + (1 << (l + 1)) - 1) >> (l + 1), lh = (h + (1 << (l + 1)) - 1) >> (l + 1);
            est += 4 * dframe_bytes(format, lw, lh);
        }
        est += dframe_bytes(format, w, h); // pred
        size_t ncoef = 0, nscan = 0;
        for (int c = 0; c < 3; c++) {
            ScanGeom g;
            make_scan(&g, cw[c], ch[c]);
            ncoef += (size_t) cw[c] * ch[c];
            nscan += (size_t) g.base[10];
        }
        est += ncoef * sizeof(int32_t); // coefficient planes
        {
            const size_t lum = SbtScratch::scratch_elems((size_t) cw[0] * ch[0], sbt_ll_elems(cw[0], ch[0]));
            const size_t chr = SbtScratch::scratch_elems((size_t) cw[1] * ch[1], sbt_ll_elems(cw[1], ch[1]));
            est += std::max(std::max(lum, 2 * chr), (nscan + 3) & ~(size_t) 3) * sizeof(int32_t); // work block
        }
        const size_t nlist = list_symbols && list_symbols < nscan ? list_symbols : nscan;
        est += nlist * 8 + ((nscan + 1023) / 1024) * 8 + 4;                                  // compaction lists, tile counts
        est += ((nlist + 1023) / 1024 + 3) * (1024 + 256 * 2 + 2 + 4 + 4 + 8) + ent_out_bytes(nlist, nscan) + 64 + 128; // entropy coder (EntBuffers::ensure)
        est += nb * (1 + sizeof(DSV_MV) + 2) + (size_t) (pyr_levels + 1) * nb * sizeof(DSV_MV);
        est += hme_counter_words(nbv) * sizeof(int) + hme_src_stats_bytes(nbh, nbv) + hme_l0_pre_bytes(nbh, nbv) + 16 + 256;
        est += 128 * 256;
        arena.create(est);
    }
    DevArenaScope arena_scope(encoder && use_arena ? &arena : nullptr);
    for (int i = 0; i < 2; i++) {
        dframe_alloc(&pics[i].recon, format, w, h);
        HIPCHK(dev_alloc((void **) &pics[i].d_final_mvs, nb * sizeof(DSV_MV)));
        dev_zero(pics[i].d_final_mvs, nb * sizeof(DSV_MV));
        if (encoder) {
            dframe_alloc(&pics[i].src, format, w, h);
            for (int l = 0; l < pyr_levels; l++) {
                int lw = (w + (1 << (l + 1)) - 1) >> (l + 1), lh = (h + (1 << (l + 1)) - 1) >> (l + 1);
                dframe_alloc(&pics[i].src_pyr[l], format, lw, lh);
                dframe_alloc(&pics[i].recon_pyr[l], format, lw, lh);
            }
        }
    }
    dframe_alloc(&pred, format, w, h);
    qv_off[0] = 0;
    for (int c = 0; c < 3; c++) {
        size_t n = (size_t) cw[c] * ch[c];
        HIPCHK(dev_alloc((void **) &coefs[c], n * sizeof(int32_t)));
        dev_zero(coefs[c], n * sizeof(int32_t));
        make_scan(&scan[c], cw[c], ch[c]);
        qv_off[c + 1] = qv_off[c] + (size_t) scan[c].base[10];
    }
    if (encoder) {
        // One work block per instance instead of seven allocations (round 4: 130 -> 109 MB per 1080p instance).  Its tenants
        // never live at the same time -- a step's kernels for one instance follow one another on one stream:
        //   forward transform of the luma plane   scratch images of the luma plane
        //   forward transform of U and V (one launch)   two chroma scratch sets, over the (dead) luma images
        //   quantiser -> ordered compaction   the dense quantised values `qv`, over the (dead) scratch images
        //   inverse transforms   the scratch images again (the compaction lists hold what the entropy coder reads)
        const size_t n0 = (size_t) cw[0] * ch[0], l0 = sbt_ll_elems(cw[0], ch[0]), n1 = (size_t) cw[1] * ch[1], l1 = sbt_ll_elems(cw[1], ch[1]);
        const size_t lum = SbtScratch::scratch_elems(n0, l0), chr = SbtScratch::scratch_elems(n1, l1);
        const size_t elems = std::max(std::max(lum, 2 * chr), (qv_off[3] + 3) & ~(size_t) 3);
        HIPCHK(dev_alloc((void **) &work, elems * sizeof(int32_t)));
        scratch.borrow(work, n0, l0);
        scratch_uv[0].borrow(work, n1, l1);
        scratch_uv[1].borrow(work + chr, n1, l1);
        qv = work;
        comp.ensure_lists(qv_off[3], list_symbols ? list_symbols : qv_off[3]);
        // (the plane sections of the packet assembled on the GPU: 4 MB section buffer, 1 MB pinned mirror -- far above any 1080p
        // picture at sane quality; larger ones are fetched by a copy)
        ent.ensure(comp.list_cap, ent_out_bytes(comp.list_cap, qv_off[3]), 1u << 20);
    } else {
        HIPCHK(dev_alloc((void **) &qv, qv_off[3] * sizeof(int32_t)));
        scratch.ensure((size_t) cw[0] * ch[0], sbt_ll_elems(cw[0], ch[0]));
    }
    HIPCHK(dev_alloc((void **) &d_blockdata, nb));
    dev_zero(d_blockdata, nb);
    HIPCHK(dev_alloc((void **) &d_mvs_stage, nb * sizeof(DSV_MV)));
    dev_zero(d_mvs_stage, nb * sizeof(DSV_MV));
    for (int l = 0; l <= pyr_levels; l++) {
        HIPCHK(dev_alloc((void **) &d_mvf[l], nb * sizeof(DSV_MV)));
    }
    HIPCHK(dev_alloc((void **) &d_counters, hme_counter_words(nbv) * sizeof(int)));
    if (encoder) { // the search's pre-pass records: nothing a decoder ever reads (advisor, round 5: 2 MB per 1080p decoder)
        HIPCHK(dev_alloc(&d_src_stats, hme_src_stats_bytes(nbh, nbv)));
        HIPCHK(dev_alloc(&d_l0_pre, hme_l0_pre_bytes(nbh, nbv)));
    }
    for (int i = 0; i < 2; i++) {
        HIPCHK(dev_alloc((void **) &d_intra_map[i], nb));
        dev_zero(d_intra_map[i], nb);
    }
    HIPCHK(dev_alloc((void **) &d_ll, 4 * sizeof(int32_t)));
    h_frame_bytes = 0;
    for (int c = 0; c < 3; c++) {
        h_frame_bytes += (size_t) pred.p[c].w * pred.p[c].h;
    }
    HIPCHK(hipHostMalloc((void **) &h_frame, h_frame_bytes, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &h_mvs, nb * sizeof(DSV_MV), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &h_intra, nb * sizeof(DSV_MV), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &h_counters, 16 * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &h_ll, 4 * sizeof(int32_t), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &h_small, (size_t) 1 << 20, hipHostMallocDefault));
}

void CodecDev::ensure_host_syms(size_t n)
{
    if (n <= h_sym_cap) {
        return;
    }
    if (h_pos) {
        HIPCHK(hipHostFree(h_pos));
        HIPCHK(hipHostFree(h_val));
    }
    size_t cap = n + n / 4 + 4096;
    HIPCHK(hipHostMalloc((void **) &h_pos, cap * sizeof(uint32_t), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &h_val, cap * sizeof(int32_t), hipHostMallocDefault));
    h_sym_cap = cap;
}

void CodecDev::ensure_dev_syms(size_t n)
{
    if (n <= sym_cap) {
        return;
    }
    if (d_sym_pos) {
        dev_release(d_sym_pos);
        dev_release(d_sym_val);
    }
    size_t cap = n + n / 4 + 4096;
    HIPCHK(dev_alloc((void **) &d_sym_pos, cap * sizeof(uint32_t)));
    HIPCHK(dev_alloc((void **) &d_sym_val, cap * sizeof(int32_t)));
    sym_cap = cap;
}

hipStream_t CodecDev::ensure_stream()
{
    if (!stream) {
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    return stream;
}

void CodecDev::destroy()
{
    if (!alive) {
        return;
    }
    alive = false;
    if (stream) {
        HIPCHK(hipStreamSynchronize(stream));
    }
    prof.destroy();
    for (int i = 0; i < 2; i++) {
        dframe_free(&pics[i].recon);
        dframe_free(&pics[i].src);
        for (int l = 0; l < DSV_MAX_PYRAMID_LEVELS; l++) {
            dframe_free(&pics[i].src_pyr[l]);
            dframe_free(&pics[i].recon_pyr[l]);
        }
        dev_release(pics[i].d_final_mvs);
    }
    dframe_free(&pred);
    for (int c = 0; c < 3; c++) {
        dev_release(coefs[c]);
    }
    if (work) {
        dev_release(work);
        work = nullptr;
    } else {
        dev_release(qv);
    }
    qv = nullptr;
    scratch.release();
    scratch_uv[0].release();
    scratch_uv[1].release();
    comp.release();
    ent.release();
    dev_release(d_blockdata);
    dev_release(d_mvs_stage);
    for (int l = 0; l <= DSV_MAX_PYRAMID_LEVELS; l++) {
        if (d_mvf[l]) {
            dev_release(d_mvf[l]);
        }
    }
    dev_release(d_counters);
    dev_release(d_src_stats);
    dev_release(d_l0_pre);
    d_src_stats = d_l0_pre = nullptr;
    for (int i = 0; i < 2; i++) {
        if (d_intra_map[i]) {
            dev_release(d_intra_map[i]);
            d_intra_map[i] = nullptr;
        }
    }
    dev_release(d_ll);
    if (d_sym_pos) {
        dev_release(d_sym_pos);
        dev_release(d_sym_val);
    }
    HIPCHK(hipHostFree(h_frame));
    HIPCHK(hipHostFree(h_mvs));
    HIPCHK(hipHostFree(h_intra));
    HIPCHK(hipHostFree(h_counters));
    HIPCHK(hipHostFree(h_ll));
    HIPCHK(hipHostFree(h_small));
    if (h_pos) {
        HIPCHK(hipHostFree(h_pos));
        HIPCHK(hipHostFree(h_val));
    }
    if (stream) {
        HIPCHK(hipStreamDestroy(stream));
        stream = nullptr;
    }
    arena.destroy(); // (last: the releases above skipped everything that lives inside it)
}

MCParams CodecDev::mc_params(int temporal_mc, int lossless) const
{
    MCParams m;
    m.blk_w = blk_w;
    m.blk_h = blk_h;
    m.nbh = nbh;
    m.nbv = nbv;
    m.hshift = DSV_FORMAT_H_SHIFT(format);
    m.vshift = DSV_FORMAT_V_SHIFT(format);
    m.temporal_mc = temporal_mc;
    m.lossless = lossless;
    return m;
}

QuantCfg CodecDev::quant_cfg(int plane, int isP, int lossless, int do_psy, const DSV_MV *d_mvs) const
{
    QuantCfg c;
    c.w = cw[plane];
    c.h = ch[plane];
    c.plane = plane;
    c.isP = isP;
    c.lossless = lossless;
    c.do_psy = do_psy;
    c.hshift = DSV_FORMAT_H_SHIFT(format);
    c.vshift = DSV_FORMAT_V_SHIFT(format);
    c.blk_w = blk_w;
    c.blk_h = blk_h;
    c.nbh = nbh;
    c.nbv = nbv;
    c.bd = d_blockdata;
    c.mvs = d_mvs;
    return c;
}

} // namespace dsv2
