// decoder.cpp -- host frame controller of the decoder (C ABI section 3 of include/dsv2_hip.h).
//
// Restates the serial parsing of reference src/dsv_decoder.c (packet header :22, metadata :52,
// stability blocks :177, intra metadata :202, motion data :82, picture packet :394) and drives
// the device pipeline: symbol scatter + dequantisation -> inverse SBT -> intra filter, or
// motion-compensated prediction + reconstruction + in-loop filters -> border extension of
// reference pictures.  Entropy *parsing* is serial adaptive-state work and stays on the host.
#include <sched.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

#include "batch.h"
#include "codec.h"
#include "dec_parse.h"
#include "dec_parse_dev.h"

using namespace dsv2;
using namespace dsv2::decparse;

namespace {

struct DecImpl {
    CodecDev dev;
    bool ready = false;
    int cur = 0;
    bool have_ref = false;
    SideBufs side;                  // per-block flag bytes and vectors as parsed (dec_parse.h)
    std::vector<DSV_MV> &mvs = side.mvs;
    std::vector<uint8_t> &blockdata = side.blockdata;
    std::vector<uint32_t> pos;
    std::vector<int32_t> val;
    bool out420p = false; // deliver every picture as 4:2:0 (the CLI's -out420p, util.c:79-153, done by the GPU on the way out)
};

// ---- lockstep batch engine ------------------------------------------------------------------------
// One step decodes ONE packet on each of n decoder instances (dsv_dec is the n = 1 case):
//   A host   (one pool task per stream) packet header, metadata, per-block side information and the
//            serial entropy parse of the three planes into (position, value) symbol lists
//   B device every picture of the step in one set of launches over job tables: zero + scatter/dequantise
//            the coefficient planes, inverse transform, intra filter or motion-compensated
//            reconstruction + in-loop filters, border extension, picture to pinned host memory
//   C host   (one pool task per stream) output frame, reference bookkeeping
// Pictures of a step whose geometry differs from the first one are decoded in a second round.
struct DecJob {
    DSV_DECODER *d;
    DSV_BUF *buf;
    DSV_FRAME **out;
    DSV_FNUM *fn;
    int ret = DSV_DEC_OK;
    bool pic = false; // a picture that takes part in the device phase
    DecImpl *im = nullptr;
    int has_ref = 0, is_ref = 0, do_filter = 0, quant = 0, lossless = 0;
    DSV_FNUM fno = 0;
    int ok[3] = {0, 0, 0};
    int seg[3][4];
    int32_t LL[3] = {0, 0, 0};
    size_t sym_first[3] = {0, 0, 0}; // first symbol of each plane within the decoder's list
    size_t nsym = 0, stage_off = 0;
    bool dev_parse = false;          // the plane sections' symbols are parsed on the device (dec_parse_dev.hip)
    PlaneHead head[3];               // ... from here
    int cap[3] = {0, 0, 0};          // ... into lists of this many entries (min(header count, coefficients of the plane))
    size_t pkt_off = 0;              // ... out of the packet as staged at this offset of the round's stage block
    DSV_FRAME *of = nullptr; // output picture: a bordered frame on pinned memory the device writes directly
};

struct DecScratch { // held by ONE device round at a time (pool below); owns the stream the round's kernels run on
    TableArena tabs;
    hipStream_t main = nullptr;
    hipStream_t main_stream()
    {
        if (!main) {
            HIPCHK(hipStreamCreateWithFlags(&main, hipStreamNonBlocking));
        }
        return main;
    }
    uint8_t *h_stage = nullptr, *d_stage = nullptr;
    size_t stage_cap = 0;
    void ensure_stage(size_t bytes)
    {
        if (bytes <= stage_cap) {
            return;
        }
        if (stage_cap) {
            HIPCHK(hipHostFree(h_stage));
            HIPCHK(hipFree(d_stage));
        }
        bytes += bytes / 4;
        HIPCHK(hipHostMalloc((void **) &h_stage, bytes, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **) &d_stage, bytes));
        stage_cap = bytes;
    }
};
// process-wide pool, last released first (same reasoning as the encoder's ScratchPool): a lockstep group keeps getting the
// same scratch and stream from whatever thread it calls, and nothing is leaked when caller threads come and go
struct DecScratchPool {
    std::mutex mu;
    std::vector<DecScratch *> idle;
    bool primed = false;
    DecScratch *acquire(DecScratch *prefer) // (prefer: the one this thread used last, so that a group keeps its stream)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!primed) { // the first four scratches' streams are made once, in a row, and kept (see the encoder's ScratchPool)
            primed = true;
            DecScratch *first[4];
            for (int k = 0; k < 4; k++) {
                first[k] = new DecScratch();
                first[k]->main_stream();
            }
            for (int k = 3; k >= 0; k--) {
                idle.push_back(first[k]);
            }
        }
        if (idle.empty()) {
            return new DecScratch();
        }
        for (size_t i = 0; i < idle.size(); i++) {
            if (idle[i] == prefer) {
                idle.erase(idle.begin() + (ptrdiff_t) i);
                return prefer;
            }
        }
        DecScratch *sc = idle.back();
        idle.pop_back();
        return sc;
    }
    void release(DecScratch *sc)
    {
        std::lock_guard<std::mutex> lk(mu);
        idle.push_back(sc);
    }
};
DecScratchPool g_dec_scratch_pool;
thread_local DecScratch *t_last_dec_scratch = nullptr;
struct DecScratchLease {
    DecScratch *sc = g_dec_scratch_pool.acquire(t_last_dec_scratch);
    DecScratchLease() { t_last_dec_scratch = sc; }
    ~DecScratchLease() { g_dec_scratch_pool.release(sc); }
};

struct DecClock { // DSV2_TRACE=2: wall-clock split of a lockstep decode step, printed every 16 steps
    bool on = (trace_mode() & 2) != 0;
    double acc[6] = {0};
    int steps = 0;
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void lap(int i)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
    }
    void done(int n)
    {
        if (!on || ++steps % 16) return;
        fprintf(stderr, "[dec batch n=%d] ms/step: parse %.2f | pack %.2f | enqueue %.2f | wait %.2f | deliver %.2f\n", n, acc[0] / 16, acc[1] / 16,
                acc[2] / 16, acc[3] / 16, acc[4] / 16);
        for (double &a : acc) a = 0;
    }
};
thread_local DecClock t_dec_clock;

// Where a picture's plane sections are parsed: DSV2_DEC_DEVICE_PARSE = 0: on the host (one pool task per picture: ~2 ms of a
// core per 1080p P picture -- the fastest decoder while there are ~16 host cores per GPU to burn); 1: P pictures on the device, one
// wavefront per section (dec_parse_dev.hip) -- slower per step, the section being one dependency chain, but with ~1.4 host cores
// per GPU instead of ~15 -- and intra pictures (1 in a GOP, ten times the symbols: a 0.4 s chain on a wavefront) on the host; 2:
// everything on the device (tests).  Unset: by the host budget of this process -- the device parses when fewer than 12 cores are
// usable (DSV2_DEC_PARSE_AUTO_CORES), e.g. eight ranks of an 8-GPU node on a 64-core host, or a rank pinned to two cores.
static int dev_parse_mode()
{
    if (const char *e = getenv("DSV2_DEC_DEVICE_PARSE")) {
        return atoi(e);
    }
    cpu_set_t set;
    CPU_ZERO(&set);
    const int need = getenv("DSV2_DEC_PARSE_AUTO_CORES") ? atoi(getenv("DSV2_DEC_PARSE_AUTO_CORES")) : 12;
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) < need) {
        return 1;
    }
    return 0;
}
static std::atomic<int> g_dev_parse{dev_parse_mode()};

// phase A: everything dsv_dec does before it touches the device (dsv_decoder.c:393-503).  The bit parsing itself is
// dec_parse.h (device-free: fuzzed on the CPU under AddressSanitizer by tests/parser_fuzz.cpp); here: the private copy of
// the packet, the device instance of the stream's geometry, the pinned mirrors the device reads.
void dec_parse(DecJob &jb)
{
    DSV_DECODER *d = jb.d;
    DSV_BUF *buffer = jb.buf;
    *jb.fn = (DSV_FNUM) -1;
    // parse from a private copy with zeroed slack behind it, so that codes can be read through a 64-bit window
    static thread_local std::vector<uint8_t> copy;
    if (buffer->len > (1u << 28)) { // (bit positions are 32-bit)
        jb.ret = DSV_DEC_ERROR;
        return;
    }
    copy.assign(buffer->data, buffer->data + buffer->len);
    copy.resize((size_t) buffer->len + 64, 0);
    const uint8_t *pkt = copy.data();
    BitReader br{pkt, 0};
    br.wide = true;
    br.limit = (buffer->len + 8) * 8; // reads stop here; the copy is zero for 56 more bytes (see BitReader)
    PictureHead hd;
    const int rc = parse_head(br, d, hd);
    if (rc != kParsePicture) {
        jb.ret = rc;
        return;
    }
    const DSV_META *meta = &d->vidmeta;
    jb.has_ref = hd.has_ref;
    jb.is_ref = hd.is_ref;
    jb.fno = hd.fno;
    const int blk_w = hd.blk_w, blk_h = hd.blk_h;
    bind_device();
    DecImpl *im = (DecImpl *) d->ref;
    if (!im) {
        im = new DecImpl();
        d->ref = im;
    }
    jb.im = im;
    if (im->ready && (im->dev.w != meta->width || im->dev.h != meta->height || im->dev.format != meta->subsamp ||
                      im->dev.blk_w != blk_w || im->dev.blk_h != blk_h)) {
        im->dev.destroy(); // stream parameters changed: start over
        im->ready = false;
        im->have_ref = false;
    }
    if (!im->ready) {
        im->dev.init(meta->subsamp, meta->width, meta->height, blk_w, blk_h, 0, false);
        im->dev.scratch_uv[0].ensure((size_t) im->dev.cw[1] * im->dev.ch[1], sbt_ll_elems(im->dev.cw[1], im->dev.ch[1]));
        im->dev.scratch_uv[1].ensure((size_t) im->dev.cw[2] * im->dev.ch[2], sbt_ll_elems(im->dev.cw[2], im->dev.ch[2]));
        im->ready = true;
    }
    CodecDev &dv = im->dev;
    PictureBody body;
    const int dev_mode = g_dev_parse.load(std::memory_order_relaxed);
    jb.dev_parse = dev_mode >= 2 || (dev_mode == 1 && hd.has_ref);
    parse_body(br, pkt, hd.has_ref, dv.nbh, dv.nbv, dv.scan, im->side, im->pos, im->val, body, jb.dev_parse);
    jb.do_filter = body.do_filter;
    jb.quant = body.quant;
    jb.lossless = body.lossless;
    for (int c = 0; c < 3; c++) {
        jb.sym_first[c] = body.sym_first[c];
        jb.LL[c] = body.LL[c];
        jb.ok[c] = body.ok[c];
        for (int k = 0; k < 4; k++) {
            jb.seg[c][k] = body.seg[c][k];
        }
    }
    jb.nsym = body.nsym;
    if (jb.dev_parse) {
        // the symbol lists live in device memory; their sizes come from the (untrusted) header counts, bounded by the planes' coefficient counts
        size_t at = 0;
        for (int c = 0; c < 3; c++) {
            jb.head[c] = body.head[c];
            jb.sym_first[c] = at;
            jb.cap[c] = jb.ok[c] > 0 ? std::min(std::max(body.head[c].runs, 0), dv.scan[c].base[10]) : 0;
            at += (size_t) jb.cap[c];
        }
        dv.ensure_dev_syms(at);
    } else {
        // the device reads the symbols straight from pinned host memory (each is read exactly once)
        dv.ensure_host_syms(body.nsym);
        memcpy(dv.h_pos, im->pos.data(), body.nsym * sizeof(uint32_t));
        memcpy(dv.h_val, im->val.data(), body.nsym * sizeof(int32_t));
    }
    *jb.fn = jb.fno;
    if (jb.has_ref && !im->have_ref) {
        jb.ret = DSV_DEC_ERROR; /* reference frame not found (dsv_decoder.c:535) */
        return;
    }
    jb.of = mk_frame_pinned(im->out420p ? DSV_SUBSAMP_420 : meta->subsamp, meta->width, meta->height);
    jb.pic = true;
}

// phases B and C for the jobs listed in `ids` (pictures of one geometry)
void dec_device_round(DecJob *jobs, const std::vector<int> &ids)
{
    const int n = (int) ids.size();
    DecScratchLease lease; // (the round ends with its stream drained: nothing of the scratch is in use after it)
    DecScratch &sc = *lease.sc;
    CodecDev &dv0 = jobs[ids[0]].im->dev;
    hipStream_t bs = sc.main_stream(); // (a stream of the scratch pool, made once: see the encoder's ScratchPool)
    const size_t nb = dv0.nblocks();
    const size_t mv_bytes = nb * sizeof(DSV_MV), bd_bytes = (nb + 15) & ~(size_t) 15;
    sc.tabs.reserve((size_t) n * 8192 + 65536);

    // stage layout: per stream {motion field, block flags}
    // stage layout: per stream {motion field, block flags}, then -- for pictures whose sections the device parses -- the packets,
    // each 16-byte aligned with 64 zero bytes behind it (the parser reads a 2 KB window: the block ends with that much slack)
    size_t total = 0;
    int n_dev = 0;
    for (int i = 0; i < n; i++) {
        DecJob &jb = jobs[ids[(size_t) i]];
        jb.stage_off = total;
        total += mv_bytes + bd_bytes;
    }
    for (int i = 0; i < n; i++) {
        DecJob &jb = jobs[ids[(size_t) i]];
        if (jb.dev_parse) {
            jb.pkt_off = total;
            total += ((size_t) jb.buf->len + 64 + 15) & ~(size_t) 15;
            n_dev++;
        }
    }
    const size_t stage_used = total;
    sc.ensure_stage(total + (n_dev ? 4096 : 0));
    parallel_for(n, [&](int i) {
        DecJob &jb = jobs[ids[(size_t) i]];
        DecImpl *im = jb.im;
        uint8_t *h = sc.h_stage + jb.stage_off;
        if (jb.has_ref) {
            memcpy(h, im->mvs.data(), mv_bytes);
        }
        memcpy(h + mv_bytes, im->blockdata.data(), nb);
        if (jb.dev_parse) {
            uint8_t *p = sc.h_stage + jb.pkt_off;
            memcpy(p, jb.buf->data, jb.buf->len);
            memset(p + jb.buf->len, 0, (((size_t) jb.buf->len + 64 + 15) & ~(size_t) 15) - jb.buf->len);
        }
    });

    t_dec_clock.lap(1);
    // order by (frame type, lossless): the kernels are specialised on those
    std::vector<int> order(ids);
    auto cls = [&](int k) { return jobs[k].has_ref * 2 + jobs[k].lossless; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cls(a) > cls(b); });

    const CopyJob *d_zero, *d_out;
    CopyJob *h_zero = sc.tabs.take<CopyJob>(3 * (size_t) n, &d_zero), *h_out = sc.tabs.take<CopyJob>((size_t) n, &d_out);
    const DequantJob *d_dq[3];
    DequantJob *h_dq[3];
    const PlaneJob *d_pj[3];
    PlaneJob *h_pj[3];
    for (int c = 0; c < 3; c++) {
        h_dq[c] = sc.tabs.take<DequantJob>((size_t) n, &d_dq[c]);
        h_pj[c] = sc.tabs.take<PlaneJob>((size_t) n, &d_pj[c]);
    }
    const McJob *d_mc_pred, *d_mc_filt, *d_mc_intra;
    McJob *h_mc_pred = sc.tabs.take<McJob>((size_t) n, &d_mc_pred), *h_mc_filt = sc.tabs.take<McJob>((size_t) n, &d_mc_filt),
          *h_mc_intra = sc.tabs.take<McJob>((size_t) n, &d_mc_intra);
    const PlanePair *d_icopy[3];
    PlanePair *h_icopy[3];
    const DPlane *d_ext[3];
    DPlane *h_ext[3];
    for (int c = 0; c < 3; c++) {
        h_icopy[c] = sc.tabs.take<PlanePair>((size_t) n, &d_icopy[c]);
        h_ext[c] = sc.tabs.take<DPlane>((size_t) n, &d_ext[c]);
    }
    const CopyJob *d_zfail;
    CopyJob *h_zfail = sc.tabs.take<CopyJob>(3 * (size_t) n, &d_zfail);
    const To420Job *d_to420;
    To420Job *h_to420 = sc.tabs.take<To420Job>(3 * (size_t) n, &d_to420);
    // device-parsed sections: one job each; a damaged one raises its flag and the plane's residual is zeroed behind the inverse transform
    const DecParseJob *d_parse;
    DecParseJob *h_parse = sc.tabs.take<DecParseJob>(3 * (size_t) n, &d_parse);
    const CopyJob *d_zcond;
    CopyJob *h_zcond = sc.tabs.take<CopyJob>(3 * (size_t) n, &d_zcond);
    const int *d_fail;
    int *h_fail = sc.tabs.take<int>(3 * (size_t) n, &d_fail);
    int n_parse = 0;
    int nP = 0, nI = 0, nIf = 0, n_ext = 0, n_zfail = 0, n_out = 0, n_to420 = 0;
    size_t max_plane_bytes = 0;
    bool any_filter = false;
    size_t max_coef_bytes = 0;
    struct Slice {
        int first, count, isP, lossless, max_seg[3][4];
    };
    std::vector<Slice> slices;
    for (int i = 0; i < n; i++) {
        DecJob &jb = jobs[order[(size_t) i]];
        DecImpl *im = jb.im;
        CodecDev &dv = im->dev;
        PicSet &cur = dv.pics[im->cur], &ref = dv.pics[im->cur ^ 1];
        DFrame &resid = dv.pred; // the decoder's residual picture
        const uint8_t *d_slot = sc.d_stage + jb.stage_off;
        const DSV_MV *d_mvs = (const DSV_MV *) d_slot;
        const uint8_t *d_bd = d_slot + mv_bytes;
        const uint32_t *d_pos = dv.h_pos; // pinned host memory, read in place
        const int32_t *d_val = dv.h_val;
        if (slices.empty() || slices.back().isP != jb.has_ref || slices.back().lossless != jb.lossless) {
            Slice sl = {};
            sl.first = i;
            sl.isP = jb.has_ref;
            sl.lossless = jb.lossless;
            slices.push_back(sl);
        }
        Slice &sl = slices.back();
        sl.count++;
        MCParams mc = dv.mc_params((int) (jb.fno % 2), jb.lossless);
        for (int c = 0; c < 3; c++) {
            size_t cbytes = (size_t) dv.cw[c] * dv.ch[c] * sizeof(int32_t);
            h_zero[3 * i + c] = CopyJob{nullptr, dv.coefs[c], cbytes};
            max_coef_bytes = cbytes > max_coef_bytes ? cbytes : max_coef_bytes;
            DequantJob &dq = h_dq[c][i];
            dq.coefs = dv.coefs[c];
            if (jb.dev_parse) {
                dq.pos = dv.d_sym_pos + jb.sym_first[c];
                dq.val = dv.d_sym_val + jb.sym_first[c];
                for (int k = 0; k < 4; k++) { // (the counts are written into the device copy of this record by the parse kernel)
                    dq.seg[k] = 0;
                    sl.max_seg[c][k] = jb.cap[c] > sl.max_seg[c][k] ? jb.cap[c] : sl.max_seg[c][k];
                }
                if (jb.ok[c] > 0) {
                    DecParseJob &pj = h_parse[n_parse];
                    pj.pkt = sc.d_stage + jb.pkt_off;
                    pj.data_bitpos = jb.head[c].data_bitpos;
                    pj.limit_bits = (jb.buf->len + 8) * 8;
                    pj.end_byte = jb.head[c].end_byte;
                    pj.runs = jb.head[c].runs;
                    pj.cap = jb.cap[c];
                    pj.chroma = c != 0;
                    pj.pos = dv.d_sym_pos + jb.sym_first[c];
                    pj.val = dv.d_sym_val + jb.sym_first[c];
                    pj.seg_out = (int *) ((uint8_t *) const_cast<DequantJob *>(d_dq[c]) + (size_t) i * sizeof(DequantJob) + offsetof(DequantJob, seg));
                    pj.fail = const_cast<int *>(d_fail) + n_parse;
                    h_fail[n_parse] = 0;
                    h_zcond[n_parse] = CopyJob{nullptr, resid.alloc + resid.plane_off[c], resid.plane_len[c]};
                    max_plane_bytes = resid.plane_len[c] > max_plane_bytes ? resid.plane_len[c] : max_plane_bytes;
                    n_parse++;
                }
            } else {
                dq.pos = d_pos + jb.sym_first[c];
                dq.val = d_val + jb.sym_first[c];
                for (int k = 0; k < 4; k++) {
                    dq.seg[k] = jb.seg[c][k];
                    sl.max_seg[c][k] = jb.seg[c][k] > sl.max_seg[c][k] ? jb.seg[c][k] : sl.max_seg[c][k];
                }
            }
            dq.bd = d_bd;
            dq.LL = jb.LL[c];
            dequant_steps(&dq, dv.quant_cfg(c, jb.has_ref, jb.lossless, 0, nullptr), jb.quant);
            PlaneJob &pj = h_pj[c][i];
            pj = PlaneJob{};
            pj.pic = resid.p[c];
            pj.coefs = dv.coefs[c];
            for (int t = 0; t < 3; t++) {
                pj.t[t] = c ? dv.scratch_uv[c - 1].t[t] : dv.scratch.t[t];
            }
            pj.bd = d_bd;
            pj.q = jb.quant;
            if (jb.ok[c] <= 0) { // "decoding error in plane": its residual plane stays zero (dsv_decoder.c:516-523)
                h_zfail[n_zfail++] = CopyJob{nullptr, resid.alloc + resid.plane_off[c], resid.plane_len[c]};
                max_plane_bytes = resid.plane_len[c] > max_plane_bytes ? resid.plane_len[c] : max_plane_bytes;
            }
        }
        McJob mj;
        mj.mvs = d_mvs;
        mj.bd = d_bd;
        mj.p = mc;
        if (jb.has_ref) {
            mj.f = make_filter_params(mc, jb.quant, jb.do_filter, jb.d->vidmeta.inter_sharpen);
            for (int c = 0; c < 3; c++) {
                mj.ref.p[c] = ref.recon.p[c];
                mj.pred.p[c] = cur.recon.p[c];
                mj.res.p[c] = resid.p[c];
            }
            h_mc_pred[nP] = mj;
            for (int c = 0; c < 3; c++) {
                mj.res.p[c] = cur.recon.p[c];
            }
            h_mc_filt[nP] = mj;
            nP++;
            any_filter = any_filter || !jb.lossless;
        } else {
            mj.f = make_filter_params(mc, jb.quant, 1, 0);
            for (int c = 0; c < 3; c++) {
                mj.ref.p[c] = mj.pred.p[c] = mj.res.p[c] = resid.p[c];
                h_icopy[c][nI] = PlanePair{resid.p[c], cur.recon.p[c]};
            }
            nI++;
            if (jb.do_filter && !jb.lossless) { // dsv_intra_filter is a no-op for lossless pictures (bmc.c:398)
                h_mc_intra[nIf++] = mj;
            }
        }
        if (jb.is_ref || !jb.has_ref) {
            for (int c = 0; c < 3; c++) {
                h_ext[c][n_ext] = cur.recon.p[c];
            }
            n_ext++;
        }
        if (im->out420p && dv.format != DSV_SUBSAMP_420) {
            // converted on the way out (dsv_main.c:1030-1048): luma copied, chroma through the reference's pair averages
            const int hs = DSV_FORMAT_H_SHIFT(dv.format), vs = DSV_FORMAT_V_SHIFT(dv.format);
            const int mode = (hs == 0 && vs == 0) ? 1 : (hs == 1 && vs == 0) ? 2 : (hs == 2 && vs == 0) ? 3 : 4;
            for (int c = 0; c < 3; c++) {
                const DSV_PLANE &op = jb.of->planes[c];
                h_to420[n_to420++] = To420Job{cur.recon.p[c], DPlane{op.data, op.stride, op.w, op.h}, c ? mode : 0};
            }
        } else {
            h_out[n_out++] = CopyJob{cur.recon.alloc, jb.of->alloc, cur.recon.bytes};
        }
    }

    // (stage spans for bench.py's decode roofline: HIP events on this step's stream when dsv2hip_prof_enable(1) is on -- the
    // encoder's stage names: QUANT = zero + scatter + dequantise, INV_SBT, RECON_FILTER = intra filter / motion-compensated
    // reconstruction + in-loop filters, EXTEND = borders + the picture's way into the caller's frame)
    static thread_local StageProf prof;
    HIPCHK(hipMemcpyAsync(sc.d_stage, sc.h_stage, stage_used, hipMemcpyHostToDevice, bs));
    sc.tabs.upload(bs);
    prof.begin(bs, ST_QUANT);
    if (n_parse) {
        DecScanBases sb_l, sb_c;
        for (int k = 0; k < 11; k++) {
            sb_l.base[k] = dv0.scan[0].base[k];
            sb_c.base[k] = dv0.scan[1].base[k];
        }
        dec_parse_planes(bs, d_parse, n_parse, sb_l, sb_c);
    }
    zero_linear_batch(bs, d_zero, 3 * n, max_coef_bytes);
    for (const Slice &sl : slices) {
        for (int c = 0; c < 3; c++) {
            dequant_jobs(bs, d_dq[c] + sl.first, sl.count, sl.max_seg[c], dv0.quant_cfg(c, sl.isP, sl.lossless, 0, nullptr));
        }
    }
    prof.end(bs, ST_QUANT, n);
    prof.begin(bs, ST_INV_SBT);
    for (const Slice &sl : slices) {
        for (int c = 0; c < 3; c++) {
            sbt_inverse_jobs(bs, d_pj[c] + sl.first, sl.count, dv0.cw[c], dv0.ch[c], c, sl.isP, sl.lossless, dv0.nbh, dv0.nbv);
        }
    }
    prof.end(bs, ST_INV_SBT, n);
    prof.begin(bs, ST_RECON_FILTER);
    zero_linear_batch(bs, d_zfail, n_zfail, max_plane_bytes);
    zero_linear_if_batch(bs, d_zcond, d_fail, n_parse, max_plane_bytes);
    intra_filter_batch(bs, d_mc_intra, nIf, dv0.w, dv0.h);
    mc_add_pred_batch(bs, d_mc_pred, d_mc_filt, nP, dv0.nbh, dv0.nbv, any_filter, dv0.w, dv0.h, dv0.blk_w, dv0.blk_h,
                      DSV_FORMAT_H_SHIFT(dv0.format) == 1 && DSV_FORMAT_V_SHIFT(dv0.format) == 1);
    prof.end(bs, ST_RECON_FILTER, n);
    prof.begin(bs, ST_EXTEND);
    for (int c = 0; c < 3; c++) {
        const DPlane &pl = dv0.pics[0].recon.p[c];
        copy_planes_batch(bs, d_icopy[c], nI, pl.w, pl.h);
        extend_planes(bs, d_ext[c], n_ext, pl.w, pl.h);
    }
    copy_linear_batch(bs, d_out, n_out, dv0.pics[0].recon.bytes);
    to420_batch(bs, d_to420, n_to420, dv0.w, dv0.h);
    prof.end(bs, ST_EXTEND, n);
    t_dec_clock.lap(2);
    stream_wait(bs);
    prof.collect();
    t_dec_clock.lap(3);

    // phase C: the pictures are already in their output frames
    for (int i = 0; i < n; i++) {
        DecJob &jb = jobs[order[(size_t) i]];
        DecImpl *im = jb.im;
        if (jb.is_ref) {
            im->cur ^= 1;
            im->have_ref = true;
        }
        *jb.out = jb.of;
        jb.ret = DSV_DEC_OK;
    }
}

void dec_batch(DecJob *jobs, int n)
{
    {
        constexpr int fine_max = 1;
        set_wait_fine(n <= fine_max); // (dev.cpp: a few streams are a latency chain, their waits poll finely)
    }
    bind_device();
    t_dec_clock.start();
    parallel_for(n, [&](int k) { dec_parse(jobs[k]); });
    t_dec_clock.lap(0);
    std::vector<int> todo;
    for (int k = 0; k < n; k++) {
        if (jobs[k].pic) {
            todo.push_back(k);
        }
    }
    while (!todo.empty()) { // one round per picture geometry present in the step
        const CodecDev &g = jobs[todo[0]].im->dev;
        std::vector<int> ids, rest;
        for (int k : todo) {
            const CodecDev &dv = jobs[k].im->dev;
            bool same = dv.w == g.w && dv.h == g.h && dv.format == g.format && dv.blk_w == g.blk_w && dv.blk_h == g.blk_h;
            (same ? ids : rest).push_back(k);
        }
        dec_device_round(jobs, ids);
        todo.swap(rest);
    }
    t_dec_clock.lap(4);
    t_dec_clock.done(n);
    for (int k = 0; k < n; k++) {
        dsv_buf_free(jobs[k].buf); /* the decoder frees its input on every path (dsv_decoder.c:414,432,438,581) */
    }
}

Coalescer<DecJob> g_dec_queue; // dsv_dec callers share lockstep steps (batch.h)

} // namespace

extern "C" {

void dsv_dec_free(DSV_DECODER *d)
{
    g_dec_queue.forget(d);
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
    DecJob jb;
    jb.d = d;
    jb.buf = buffer;
    jb.out = out;
    jb.fn = fn;
    if (!Coalescer<DecJob>::enabled()) {
        dec_batch(&jb, 1);
        return jb.ret;
    }
    // concurrent callers (a decoder per thread) share one lockstep step; dec_batch itself sorts mixed geometries into rounds,
    // so every caller carries the same key
    g_dec_queue.submit(jb, 0, d, dec_batch);
    if (jb.ret == DSV_DEC_EOS || jb.ret == DSV_DEC_ERROR) { // (a caller that gives up after an error must not stay expected; one that goes on is seen again)
        g_dec_queue.forget(d);
    }
    return jb.ret;
}

/* what the submit queue of dsv_dec did so far (see dsv2hip_enc_queue_stats) */
void dsv2hip_dec_queue_stats(unsigned long long *out4, int reset)
{
    Coalescer<DecJob>::Stats st = g_dec_queue.stats();
    if (out4) {
        out4[0] = st.calls;
        out4[1] = st.steps;
        out4[2] = st.largest;
        out4[3] = st.waited_us;
    }
    if (reset) {
        g_dec_queue.reset_stats();
    }
}

/* every picture this decoder returns from now on is converted to 4:2:0 by the GPU while it is written to the output
 * frame -- what the reference CLI's -out420p does on the host afterwards (dsv_main.c:1030-1048, util.c:79-153) */
int dsv2hip_dec_set_out420p(DSV_DECODER *d, int on)
{
    if (!d) {
        return -1;
    }
    if (!d->ref) {
        d->ref = new DecImpl();
    }
    ((DecImpl *) d->ref)->out420p = on != 0;
    return 0;
}

int dsv2hip_dec_parse_mode(void) { return g_dev_parse.load(); }
int dsv2hip_dec_set_parse_mode(int mode)
{
    if (mode < 0) {
        mode = dev_parse_mode(); // back to the environment / the host budget
    }
    g_dev_parse.store(mode > 2 ? 2 : mode);
    return g_dev_parse.load();
}

// lockstep decode: packet bufs[k] on decoder decs[k]; ret[k], out[k], fn[k] are what dsv_dec would return
int dsv2hip_dec_batch(int n, DSV_DECODER **decs, DSV_BUF *bufs, DSV_FRAME **out, DSV_FNUM *fn, int *ret)
{
    if (n <= 0 || !decs || !bufs || !out || !fn || !ret) {
        return 0;
    }
    std::vector<DecJob> jobs((size_t) n);
    for (int k = 0; k < n; k++) {
        jobs[(size_t) k].d = decs[k];
        jobs[(size_t) k].buf = &bufs[k];
        jobs[(size_t) k].out = &out[k];
        jobs[(size_t) k].fn = &fn[k];
        out[k] = NULL;
    }
    dec_batch(jobs.data(), n);
    for (int k = 0; k < n; k++) {
        ret[k] = jobs[(size_t) k].ret;
    }
    return n;
}

} // extern "C"
