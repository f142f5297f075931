// encoder.cpp -- host frame controller of the encoder (C ABI section 2 of include/dsv2_hip.h).
//
// Restates the serial control logic of reference src/dsv_encoder.c around the device pipeline:
// GOP / I-P decision (encode_one_frame :1185), rate control (quality2quant :253, qual_to_qp :91),
// scene-change detection (:546, avg_motion :130, scene_complexity :180), auto filter decision
// (:519), per-block metadata coders (encode_stable_blocks :798, encode_motion :693,
// encode_intra_meta :887, gather_stats :993), packet framing (:934-990, set_link_offsets :471)
// and statistics (dsv_enc :1431).  Its outputs -- quantiser, flags, I/P decisions -- parameterise
// every kernel, so it is restated exactly; the pixel work itself never runs on the host:
//   upload -> extend / pyramid -> [P: HME] -> (host decisions) -> [P: predict+subtract]
//   -> fwd SBT -> quantise + compact -> (host entropy packing) -> inv SBT -> [I: intra filter |
//   P: reconstruct + in-loop filters] -> extend.
#include <limits.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "batch.h"
#include "codec.h"
#include "sideinfo.h"

using namespace dsv2;

namespace {

struct EncImpl {
    CodecDev dev;
    bool dead = false; // a step of this instance failed (search time-out, GPU unusable): every later call returns "no packets"
    std::vector<uint8_t> pkt; // picture packet under construction (all zero between frames)
    bool ready = false;
    int cur = 0;          // picture set receiving the current frame
    bool have_ref = false; // pics[cur ^ 1] holds a usable reference
    std::vector<DSV_MV> mvs; // host copy of the current motion field
    std::vector<DSV_MV> intramv;
    // the running intra map of scene_change_detection lives on the device (CodecDev::d_intra_map, hme.h BlockStatsJob):
    // which of the two buffers holds the committed map, and whether there is one (none right after an intra picture)
    int map_cur = 0;
    bool map_valid = false;
    // pictures handed over in HOST memory (dsv2hip_enc_batch_host): two packed planar staging pictures in HBM;
    // the one not being ingested receives the next step's upload on the group's copy stream meanwhile
    uint8_t *d_stage[2] = {nullptr, nullptr};
    int stage_cur = 0;
    const void *staged_src = nullptr; // host picture whose upload into d_stage[stage_cur ^ 1] is in flight / done
    hipEvent_t staged_ev = nullptr;   // ... and the event (of the uploading group) that marks its completion
    uint8_t *h_pack = nullptr;        // dsv_enc: the caller's DSV_FRAME packed (Y, U, V rows without padding) into pinned host memory
    size_t h_pack_bytes = 0;
    std::vector<uint8_t> big_bytes;   // plane sections of a picture too large for the pinned mirror of the GPU entropy coder
    bool input_uyvy = false;          // packed pictures arrive as interleaved UYVY rows (de-interleaved by the ingest kernel)
};

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline int sar(int v, int s) { return v < 0 ? ~(~v >> s) : v >> s; }
inline int sar_r(int v, int s) { return sar(v + (1 << (s - 1)), s); }
inline int sqr(int x) { return x * x; }
inline bool mvflag(const DSV_MV &m, int bit) { return (m.flags >> bit) & 1; }

// ---- motion-vector helpers shared with the bitstream (dsv.c:324-447) ----------------------
int mv_pred1(int left, int top, int topleft)
{
    int dif = left + top - topleft;
    return abs(dif - left) < abs(dif - top) ? left : top;
}

void movec_pred(const DSV_MV *v, int nbh, int x, int y, int *px, int *py)
{
    int vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0};
    if (x > 0) {
        const DSV_MV *m = &v[y * nbh + x - 1];
        vx[0] = m->u.mv.x;
        vy[0] = m->u.mv.y;
    }
    if (y > 0) {
        const DSV_MV *m = &v[(y - 1) * nbh + x];
        vx[1] = m->u.mv.x;
        vy[1] = m->u.mv.y;
    }
    if (x > 0 && y > 0) {
        const DSV_MV *m = &v[(y - 1) * nbh + x - 1];
        vx[2] = m->u.mv.x;
        vy[2] = m->u.mv.y;
    }
    *px = mv_pred1(vx[0], vx[1], vx[2]);
    *py = mv_pred1(vy[0], vy[1], vy[2]);
}

int seg_bits(int v)
{
    if (v < 0) {
        v = -v;
    }
    v++;
    int nb = 31 - __builtin_clz((unsigned) v);
    return nb * 2 + 2;
}

int mv_cost(const DSV_MV *v, const DSV_PARAMS *p, int i, int j, int mx, int my, int q, int sqr_)
{
    int px, py;
    movec_pred(v, p->nblocks_h, i, j, &px, &py);
    int bits = seg_bits(mx - px) + seg_bits(my - py);
    int b2sr = (256 * (q * q >> DSV_MAX_QP_BITS) * p->blk_w * p->blk_h) / (p->vidmeta->width * p->vidmeta->height);
    bits += bits * b2sr >> 7;
    return sqr_ ? bits * bits : bits;
}

void neighbordif2(const DSV_MV *v, int nbh, int x, int y, int *dx, int *dy)
{
    const DSV_MV *c = &v[x + y * nbh];
    int cx = c->u.mv.x, cy = c->u.mv.y, lx = cx, ly = cy, tx = cx, ty = cy;
    if (abs(cx) < 2 && abs(cy) < 2) {
        *dx = *dy = 0;
        return;
    }
    if (x > 0) {
        const DSV_MV *m = c - 1;
        if (m->u.all && !mvflag(*m, DSV_MV_BIT_SKIP)) {
            lx = m->u.mv.x;
            ly = m->u.mv.y;
        }
    }
    if (y > 0) {
        const DSV_MV *m = c - nbh;
        if (m->u.all && !mvflag(*m, DSV_MV_BIT_SKIP)) {
            tx = m->u.mv.x;
            ty = m->u.mv.y;
        }
    }
    *dx = abs(lx - cx) + abs(ly - cy);
    *dy = abs(tx - cx) + abs(ty - cy);
}

int neighbordif(const DSV_MV *v, int nbh, int x, int y)
{
    int a, b;
    neighbordif2(v, nbh, x, y, &a, &b);
    return (a + b) / 3;
}

// ---- rate control ---------------------------------------------------------------------------
int sample_point(int v) // dsv_encoder.c:72
{
    const int unit = 10 * DSV_RC_QUAL_SCALE;
    v = 100 * DSV_RC_QUAL_SCALE - v;
    int whole = v / unit, frac = v % unit;
    int lo = 1 << whole, hi = 1 << (whole + 1);
    int qp = ((unit - frac) * lo + frac * hi) / unit - 1;
    return clampi(qp * 4, 0, DSV_MAX_QP);
}

int qual_to_qp(int v) // dsv_encoder.c:90
{
    int d_hi = 100 * DSV_RC_QUAL_SCALE - v;
    if (d_hi < 60) {
        return d_hi + 16;
    }
    v *= 2;
    int actv = v / 3, frac = v % 3;
    return (sample_point(actv) * (3 - frac) + frac * sample_point(actv + 1)) / 3;
}

#define RC_PCT(p) ((p) * DSV_RC_QUAL_SCALE)

// mean luma of the coarsest source pyramid level, row means first (dsv_encoder.c:108)
unsigned luma_avg_rows(const uint8_t *pix, int w, int h)
{
    unsigned avg = 0;
    for (int j = 0; j < h; j++) {
        unsigned r = 0;
        for (int i = 0; i < w; i++) {
            r += pix[j * w + i];
        }
        avg += r / (unsigned) w;
    }
    return avg / (unsigned) h;
}

struct FrameCtl { // per-frame control block (the reference's DSV_ENCDATA minus the pictures)
    DSV_FNUM fnum;
    DSV_PARAMS params;
    int quant;
    unsigned coarse_luma_avg;
};

void quality2quant(DSV_ENCODER *enc, FrameCtl *d, DSV_FNUM prev_I, int forced_intra) // dsv_encoder.c:252
{
    int q = (int) enc->rc_qual;
    const DSV_META *vf = d->params.vidmeta;
    bool isP = d->params.has_ref;

    if (enc->rc_mode == DSV_RATE_CONTROL_CRF) {
        int bound = RC_PCT(25);
        int minq = isP ? enc->min_quality : enc->min_I_frame_quality;
        int maxq = enc->max_quality;
        int anchor = clampi(enc->quality, minq, maxq);
        int fps = (vf->fps_num << 5) / vf->fps_den;
        int gop = clampi(enc->gop, 1, (10 * fps >> 5));
        int sqst = sqr(enc->motion_static) / 75;
        if (sqst < enc->motion_static) {
            sqst = enc->motion_static;
        }
        int plex;
        if (!isP) {
            plex = (forced_intra ? 2 : 1) * sqst - enc->motion_chaos;
        } else {
            int m = enc->avg_err < enc->motion_chaos / 3 ? enc->avg_err : enc->motion_chaos / 3;
            plex = sqr(m) / 2 + sqst - 3 * enc->motion_chaos;
        }
        plex = (plex * gop * vf->fps_den) / (vf->fps_num << 4);
        plex = clampi(plex, -bound / 4, bound / 4);
        int clamped_avg = enc->rf_avg > enc->quality ? enc->rf_avg : enc->quality;
        int target = (anchor + 3 * clamped_avg + 2) >> 2;
        target = clampi(target, enc->quality - bound, enc->quality + bound);
        if (enc->do_dark_intra_boost) {
            unsigned la = d->coarse_luma_avg;
            if (la < 80) {
                int step = (int) (80 - la) / 5;
                step = clampi(step, 5, 16) - 5;
                plex += sqr(step) / 4;
            }
        }
        q = target + plex;
        if (!isP) {
            int back = (DSV_RC_QUAL_MAX - q) / (1 + enc->motion_chaos / 4);
            q += (back * gop * vf->fps_den) / (vf->fps_num << 4);
        }
        q = clampi(q, enc->quality - bound, enc->quality + bound);
        q = clampi(q, minq, maxq);
        enc->rc_qual = (unsigned) (q > 0 ? q : 0);
    } else if (enc->rc_mode == DSV_RATE_CONTROL_ABR) {
        int fps = (vf->fps_num << 5) / vf->fps_den;
        if (fps == 0) {
            fps = 1;
        }
        if (enc->prev_complexity < 0) {
            enc->prev_complexity = enc->curr_complexity;
        }
        int target_rf = (int) (((enc->bitrate << 5) / (unsigned) fps) >> 3);
        int rf = enc->rf_avg ? enc->rf_avg : target_rf;
        int dir = (rf - target_rf) > 0 ? -1 : 1;
        int delta;
        enc->min_q_step = clampi(enc->min_q_step, 1, DSV_RC_QUAL_MAX);
        enc->max_q_step = clampi(enc->max_q_step, 1, DSV_RC_QUAL_MAX);
        if (!isP) {
            unsigned dif = (unsigned) abs(rf - target_rf);
            if (dif > 32768) {
                dif = 32768;
            }
            delta = (int) ((dif * dif) / (unsigned) ((dir > 0 ? 32 : 64) * target_rf));
            if (delta > RC_PCT(12)) {
                delta -= RC_PCT(8);
            } else if (delta > RC_PCT(8)) {
                delta -= RC_PCT(4);
            } else if (delta > RC_PCT(4)) {
                delta -= RC_PCT(2);
            }
            delta = delta < RC_PCT(25) ? delta : RC_PCT(25);
            q = (q > enc->avg_P_frame_q ? q : enc->avg_P_frame_q) + dir * delta;
            if (enc->prev_complexity < 15) {
                q += RC_PCT(2);
            } else if (enc->prev_complexity < 30) {
                q += RC_PCT(1);
            } else if (enc->prev_complexity > 40) {
                q -= RC_PCT(1);
            } else if (enc->prev_complexity > 60) {
                q -= RC_PCT(2);
            }
            enc->prev_I_frame_quality = q;
        } else {
            delta = (abs(rf - target_rf) * RC_PCT(100)) / target_rf;
            if (dir < 0 && delta < enc->min_q_step) {
                delta = 0;
            }
            int cap = enc->max_q_step * (dir > 0 ? 1 : 8);
            delta = delta < cap ? delta : cap;
            q += dir * delta;
        }
        int low_p = clampi(enc->avg_P_frame_q - RC_PCT(4), enc->min_quality, enc->max_quality);
        int minq = isP ? low_p : enc->min_I_frame_quality;
        if (enc->do_dark_intra_boost && !isP) {
            unsigned la = d->coarse_luma_avg;
            if (la < 80) {
                q += clampi((int) (80 - la) / 5, 5, 16);
            }
        }
        q = clampi(q, minq, enc->max_quality);
        q = clampi(q, 0, DSV_RC_QUAL_MAX);
        enc->rc_qual = (unsigned) q;
        enc->prev_complexity = enc->curr_complexity;
        if (enc->rc_pergop) {
            q = clampi(enc->prev_I_frame_quality, enc->min_quality, enc->max_quality);
        } else if (d->fnum > 0 && isP) {
            int step = RC_PCT(8), closeness;
            int gop = clampi(enc->gop, 1, 60);
            int dist = abs((int) d->fnum - (int) prev_I);
            if (dist >= enc->gop / 2) {
                dist = abs((int) d->fnum - ((int) prev_I + gop / 2));
                closeness = step - (step * dist / (gop / 2 > 1 ? gop / 2 : 1));
            } else {
                closeness = step * dist / (gop / 2 > 1 ? gop / 2 : 1);
            }
            q += clampi(closeness, 0, step) / 2;
            q -= clampi((enc->avg_err * enc->avg_err) >> 1, 0, RC_PCT(16));
            q = clampi(q, low_p, enc->max_quality);
            if (enc->gop <= (2 * fps >> 5)) {
                if (enc->prev_I_frame_quality < q) {
                    q = enc->prev_I_frame_quality;
                } else {
                    q = (3 * q + enc->prev_I_frame_quality) >> 2;
                }
                q = clampi(q, enc->min_quality, enc->max_quality);
            }
        }
    } else {
        q = enc->quality;
        enc->rc_qual = (unsigned) q;
    }
    d->quant = d->params.lossless ? 1 : qual_to_qp(q);
    enc->prev_quant = d->quant;
}

// ---- scene statistics (dsv_encoder.c:129-250) ---------------------------------------------------
// The per-block sums these start from are taken on the device right behind the search (k_block_stats_b, hme.h
// BlockStatsJob: `bs` = its BS_* words); what is left here is the scalar arithmetic on them.
int avg_motion(DSV_ENCODER *enc, const DSV_PARAMS *p, const int *bs)
{
    int nblk = p->nblocks_h * p->nblocks_v;
    int ax = (abs(bs[BS_AX]) + abs(bs[BS_AY])) / (nblk * 2);
    if (ax < 1) {
        ax = 1;
    }
    enc->curr_avgmot = ax;
    enc->motion_static = bs[BS_STAT] * 100 / nblk;
    int chaos = bs[BS_CHAOS] * 100 / nblk;
    if (enc->prev_chaos < 0) {
        enc->motion_chaos = chaos;
        enc->prev_chaos = chaos;
    } else {
        enc->prev_chaos = (enc->prev_chaos + enc->motion_chaos) / 2;
        enc->motion_chaos = chaos;
    }
    return ax;
}

int mv_cost_b2sr(const DSV_PARAMS *p, int q) // the bits-to-SSE ratio of dsv_mv_cost (dsv.c:357)
{
    return (256 * (q * q >> DSV_MAX_QP_BITS) * p->blk_w * p->blk_h) / (p->vidmeta->width * p->vidmeta->height);
}

int scene_complexity(DSV_ENCODER *enc, const DSV_PARAMS *p, const int *bs)
{
    int complexity = bs[BS_COMPLEXITY], maxpot;
    int nblk = p->nblocks_h * p->nblocks_v;
    if (enc->rc_mode == DSV_RATE_CONTROL_ABR) {
        // mv_cost of the vector (64, 64) at block (0, 0): the predictor there is zero
        int bits = seg_bits(64) + seg_bits(64);
        bits += bits * mv_cost_b2sr(p, enc->prev_quant) >> 7;
        maxpot = bits + 12 + 64;
        maxpot = (maxpot * nblk + 1) >> 1;
    } else if (enc->rc_mode == DSV_RATE_CONTROL_CRF) {
        maxpot = 70 * nblk;
    } else {
        return 0;
    }
    return complexity <= 0 ? 0 : complexity * 100 / maxpot;
}

void compute_auto_filter(DSV_ENCODER *enc, const FrameCtl *d) // dsv_encoder.c:518
{
    const DSV_PARAMS *p = &d->params;
    int chaos = enc->motion_chaos;
    int psy = spatial_psy_factor(p->blk_w, p->blk_h, p->nblocks_h, p->nblocks_v, -1);
    int norm = sqr(d->quant) >> 15;
    int relerr = (sqr(enc->curr_intra_pct) + enc->curr_scblocks + enc->avg_err * chaos) / (norm > 1 ? norm : 1);
    relerr += relerr * psy >> 7;
    int avg_chaos = (enc->prev_chaos + chaos + 1) >> 1;
    int thresh = 8;
    thresh += thresh * psy >> 5;
    int ae = enc->avg_err / 2 > 1 ? enc->avg_err / 2 : 1;
    thresh -= ((avg_chaos < 48 ? avg_chaos : 48) * psy * ae / (128 * (thresh - 2)));
    enc->auto_filter = chaos <= 1 || relerr > thresh;
}

// returns 1 when the P frame must be re-coded as an I frame (dsv_encoder.c:545); *map_written: the running intra map
// took this frame's intra blocks in (the reference updates it between its two tests)
int scene_change_detection(DSV_ENCODER *enc, FrameCtl *d, const int *bs, bool *map_written)
{
    DSV_PARAMS *p = &d->params;
    *map_written = false;
    int intra_pct = enc->curr_intra_pct, scblocks = enc->curr_scblocks;
    int avgmot = avg_motion(enc, p, bs);
    int chaos = enc->motion_chaos;
    int dchaos = abs(chaos - enc->prev_chaos);
    int gopdiv = abs(enc->gop) * 3 / 4;
    int closeness = (int) d->fnum - (int) enc->prev_gop;
    int complexity = scene_complexity(enc, p, bs);
    int closefac = closeness / (gopdiv > 1 ? gopdiv : 1);
    int shift;
    if (complexity > 256 && chaos < 5) {
        shift = 9;
    } else if (complexity > chaos * 2) {
        shift = 8;
    } else if (complexity > chaos) {
        shift = 7;
    } else {
        shift = 6;
    }
    int tipct = sqr(intra_pct) >> 5;
    int likely_sc = (intra_pct * 3 / 2 > scblocks) + (tipct > scblocks);
    int scp = enc->scene_change_pct > 1 ? enc->scene_change_pct : 1;
    if (scblocks > enc->scene_change_pct && chaos < 34) {
        scblocks = sqr(scblocks * 2) / scp;
        likely_sc++;
    } else {
        scblocks = sqr(scblocks) / scp;
    }
    shift = shift - likely_sc > 5 ? shift - likely_sc : 5;
    int a = dchaos / 16 + enc->avg_err / 8;
    int blks = (a > 1 ? a : 1) * scblocks * (complexity > 1 ? complexity : 1) * (closefac > 1 ? closefac : 1) >> (shift + 1);
    int pc = enc->prev_chaos - 10 > 30 ? enc->prev_chaos - 10 : 30;
    bool sc = enc->do_scd && (blks > 120 || (blks > enc->scene_change_pct && avgmot < 20 && enc->motion_chaos <= pc));
    bool high_intra = intra_pct > enc->intra_pct_thresh;
    if (sc || high_intra) {
        p->has_ref = 0;
        return 1;
    }
    enc->curr_complexity = complexity;
    *map_written = true;
    int nblk = p->nblocks_h * p->nblocks_v;
    int nintra = bs[BS_NINTRA] * 100 / nblk;
    int skipn = bs[BS_SKIPN] * 100 / nblk;
    if (nintra > enc->intra_pct_thresh && enc->curr_avgmot < 10 &&
        enc->motion_chaos <= clampi(enc->prev_chaos / 2 + skipn, 20, 40)) {
        p->has_ref = 0;
        return 1;
    }
    return 0;
}

// ---- packet framing -------------------------------------------------------------------------
void put_packet_hdr(BitWriter &bw, int type) // dsv_encoder.c:934
{
    bw.put_bits(8, 'D');
    bw.put_bits(8, 'S');
    bw.put_bits(8, 'V');
    bw.put_bits(8, '2');
    bw.put_bits(8, 8); // DSV_VERSION_MINOR
    bw.put_bits(8, (unsigned) type);
    bw.put_bits(32, 0);
    bw.put_bits(32, 0);
}

void put_be32(uint8_t *p, unsigned v)
{
    p[0] = (uint8_t) (v >> 24);
    p[1] = (uint8_t) (v >> 16);
    p[2] = (uint8_t) (v >> 8);
    p[3] = (uint8_t) v;
}

void set_link_offsets(DSV_ENCODER *enc, DSV_BUF *buf, int is_eos) // dsv_encoder.c:470
{
    unsigned next = is_eos ? 0 : buf->len;
    put_be32(buf->data + DSV_PACKET_PREV_OFFSET, (unsigned) enc->prev_link);
    put_be32(buf->data + DSV_PACKET_NEXT_OFFSET, next);
    enc->prev_link = (int) next;
}

void encode_metadata(DSV_ENCODER *enc, DSV_BUF *buf) // dsv_encoder.c:951
{
    const DSV_META *m = &enc->vidmeta;
    dsv_mk_buf(buf, 64);
    BitWriter bw{buf->data, 0};
    put_packet_hdr(bw, DSV_PT_META);
    bw.put_ueg((unsigned) m->width);
    bw.put_ueg((unsigned) m->height);
    bw.put_ueg((unsigned) m->subsamp);
    bw.put_ueg((unsigned) m->fps_num);
    bw.put_ueg((unsigned) m->fps_den);
    bw.put_ueg((unsigned) m->aspect_num);
    bw.put_ueg((unsigned) m->aspect_den);
    bw.put_ueg((unsigned) m->inter_sharpen);
    bw.put_bit(0);
    bw.align();
    unsigned len = bw.byte_pos();
    put_be32(buf->data + DSV_PACKET_NEXT_OFFSET, len);
    buf->len = len;
}

// ---- per-block metadata ---------------------------------------------------------------------
enum { ST_STABLE = 0, ST_MAINTAIN, ST_RINGING, ST_MODE, ST_EPRM, ST_MAX };

void gather_stats(DSV_ENCODER *enc, const FrameCtl *d, const int *bs, const DSV_MV *intramv, int *stats) // :992
{
    if (d->params.has_ref) { // the votes over the motion field were counted on the device (BlockStatsJob)
        stats[ST_MODE] += bs[BS_MODE];
        stats[ST_EPRM] += bs[BS_EPRM];
        stats[ST_STABLE] += bs[BS_STABLE];
        return;
    }
    int nblk = d->params.nblocks_h * d->params.nblocks_v;
    int avgdiv = (int) enc->refresh_ctr;
    if (enc->refresh_ctr >= enc->stable_refresh) {
        avgdiv = 0;
    }
    if (avgdiv <= 0) {
        avgdiv = 1;
    }
    for (int i = 0; i < nblk; i++) {
        int stable = 0;
        {
            const DSV_MV *mv = &intramv[i];
            if (d->fnum > 0 && enc->do_temporal_aq) {
                stable = (enc->stability[i].x / avgdiv == 0) && (enc->stability[i].y / avgdiv == 0);
            } else {
                stable = mvflag(*mv, DSV_MV_BIT_SKIP);
            }
            stats[ST_MAINTAIN] += mvflag(*mv, DSV_MV_BIT_MAINTAIN) ? 1 : -1;
            stats[ST_RINGING] += mvflag(*mv, DSV_MV_BIT_RINGING) ? 1 : -1;
        }
        stats[ST_STABLE] += (stable & 1) ? 1 : -1;
    }
}

// the five majority bits of a picture's header (dsv_encoder.c:1066-1078)
void picture_votes(DSV_ENCODER *enc, const FrameCtl *d, const int *bs, const DSV_MV *intramv, int *stats)
{
    for (int i = 0; i < ST_MAX; i++) {
        stats[i] = 0;
    }
    if (enc->effort >= 7) {
        gather_stats(enc, d, bs, intramv, stats);
        for (int i = 0; i < ST_MAX; i++) {
            stats[i] = stats[i] > 0 ? 1 : 0; // 1 = DSV_ZERO_MARKER
        }
    } else {
        stats[ST_MAINTAIN] = stats[ST_RINGING] = 1;
    }
}

// Zero-filled scratch for the bit writers of the per-block side information (bs.c:143: the writers skip zero bits).  The
// sub-streams are sized for the worst case (32 bytes a block) but a frame writes a few hundred bytes of each: the
// buffers live per worker thread and only what a frame wrote is cleared again (a fresh zero-filled vector per
// sub-stream was 1.6 MB of memset per 1080p frame -- most of a host core at 5 000 frames/s).
struct ZeroScratch {
    std::vector<uint8_t> mem;
    uint8_t *get(size_t bytes)
    {
        if (mem.size() < bytes) {
            mem.assign(bytes, 0);
        }
        return mem.data();
    }
    void clear(size_t used) { memset(mem.data(), 0, std::min(used + 24, mem.size())); }
};
thread_local ZeroScratch t_side[6];

void append_sub(BitWriter &bs, const uint8_t *data, int bytes)
{
    bs.put_ueg((unsigned) bytes);
    bs.align();
    bs.concat(data, bytes);
}

void encode_stable_blocks(DSV_ENCODER *enc, const FrameCtl *d, BitWriter &bs, DSV_MV *mvs, const DSV_MV *intramv,
                          const int *stats) // dsv_encoder.c:797
{
    int nblk = d->params.nblocks_h * d->params.nblocks_v;
    uint8_t *buf = t_side[5].get((size_t) nblk * 32);
    RleWriter rle;
    rle.bw = BitWriter{buf, 0};
    rle.bw.wide = true; // (the scratch is sized for the worst case, 32 bytes a block: every 64-bit store stays inside)
    if (enc->refresh_ctr >= enc->stable_refresh) {
        enc->refresh_ctr = 0;
        memset(enc->stability, 0, sizeof(*enc->stability) * (size_t) nblk);
    }
    int avgdiv = (int) enc->refresh_ctr;
    if (avgdiv <= 0) {
        avgdiv = 1;
    }
    int fps = (d->params.vidmeta->fps_num + d->params.vidmeta->fps_den / 2) / d->params.vidmeta->fps_den;
    int dsf = fps <= 24 ? 6 : (fps <= 30 ? 4 : (fps <= 60 ? 2 : 0));
    for (int i = 0; i < nblk; i++) {
        int stable = 0;
        if (d->params.has_ref) {
            DSV_MV *mv = &mvs[i];
            enc->blockdata[i] = 0;
            if (mvflag(*mv, DSV_MV_BIT_SKIP)) {
                mv->u.all = 0;
            }
            if (mvflag(*mv, DSV_MV_BIT_INTRA)) {
                enc->blockdata[i] |= DSV_IS_INTRA;
            } else {
                stable = mvflag(*mv, DSV_MV_BIT_SKIP);
                if (!stable) {
                    enc->stability[i].x += abs(mv->u.mv.x) >> dsf;
                    enc->stability[i].y += abs(mv->u.mv.y) >> dsf;
                } else {
                    mv->u.all = 0;
                }
            }
            enc->blockdata[i] |= (uint8_t) (stable << 2);                               /* DSV_SKIP_BIT */
            enc->blockdata[i] |= (uint8_t) (mvflag(*mv, DSV_MV_BIT_SIMCMPLX) << 6);      /* DSV_SIMCMPLX_BIT */
        } else {
            const DSV_MV *mv = &intramv[i];
            if (d->fnum > 0 && enc->do_temporal_aq) {
                stable = (enc->stability[i].x / avgdiv == 0) && (enc->stability[i].y / avgdiv == 0);
            }
            stable |= mvflag(*mv, DSV_MV_BIT_SKIP);
            enc->blockdata[i] = (uint8_t) (stable << 0); /* DSV_STABLE_BIT */
        }
        rle.put(stats[ST_STABLE] == 0 ? (stable & 1) : !(stable & 1));
    }
    bs.align();
    int bytes = rle.finish();
    append_sub(bs, buf, bytes);
    t_side[5].clear((size_t) bytes);
}

void encode_motion(DSV_ENCODER *enc, const FrameCtl *d, BitWriter &bs, DSV_MV *mvs, const int *stats) // dsv_encoder.c:692
{
    const DSV_PARAMS *p = &d->params;
    int nblk = p->nblocks_h * p->nblocks_v;
    size_t ub = (size_t) nblk * 32;
    uint8_t *bufs[5];
    for (int k = 0; k < 5; k++) {
        bufs[k] = t_side[k].get(ub);
    }
    RleWriter mode_rle, eprm_rle;
    mode_rle.bw = BitWriter{bufs[0], 0};
    eprm_rle.bw = BitWriter{bufs[4], 0};
    BitWriter mvx{bufs[1], 0}, mvy{bufs[2], 0}, sbim{bufs[3], 0};
    mode_rle.bw.wide = eprm_rle.bw.wide = mvx.wide = mvy.wide = sbim.wide = true; // worst-case sized, zero-filled scratch

    for (int j = 0; j < p->nblocks_v; j++) {
        for (int i = 0; i < p->nblocks_h; i++) {
            int idx = i + j * p->nblocks_h;
            DSV_MV *mv = &mvs[idx];
            enc->blockdata[idx] |= (uint8_t) (mvflag(*mv, DSV_MV_BIT_EPRM) << 5);
            if (mvflag(*mv, DSV_MV_BIT_SKIP)) {
                enc->blockdata[idx] |= DSV_IS_STABLE;
                continue;
            }
            int intra = mvflag(*mv, DSV_MV_BIT_INTRA);
            int px, py, cvx, cvy;
            movec_pred(mvs, p->nblocks_h, i, j, &px, &py);
            if (intra) {
                px = sar_r(px, 2);
                py = sar_r(py, 2);
                cvx = sar(mv->u.mv.x, 2);
                cvy = sar(mv->u.mv.y, 2);
                mv->u.mv.x = (int16_t) (cvx * 4);
                mv->u.mv.y = (int16_t) (cvy * 4);
                if (mv->submask == DSV_MASK_ALL_INTRA) {
                    sbim.put_bit(1);
                } else {
                    sbim.put_bit(0);
                    sbim.put_bits(4, mv->submask);
                }
                if (mv->dc & DSV_SRC_DC_PRED) {
                    sbim.put_bit(1);
                    sbim.put_bits(8, mv->dc & 0xff);
                } else {
                    sbim.put_bit(0);
                }
            } else {
                cvx = mv->u.mv.x;
                cvy = mv->u.mv.y;
            }
            mvx.put_seg(cvx - px);
            mvy.put_seg(cvy - py);
            if (neighbordif(mvs, p->nblocks_h, i, j) > 8) { /* DSV_NDIF_THRESH */
                enc->blockdata[idx] |= DSV_IS_STABLE;
            }
            int eprm = mvflag(*mv, DSV_MV_BIT_EPRM);
            mode_rle.put(stats[ST_MODE] == 0 ? intra : !intra);
            eprm_rle.put(stats[ST_EPRM] == 0 ? eprm : !eprm);
        }
    }
    for (int s = 0; s < 5; s++) {
        int bytes;
        bs.align();
        if (s == 0) {
            bytes = mode_rle.finish();
        } else if (s == 4) {
            bytes = eprm_rle.finish();
        } else {
            BitWriter &w = s == 1 ? mvx : (s == 2 ? mvy : sbim);
            w.align();
            bytes = (int) w.byte_pos();
        }
        append_sub(bs, bufs[s], bytes);
        t_side[s].clear((size_t) bytes);
    }
}

void encode_intra_meta(DSV_ENCODER *enc, const FrameCtl *d, BitWriter &bs, const DSV_MV *intramv, const int *stats) // :886
{
    int nblk = d->params.nblocks_h * d->params.nblocks_v;
    uint8_t *br = t_side[0].get((size_t) nblk * 32), *bm = t_side[1].get((size_t) nblk * 32);
    RleWriter rr, rm;
    rr.bw = BitWriter{br, 0};
    rm.bw = BitWriter{bm, 0};
    rr.bw.wide = rm.bw.wide = true;
    for (int i = 0; i < nblk; i++) {
        int ring = mvflag(intramv[i], DSV_MV_BIT_RINGING), maintain = mvflag(intramv[i], DSV_MV_BIT_MAINTAIN);
        enc->blockdata[i] |= (uint8_t) (ring << 3);
        enc->blockdata[i] |= (uint8_t) (maintain << 1);
        rr.put(stats[ST_RINGING] == 0 ? ring : !ring);
        rm.put(stats[ST_MAINTAIN] == 0 ? maintain : !maintain);
    }
    bs.align();
    int bytes = rr.finish();
    append_sub(bs, br, bytes);
    t_side[0].clear((size_t) bytes);
    bs.align();
    bytes = rm.finish();
    append_sub(bs, bm, bytes);
    t_side[1].clear((size_t) bytes);
}

// ---- device pipeline pieces -----------------------------------------------------------------
void build_pyramid_on(hipStream_t s, CodecDev &dv, const DFrame &base, DFrame *pyr) // mk_pyramid, dsv_encoder.c:493
{
    const DFrame *prev = &base;
    for (int l = 0; l < dv.pyr_levels; l++) {
        ds2x_luma(s, prev->p[0], pyr[l].p[0]);
        extend_plane(s, pyr[l].p[0]);
        prev = &pyr[l];
    }
}

AnalysisParams analysis_params(const CodecDev &dv, int do_psy)
{
    AnalysisParams a;
    a.width = dv.w;
    a.height = dv.h;
    a.blk_w = dv.blk_w;
    a.blk_h = dv.blk_h;
    a.nbh = dv.nbh;
    a.nbv = dv.nbv;
    a.hshift = DSV_FORMAT_H_SHIFT(dv.format);
    a.vshift = DSV_FORMAT_V_SHIFT(dv.format);
    a.do_psy = do_psy;
    a.scale = 2 * spatial_psy_factor(dv.blk_w, dv.blk_h, dv.nbh, dv.nbv, -1);
    return a;
}

// DC coefficient of every plane of every stream of the batch (sent raw, hzcc.c:599-602): table
// entry i of `luma`, entries 2i / 2i+1 of `chroma` -> out[3i .. 3i+2]
__global__ void k_grab_ll(const PlaneJob *luma, const PlaneJob *chroma, int n, int32_t *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[3 * i] = luma[i].coefs[0];
        out[3 * i + 1] = chroma[2 * i].coefs[0];
        out[3 * i + 2] = chroma[2 * i + 1].coefs[0];
    }
}

// dsv_encode_picture (dsv_encoder.c:1039): returns the picture packet in `out`

void account(DSV_ENCODER *enc, EncImpl *im, const FrameCtl *d, unsigned len, const int *bs) // dsv_enc tail, dsv_encoder.c:1471-1570
{
    const DSV_PARAMS *p = &d->params;
    struct DSV_STATS *st = &enc->stats;
    if (p->has_ref) {
        st->pnum++;
        st->pfnum += !!enc->auto_filter;
        st->psize += len;
        st->pqual += enc->rc_qual;
        st->pmaxq = enc->rc_qual > st->pmaxq ? enc->rc_qual : st->pmaxq;
        st->pmaxs = len > st->pmaxs ? len : st->pmaxs;
        st->pminq = enc->rc_qual < st->pminq ? enc->rc_qual : st->pminq;
        st->pmins = len < st->pmins ? len : st->pmins;
        int nblk = p->nblocks_h * p->nblocks_v;
        // the block counts over the field were taken on the device (k_block_stats_b, BS_ST_*)
        st->eprm += (unsigned) bs[BS_ST_EPRM];
        st->skip += (unsigned) bs[BS_ST_SKIP];
        st->mbI += (unsigned) bs[BS_ST_MBI];
        st->mbdc += (unsigned) bs[BS_ST_MBDC];
        st->mbsub += (unsigned) bs[BS_ST_MBSUB];
        for (int k = 0; k < 4; k++) {
            st->mbsubs[k] += (unsigned) bs[BS_ST_SUB0 + k];
        }
        st->mbP += (unsigned) bs[BS_ST_MBP];
        st->qpx += (unsigned) bs[BS_ST_QPX];
        st->hpx += (unsigned) bs[BS_ST_HPX];
        st->fpx += (unsigned) bs[BS_ST_FPX];
        st->qpy += (unsigned) bs[BS_ST_QPY];
        st->hpy += (unsigned) bs[BS_ST_HPY];
        st->fpy += (unsigned) bs[BS_ST_FPY];
        st->mb += (unsigned) nblk;
        enc->refresh_ctr++;
    } else {
        st->inum++;
        st->ifnum += !!enc->do_intra_filter;
        st->isize += len;
        st->iqual += enc->rc_qual;
        st->imaxq = enc->rc_qual > st->imaxq ? enc->rc_qual : st->imaxq;
        st->imaxs = len > st->imaxs ? len : st->imaxs;
        st->iminq = enc->rc_qual < st->iminq ? enc->rc_qual : st->iminq;
        st->imins = len < st->imins ? len : st->imins;
    }
    if (enc->rc_mode != DSV_RATE_CONTROL_CQP) {
        enc->rf_total += enc->rc_mode == DSV_RATE_CONTROL_CRF ? enc->rc_qual : len;
        enc->rf_reset++;
        if (p->has_ref) {
            enc->total_P_frame_q += (int) enc->rc_qual;
            enc->avg_P_frame_q = (int) ((unsigned) enc->total_P_frame_q / enc->rf_reset);
        }
        enc->rf_avg = (int) (enc->rf_total / enc->rf_reset);
        if (enc->rf_reset >= 256) {
            enc->rf_total = (unsigned) enc->rf_avg;
            enc->total_P_frame_q = (int) ((unsigned) enc->total_P_frame_q / enc->rf_reset);
            enc->rf_reset = 1;
        }
    }
}


// ---- lockstep batch engine ----------------------------------------------------------------------
// One step encodes ONE frame on each of n independent encoder instances of identical geometry.
// The per-frame control flow of the reference (encode_one_frame dsv_encoder.c:1185, encode_picture
// :1040) is cut into phases; device phases enqueue the work of all streams on one HIP stream with
// the latency-bound kernels (ME fronts, MC, in-loop filters) launched ONCE for all streams
// (stream index = a grid dimension), host phases run the per-stream serial logic on a thread per
// stream.  n = 1 is the plain dsv_enc() path: same code, same results.
//
//   P0 host   frame number, params, GOP decision
//   G1 device ingest + extend + pyramid, intra analysis, [P] reference pyramid, batched HME, read-backs
//   H1 host   scene-change decision, rate control, packet header, per-block metadata coding
//   G2 device [P] batched predict+subtract, forward SBT, quantise+compact, inverse SBT,
//             batched reconstruct + filters, border extension, symbol read-back
//   H2 host   entropy packing, packet framing, statistics

struct Job {
    DSV_ENCODER *enc;
    EncImpl *im;
    DSV_FRAME *frame;          // host picture (dsv_enc) ...
    const uint8_t *dev_planar; // ... or packed planar picture already in HBM
    const uint8_t *host_planar; // ... or packed planar picture in host memory, uploaded by the batch engine (pinned: asynchronously)
    const uint8_t *host_next;   // the picture this stream will bring to the NEXT step: uploaded under this step's kernels
    bool from_frame;            // host_planar is a DSV_FRAME packed by dsv_enc: planar whatever the encoder's packed-input format
    DSV_BUF *bufs;
    int nbuf;
    FrameCtl d;
    DSV_BUF out;
    BitWriter bs;
    int gop_start, forced_intra, ran_hme, inter_filter, nsym;
    int failed; // phase_h1a: the search did not deliver (1: a row timed out, 2: no counters)
    const int *bstats; // BS_* sums over the search result (pinned, written by k_block_stats_b), P frames only
    const uint8_t *side_out; // P frames: the six side-information sub-streams as coded by k_side_info (pinned), or null
    const int *side_info;    // ... their byte lengths ([1 + sub]) and the fall-back flag ([0])
    const uint8_t *gpu_bytes; // the three plane sections as assembled by the GPU (entropy_gpu.hip), or null: the host codes them
    unsigned gpu_plane_bytes[3];
    DSV_FNUM prev_I;
    int stats[ST_MAX];
};

// A stream whose hardware queue is chosen NOW, one stream at a time.  The runtime binds a stream to one of its
// GPU_MAX_HW_QUEUES hardware queues when the stream is first used; lockstep groups start together, and streams that are
// created and first used by several threads at the same moment were seen to land on the SAME queue (two groups' kernels
// then run one after the other: every device phase of a small batch took twice as long).  So creation and a first,
// completed, operation happen under one process-wide lock.
hipStream_t new_bound_stream()
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    static int *d_word = nullptr;
    if (!d_word) {
        HIPCHK(hipMalloc((void **) &d_word, 64));
    }
    HIPCHK(hipMemsetAsync(d_word, 0, 64, s));
    HIPCHK(hipStreamSynchronize(s));
    return s;
}

struct BatchScratch { // pinned + device memory for the job tables, the step's HIP streams: held by ONE enc_batch call at a time
    void *h_hme = nullptr, *d_hme = nullptr;
    McJob *h_mc = nullptr, *d_mc = nullptr;
    uint8_t *h_stage = nullptr, *d_stage = nullptr; // per stream: transmitted motion field + block flag bytes
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
        HIPCHK(hipHostMalloc((void **) &h_stage, bytes, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **) &d_stage, bytes));
        stage_cap = bytes;
    }
    int32_t *h_ll = nullptr, *d_ll = nullptr; // [3 * n] DC coefficients
    int *h_totals = nullptr, *d_totals = nullptr; // [n] symbol counts
    int *h_bstats = nullptr, *d_bstats = nullptr; // [n][BS_WORDS] block statistics of the search results
    uint8_t *h_side = nullptr;                    // [n][SIDE_IMG_BYTES] pinned: the side-information sub-streams (k_side_info)
    int *h_side_info = nullptr;                   // [n][SIDE_INFO_WORDS] pinned: their lengths and fall-back flags
    hipEvent_t ev_side = nullptr;                 // the sub-streams of this step are in host memory
    // the step's kernels run on the SCRATCH's stream, not on one of the encoders': a lockstep group then owns one stream (one
    // hardware queue) however many encoder instances exist (a step starts and ends with that stream drained, so which
    // stream carried an encoder's previous step does not matter)
    hipStream_t main = nullptr;
    hipStream_t main_stream()
    {
        if (!main) {
            main = new_bound_stream();
        }
        return main;
    }
    // uploads of the NEXT step's host pictures run on a stream of their own, under this step's kernels
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;
    // side streams of a step: work that does not depend on the main chain runs beside it (the plane sections are
    // assembled while the inverse transform / reconstruction / in-loop filters run; the intra pictures' filter sweeps
    // beside the inter pictures')
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork[2], ev_join[2];
    void ensure_aux()
    {
        if (!aux[0]) {
            for (int i = 0; i < 2; i++) {
                HIPCHK(hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&ev_fork[i], hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
            }
        }
    }
    // the rest of `side` happens after everything enqueued on `main` so far
    void fork(hipStream_t main, int i)
    {
        HIPCHK(hipEventRecord(ev_fork[i], main));
        HIPCHK(hipStreamWaitEvent(aux[i], ev_fork[i], 0));
    }
    void join(hipStream_t main, int i)
    {
        HIPCHK(hipEventRecord(ev_join[i], aux[i]));
        HIPCHK(hipStreamWaitEvent(main, ev_join[i], 0));
    }
    void ensure_copy_stream()
    {
        if (!copy_stream) {
            copy_stream = new_bound_stream();
            HIPCHK(hipEventCreateWithFlags(&copy_done, hipEventDisableTiming));
        }
    }
    // working picture of a picture whose symbols are worked out a second time (redo_overflow: its compaction lists were too short)
    DFrame redo;
    DFrame &redo_frame(int format, int w, int h)
    {
        if (redo.alloc && (redo.format != format || redo.w != w || redo.h != h)) {
            dframe_free(&redo);
        }
        if (!redo.alloc) {
            dframe_alloc(&redo, format, w, h);
        }
        return redo;
    }
    TableArena tabs;
    int cap = 0;
    void ensure(int n)
    {
        tabs.reserve((size_t) n * 12288 + 65536); // (the last 4 KB a stream: the tables of a redone picture, enc_batch)
        if (n <= cap) {
            return;
        }
        if (cap) {
            HIPCHK(hipHostFree(h_hme));
            HIPCHK(hipFree(d_hme));
            HIPCHK(hipHostFree(h_mc));
            HIPCHK(hipFree(d_mc));
            HIPCHK(hipHostFree(h_ll));
            HIPCHK(hipFree(d_ll));
            HIPCHK(hipHostFree(h_totals));
            HIPCHK(hipFree(d_totals));
            HIPCHK(hipHostFree(h_bstats));
            HIPCHK(hipFree(d_bstats));
            HIPCHK(hipHostFree(h_side));
            HIPCHK(hipHostFree(h_side_info));
        }
        HIPCHK(hipHostMalloc(&h_hme, hme_table_bytes(n), hipHostMallocDefault));
        HIPCHK(hipMalloc(&d_hme, hme_table_bytes(n)));
        HIPCHK(hipHostMalloc((void **) &h_mc, 2 * (size_t) n * sizeof(McJob), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **) &d_mc, 2 * (size_t) n * sizeof(McJob)));
        HIPCHK(hipHostMalloc((void **) &h_ll, 3 * (size_t) n * sizeof(int32_t), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **) &d_ll, 3 * (size_t) n * sizeof(int32_t)));
        HIPCHK(hipHostMalloc((void **) &h_totals, (size_t) n * sizeof(int), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **) &d_totals, (size_t) n * sizeof(int)));
        HIPCHK(hipHostMalloc((void **) &h_bstats, (size_t) n * BS_WORDS * sizeof(int), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **) &d_bstats, (size_t) n * BS_WORDS * sizeof(int)));
        HIPCHK(hipHostMalloc((void **) &h_side, (size_t) n * SIDE_IMG_BYTES, hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **) &h_side_info, (size_t) n * SIDE_INFO_WORDS * sizeof(int), hipHostMallocDefault));
        if (!ev_side) {
            HIPCHK(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
        }
        cap = n;
    }
};
// Process-wide pool, last released first: a lockstep group that calls step after step keeps getting the same scratch (its
// tables are sized, its streams mapped to a hardware queue) whatever thread it calls from, and scratches -- with their
// streams and events -- are never destroyed, so an upload left in flight by one call (EncImpl::staged_ev) can be awaited
// by the next, from any thread.  A scratch's copy stream and copy_done event travel together: a later re-record of the
// event on that stream covers every earlier upload on it.
inline bool use_own_streams()
{
    return true;
}

struct ScratchPool {
    std::mutex mu;
    std::vector<BatchScratch *> idle;
    // `prefer`: the scratch this thread used last -- a lockstep group driven by one long-lived thread keeps its scratch (and
    // with it its streams and their hardware queues) from step to step; a new thread takes whatever is idle
    bool primed = false;
    BatchScratch *acquire(BatchScratch *prefer)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!primed) {
            // The first four scratches and their streams are made HERE, once, in a fixed order -- main streams first, then
            // the copy streams -- and live for the rest of the process.  The runtime binds a stream to one of its hardware
            // queues when it is created, by the queues' load at that moment (tools/probe/queue_pairs.cpp: sixteen streams made
            // in a row land on queues 0 .. 7, 7 .. 0; four made beside six long-lived ones share TWO queues).  Streams that
            // come and go with the encoder instances of a run therefore end up two to a queue sooner or later and two
            // lockstep groups then execute one after the other; a fixed set made up front does not.
            primed = true;
            if (use_own_streams()) {
                BatchScratch *first[4];
                for (int k = 0; k < 4; k++) {
                    first[k] = new BatchScratch();
                    first[k]->main_stream();
                }
                for (int k = 0; k < 4; k++) {
                    first[k]->ensure_copy_stream();
                }
                for (int k = 3; k >= 0; k--) {
                    idle.push_back(first[k]);
                }
            }
        }
        if (idle.empty()) {
            return new BatchScratch();
        }
        for (size_t i = 0; i < idle.size(); i++) {
            if (idle[i] == prefer) {
                idle.erase(idle.begin() + (ptrdiff_t) i);
                return prefer;
            }
        }
        BatchScratch *sc = idle.back();
        idle.pop_back();
        return sc;
    }
    void release(BatchScratch *sc)
    {
        std::lock_guard<std::mutex> lk(mu);
        idle.push_back(sc);
    }
};
ScratchPool g_scratch_pool;
thread_local BatchScratch *t_last_scratch = nullptr;
struct ScratchLease {
    BatchScratch *sc = g_scratch_pool.acquire(t_last_scratch);
    const int unwinding = std::uncaught_exceptions();
    ScratchLease() { t_last_scratch = sc; }
    ~ScratchLease()
    {
        // A step that FAILS (StepFailed) leaves with work enqueued or running on this scratch's streams, reading the pinned and
        // device job tables: they are drained before another group may overwrite the tables; a scratch whose streams do not
        // drain is never handed out again (leaked on purpose: its owner's encoders are dead, the process is not).
        if (std::uncaught_exceptions() > unwinding) {
            bool ok = true;
            for (hipStream_t s : {sc->main, sc->aux[0], sc->aux[1], sc->copy_stream}) {
                ok = ok && (!s || hipStreamSynchronize(s) == hipSuccess);
            }
            if (!ok) {
                fprintf(stderr, "[dsv2hip] a failed step's streams did not drain: its batch scratch is retired\n");
                t_last_scratch = nullptr;
                return;
            }
        }
        g_scratch_pool.release(sc);
    }
};

// A step that cannot finish (the search token never comes, a search reports a time-out) fails the CALLS that are part of it --
// every job of the step returns no packets, its encoder is marked dead -- not the process: the library lives inside somebody
// else's program (the reference's dsv_enc has no failure path at all; "no packets" is the closest thing its callers handle).
struct StepFailed {
    const char *what;
};

// what the library cannot encode; said once per reason, the call then returns "no packets" (dsv_enc: 0, the batch calls: -1)
static bool enc_usable(const DSV_ENCODER *enc)
{
    if (enc == nullptr) {
        return false;
    }
    const int w = enc->vidmeta.width, h = enc->vidmeta.height;
    if ((w & 1) || (h & 1) || w < 16 || h < 16) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true)) {
            fprintf(stderr, "[dsv2hip] %dx%d: DSV2 needs even picture dimensions of at least 16x16 (dsv_main.c:621, sbt.c:384-388); the call is refused\n", w, h);
        }
        return false;
    }
    if (enc->ref && ((const EncImpl *) enc->ref)->dead) {
        return false;
    }
    return true;
}

// the pyramid depth an encoder runs with (dsv_encoder.c:1229-1241): 0 in the public struct means "work it out"
static int resolved_pyramid_levels(const DSV_ENCODER *enc)
{
    if (enc->pyramid_levels != 0) {
        return enc->pyramid_levels;
    }
    const int w = enc->vidmeta.width, h = enc->vidmeta.height;
    int bw, bh, nbh, nbv;
    block_geometry(w, h, enc->block_size_override_x, enc->block_size_override_y, &bw, &bh, &nbh, &nbv);
    int lvls = dsv_lb2((unsigned) (w < h ? w : h));
    int maxdim = nbh > nbv ? nbh : nbv;
    while ((1 << lvls) > maxdim) {
        lvls--;
    }
    return clampi(lvls, 3, DSV_MAX_PYRAMID_LEVELS);
}

// what a lockstep step must agree on: picture geometry, block size, pyramid depth (the RESOLVED one: an encoder's first frame
// and its later ones, and encoders that were and were not started yet, then share a key), psy switch
static unsigned long long step_key(const DSV_ENCODER *enc)
{
    unsigned long long key = 1469598103934665603ull;
    for (unsigned long long v : {(unsigned long long) enc->vidmeta.width, (unsigned long long) enc->vidmeta.height, (unsigned long long) enc->vidmeta.subsamp,
                                 (unsigned long long) (unsigned) enc->block_size_override_x, (unsigned long long) (unsigned) enc->block_size_override_y,
                                 (unsigned long long) resolved_pyramid_levels(enc), (unsigned long long) enc->do_psy}) {
        key = (key ^ v) * 1099511628211ull;
    }
    return key;
}

void ensure_ready(DSV_ENCODER *enc, EncImpl *im)
{
    if (im->ready) {
        return;
    }
    int w = enc->vidmeta.width, h = enc->vidmeta.height;
    int bw, bh, nbh, nbv;
    block_geometry(w, h, enc->block_size_override_x, enc->block_size_override_y, &bw, &bh, &nbh, &nbv);
    enc->pyramid_levels = resolved_pyramid_levels(enc);
    // Compaction lists: HALF the worst case to begin with -- a detail-rich 1080p intra picture at qp 60 has a symbol for 33 % of
    // its coefficients, its P pictures for 5 % -- and the worst case at once for lossless streams, where nearly every coefficient
    // is a symbol.  DSV2_COMPACT_CAP (symbols) forces a figure (tests: so small that pictures overflow and take the redo path).
    size_t list_syms = 0;
    if (enc->quality != DSV_RC_QUAL_MAX) {
        const long forced = getenv("DSV2_COMPACT_CAP") ? atol(getenv("DSV2_COMPACT_CAP")) : -1; // (read per instance: tests set it)
        const int fmt = enc->vidmeta.subsamp;
        const size_t ncoef = (size_t) w * h + 2 * (size_t) ((w + (1 << DSV_FORMAT_H_SHIFT(fmt)) - 1) >> DSV_FORMAT_H_SHIFT(fmt)) * ((h + (1 << DSV_FORMAT_V_SHIFT(fmt)) - 1) >> DSV_FORMAT_V_SHIFT(fmt));
        list_syms = forced >= 0 ? (size_t) forced : std::max<size_t>(65536, ncoef / 2);
    }
    im->dev.init(enc->vidmeta.subsamp, w, h, bw, bh, enc->pyramid_levels, true, list_syms);
    im->ready = true;
    im->mvs.assign((size_t) nbh * nbv, DSV_MV{});
    enc->stability = (struct DSV_STAB_ACC *) dsv_alloc((int) (sizeof(struct DSV_STAB_ACC) * (size_t) nbh * nbv));
    enc->blockdata = (uint8_t *) dsv_alloc(nbh * nbv);
}

// P0: frame number, parameters, GOP / reference decision (dsv_encoder.c:1193-1278)
void phase_p0(Job &jb)
{
    DSV_ENCODER *enc = jb.enc;
    EncImpl *im = jb.im;
    memset(&jb.d, 0, sizeof(jb.d));
    jb.d.fnum = enc->next_fnum++;
    DSV_PARAMS *p = &jb.d.params;
    p->vidmeta = &enc->vidmeta;
    p->effort = enc->effort;
    p->do_psy = enc->do_psy;
    p->temporal_mc = (int) (jb.d.fnum % 2);
    p->lossless = enc->quality == DSV_RC_QUAL_MAX;
    p->blk_w = im->dev.blk_w;
    p->blk_h = im->dev.blk_h;
    p->nblocks_h = im->dev.nbh;
    p->nblocks_v = im->dev.nbv;
    jb.gop_start = jb.forced_intra = jb.ran_hme = jb.inter_filter = jb.nsym = 0;
    jb.prev_I = enc->prev_gop;
    if (enc->force_metadata || ((DSV_FNUM) (enc->prev_gop + (DSV_FNUM) enc->gop) <= jb.d.fnum)) {
        jb.gop_start = 1;
        enc->prev_gop = jb.d.fnum;
        enc->force_metadata = 0;
    }
    if (enc->gop == DSV_GOP_INTRA) {
        p->is_ref = 0;
        p->has_ref = 0;
    } else {
        p->is_ref = 1;
        p->has_ref = jb.gop_start ? 0 : 1;
        if (p->has_ref && !im->have_ref) {
            fatal("P frame requested without a reference picture", __FILE__, __LINE__);
        }
    }
    enc->avg_err = 0;
    im->dev.pics[im->cur].has_final_mvs = false;
    if (!p->has_ref && !enc->intra_map) {
        enc->intra_map = (uint8_t *) dsv_alloc((int) im->dev.nblocks());
    }
}

// H1, first half: scene-change decision and rate control (dsv_encoder.c:1279-1290)
void phase_h1a(Job &jb)
{
    DSV_ENCODER *enc = jb.enc;
    EncImpl *im = jb.im;
    CodecDev &dv = im->dev;
    FrameCtl *d = &jb.d;
    DSV_PARAMS *p = &d->params;
    size_t nb = dv.nblocks();
    if (jb.ran_hme) {
        if (dv.h_counters[7]) { // (a pool thread: the step is failed by enc_batch, behind this phase)
            jb.failed = dv.h_counters[7] == 1 ? 1 : 2;
            return;
        }
        int nintra = dv.h_counters[0], ndiff = dv.h_counters[1], eligible = dv.h_counters[2]; // hme.c:1825-1832, 2015
        unsigned total_err = (unsigned) dv.h_counters[3];
        enc->curr_scblocks = ndiff * 100 / (eligible ? eligible : 1);
        enc->avg_err = (int) (total_err / (unsigned) nb);
        enc->curr_intra_pct = nintra * 100 / (int) nb;
        bool map_written = false;
        jb.forced_intra = scene_change_detection(enc, d, jb.bstats, &map_written);
        // the device wrote (committed map | this frame's intra blocks) into the other buffer: it becomes the committed map
        // exactly when the reference's loop ran and the frame stays a P frame (an intra picture clears the map)
        if (map_written && !jb.forced_intra) {
            im->map_cur ^= 1;
            im->map_valid = true;
        }
    }
    if (enc->variable_i_interval && jb.forced_intra) {
        enc->prev_gop = d->fnum;
    }
    if (!p->has_ref) {
        memset(enc->intra_map, 0, nb);
        im->map_valid = false;
    }
    {
        const DPlane &cp = dv.pics[im->cur].src_pyr[dv.pyr_levels - 1].p[0];
        d->coarse_luma_avg = luma_avg_rows(dv.h_small, cp.w, cp.h);
    }
    quality2quant(enc, d, jb.prev_I, jb.forced_intra);
    compute_auto_filter(enc, d);
    if (p->has_ref) {
        // what the side-information coder on the device and the G2 tables need of H1's second half: the majority votes
        // (gather_stats, dsv_encoder.c:992: counted by k_block_stats_b) and the in-loop filter switch
        picture_votes(enc, d, jb.bstats, nullptr, jb.stats);
        jb.inter_filter = enc->do_inter_filter == 1 || (enc->do_inter_filter == -1 && enc->auto_filter);
    }
}

// H1, second half: encode_picture's host part (dsv_encoder.c:1051-1132); needs the intra analysis of
// a picture that is coded as intra -- for a P frame flipped by H1a that analysis runs in between
void phase_h1b(Job &jb)
{
    DSV_ENCODER *enc = jb.enc;
    EncImpl *im = jb.im;
    CodecDev &dv = im->dev;
    FrameCtl *d = &jb.d;
    DSV_PARAMS *p = &d->params;
    size_t nb = dv.nblocks();
    bool isP = p->has_ref;
    unsigned upper = (unsigned) (dv.w * dv.h);
    switch (enc->vidmeta.subsamp) {
        case DSV_SUBSAMP_444: upper *= 6; break;
        case DSV_SUBSAMP_422:
        case DSV_SUBSAMP_UYVY: upper *= 4; break;
        default: upper *= 2; break;
    }
    // the packet is assembled in a per-encoder buffer that stays mapped and zeroed between frames (the bit
    // writer needs zeroed memory, bs.c:143); H2 hands the caller a right-sized copy
    if (im->pkt.size() < (size_t) upper) {
        im->pkt.assign((size_t) upper, 0);
    }
    jb.bs = BitWriter{im->pkt.data(), 0};
    jb.bs.wide = true; // the packet scratch is sized for the worst case, far beyond any write
    BitWriter &bs = jb.bs;
    put_packet_hdr(bs, DSV_PT_PIC | (p->is_ref << 1) | p->has_ref);
    bs.align();
    bs.put_bits(32, d->fnum);
    DSV_MV *intramv = nullptr;
    if (!isP) {
        im->intramv.assign(dv.h_intra, dv.h_intra + nb);
        intramv = im->intramv.data();
    }
    int *stats = jb.stats;
    if (!isP) { // (a P frame's votes were taken in H1a)
        picture_votes(enc, d, nullptr, intramv, stats);
    }
    bs.align();
    bs.put_ueg((unsigned) (dsv_lb2((unsigned) p->blk_w) - 4));
    bs.put_ueg((unsigned) (dsv_lb2((unsigned) p->blk_h) - 4));
    bs.align();
    bs.put_bit(stats[ST_STABLE]);
    if (isP) {
        bs.put_bit(stats[ST_MODE]);
        bs.put_bit(stats[ST_EPRM]);
        bs.put_bit(jb.inter_filter);
    } else {
        bs.put_bit(stats[ST_MAINTAIN]);
        bs.put_bit(stats[ST_RINGING]);
        bs.put_bit(enc->do_intra_filter);
    }
    bs.put_bits(DSV_MAX_QP_BITS, (unsigned) d->quant);
    bs.put_bit(0);
    bs.align();
    static const bool side_fallback = getenv("DSV2_SIDE_FORCE_FALLBACK") && atoi(getenv("DSV2_SIDE_FORCE_FALLBACK")) != 0; // (tests)
    if (isP && jb.side_out && jb.side_info[0] == 0 && !side_fallback) {
        // the six sub-streams arrive coded (k_side_info); what stays here of encode_stable_blocks (dsv_encoder.c:797) is the
        // stability accumulator of the temporal AQ, which the next intra picture reads on the host
        if (enc->refresh_ctr >= enc->stable_refresh) {
            enc->refresh_ctr = 0;
            memset(enc->stability, 0, sizeof(*enc->stability) * nb);
        }
        int fps = (p->vidmeta->fps_num + p->vidmeta->fps_den / 2) / p->vidmeta->fps_den;
        int dsf = fps <= 24 ? 6 : (fps <= 30 ? 4 : (fps <= 60 ? 2 : 0));
        const DSV_MV *mv = dv.h_mvs;
        for (size_t i = 0; i < nb; i++) {
            if (!(mv[i].flags & ((1u << DSV_MV_BIT_INTRA) | (1u << DSV_MV_BIT_SKIP)))) {
                enc->stability[i].x += abs(mv[i].u.mv.x) >> dsf;
                enc->stability[i].y += abs(mv[i].u.mv.y) >> dsf;
            }
        }
        for (int s2 = 0; s2 < SIDE_SUBS; s2++) {
            bs.align();
            append_sub(bs, jb.side_out + side_image_offset(s2), jb.side_info[1 + s2]);
        }
        return;
    }
    if (isP) { // the host coders work on (and finalise) a copy of the field
        im->mvs.assign(dv.h_mvs, dv.h_mvs + nb);
    }
    encode_stable_blocks(enc, d, bs, im->mvs.data(), intramv, stats);
    if (isP) {
        // (the reference predicts between these two calls; the vectors are final after the first)
        bs.align();
        encode_motion(enc, d, bs, im->mvs.data(), stats);
    } else {
        encode_intra_meta(enc, d, bs, intramv, stats);
    }
}

// H2: entropy packing + packet framing + statistics (dsv_encoder.c:1147-1165, 1461-1570)
void phase_h2(Job &jb)
{
    DSV_ENCODER *enc = jb.enc;
    EncImpl *im = jb.im;
    CodecDev &dv = im->dev;
    BitWriter &bs = jb.bs;
    bs.align();
    if (jb.gpu_bytes) { // the sections arrive finished: byte-aligned, length field, DC, count, codes, 0x55 (hzcc.c:586-613)
        // header + side information were written into the packet scratch by H1; the sections go straight from the GPU's
        // pinned mirror into the caller's buffer (no pass through the scratch, which then only has its head to clear)
        const unsigned head = bs.byte_pos();
        const unsigned len = head + jb.gpu_plane_bytes[0] + jb.gpu_plane_bytes[1] + jb.gpu_plane_bytes[2];
        dsv_mk_buf(&jb.out, (int) len);
        memcpy(jb.out.data, im->pkt.data(), head);
        memcpy(jb.out.data + head, jb.gpu_bytes, len - head);
        memset(im->pkt.data(), 0, (size_t) head + 8);
        jb.out.len = len;
    } else {
        int at = 0;
        for (int c = 0; c < 3; c++) {
            uint32_t lo = (uint32_t) dv.qv_off[c], hi = (uint32_t) dv.qv_off[c + 1];
            int begin = at;
            while (at < jb.nsym && dv.h_pos[at] < hi) {
                dv.h_pos[at] -= lo;
                at++;
            }
            entropy_encode_plane(bs, dv.h_ll[c], dv.h_pos + begin, dv.h_val + begin, at - begin, dv.scan[c]);
        }
        bs.align();
        unsigned len = bs.byte_pos();
        dsv_mk_buf(&jb.out, (int) len);
        memcpy(jb.out.data, im->pkt.data(), len);
        memset(im->pkt.data(), 0, (size_t) len + 8);
        jb.out.len = len;
    }
    jb.nbuf = 0;
    if (jb.gop_start) {
        DSV_BUF metabuf;
        encode_metadata(enc, &metabuf);
        jb.bufs[jb.nbuf++] = metabuf;
        set_link_offsets(enc, &jb.bufs[jb.nbuf - 1], 0);
    }
    jb.bufs[jb.nbuf++] = jb.out;
    set_link_offsets(enc, &jb.bufs[jb.nbuf - 1], 0);
    account(enc, im, &jb.d, jb.out.len, jb.bstats);
    if (jb.d.params.is_ref && enc->gop != DSV_GOP_INTRA) {
        im->cur ^= 1; // this picture set becomes the reference of the next frame
        im->have_ref = true;
    }
}

struct PhaseClock { // DSV2_TRACE=2: wall-clock split of a lockstep step, printed every 16 steps
    bool on = (trace_mode() & 2) != 0;
    // =3: every phase boundary of every step with its absolute time (CLOCK_MONOTONIC, ms) -- lines up the groups' host phases
    bool abs_on = (trace_mode() & 4) != 0;
    void mark(const char *what, int n)
    {
        if (!abs_on) return;
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        fprintf(stderr, "[t %p] %.3f %s n=%d\n", (void *) this, ts.tv_sec * 1e3 + ts.tv_nsec / 1e6, what, n);
    }
    int every = (trace_mode() & 8) ? 1 : 16; // bit 3: print every step
    double acc[10] = {0};
    int steps = 0;
    std::chrono::steady_clock::time_point t0;
    void start()
    {
        if (on) {
            mark("enter", 0);
            t0 = std::chrono::steady_clock::now();
        }
    }
    void lap(int i)
    {
        if (!on) return;
        static const char *const names[10] = {"p0", "g1-enqueued", "g1-done", "h1-done", "g2-enqueued", "g2-done", "syms", "h2-done", "h1b-done", ""};
        mark(names[i], 0);
        auto t1 = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
    }
    void done(int n)
    {
        if (!on || ++steps % every) return;
        fprintf(stderr, "[batch n=%d] ms/step: p0 %.2f | g1 enqueue %.2f wait %.2f | h1 %.2f | g2 enqueue %.2f, h1b under it %.2f, wait %.2f | syms %.2f | h2 %.2f\n", n,
                acc[0] / every, acc[1] / every, acc[2] / every, acc[3] / every, acc[4] / every, acc[8] / every, acc[5] / every, acc[6] / every, acc[7] / every);
        for (double &a : acc) a = 0;
    }
};
thread_local PhaseClock t_clock;

// DSV2_TRACE bit 1: CPU time (thread clock) the pool tasks of a host phase consume, summed over the streams of a step
struct TaskCpu {
    std::atomic<long long> ns[3] = {{0}, {0}, {0}};
    std::atomic<long long> tasks{0};
    bool on = (trace_mode() & 2) != 0;
    static long long now()
    {
        timespec ts;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return (long long) ts.tv_sec * 1000000000ll + ts.tv_nsec;
    }
    template <class F> void run(int which, F fn)
    {
        if (!on) {
            fn();
            return;
        }
        long long t0 = now();
        fn();
        ns[which] += now() - t0;
        if (which == 2 && (++tasks % 12288) == 0) {
            fprintf(stderr, "[batch] host task CPU per frame: h1a %.1f us, h1b %.1f us, h2 %.1f us\n", ns[0] / 1e3 / 12288, ns[1] / 1e3 / 12288, ns[2] / 1e3 / 12288);
            ns[0] = ns[1] = ns[2] = 0;
        }
    }
};
TaskCpu g_task_cpu;

// The search token.  The motion search runs at a fixed number of wavefronts per SIMD (3 072 persistent workers: three) and is
// bound by how many of those slots it holds for how long; every other kernel of a step lives in the rest of the register file.
// Lockstep groups that reach their search phase together share the slots (each search takes twice as long) and then reach their
// G2 phases together too, with the search slots idle meanwhile: a convoy, and a stable one.  With the token at most
// ONE group has a search in flight (re-measured with the round-5 kernels: without it 8 780 against 8 950 frames/s; the coarse
// levels outside it: 8 810); the others wait with their pre-search work
// (ingest, pyramids) already enqueued, and the searches of the groups follow one another back to back while the rest of
// each step runs beside them.
struct SearchToken { // first come, first served: the group that has waited longest searches next, so the groups keep their rotation
    FifoToken tok;
    void acquire()
    {
        // bounded: a holder that never lets go (a search that hangs, a bug between acquire and release) must not park every
        // other lockstep group of the process silently; the waiter that gives up retires its ticket (batch.h)
        try {
            tok.acquire(std::chrono::seconds(120));
        } catch (const FifoToken::TimedOut &) {
            throw StepFailed{"search token not released within 120 s (another lockstep group's motion search never finished)"};
        }
    }
    void release() { tok.release(); }
};
SearchToken g_search_token;
struct SearchTokenGuard { // releases on every way out of the scope that took the token
    bool held = false;
    void acquire()
    {
        g_search_token.acquire();
        held = true;
    }
    void release()
    {
        if (held) {
            held = false;
            g_search_token.release();
        }
    }
    ~SearchTokenGuard() { release(); }
};

// the plane sections of the packet are assembled on the GPU (DSV2_GPU_ENTROPY=0: the host codes them from the symbol list)
static const bool kGpuEntropy = !(getenv("DSV2_GPU_ENTROPY") && atoi(getenv("DSV2_GPU_ENTROPY")) == 0);
// side streams within a step: bit 0 entropy coder, bit 1 intra filter; unset: the entropy kernels of a SMALL batch (fewer than
// 12 streams: the step is a chain of latency-bound launches on a mostly idle GPU) run beside inverse transform /
// reconstruction / filters, a large batch keeps one chain (no throughput gain there, more host work)
static const int kAuxStreamsEnv = -1; // (re-measured in round 6 with 192-picture launches: 1 -> -0.6 %, 3 -> +0.2 %: a group's chain in parallel with itself buys nothing)
static const bool kEntForceFallback = getenv("DSV2_GPU_ENTROPY_FORCE_FALLBACK") && atoi(getenv("DSV2_GPU_ENTROPY_FORCE_FALLBACK")) != 0; // (tests)
// the quantiser tallies nonzeros per compaction tile while it writes the values (DSV2_FUSED_COUNT=0: separate pass)
static std::atomic<long> g_list_growths{0}; // pictures that had more symbols than their stream's compaction lists (dsv2hip_enc_list_growths)
static const bool kFusedCount = true;

// tests: fail the next step of this process on purpose (dsv2hip_test_fail_next_step): 1 = as a search that did not deliver its
// counters (thrown behind H1a, the search drained), 2 = as a search token that never came (thrown with the step's ingest,
// pyramids and source pre-pass still enqueued on the stream)
static std::atomic<int> g_fail_next_step{0};

// workgroups per (picture, plane) of the entropy coder's chunk kernels; each walks its share of the plane's 1 024-symbol chunks
static const int kEntSlots = getenv("DSV2_ENT_SLOTS") ? atoi(getenv("DSV2_ENT_SLOTS")) : 32; // (192 until round 6: four in five of those workgroups found no chunk)

static void enc_batch_step(Job *jobs, int n);
bool enc_batch_ok(Job *jobs, int n)
{
    try {
        enc_batch_step(jobs, n);
        return true;
    } catch (const StepFailed &e) {
        // (the step's scratch has drained its streams on the way here: ScratchLease)
        fprintf(stderr, "[dsv2hip] encode step of %d stream(s) FAILED: %s; these encoders return no packets from now on\n", n, e.what);
        for (int k = 0; k < n; k++) {
            jobs[k].nbuf = 0;
            if (jobs[k].frame) { // not yet released by the step (dsv_encoder.c:1457 releases it on every path)
                dsv_frame_ref_dec(jobs[k].frame);
                jobs[k].frame = nullptr;
            }
            if (jobs[k].enc->ref) {
                ((EncImpl *) jobs[k].enc->ref)->dead = true;
            }
        }
        return false;
    }
}
void enc_batch(Job *jobs, int n) { (void) enc_batch_ok(jobs, n); }

static void enc_batch_step(Job *jobs, int n)
{
    {
        constexpr int fine_max = 1;
        set_wait_fine(n <= fine_max);
    }
    const int kAuxStreams = kAuxStreamsEnv >= 0 ? kAuxStreamsEnv : (n < 12 ? 1 : 0);
    static bool first_step = true; // (DSV2_TRACE=1 only; a benign race)
    const bool trace_startup = first_step;
    first_step = false;
    if (trace_startup) {
        startup_mark("first step entered");
    }
    bind_device();
    if (trace_startup) {
        startup_mark("device bound (HIP runtime up)");
    }
    t_clock.start();
    for (int k = 0; k < n; k++) {
        Job &jb = jobs[k];
        if (!jb.enc->ref) {
            jb.enc->ref = new EncImpl();
        }
        jb.im = (EncImpl *) jb.enc->ref;
        ensure_ready(jb.enc, jb.im);
        if (k > 0 && (jb.im->dev.w != jobs[0].im->dev.w || jb.im->dev.h != jobs[0].im->dev.h || jb.im->dev.format != jobs[0].im->dev.format ||
                      jb.im->dev.blk_w != jobs[0].im->dev.blk_w || jb.im->dev.blk_h != jobs[0].im->dev.blk_h ||
                      jb.im->dev.pyr_levels != jobs[0].im->dev.pyr_levels)) {
            fatal("dsv2hip_enc_batch: all encoders of a batch must share one picture geometry", __FILE__, __LINE__);
        }
        phase_p0(jb);
    }
    t_clock.lap(0);
    if (trace_startup) {
        startup_mark("encoder instances allocated");
    }
    StageProf &prof = jobs[0].im->dev.prof;
    ScratchLease lease;
    BatchScratch &sc = *lease.sc;
    if (trace_startup) {
        startup_mark("batch scratch + streams ready");
    }
    // The step's kernels run on a stream that belongs to the batch scratch (made once, see ScratchPool::acquire), not on one
    // of the encoders': a step starts and ends with that stream drained, so which stream carried an encoder's previous step
    // does not matter.  (DSV2_SCRATCH_STREAM=0: the first encoder's stream, created with the instance: A/B.)
    hipStream_t bs = use_own_streams() ? sc.main_stream() : jobs[0].im->dev.ensure_stream();
    sc.ensure(n);
    const int nbh = jobs[0].im->dev.nbh, nbv = jobs[0].im->dev.nbv;

    // ---- G1 ----
    // every per-picture helper runs ONCE for the whole batch over a device table of jobs
    CodecDev &dv0 = jobs[0].im->dev;
    const int L = dv0.pyr_levels;
    prof.begin(bs, ST_INGEST);
    std::vector<HmeFrames> hf;
    std::vector<HmeParams> hp;
    std::vector<int> pjobs;
    const IngestJob *d_ing, *d_ingu;
    IngestJob *h_ing = sc.tabs.take<IngestJob>((size_t) n, &d_ing), *h_ingu = sc.tabs.take<IngestJob>((size_t) n, &d_ingu);
    int n_ingu = 0;
    const DPlane *d_ext_y, *d_ext_c;
    DPlane *h_ext_y = sc.tabs.take<DPlane>((size_t) n, &d_ext_y), *h_ext_c = sc.tabs.take<DPlane>(2 * (size_t) n, &d_ext_c);
    const PlanePair *d_pair[DSV_MAX_PYRAMID_LEVELS];
    PlanePair *h_pair[DSV_MAX_PYRAMID_LEVELS];
    const DPlane *d_pext[DSV_MAX_PYRAMID_LEVELS];
    DPlane *h_pext[DSV_MAX_PYRAMID_LEVELS];
    for (int l = 0; l < L; l++) {
        h_pair[l] = sc.tabs.take<PlanePair>(2 * (size_t) n, &d_pair[l]);
        h_pext[l] = sc.tabs.take<DPlane>(2 * (size_t) n, &d_pext[l]);
    }
    const IntraJob *d_intra;
    IntraJob *h_intra = sc.tabs.take<IntraJob>((size_t) n, &d_intra);
    const PlaneOutJob *d_small;
    PlaneOutJob *h_small = sc.tabs.take<PlaneOutJob>((size_t) n, &d_small);
    const BlockStatsJob *d_bsj;
    BlockStatsJob *h_bsj = sc.tabs.take<BlockStatsJob>((size_t) n, &d_bsj);
    int n_ing = 0, n_pyr = 0, n_intra = 0, n_bsj = 0;
    std::function<void()> upload_next;
    {
        // pictures that arrive in host memory: this step's either came up during the previous step (prefetched
        // through host_next) or is uploaded now; the next step's goes up on the copy stream under this step's kernels
        const DFrame &f0 = dv0.pics[0].src;
        const size_t pbytes = (size_t) f0.p[0].w * f0.p[0].h + (size_t) f0.p[1].w * f0.p[1].h + (size_t) f0.p[2].w * f0.p[2].h;
        hipEvent_t waited[4] = {nullptr, nullptr, nullptr, nullptr};
        int nwaited = 0;
        bool any_next = false;
        for (int k = 0; k < n; k++) {
            Job &jb = jobs[k];
            EncImpl *im = jb.im;
            if (!jb.host_planar) {
                continue;
            }
            if (!im->d_stage[0]) {
                HIPCHK(hipMalloc((void **) &im->d_stage[0], pbytes));
                HIPCHK(hipMalloc((void **) &im->d_stage[1], pbytes));
            }
            if (im->staged_src == (const void *) jb.host_planar) {
                im->stage_cur ^= 1;
                bool seen = false;
                for (int e = 0; e < nwaited; e++) {
                    seen = seen || waited[e] == im->staged_ev;
                }
                if (!seen) {
                    HIPCHK(hipStreamWaitEvent(bs, im->staged_ev, 0));
                    if (nwaited < 4) {
                        waited[nwaited++] = im->staged_ev;
                    }
                }
            } else {
                HIPCHK(hipMemcpyAsync(im->d_stage[im->stage_cur], jb.host_planar, pbytes, hipMemcpyHostToDevice, bs));
            }
            im->staged_src = nullptr;
            jb.dev_planar = im->d_stage[im->stage_cur];
        }
        // the next step's pictures go up on the copy stream under this step's kernels.  The calls themselves -- one per stream --
        // cost the host several milliseconds for a large batch: they are made once this step's pre-search work and its search
        // have been handed to the GPU (DSV2_UPLOAD_EARLY=1: before anything else of the step, as up to round 3)
        upload_next = [&jobs, n, pbytes, &sc] {
            bool any = false;
            for (int k = 0; k < n; k++) {
                Job &jb = jobs[k];
                if (jb.host_planar && jb.host_next) {
                    sc.ensure_copy_stream();
                    HIPCHK(hipMemcpyAsync(jb.im->d_stage[jb.im->stage_cur ^ 1], jb.host_next, pbytes, hipMemcpyHostToDevice, sc.copy_stream));
                    jb.im->staged_src = jb.host_next;
                    jb.im->staged_ev = sc.copy_done;
                    any = true;
                }
            }
            if (any) {
                HIPCHK(hipEventRecord(sc.copy_done, sc.copy_stream));
            }
        };
        constexpr bool upload_early = false;
        if (upload_early) {
            upload_next();
            upload_next = nullptr;
        }
        (void) any_next;
    }
    for (int k = 0; k < n; k++) {
        Job &jb = jobs[k];
        CodecDev &dv = jb.im->dev;
        PicSet &cur = dv.pics[jb.im->cur], &ref = dv.pics[jb.im->cur ^ 1];
        if (jb.d.params.do_psy != jobs[0].d.params.do_psy) {
            fatal("dsv2hip_enc_batch: all encoders of a batch must share one do_psy setting", __FILE__, __LINE__);
        }
        if (jb.frame) {
            dframe_upload(&cur.src, jb.frame, bs);
        } else {
            IngestJob &ij = jb.im->input_uyvy && !jb.from_frame ? h_ingu[n_ingu++] : h_ing[n_ing++];
            ij.src = jb.dev_planar;
            for (int c = 0; c < 3; c++) {
                ij.dst[c] = cur.src.p[c];
            }
        }
        h_ext_y[k] = cur.src.p[0];
        h_ext_c[2 * k] = cur.src.p[1];
        h_ext_c[2 * k + 1] = cur.src.p[2];
        for (int l = 0; l < L; l++) { // mk_pyramid, dsv_encoder.c:493
            h_pair[l][n_pyr] = PlanePair{l ? cur.src_pyr[l - 1].p[0] : cur.src.p[0], cur.src_pyr[l].p[0]};
            h_pext[l][n_pyr] = cur.src_pyr[l].p[0];
        }
        n_pyr++;
        if (jb.d.params.has_ref && !ref.recon_pyr_valid) {
            for (int l = 0; l < L; l++) {
                h_pair[l][n_pyr] = PlanePair{l ? ref.recon_pyr[l - 1].p[0] : ref.recon.p[0], ref.recon_pyr[l].p[0]};
                h_pext[l][n_pyr] = ref.recon_pyr[l].p[0];
            }
            n_pyr++;
            ref.recon_pyr_valid = true;
        }
        // the block analysis of an intra picture, written straight into the host's pinned array (a P
        // frame that H1 flips to intra gets its analysis then)
        if (!jb.d.params.has_ref) {
            for (int c = 0; c < 3; c++) {
                h_intra[n_intra].src.p[c] = cur.src.p[c];
            }
            h_intra[n_intra].out = dv.h_intra;
            n_intra++;
        }
        h_small[k] = PlaneOutJob{cur.src_pyr[L - 1].p[0], dv.h_small};
        jb.bstats = nullptr;
        if (jb.d.params.has_ref) { // the controller's sums over the field this step's search is about to produce
            BlockStatsJob &bj = h_bsj[n_bsj];
            bj.mvs = dv.d_mvf[0];
            bj.counters = dv.d_counters;
            bj.map_in = jb.im->map_valid ? dv.d_intra_map[jb.im->map_cur] : nullptr;
            bj.map_out = dv.d_intra_map[jb.im->map_cur ^ 1];
            bj.host_mvs = dv.h_mvs;
            bj.out = sc.d_bstats + (size_t) n_bsj * BS_WORDS;
            bj.b2sr = mv_cost_b2sr(&jb.d.params, jb.enc->prev_quant);
            bj.rc_mode = jb.enc->rc_mode;
            jb.bstats = sc.h_bstats + (size_t) n_bsj * BS_WORDS;
            n_bsj++;
        }
    }
    sc.tabs.upload(bs);
    {
        const DFrame &f0 = dv0.pics[0].src;
        {
            bool wide = f0.p[0].w <= 2048 && (f0.p[0].w % 16) == 0 && (f0.p[1].w % 16) == 0 && (f0.p[2].w % 16) == 0 &&
                        (((size_t) f0.p[0].w * f0.p[0].h) % 16) == 0 && (((size_t) f0.p[1].w * f0.p[1].h) % 16) == 0;
            for (int k = 0; k < n_ing && wide; k++) {
                wide = ((uintptr_t) h_ing[k].src % 16) == 0;
            }
            if (wide) {
                ingest_batch16(bs, d_ing, n_ing, f0.p[0].h + f0.p[1].h + f0.p[2].h);
            } else {
                ingest_batch(bs, d_ing, n_ing, f0.p[0].w, f0.p[0].h + f0.p[1].h + f0.p[2].h);
            }
        }
        ingest_uyvy_batch(bs, d_ingu, n_ingu, f0.p[0].w, f0.p[0].h);
        extend_planes(bs, d_ext_y, n, f0.p[0].w, f0.p[0].h);
        extend_planes(bs, d_ext_c, 2 * n, f0.p[1].w, f0.p[1].h);
        for (int l = 0; l < L; l++) {
            const DPlane &lp = dv0.pics[0].src_pyr[l].p[0];
            ds2x_planes4(bs, d_pair[l], n_pyr, lp.w, lp.h); // (picture sets are dframe_alloc'd: aligned)
            extend_planes(bs, d_pext[l], n_pyr, lp.w, lp.h);
        }
        planes_to_host_batch(bs, d_small, n, dv0.pics[0].src_pyr[L - 1].p[0].h);
        intra_analysis_batch(bs, d_intra, n_intra, analysis_params(dv0, jobs[0].d.params.do_psy));
    }
    for (int k = 0; k < n; k++) {
        Job &jb = jobs[k];
        CodecDev &dv = jb.im->dev;
        PicSet &cur = dv.pics[jb.im->cur], &ref = dv.pics[jb.im->cur ^ 1];
        if (jb.d.params.has_ref) { // motion_est (dsv_encoder.c:653)
            HmeFrames f;
            f.src[0] = cur.src.p[0];
            f.ref[0] = ref.recon.p[0];
            f.ogr[0] = ref.src.p[0];
            for (int l = 0; l < dv.pyr_levels; l++) {
                f.src[l + 1] = cur.src_pyr[l].p[0];
                f.ref[l + 1] = ref.recon_pyr[l].p[0];
                f.ogr[l + 1] = ref.src_pyr[l].p[0];
            }
            for (int c = 0; c < 2; c++) {
                f.srcc[c] = cur.src.p[c + 1];
                f.refc[c] = ref.recon.p[c + 1];
            }
            for (int l = 0; l <= dv.pyr_levels; l++) {
                f.mvf[l] = dv.d_mvf[l];
            }
            f.ref_mvf = ref.has_final_mvs ? ref.d_final_mvs : nullptr;
            f.counters = dv.d_counters;
            f.src_stats = dv.d_src_stats;
            f.l0_pre = dv.d_l0_pre;
            f.host_mvs = nullptr; // (the field reaches the host through k_block_stats_b right behind the search: BlockStatsJob::host_mvs)
            f.host_counters = dv.h_counters;
            dv.h_counters[7] = -1; // overwritten with 0 by the search's last row (1: a row timed out); -1 left = it never finished
            dv.h_counters[kHmeHostTailWord] = 0;
            HmeParams h;
            h.a = analysis_params(dv, jb.d.params.do_psy);
            h.effort = jb.d.params.effort;
            h.lossless = jb.d.params.lossless;
            h.quant = jb.enc->prev_quant;
            h.skip_block_thresh = jb.enc->skip_block_thresh;
            h.pyr_levels = dv.pyr_levels;
            hf.push_back(f);
            hp.push_back(h);
            pjobs.push_back(k);
            jb.ran_hme = 1;
        }
    }
    if (!pjobs.empty()) { // job table, clears, the source blocks' statistics: nothing the token is for
        hme_run_batch(bs, hf.data(), hp.data(), (int) pjobs.size(), sc.h_hme, sc.d_hme, nullptr, -1, 0, HME_PREPARE);
    }
    prof.end(bs, ST_INGEST, n);
    // (only launches that keep the chip's search slots -- 3 072 persistent workers -- full for most of their length take the token: a row-pipelined
    // launch ramps up and down over one picture's critical path, ~2 ms whatever the batch, and launches of a few dozen
    // pictures hide each other's ramps when they overlap)
    // (Until the end of round 4 the bound was 8 192 rows.  Re-measured with the persistent kernels: 4 groups of 3 264 rows -- 192
    // streams -- gain 3 % from the token, 6 880 -> 7 090 frames/s, 4 groups of 6 528 rows 3.4 %; at 2 176 rows a group the token
    // costs 3 %, at 816 it makes no difference.  3 072 rows it is: one set of the persistent workers.)
    constexpr int min_rows = 3072;
    const bool searching = !pjobs.empty() && (int) pjobs.size() * nbv >= min_rows;
    SearchTokenGuard token;
    if (g_fail_next_step.load() == 2 && g_fail_next_step.exchange(0) == 2) {
        throw StepFailed{"search token not released (test hook)"};
    }
    if (searching) {
        // the token is for the search alone: what precedes it on the stream (this step's upload, ingest, pyramids) is waited
        // for BEFORE taking it, or the holder would sit on the token while its own pictures are still crossing PCIe.
        // (Running the coarse levels -- launches that cannot fill the slots -- outside the token, beside another group's
        // level-0 launch, was tried: hme_run_batch takes a level range for it; no gain, they slow the holder's launch.)
        stream_wait(bs);
        t_clock.mark("pre-search-drained", n);
    }
    // DSV2_COARSE_OUTSIDE: the coarse levels -- five launches that are dependency chains and cannot fill the chip -- run BEFORE the
    // token is taken, beside whatever level-0 launch holds it; the token then covers the level-0 launch alone.  1: queue for the
    // token at once (the level-0 launch follows the coarse levels on the stream); 2: when the coarse levels have finished.
    constexpr int coarse_outside = 0;
    const bool split_levels = searching && coarse_outside && dv0.pyr_levels >= 1;
    int nfronts_coarse = 0;
    if (split_levels) {
        prof.begin(bs, ST_HME);
        nfronts_coarse = hme_run_batch(bs, hf.data(), hp.data(), (int) pjobs.size(), sc.h_hme, sc.d_hme, &prof, -1, 1, HME_LEVELS);
        if (coarse_outside >= 2) {
            stream_wait(bs);
        }
        t_clock.mark("coarse-levels", n);
    }
    if (searching) {
        token.acquire(); // (released once the search has drained, below)
        t_clock.mark("token", n);
    }
    if (!pjobs.empty()) {
        if (!split_levels) {
            prof.begin(bs, ST_HME);
        }
        int nfronts = nfronts_coarse + hme_run_batch(bs, hf.data(), hp.data(), (int) pjobs.size(), sc.h_hme, sc.d_hme, &prof, split_levels ? 0 : -1, 0, HME_LEVELS);
        prof.end(bs, ST_HME, (int) pjobs.size(), nfronts); // launches = the per-level search kernels
        HIPCHK(hipMemsetAsync(sc.d_bstats, 0, (size_t) n_bsj * BS_WORDS * sizeof(int), bs));
        block_stats_batch(bs, d_bsj, n_bsj, nbh, nbv);
        HIPCHK(hipMemcpyAsync(sc.h_bstats, sc.d_bstats, (size_t) n_bsj * BS_WORDS * sizeof(int), hipMemcpyDeviceToHost, bs));
    }
    if (upload_next) {
        upload_next();
    }
    t_clock.lap(1);
    if (searching) {
        // The token is passed on when the level-0 launch has handed out its last block row (the kernel says so in the first
        // stream's pinned counter block): what is left of it is a tail of draining wavefronts -- one row's walk, ~2 ms --
        // whose freed slots the next group's search can take.  (Launch-per-front form, or a launch that ends first: the
        // stream's completion.)
        volatile int *tail = &jobs[pjobs[0]].im->dev.h_counters[kHmeHostTailWord];
        // DSV2_SEARCH_EARLY_RELEASE: 1 (default) at the tail; 0 when the launch has finished; 2 right after it was enqueued
        constexpr int early = 1;
        if (early == 0) {
            stream_wait(bs);
        } else if (early == 1) {
            while (!*tail && hipStreamQuery(bs) == hipErrorNotReady) {
                timespec ts = {0, 100000};
                nanosleep(&ts, nullptr);
            }
        }
        token.release();
        t_clock.mark("token-released", n);
    }
    stream_wait(bs);
    t_clock.lap(2);
    for (int k = 0; k < n; k++) {
        if (jobs[k].frame) {
            dsv_frame_ref_dec(jobs[k].frame); // the caller's pixels are in HBM now (dsv_encoder.c:1457)
            jobs[k].frame = nullptr;
        }
    }

    // ---- H1 ----
    parallel_for(n, [&](int k) { g_task_cpu.run(0, [&] { phase_h1a(jobs[k]); }); });
    if (g_fail_next_step.load() == 1 && g_fail_next_step.exchange(0) == 1) {
        jobs[0].failed = 2;
    }
    for (int k = 0; k < n; k++) {
        if (jobs[k].failed) {
            throw StepFailed{jobs[k].failed == 1 ? "motion estimation row pipeline timed out (a row waited > 4 s for the row above)"
                                                 : "motion estimation did not deliver its counters (search incomplete)"};
        }
    }
    {
        // P frames flipped to intra by the scene-change test: their block analysis is due now
        const IntraJob *d_late;
        IntraJob *h_late = sc.tabs.take<IntraJob>((size_t) n, &d_late);
        int n_late = 0;
        for (int k = 0; k < n; k++) {
            Job &jb = jobs[k];
            if (jb.ran_hme && !jb.d.params.has_ref) {
                PicSet &cur = jb.im->dev.pics[jb.im->cur];
                for (int c = 0; c < 3; c++) {
                    h_late[n_late].src.p[c] = cur.src.p[c];
                }
                h_late[n_late].out = jb.im->dev.h_intra;
                n_late++;
            }
        }
        if (n_late) {
            sc.tabs.upload(bs);
            intra_analysis_batch(bs, d_late, n_late, analysis_params(dv0, jobs[0].d.params.do_psy));
            stream_wait(bs);
        }
    }
    // The pictures that stay P frames: their side information is coded on the device (k_side_info), which also finalises
    // the motion field and forms the block flag bytes in HBM for G2.  It is enqueued now, G2 right behind it; the host
    // assembles the P packets' heads (H1b) from the pinned sub-streams while G2 runs.  Intra pictures (1 in a GOP) keep
    // the host coders, whose flag bytes G2 needs: their H1b runs before G2 is built.
    std::vector<int> p_jobs, i_jobs;
    for (int k = 0; k < n; k++) {
        jobs[k].side_out = nullptr;
        jobs[k].side_info = nullptr;
        (jobs[k].d.params.has_ref ? p_jobs : i_jobs).push_back(k);
    }
    if (!p_jobs.empty()) {
        const SideJob *d_side;
        SideJob *h_sj = sc.tabs.take<SideJob>(p_jobs.size(), &d_side);
        for (size_t q = 0; q < p_jobs.size(); q++) {
            Job &jb = jobs[p_jobs[q]];
            CodecDev &dv = jb.im->dev;
            SideJob &sj = h_sj[q];
            sj.raw = dv.d_mvf[0];
            sj.final_mvs = dv.pics[jb.im->cur].d_final_mvs;
            sj.bd = dv.d_blockdata;
            sj.out = sc.h_side + q * SIDE_IMG_BYTES;
            sj.info = sc.h_side_info + q * SIDE_INFO_WORDS;
            sj.inv_stable = jb.stats[ST_STABLE] != 0;
            sj.inv_mode = jb.stats[ST_MODE] != 0;
            sj.inv_eprm = jb.stats[ST_EPRM] != 0;
            jb.side_out = sj.out;
            jb.side_info = sj.info;
        }
        sc.tabs.upload(bs);
        side_info_batch(bs, d_side, (int) p_jobs.size(), nbh, nbv);
        HIPCHK(hipEventRecord(sc.ev_side, bs));
    }
    parallel_for((int) i_jobs.size(), [&](int q) { g_task_cpu.run(1, [&] { phase_h1b(jobs[i_jobs[(size_t) q]]); }); });
    t_clock.lap(3);

    // ---- G2 ----
    // streams are ordered by (frame type, lossless): the transform / quantiser kernels are specialised
    // on those, so each class is one set of launches over its slice of the job tables
    std::vector<int> order((size_t) n);
    for (int k = 0; k < n; k++) {
        order[(size_t) k] = k;
    }
    auto cls = [&](int k) { return jobs[k].d.params.has_ref * 2 + jobs[k].d.params.lossless; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cls(a) > cls(b); });
    const CopyJob *d_copy;
    CopyJob *h_copy = sc.tabs.take<CopyJob>((size_t) n, &d_copy);
    const PlaneJob *d_py, *d_pc;
    PlaneJob *h_py = sc.tabs.take<PlaneJob>((size_t) n, &d_py), *h_pc = sc.tabs.take<PlaneJob>(2 * (size_t) n, &d_pc);
    const CompactJob *d_comp;
    CompactJob *h_comp = sc.tabs.take<CompactJob>((size_t) n, &d_comp);
    const EntJob *d_ent;
    EntJob *h_ent = sc.tabs.take<EntJob>((size_t) n, &d_ent);
    const DPlane *d_rext_y, *d_rext_c;
    DPlane *h_rext_y = sc.tabs.take<DPlane>((size_t) n, &d_rext_y), *h_rext_c = sc.tabs.take<DPlane>(2 * (size_t) n, &d_rext_c);
    // host -> device hand-over of what H1 decided: per stream the transmitted motion field and the
    // block flag bytes, packed into ONE pinned buffer and shipped with one copy
    const size_t nb0 = dv0.nblocks();
    const size_t mv_bytes = nb0 * sizeof(DSV_MV), slot = (nb0 + 15) & ~(size_t) 15; // (a staging slot: one picture's flag bytes)
    sc.ensure_stage(slot * (i_jobs.size() + 1));
    const CopyJob *d_mvcopy;
    CopyJob *h_mvcopy = sc.tabs.take<CopyJob>((size_t) n, &d_mvcopy);
    int nP = 0, nI = 0, n_rext = 0, n_mvcopy = 0, n_slots = 0, n_copy = 0;
    bool any_filter = false;
    for (int i = 0; i < n; i++) {
        int k = order[(size_t) i];
        Job &jb = jobs[k];
        CodecDev &dv = jb.im->dev;
        PicSet &cur = dv.pics[jb.im->cur], &ref = dv.pics[jb.im->cur ^ 1];
        const DSV_PARAMS *p = &jb.d.params;
        size_t nb = dv.nblocks();
        // the working ("residual") picture starts as a copy of the padded source (dsv_encoder.c:1292) -- of an INTRA picture:
        // its forward transform reads the working picture.  A P picture's predict + subtract reads the source itself and
        // writes the residual over every block of the working picture (McJob::src), so nothing needs copying there.
        if (!p->has_ref) {
            h_copy[n_copy++] = CopyJob{cur.src.alloc, cur.recon.alloc, cur.src.bytes};
        }
        cur.recon_pyr_valid = false;
        const uint8_t *d_bd = dv.d_blockdata; // P: formed on the device (k_side_info)
        if (!p->has_ref) {                      // I: the host coders' flag bytes, staged and shipped with one copy
            uint8_t *h_slot = sc.h_stage + slot * (size_t) n_slots;
            d_bd = sc.d_stage + slot * (size_t) n_slots;
            memcpy(h_slot, jb.enc->blockdata, nb);
            n_slots++;
        }
        McJob mj;
        mj.mvs = cur.d_final_mvs;
        mj.bd = d_bd;
        mj.p = dv.mc_params(p->temporal_mc, p->lossless);
        for (int c = 0; c < 3; c++) {
            mj.ref.p[c] = ref.recon.p[c];
            mj.pred.p[c] = dv.pred.p[c];
            mj.res.p[c] = cur.recon.p[c];
            mj.src[c] = cur.src.p[c].data;
        }
        if (p->has_ref) {
            // the motion field as transmitted (written into cur.d_final_mvs by k_side_info): used by MC now and as temporal
            // candidates of the next frame
            cur.has_final_mvs = true;
            mj.f = make_filter_params(mj.p, jb.d.quant, jb.inter_filter, jb.enc->vidmeta.inter_sharpen);
            any_filter = any_filter || !p->lossless;
            sc.h_mc[nP++] = mj;
        } else {
            mj.f = make_filter_params(mj.p, jb.d.quant, 1, 0);
            if (jb.enc->do_intra_filter && !p->lossless) {
                sc.h_mc[n + nI++] = mj;
            }
            if (jb.ran_hme) {
                // the search result of a frame that H1 flipped to intra still serves as the next frame's
                // temporal candidates (dsv_encoder.c:680, hme.c:1651)
                h_mvcopy[n_mvcopy++] = CopyJob{dv.d_mvf[0], cur.d_final_mvs, mv_bytes};
                cur.has_final_mvs = true;
            }
        }
        for (int c = 0; c < 3; c++) {
            PlaneJob &pj = c ? h_pc[2 * i + c - 1] : h_py[i];
            pj.pic = cur.recon.p[c];
            pj.coefs = dv.coefs[c];
            for (int t = 0; t < 3; t++) {
                pj.t[t] = c ? dv.scratch_uv[c - 1].t[t] : dv.scratch.t[t];
            }
            pj.bd = d_bd;
            pj.qv = dv.qv + dv.qv_off[c];
            pj.tile_count = kFusedCount ? dv.comp.tile_count : nullptr;
            pj.qv_base = (unsigned) dv.qv_off[c];
            pj.mvs = cur.d_final_mvs;
            quant_steps(&pj, dv.quant_cfg(c, p->has_ref, p->lossless, p->do_psy, nullptr), jb.d.quant);
        }
        h_comp[i] = dv.comp.job(dv.qv, dv.qv_off[3]);
        h_comp[i].total = sc.d_totals + i;
        if (kGpuEntropy) {
            // the symbols stay in HBM; what comes back is the finished plane sections (pinned mirror: 1 MB, far above
            // any 1080p picture at sane quality -- larger ones are fetched by a copy)
            dv.ent.ensure(dv.comp.list_cap, 4u << 20, 1u << 20);
            h_ent[i] = dv.ent.job(dv.comp.d_pos, dv.comp.d_val, sc.d_totals + i, sc.d_ll + 3 * i, dv.comp.list_cap);
        } else {
            dv.ensure_host_syms(dv.qv_off[3] / 8); // P pictures fit; the first intra picture grows it (one fallback copy)
            h_comp[i].host_pos = dv.h_pos;
            h_comp[i].host_val = dv.h_val;
            h_comp[i].host_cap = (int) dv.h_sym_cap;
        }
        if (jb.enc->frame_callback || (p->is_ref && jb.enc->gop != DSV_GOP_INTRA)) {
            h_rext_y[n_rext] = cur.recon.p[0];
            h_rext_c[2 * n_rext] = cur.recon.p[1];
            h_rext_c[2 * n_rext + 1] = cur.recon.p[2];
            n_rext++;
        }
    }
    sc.tabs.upload(bs);
    HIPCHK(hipMemcpyAsync(sc.d_mc, sc.h_mc, 2 * (size_t) n * sizeof(McJob), hipMemcpyHostToDevice, bs));
    if (n_slots) {
        HIPCHK(hipMemcpyAsync(sc.d_stage, sc.h_stage, slot * (size_t) n_slots, hipMemcpyHostToDevice, bs));
    }
    copy_linear_batch(bs, d_mvcopy, n_mvcopy, mv_bytes);
    copy_linear_batch(bs, d_copy, n_copy, dv0.pics[0].src.bytes);
    prof.begin(bs, ST_PREDICT);
    mc_sub_pred_batch(bs, sc.d_mc, nP, nbh, nbv, dv0.blk_w, dv0.blk_h, DSV_FORMAT_H_SHIFT(dv0.format) == 1 && DSV_FORMAT_V_SHIFT(dv0.format) == 1);
    prof.end(bs, ST_PREDICT, nP);
    struct Slice {
        int first, count, isP, lossless;
    };
    std::vector<Slice> slices;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && cls(order[(size_t) j]) == cls(order[(size_t) i])) {
            j++;
        }
        slices.push_back(Slice{i, j - i, jobs[order[(size_t) i]].d.params.has_ref, jobs[order[(size_t) i]].d.params.lossless});
        i = j;
    }
    const int do_psy = jobs[0].d.params.do_psy;
    prof.begin(bs, ST_FWD_SBT);
    for (const Slice &sl : slices) {
        sbt_forward_jobs(bs, d_py + sl.first, sl.count, dv0.cw[0], dv0.ch[0], 0, sl.isP, sl.lossless, nbh, nbv);
        sbt_forward_jobs(bs, d_pc + 2 * sl.first, 2 * sl.count, dv0.cw[1], dv0.ch[1], 1, sl.isP, sl.lossless, nbh, nbv);
    }
    prof.end(bs, ST_FWD_SBT, n);
    prof.begin(bs, ST_QUANT);
    DSV2_LAUNCH(k_grab_ll, dim3((n + 63) / 64), dim3(64), 0, bs, d_py, d_pc, n, sc.d_ll);
    HIPCHK(hipMemcpyAsync(sc.h_ll, sc.d_ll, 3 * (size_t) n * sizeof(int32_t), hipMemcpyDeviceToHost, bs));
    for (const Slice &sl : slices) {
        quant_jobs(bs, d_py + sl.first, sl.count, dv0.quant_cfg(0, sl.isP, sl.lossless, do_psy, nullptr));
        quant_jobs(bs, d_pc + 2 * sl.first, 2 * sl.count, dv0.quant_cfg(1, sl.isP, sl.lossless, do_psy, nullptr));
    }
    compact_jobs(bs, d_comp, n, dv0.qv_off[3], kFusedCount);
    HIPCHK(hipMemcpyAsync(sc.h_totals, sc.d_totals, (size_t) n * sizeof(int), hipMemcpyDeviceToHost, bs));
    if (kGpuEntropy) {
        if (kAuxStreams & 1) {
            sc.ensure_aux();
            sc.fork(bs, 0);
            entropy_gpu_jobs(sc.aux[0], d_ent, n, ent_geom(dv0.qv_off, dv0.scan), kEntSlots);
        } else {
            entropy_gpu_jobs(bs, d_ent, n, ent_geom(dv0.qv_off, dv0.scan), kEntSlots);
        }
    }
    prof.end(bs, ST_QUANT, n);
    prof.begin(bs, ST_INV_SBT);
    for (const Slice &sl : slices) {
        sbt_inverse_jobs(bs, d_py + sl.first, sl.count, dv0.cw[0], dv0.ch[0], 0, sl.isP, sl.lossless, nbh, nbv);
        sbt_inverse_jobs(bs, d_pc + 2 * sl.first, 2 * sl.count, dv0.cw[1], dv0.ch[1], 1, sl.isP, sl.lossless, nbh, nbv);
    }
    prof.end(bs, ST_INV_SBT, n);
    prof.begin(bs, ST_RECON_FILTER);
    const bool side_intra = (kAuxStreams & 2) && nI > 0 && nP > 0; // two latency-bound sweeps over disjoint pictures: side by side
    if (side_intra) {
        sc.ensure_aux();
        sc.fork(bs, 1);
        intra_filter_batch(sc.aux[1], sc.d_mc + n, nI, dv0.w, dv0.h);
    } else {
        intra_filter_batch(bs, sc.d_mc + n, nI, dv0.w, dv0.h);
    }
    mc_add_res_batch(bs, sc.d_mc, nP, nbh, nbv, any_filter, dv0.w, dv0.h, dv0.blk_w, dv0.blk_h);
    if (side_intra) {
        sc.join(bs, 1);
    }
    prof.end(bs, ST_RECON_FILTER, nP + nI);
    prof.begin(bs, ST_EXTEND);
    extend_planes(bs, d_rext_y, n_rext, dv0.pics[0].recon.p[0].w, dv0.pics[0].recon.p[0].h);
    extend_planes(bs, d_rext_c, 2 * n_rext, dv0.pics[0].recon.p[1].w, dv0.pics[0].recon.p[1].h);
    prof.end(bs, ST_EXTEND, n_rext);
    if (kGpuEntropy && (kAuxStreams & 1)) {
        sc.join(bs, 0);
    }
    t_clock.lap(4);
    if (!p_jobs.empty()) { // H1b of the P pictures, under G2's kernels: header + the sub-streams the device coded
        event_wait(sc.ev_side);
        parallel_for((int) p_jobs.size(), [&](int q) { g_task_cpu.run(1, [&] { phase_h1b(jobs[p_jobs[(size_t) q]]); }); });
    }
    t_clock.lap(8);
    stream_wait(bs);
    t_clock.lap(5);
    bool late_copy = false;
    const bool force_redo = getenv("DSV2_COMPACT_REDO") && atoi(getenv("DSV2_COMPACT_REDO"));
    for (int k = 0; k < n; k++) {
        Job &jb = jobs[k];
        CodecDev &dv = jb.im->dev;
        int ti = (int) (std::find(order.begin(), order.end(), k) - order.begin()); // this stream's table slot
        jb.nsym = sc.h_totals[ti];
        for (int c = 0; c < 3; c++) {
            dv.h_ll[c] = sc.h_ll[3 * ti + c];
        }
        jb.gpu_bytes = nullptr;
        bool need_syms = !kGpuEntropy;
        // More symbols than this stream's compaction lists hold (they start at half the worst case, at least 65 536 symbols: ensure_ready): the lists are
        // enlarged to the worst case for good, and this picture's symbols worked out again -- predict + subtract (or the source
        // copy of an intra picture) into a spare working picture, forward transform, quantiser, compaction: the same kernels
        // on the same operands, so the same symbols -- which the host then codes.  The reconstruction is untouched.
        const bool overflow = (size_t) jb.nsym > dv.comp.list_cap || force_redo; // (DSV2_COMPACT_REDO=1: every picture, a test switch)
        if (overflow) {
            if (trace_mode() & 2) {
                fprintf(stderr, "[batch] stream %d: %d symbols > compaction lists of %zu: redone\n", k, jb.nsym, dv.comp.list_cap);
            }
            const bool isP = jb.d.params.has_ref, lossless = jb.d.params.lossless;
            PicSet &cur = dv.pics[jb.im->cur];
            g_list_growths += (size_t) jb.nsym > dv.comp.list_cap;
            dv.comp.grow_lists(dv.qv_off[3]);
            if (kGpuEntropy) {
                dv.ent.ensure(dv.comp.list_cap, 4u << 20, 1u << 20);
            }
            DFrame &tmp = sc.redo_frame(dv.format, dv.w, dv.h);
            static_assert(sizeof(McJob) + sizeof(CopyJob) + 3 * sizeof(PlaneJob) + sizeof(CompactJob) + 6 * 16 <= 4096,
                          "a redone picture's tables outgrow the 4 KB a stream BatchScratch::ensure sets aside for them");
            const McJob *d_m2;
            McJob *h_m2 = sc.tabs.take<McJob>(1, &d_m2);
            const CopyJob *d_c2;
            CopyJob *h_c2 = sc.tabs.take<CopyJob>(1, &d_c2);
            const PlaneJob *d_y2, *d_uv2;
            PlaneJob *h_y2 = sc.tabs.take<PlaneJob>(1, &d_y2), *h_uv2 = sc.tabs.take<PlaneJob>(2, &d_uv2);
            const CompactJob *d_k2;
            CompactJob *h_k2 = sc.tabs.take<CompactJob>(1, &d_k2);
            if (isP) {
                *h_m2 = sc.h_mc[ti]; // (P pictures lead the sorted order: table slot = index among the P jobs)
                for (int c = 0; c < 3; c++) {
                    h_m2->res.p[c] = tmp.p[c];
                }
            } else {
                *h_c2 = CopyJob{cur.src.alloc, tmp.alloc, cur.src.bytes};
            }
            *h_y2 = h_py[ti];
            h_y2->pic = tmp.p[0];
            for (int c = 1; c < 3; c++) {
                h_uv2[c - 1] = h_pc[2 * ti + c - 1];
                h_uv2[c - 1].pic = tmp.p[c];
            }
            dv.ensure_host_syms((size_t) jb.nsym);
            *h_k2 = dv.comp.job(dv.qv, dv.qv_off[3]);
            h_k2->total = sc.d_totals + ti;
            sc.tabs.upload(bs);
            if (isP) {
                mc_sub_pred_batch(bs, d_m2, 1, nbh, nbv, dv0.blk_w, dv0.blk_h, DSV_FORMAT_H_SHIFT(dv0.format) == 1 && DSV_FORMAT_V_SHIFT(dv0.format) == 1);
            } else {
                copy_linear_batch(bs, d_c2, 1, cur.src.bytes);
            }
            sbt_forward_jobs(bs, d_y2, 1, dv.cw[0], dv.ch[0], 0, isP, lossless, nbh, nbv);
            sbt_forward_jobs(bs, d_uv2, 2, dv.cw[1], dv.ch[1], 1, isP, lossless, nbh, nbv);
            quant_jobs(bs, d_y2, 1, dv.quant_cfg(0, isP, lossless, do_psy, nullptr));
            quant_jobs(bs, d_uv2, 2, dv.quant_cfg(1, isP, lossless, do_psy, nullptr));
            compact_jobs(bs, d_k2, 1, dv.qv_off[3], kFusedCount);
            need_syms = true;
            late_copy = true;
        }
        if (kGpuEntropy && !overflow) {
            const int *info = dv.ent.host_info;
            if ((info[0] & ENT_FALLBACK_MASK) || kEntForceFallback) {
                need_syms = true; // (state outside the tabulated range / no room: code this picture on the host)
                if (trace_mode() & 2) {
                    fprintf(stderr, "[batch] stream %d: GPU entropy coder fell back (flags %d)\n", k, info[0]);
                }
            } else {
                for (int c = 0; c < 3; c++) {
                    jb.gpu_plane_bytes[c] = (unsigned) info[ENT_INFO_PBYTES + c];
                }
                if (info[0] & ENT_NOT_MIRRORED) { // finished, but larger than the pinned mirror
                    jb.im->big_bytes.resize((size_t) info[ENT_INFO_TOTAL]);
                    HIPCHK(hipMemcpyAsync(jb.im->big_bytes.data(), dv.ent.out, (size_t) info[ENT_INFO_TOTAL], hipMemcpyDeviceToHost, bs));
                    late_copy = true;
                    jb.gpu_bytes = jb.im->big_bytes.data();
                } else {
                    jb.gpu_bytes = dv.ent.host_out;
                }
            }
        }
        // the symbols are needed on the host: copied when the mirror did not hold them (always so with the GPU coder on)
        if (need_syms && jb.nsym > 0 && (kGpuEntropy || overflow || (size_t) jb.nsym > dv.h_sym_cap)) {
            if (trace_mode() & 2) {
                fprintf(stderr, "[batch] stream %d: %d symbols > pinned mirror of %zu, copying\n", k, jb.nsym, dv.h_sym_cap);
            }
            dv.ensure_host_syms((size_t) jb.nsym);
            HIPCHK(hipMemcpyAsync(dv.h_pos, dv.comp.d_pos, (size_t) jb.nsym * sizeof(uint32_t), hipMemcpyDeviceToHost, bs));
            HIPCHK(hipMemcpyAsync(dv.h_val, dv.comp.d_val, (size_t) jb.nsym * sizeof(int32_t), hipMemcpyDeviceToHost, bs));
            late_copy = true;
        }
    }
    if (late_copy) {
        stream_wait(bs);
    }
    t_clock.lap(6);
    prof.collect();

    // ---- H2 ----
    for (int k = 0; k < n; k++) {
        Job &jb = jobs[k];
        if (jb.enc->frame_callback) { // before the picture sets swap
            CodecDev &dv = jb.im->dev;
            PicSet &cur = dv.pics[jb.im->cur];
            DSV_FRAME *orig = dsv_mk_frame(dv.format, dv.w, dv.h, 1), *rec = dsv_mk_frame(dv.format, dv.w, dv.h, 1);
            dframe_download_full(&cur.src, orig, bs);
            dframe_download_full(&cur.recon, rec, bs);
            stream_wait(bs);
            jb.enc->frame_callback(&jb.enc->vidmeta, orig, rec);
            dsv_frame_ref_dec(orig);
            dsv_frame_ref_dec(rec);
        }
    }
    parallel_for(n, [&](int k) { g_task_cpu.run(2, [&] { phase_h2(jobs[k]); }); });
    if (trace_startup) {
        startup_mark("first step done");
    }
    t_clock.lap(7);
    t_clock.done(n);
}

Coalescer<Job> g_enc_queue; // dsv_enc callers share lockstep steps (batch.h)

} // namespace



extern "C" {

void dsv_enc_init(DSV_ENCODER *enc) // dsv_encoder.c:1319
{
    memset(enc, 0, sizeof(*enc));
    enc->prev_gop = (DSV_FNUM) -1;
    enc->quality = 80;
    enc->gop = 48;
    enc->effort = DSV_MAX_EFFORT;
    enc->rc_mode = DSV_RATE_CONTROL_CRF;
    enc->bitrate = INT_MAX;
    enc->min_q_step = 4;
    enc->max_q_step = 1;
    enc->min_quality = enc->quality - DSV_USER_QUAL_TO_RC_QUAL(5);
    enc->max_quality = DSV_RC_QUAL_MAX;
    enc->min_I_frame_quality = enc->quality - DSV_USER_QUAL_TO_RC_QUAL(2);
    enc->prev_chaos = -1;
    enc->prev_complexity = -1;
    enc->curr_complexity = -1;
    enc->intra_pct_thresh = 90;
    enc->stable_refresh = 24;
    enc->scene_change_pct = 85;
    enc->do_scd = 1;
    enc->variable_i_interval = 1;
    enc->block_size_override_x = -1;
    enc->block_size_override_y = -1;
    enc->do_temporal_aq = 1;
    enc->do_psy = DSV_PSY_ALL;
    enc->do_dark_intra_boost = 1;
    enc->do_intra_filter = 1;
    enc->do_inter_filter = -1;
}

void dsv_enc_start(DSV_ENCODER *enc) // dsv_encoder.c:1360
{
    enc->quality = clampi(enc->quality, 0, DSV_RC_QUAL_MAX);
    switch (enc->rc_mode) {
        case DSV_RATE_CONTROL_CRF:
            enc->rc_qual = (unsigned) clampi(enc->quality + RC_PCT(5), enc->min_I_frame_quality, enc->max_quality);
            enc->rf_avg = (int) enc->rc_qual;
            enc->avg_P_frame_q = enc->quality;
            break;
        case DSV_RATE_CONTROL_ABR:
            enc->rc_qual = (unsigned) enc->quality;
            enc->avg_P_frame_q = enc->quality * 4 / 5;
            break;
        default:
            break;
    }
    enc->stats.iminq = enc->stats.pminq = enc->stats.imins = enc->stats.pmins = INT_MAX;
    enc->force_metadata = 1;
}

void dsv_enc_free(DSV_ENCODER *enc)
{
    g_enc_queue.forget(enc);
    if (enc->ref) {
        EncImpl *im = (EncImpl *) enc->ref;
        if (im->ready) {
            im->dev.destroy();
        }
        if (im->h_pack) {
            pinned_pool_release(im->h_pack);
            im->h_pack = nullptr;
        }
        for (int i = 0; i < 2; i++) {
            if (im->d_stage[i]) {
                HIPCHK(hipDeviceSynchronize()); // an upload into it may still be in flight on a copy stream
                HIPCHK(hipFree(im->d_stage[i]));
                im->d_stage[i] = nullptr;
            }
        }
        delete im;
        enc->ref = NULL;
    }
    if (enc->stability) {
        dsv_free(enc->stability);
        enc->stability = NULL;
    }
    if (enc->blockdata) {
        dsv_free(enc->blockdata);
        enc->blockdata = NULL;
    }
    if (enc->intra_map) {
        dsv_free(enc->intra_map);
        enc->intra_map = NULL;
    }
}

void dsv_enc_set_metadata(DSV_ENCODER *enc, DSV_META *md) { memcpy(&enc->vidmeta, md, sizeof(DSV_META)); }
void dsv_enc_force_metadata(DSV_ENCODER *enc) { enc->force_metadata = 1; }

void dsv_enc_end_of_stream(DSV_ENCODER *enc, DSV_BUF *bufs) // dsv_encoder.c:1416
{
    g_enc_queue.forget(enc); // (this caller will not join another step: leaders stop waiting for it)
    dsv_mk_buf(&bufs[0], DSV_PACKET_HDR_SIZE);
    BitWriter bw{bufs[0].data, 0};
    put_packet_hdr(bw, DSV_PT_EOS);
    set_link_offsets(enc, &bufs[0], 1);
}


int dsv_enc(DSV_ENCODER *enc, DSV_FRAME *frame, DSV_BUF *bufs) // dsv_encoder.c:1430
{
    if (frame == NULL || bufs == NULL) {
        return 0;
    }
    if (!enc_usable(enc)) {
        dsv_frame_ref_dec(frame); // (the reference releases the caller's frame on every path: dsv_encoder.c:1457)
        return 0;
    }
    Job jb;
    memset(&jb, 0, sizeof(jb));
    jb.enc = enc;
    jb.bufs = bufs;
    if (!Coalescer<Job>::enabled()) {
        jb.frame = frame;
        enc_batch(&jb, 1);
        return jb.nbuf;
    }
    // The calling thread does what is its own: the picture, whatever its strides, is packed into this encoder's pinned
    // staging block (concurrent callers pack side by side; the step then uploads every picture asynchronously) and released
    // (dsv_encoder.c:1457).  The step itself is shared with whoever else is calling right now (batch.h: Coalescer).
    bind_device();
    if (!enc->ref) {
        enc->ref = new EncImpl();
    }
    EncImpl *im = (EncImpl *) enc->ref;
    const int fmt = enc->vidmeta.subsamp, w = enc->vidmeta.width, h = enc->vidmeta.height;
    const int hs = DSV_FORMAT_H_SHIFT(fmt), vs = DSV_FORMAT_V_SHIFT(fmt);
    const int pw[3] = {w, (w + (1 << hs) - 1) >> hs, (w + (1 << hs) - 1) >> hs}, ph[3] = {h, (h + (1 << vs) - 1) >> vs, (h + (1 << vs) - 1) >> vs};
    const size_t pbytes = (size_t) pw[0] * ph[0] + 2 * (size_t) pw[1] * ph[1];
    if (im->h_pack_bytes != pbytes) {
        if (im->h_pack) {
            pinned_pool_release(im->h_pack);
        }
        im->h_pack = (uint8_t *) pinned_pool_take(pbytes);
        im->h_pack_bytes = pbytes;
    }
    uint8_t *dst = im->h_pack;
    for (int c = 0; c < 3; c++) {
        const DSV_PLANE *sp = &frame->planes[c];
        const int cw = sp->w < pw[c] ? sp->w : pw[c], rows = sp->h < ph[c] ? sp->h : ph[c];
        for (int y = 0; y < ph[c]; y++, dst += pw[c]) {
            if (y < rows) {
                memcpy(dst, sp->data + (size_t) y * sp->stride, (size_t) cw);
                if (cw < pw[c]) {
                    memset(dst + cw, 0, (size_t) (pw[c] - cw));
                }
            } else {
                memset(dst, 0, (size_t) pw[c]);
            }
        }
    }
    dsv_frame_ref_dec(frame);
    jb.host_planar = im->h_pack;
    jb.from_frame = true;
    g_enc_queue.submit(jb, step_key(enc), enc, enc_batch);
    return jb.nbuf;
}

/* what the submit queue of dsv_enc did so far: [0] calls, [1] lockstep steps they were run as, [2] the largest step, [3] total
 * microseconds leaders spent waiting for expected callers; reset != 0 clears the counts afterwards */
void dsv2hip_test_fail_next_step(int how) { g_fail_next_step.store(how); }
long dsv2hip_enc_list_growths(void) { return g_list_growths.load(); }
long dsv2hip_arena_fallbacks(void) { return dsv2::arena_fallbacks(); }

void dsv2hip_enc_queue_stats(unsigned long long *out4, int reset)
{
    Coalescer<Job>::Stats st = g_enc_queue.stats();
    if (out4) {
        out4[0] = st.calls;
        out4[1] = st.steps;
        out4[2] = st.largest;
        out4[3] = st.waited_us;
    }
    if (reset) {
        g_enc_queue.reset_stats();
    }
}

/* same as dsv_enc for a packed planar 8-bit picture (Y, then U, then V, no padding) that is
 * already resident in device memory: no host->device copy is made */
int dsv2hip_enc_device_frame(DSV_ENCODER *enc, const void *dev_planar, DSV_BUF *bufs)
{
    if (dev_planar == NULL || bufs == NULL || !enc_usable(enc)) {
        return 0;
    }
    Job jb;
    memset(&jb, 0, sizeof(jb));
    jb.enc = enc;
    jb.dev_planar = (const uint8_t *) dev_planar;
    jb.bufs = bufs;
    enc_batch(&jb, 1);
    return jb.nbuf;
}

/* lockstep step: one frame on each of n encoders (identical geometry).  dev_planar[k] is stream k's
 * picture in device memory; bufs holds 4 DSV_BUF slots per stream, nbufs[k] receives the packet count.
 * Results are identical to calling dsv2hip_enc_device_frame on every encoder separately. */
int dsv2hip_enc_batch(int n, DSV_ENCODER **encs, const void *const *dev_planar, DSV_BUF *bufs, int *nbufs)
{
    if (n <= 0 || !encs || !dev_planar || !bufs || !nbufs) {
        return -1;
    }
    for (int k = 0; k < n; k++) { // one geometry per step, every encoder usable: refused as a whole otherwise (nothing was touched)
        if (!enc_usable(encs[k]) || step_key(encs[k]) != step_key(encs[0])) {
            return -1;
        }
    }
    std::vector<Job> jobs((size_t) n);
    for (int k = 0; k < n; k++) {
        memset(&jobs[(size_t) k], 0, sizeof(Job));
        jobs[(size_t) k].enc = encs[k];
        jobs[(size_t) k].dev_planar = (const uint8_t *) dev_planar[k];
        jobs[(size_t) k].bufs = bufs + 4 * k;
    }
    const bool ok = enc_batch_ok(jobs.data(), n);
    for (int k = 0; k < n; k++) {
        nbufs[k] = jobs[(size_t) k].nbuf;
    }
    return ok ? 0 : -1; // a failed step: every nbufs[k] is 0, the encoders are dead
}

/* the same with the pictures in HOST memory (packed planar Y, U, V).  host_planar[k]: stream k's picture of this
 * step; host_next (may be NULL, entries may be NULL): the picture stream k will bring to the NEXT call -- it is
 * uploaded on a copy stream while this step's kernels run, and the next call finds it in HBM when it passes the same
 * pointer as host_planar[k] (the bytes must stay unchanged until then).  Memory from dsv2hip_host_alloc is pinned,
 * which makes the uploads asynchronous; any other host memory works, synchronously. */
int dsv2hip_enc_batch_host(int n, DSV_ENCODER **encs, const void *const *host_planar, const void *const *host_next, DSV_BUF *bufs, int *nbufs)
{
    if (n <= 0 || !encs || !host_planar || !bufs || !nbufs) {
        return -1;
    }
    for (int k = 0; k < n; k++) { // one geometry per step, every encoder usable: refused as a whole otherwise (nothing was touched)
        if (!enc_usable(encs[k]) || step_key(encs[k]) != step_key(encs[0])) {
            return -1;
        }
    }
    std::vector<Job> jobs((size_t) n);
    for (int k = 0; k < n; k++) {
        if (!host_planar[k]) {
            return -1;
        }
        memset(&jobs[(size_t) k], 0, sizeof(Job));
        jobs[(size_t) k].enc = encs[k];
        jobs[(size_t) k].host_planar = (const uint8_t *) host_planar[k];
        jobs[(size_t) k].host_next = host_next ? (const uint8_t *) host_next[k] : nullptr;
        jobs[(size_t) k].bufs = bufs + 4 * k;
    }
    const bool ok = enc_batch_ok(jobs.data(), n);
    for (int k = 0; k < n; k++) {
        nbufs[k] = jobs[(size_t) k].nbuf;
    }
    return ok ? 0 : -1; // a failed step: every nbufs[k] is 0, the encoders are dead
}

/* packed pictures handed to this encoder (dsv2hip_enc_device_frame / _batch / _batch_host) are interleaved UYVY 4:2:2
 * rows instead of planar: the de-interleave of dsv_yuv_read (dsv.c:177-205) moves into the ingest kernel.  The
 * stream's metadata must say DSV_SUBSAMP_UYVY (or 4:2:2: same plane geometry). */
int dsv2hip_enc_set_uyvy_input(DSV_ENCODER *enc, int on)
{
    if (!enc) {
        return -1;
    }
    if (DSV_FORMAT_H_SHIFT(enc->vidmeta.subsamp) != 1 || DSV_FORMAT_V_SHIFT(enc->vidmeta.subsamp) != 0 || (enc->vidmeta.width & 1)) {
        return -1;
    }
    if (!enc->ref) {
        enc->ref = new EncImpl();
    }
    ((EncImpl *) enc->ref)->input_uyvy = on != 0;
    return 0;
}

/* pinned host memory for pictures handed to dsv2hip_enc_batch_host */
void *dsv2hip_host_alloc(size_t bytes)
{
    bind_device();
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        return nullptr;
    }
    return p;
}

void dsv2hip_host_free(void *p)
{
    if (p) {
        HIPCHK(hipHostFree(p));
    }
}

} // extern "C"
