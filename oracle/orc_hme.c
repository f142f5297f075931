/*
 * oracle/orc_hme.c -- TEST INFRASTRUCTURE (see orc_common.h).
 *
 * Hierarchical motion estimation + P-frame mode decision restated from reference
 * src/hme.c (dsv_hme :2001, refine_level :1373, refine_best_fpel_cand :1301,
 * subpixel_ME :1052, hpel :788, qpel :816, qpsad :245, mode decision :1636-1821,
 * test_subblock_intra_y/c :892/:988, calc_EPRM :453, global_motion :1974) in the
 * decomposition used by csrc/hme.hip:
 *   - one independent work item per block, written as straight-line control code over a
 *     small set of block primitives (metric / SSE / statistics), which the GPU evaluates
 *     cooperatively across one wavefront;
 *   - the ORDER in which blocks of a level may be evaluated: a block only reads the vectors of
 *     its left, top and top-left neighbours at the same level (candidate list :1218-1226, MV
 *     predictor dsv.c:375, neighbour difference dsv.c:404).  This restatement walks the
 *     anti-diagonal fronts (all blocks with equal i/step + j/step), one legal order; the
 *     kernels (csrc/hme.hip: k_hme_rows_*) walk another -- a ROW PIPELINE: one wavefront per
 *     block row, left to right, each block waiting for the heads of its top / top-left
 *     neighbours.  Both reproduce the reference's raster order;
 *   - the half/quarter-pel search never materialises the 68x68 quarter-pel image: every
 *     quarter-pel sample is derived on the fly from the 34x34 half-pel image.
 */
#include "orc_blockstat.h"

extern int orc_spatial_psy_factor(int blk_w, int blk_h, int nbh, int nbv, int sub);

typedef struct {
    const uint8_t *data;
    int stride, w, h;
} hplane;

typedef struct {
    orc_params p;
    int quant, skip_block_thresh, pyr_levels;
    hplane src[6], ref[6], ogr[6]; /* luma of pyramid level 0..pyr_levels */
    hplane srcc[2], refc[2];       /* chroma (U, V) of the full-size source / reconstructed reference */
    orc_mv *mvf[6];
    const orc_mv *ref_mvf;
    /* frame totals */
    int nintra, ndiff, eligible;
    unsigned total_err;
} hme_ctx;

typedef struct {
    int err_weight, tex_weight, avg_weight;
} psy_t;

#define UAVG4(a, b, c, d) ((unsigned) ((a) + (b) + (c) + (d) + 2) >> 2)
#define AVG2(a, b) (((a) + (b) + 1) >> 1)
#define SQR(x) ((x) * (x))

/* ---------------- block primitives ---------------- */

static unsigned
quad_metric(int a1, int a2, int a3, int a4, int b1, int b2, int b3, int b4, const psy_t *psy) /* hme.c:126 METR_CALC */
{
    int s0 = (int) UAVG4(a1, a2, a3, a4), s1 = (int) UAVG4(b1, b2, b3, b4);
    int se = (int) UAVG4(bs_abs(a1 - b1), bs_abs(a2 - b2), bs_abs(a3 - b3), bs_abs(a4 - b4));
    int ta = (int) UAVG4(bs_abs(a1 - a2), bs_abs(a2 - a3), bs_abs(a3 - a4), bs_abs(a4 - a1));
    int tb = (int) UAVG4(bs_abs(b1 - b2), bs_abs(b2 - b3), bs_abs(b3 - b4), bs_abs(b4 - b1));
    unsigned acc = 0;
    acc += (unsigned) (SQR(se) << psy->err_weight);
    acc += (unsigned) (SQR(ta - tb) << psy->tex_weight);
    acc += (unsigned) (SQR(s0 - s1) << psy->avg_weight);
    return acc;
}

static unsigned
P_umetr(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h, const psy_t *psy) /* raw accumulator, hme.c:191 */
{
    unsigned acc = 0;
    int i, j;
    for (j = 0; j < h / 2; j++) {
        for (i = 0; i < w / 2; i++) {
            const uint8_t *p = a + 2 * j * as + 2 * i, *q = b + 2 * j * bs + 2 * i;
            acc += quad_metric(p[0], p[1], p[as], p[as + 1], q[0], q[1], q[bs], q[bs + 1], psy);
        }
    }
    return acc;
}

static unsigned
metric_return(unsigned acc, int w, int h) /* hme.c:97 */
{
    return bs_isqrt(acc) * (unsigned) w * (unsigned) h / (unsigned) AVG2(w, h);
}

static unsigned
P_metr(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h, const psy_t *psy) /* fastmetr, hme.c:271 */
{
    if (w == 0 || h == 0) {
        return INT_MAX;
    }
    return metric_return(P_umetr(a, as, b, bs, w, h, psy), w, h);
}

static unsigned
P_sse(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h) /* hme.c:198 */
{
    unsigned acc = 0;
    int i, j;
    if (w == 0 || h == 0) {
        return INT_MAX;
    }
    for (j = 0; j < h; j++) {
        for (i = 0; i < w; i++) {
            int d = a[j * as + i] - b[j * bs + i];
            acc += (unsigned) (d * d);
        }
    }
    return acc;
}

static unsigned
P_hier_metr(int level, const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h, const psy_t *psy)
{
    return level > 1 ? P_sse(a, as, b, bs, w, h) : P_metr(a, as, b, bs, w, h, psy);
}

static const uint8_t *
at(const hplane *p, int x, int y)
{
    return p->data + y * p->stride + x;
}

static int
invalid_block(const hplane *f, int bx, int by, int bw, int bh, int pad) /* hme.c:426 */
{
    int b = ORC_BORDER;
    return (bx - pad) < -b || (by - pad) < -b || (bx + bw + pad) >= (f->w + b) || (by + bh + pad) >= (f->h + b);
}

/* ---------------- motion vector cost ---------------- */

static int
pred1(int left, int top, int topleft)
{
    int dif = left + top - topleft;
    return bs_abs(dif - left) < bs_abs(dif - top) ? left : top;
}

static void
movec_pred(const orc_mv *v, int nbh, int x, int y, int *px, int *py) /* dsv.c:375 */
{
    int vx[3] = { 0, 0, 0 }, vy[3] = { 0, 0, 0 };
    if (x > 0) {
        vx[0] = v[y * nbh + x - 1].x;
        vy[0] = v[y * nbh + x - 1].y;
    }
    if (y > 0) {
        vx[1] = v[(y - 1) * nbh + x].x;
        vy[1] = v[(y - 1) * nbh + x].y;
    }
    if (x > 0 && y > 0) {
        vx[2] = v[(y - 1) * nbh + x - 1].x;
        vy[2] = v[(y - 1) * nbh + x - 1].y;
    }
    *px = pred1(vx[0], vx[1], vx[2]);
    *py = pred1(vy[0], vy[1], vy[2]);
}

static int
seg_bits(int v) /* dsv.c:335 */
{
    int n = -1;
    unsigned x;
    if (v < 0) {
        v = -v;
    }
    v++;
    for (x = (unsigned) v; x; x >>= 1) {
        n++;
    }
    return n * 2 + 2;
}

static int
mv_cost(const hme_ctx *c, const orc_mv *v, int i, int j, int mx, int my, int level) /* hme.c:354 + dsv.c:357 */
{
    int px, py, bits, b2sr, q = c->quant, sqr = level > 1, cost;
    movec_pred(v, c->p.nblocks_h, i, j, &px, &py);
    bits = seg_bits(mx - px) + seg_bits(my - py);
    b2sr = (256 * (q * q >> 12) * c->p.blk_w * c->p.blk_h) / (c->p.width * c->p.height);
    bits += bits * b2sr >> 7;
    if (sqr) {
        bits *= bits;
    }
    cost = ORC_MIN(bits, 1 << 19);
    if (sqr) {
        return (int) ((unsigned) cost * (unsigned) (q * q >> 12)) >> 10;
    }
    return 3 * cost * q >> 12;
}

static void
neighbordif2(const orc_mv *v, int nbh, int x, int y, int *dx, int *dy) /* dsv.c:403 */
{
    const orc_mv *c = &v[x + y * nbh];
    int cx = c->x, cy = c->y, lx = cx, ly = cy, tx = cx, ty = cy;
    if (bs_abs(cx) < 2 && bs_abs(cy) < 2) {
        *dx = *dy = 0;
        return;
    }
    if (x > 0) {
        const orc_mv *m = c - 1;
        if ((m->x || m->y) && !(m->flags & ORC_MV_SKIP)) {
            lx = m->x;
            ly = m->y;
        }
    }
    if (y > 0) {
        const orc_mv *m = c - nbh;
        if ((m->x || m->y) && !(m->flags & ORC_MV_SKIP)) {
            tx = m->x;
            ty = m->y;
        }
    }
    *dx = bs_abs(lx - cx) + bs_abs(ly - cy);
    *dy = bs_abs(tx - cx) + bs_abs(ty - cy);
}

/* ---------------- candidate list ---------------- */

typedef struct {
    int x, y;
} vec2;

#define MAXC 40

static int
qp2fp(int v) /* DSV_SAR_R(v, 2), hme.c:39 */
{
    return orc_sar(v + 2, 2);
}

static int
find_inliers(const vec2 *list, vec2 *out, int n, int *ax, int *ay) /* hme.c:1260 */
{
    int i, dist[16], avgd = 0, ssd = 0, thresh, nin = 0, sx = 0, sy = 0;
    if (n == 0) {
        return 0;
    }
    for (i = 0; i < n; i++) {
        dist[i] = SQR(list[i].x - *ax) + SQR(list[i].y - *ay);
        avgd += dist[i];
    }
    avgd /= n;
    for (i = 0; i < n; i++) {
        ssd += SQR(dist[i] - avgd);
    }
    thresh = avgd + (int) bs_isqrt((unsigned) (ssd / n));
    for (i = 0; i < n; i++) {
        if (dist[i] <= thresh) {
            sx += list[i].x;
            sy += list[i].y;
            out[nin++] = list[i];
        }
    }
    if (nin == 0) {
        return 0;
    }
    *ax = sx / nin;
    *ay = sy / nin;
    return nin;
}

/* ---------------- half / quarter-pel refinement ---------------- */

static uint8_t
clamp_u8(int v)
{
    return (uint8_t) (v > 255 ? 255 : (v < 0 ? 0 : v));
}

#define HPF_ME(a, b, c, d) ((5 * ((b) + (c))) - ((a) + (d)))

/* 34x34 half-pel image of the 17x17 window whose top-left full-pel sample is r (hme.c:787) */
static void
build_hpel(uint8_t *h, const uint8_t *r, int rs)
{
    int i, j, k;
    for (j = 0; j < 17; j++) {
        for (i = 0; i < 17; i++) {
            const uint8_t *p = r + j * rs + i;
            int hz[4], c;
            for (k = 0; k < 4; k++) {
                const uint8_t *q = p + (k - 1) * rs;
                hz[k] = HPF_ME(q[-1], q[0], q[1], q[2]);
            }
            c = HPF_ME(hz[0], hz[1], hz[2], hz[3]);
            h[(2 * j) * 34 + 2 * i] = p[0];
            h[(2 * j) * 34 + 2 * i + 1] = clamp_u8((hz[1] + 4) >> 3);
            h[(2 * j + 1) * 34 + 2 * i] = clamp_u8((HPF_ME(p[-rs], p[0], p[rs], p[2 * rs]) + 4) >> 3);
            h[(2 * j + 1) * 34 + 2 * i + 1] = clamp_u8((c + 32) >> 6);
        }
    }
}

/* quarter-pel sample (X, Y) of the 68x68 image implied by h (hme.c:815) */
static int
qsample(const uint8_t *h, int X, int Y)
{
    const uint8_t *p = h + (Y >> 1) * 34 + (X >> 1);
    switch ((X & 1) | ((Y & 1) << 1)) {
        case 0: return p[0];
        case 1: return AVG2(p[0], p[1]);
        case 2: return AVG2(p[0], p[34]);
        default: return (p[0] + p[1] + p[34] + p[35] + 2) >> 2;
    }
}

/* psy metric of the 16x16 source window against the block sampled at quarter-pel offset (tx, ty) (hme.c:244) */
static unsigned
P_qpsad(const uint8_t *a, int as, const uint8_t *h, int tx, int ty, const psy_t *psy)
{
    unsigned acc = 0;
    int i, j;
    for (j = 0; j < 8; j++) {
        for (i = 0; i < 8; i++) {
            const uint8_t *p = a + 2 * j * as + 2 * i;
            int X = 4 + 8 * i + tx, Y = 4 + 8 * j + ty;
            acc += quad_metric(p[0], p[1], p[as], p[as + 1], qsample(h, X, Y), qsample(h, X + 4, Y), qsample(h, X, Y + 4),
                               qsample(h, X + 4, Y + 4), psy);
        }
    }
    return metric_return(acc, 16, 16);
}

static unsigned
subpixel_me(const hme_ctx *c, const orc_mv *mvf, int *sub_x, int *sub_y, int fpelx, int fpely, int i, int j, unsigned best,
            int bx, int by, int bw, int bh, const psy_t *psy) /* hme.c:1051 */
{
    static const int dxs[4] = { 1, -1, 0, 0 }, dys[4] = { 0, 0, 1, -1 };
    uint8_t h[34 * 34 + 36];
    const hplane *src = &c->src[0], *ref = &c->ref[0];
    unsigned quad[4], yarea = (unsigned) (bw * bh), ms1, ms2, score;
    int n, pri[2], sec[2], diag[2], bestv[2] = { 0, 0 }, xx, yy, area_ratio, iarea_ratio;
    const uint8_t *srcw;

    *sub_x = *sub_y = 0;
    if (best == 0) {
        return best;
    }
    for (n = 0; n < 4; n++) {
        quad[n] = P_sse(at(src, bx, by), src->stride, at(ref, bx + fpelx + dxs[n], by + fpely + dys[n]), ref->stride, bw, bh);
    }
    area_ratio = (int) (8 * 256 / yarea);
    iarea_ratio = (int) (8 * yarea / 256);
    best = best * (unsigned) area_ratio >> 3;
    xx = bx + ((bw >> 1) - 8);
    yy = by + ((bh >> 1) - 8);
    srcw = at(src, xx, yy);
    build_hpel(h, at(ref, xx + fpelx - 1, yy + fpely - 1), ref->stride);

    pri[0] = 0, pri[1] = -1;
    sec[0] = -1, sec[1] = 0;
    ms1 = quad[1];
    ms2 = quad[3];
    if (quad[3] >= quad[2]) {
        pri[1] = 1;
        ms2 = quad[2];
    }
    if (quad[1] >= quad[0]) {
        sec[0] = 1;
        ms1 = quad[0];
    }
    if (ms2 > ms1) {
        int t0 = sec[0], t1 = sec[1];
        sec[0] = pri[0], sec[1] = pri[1];
        pri[0] = t0, pri[1] = t1;
    }
    diag[0] = pri[0] + sec[0];
    diag[1] = pri[1] + sec[1];
    for (n = 0; n <= 6; n++) {
        int t[2];
        if (n == 6) {
            t[0] = pri[0] + diag[0];
            t[1] = pri[1] + diag[1];
        } else {
            const int *v = (n >> 1) == 0 ? pri : ((n >> 1) == 1 ? sec : diag);
            int hp = !(n & 1);
            t[0] = v[0] * (1 << hp);
            t[1] = v[1] * (1 << hp);
        }
        if (((t[0] | t[1]) & 1) && c->p.effort < 8) {
            continue;
        }
        score = P_qpsad(srcw, src->stride, h, t[0], t[1], psy);
        score += (unsigned) mv_cost(c, mvf, i, j, fpelx * 4 + t[0], fpely * 4 + t[1], 0);
        if (best > score) {
            best = score;
            bestv[0] = t[0];
            bestv[1] = t[1];
        }
    }
    *sub_x = bestv[0];
    *sub_y = bestv[1];
    return best * (unsigned) iarea_ratio >> 3;
}

/* ---------------- mode decision helpers ---------------- */

static void
P_yuv_max_subblock_err(unsigned out[3], const hme_ctx *c, int bx, int by, int brx, int bry, int bw, int bh, int cbx, int cby,
                       int cbrx, int cbry, int cbw, int cbh, const psy_t *psy) /* hme.c:369 */
{
    int z, k;
    for (z = 0; z < 3; z++) {
        const uint8_t *sp, *rp;
        int ss, rs, w, h;
        unsigned mx = 0;
        if (z == 0) {
            sp = at(&c->src[0], bx, by);
            ss = c->src[0].stride;
            rp = at(&c->ref[0], brx, bry);
            rs = c->ref[0].stride;
            w = bw / 2;
            h = bh / 2;
        } else {
            sp = at(&c->srcc[z - 1], cbx, cby);
            ss = c->srcc[z - 1].stride;
            rp = at(&c->refc[z - 1], cbrx, cbry);
            rs = c->refc[z - 1].stride;
            w = cbw / 2;
            h = cbh / 2;
        }
        for (k = 0; k < 4; k++) {
            int f = (k & 1) ? w : 0, g = (k & 2) ? h : 0;
            unsigned e = P_umetr(sp + f + g * ss, ss, rp + f + g * rs, rs, w, h, psy);
            mx = ORC_MAX(mx, e);
        }
        out[z] = mx;
    }
}

static void
P_calc_eprm(const uint8_t *src, int ss, const uint8_t *mvr, int rs, int avg_src, int avg_ref, int w, int h, int *eprmi, int *eprmd,
            int *eprmr) /* hme.c:452: any-pixel clip tests */
{
    int i, j, ci = 0, cd = 0, cr = 0;
    avg_src -= 128;
    avg_ref -= 128;
    for (j = 0; j < h; j++) {
        for (i = 0; i < w; i++) {
            int s = src[j * ss + i];
            cr |= ((s - mvr[j * rs + i]) + 128) & ~0xff;
            ci |= (s - avg_ref) & ~0xff;
            cd |= (s - avg_src) & ~0xff;
        }
    }
    *eprmi = !!ci;
    *eprmd = !!cd;
    *eprmr = !!cr;
}

static void
P_err_intra(const uint8_t *a, int as, const uint8_t *b, int bs, int avg_sb, int avg_src, int w, int h, unsigned *intra_err,
            unsigned *intrasrc_err, unsigned *inter_err, const psy_t *psy, int ratio) /* hme.c:839 */
{
    unsigned isb = 0, isrc = 0, inter = 0;
    int i, j;
    for (j = 0; j < h / 2; j++) {
        for (i = 0; i < w / 2; i++) {
            const uint8_t *p = a + 2 * j * as + 2 * i, *q = b + 2 * j * bs + 2 * i;
            int a1 = p[0], a2 = p[1], a3 = p[as], a4 = p[as + 1];
            int b1 = q[0], b2 = q[1], b3 = q[bs], b4 = q[bs + 1];
            int s0 = (int) UAVG4(a1, a2, a3, a4), s1 = (int) UAVG4(b1, b2, b3, b4);
            int ae = (int) UAVG4(bs_abs(a1 - b1), bs_abs(a2 - b2), bs_abs(a3 - b3), bs_abs(a4 - b4));
            int ta = (int) UAVG4(bs_abs(a1 - a2), bs_abs(a2 - a3), bs_abs(a3 - a4), bs_abs(a4 - a1));
            int tb = (int) UAVG4(bs_abs(b1 - b2), bs_abs(b2 - b3), bs_abs(b3 - b4), bs_abs(b4 - b1));
            inter += (unsigned) (SQR(ae) * ratio >> (5 - psy->err_weight));
            inter += (unsigned) (SQR(ta - tb) << psy->tex_weight);
            inter += (unsigned) (SQR(s0 - s1) << psy->avg_weight);
            ae = (int) UAVG4(bs_abs(a1 - avg_sb), bs_abs(a2 - avg_sb), bs_abs(a3 - avg_sb), bs_abs(a4 - avg_sb));
            isb += (unsigned) (SQR(ae) << psy->err_weight);
            isb += (unsigned) (SQR(ta) << psy->tex_weight);
            isb += (unsigned) (SQR(s0 - avg_sb) << (psy->avg_weight + 1));
            ae = (int) UAVG4(bs_abs(a1 - avg_src), bs_abs(a2 - avg_src), bs_abs(a3 - avg_src), bs_abs(a4 - avg_src));
            isrc += (unsigned) (SQR(ae) << psy->err_weight);
            isrc += (unsigned) (SQR(ta) << psy->tex_weight);
            isrc += (unsigned) (SQR(s0 - avg_src) << (psy->avg_weight + 1));
        }
    }
    *intra_err = isb;
    *intrasrc_err = isrc;
    *inter_err = inter * (unsigned) ratio >> 5;
}

static int
P_plane_avg(const hplane *p, int x, int y, int w, int h)
{
    return bs_block_avg(at(p, x, y), p->stride, w, h);
}

static void
test_subblock_intra_y(const hme_ctx *c, const orc_mv *refmv, orc_mv *mv, const uint8_t *srcd, int ss, const uint8_t *refd, int rs,
                      int detail_src, int avg_src, int neidif, unsigned ratio, int bw, int bh) /* hme.c:891 */
{
    int k, sbw = bw / 2, sbh = bh / 2, nsub = 0, psyscale;
    unsigned avg_tot = 0, err_sub = 0, err_src = 0;
    psy_t psy = { 0, 1, 2 };
    if (refmv == NULL) {
        refmv = mv;
    }
    if ((mv->x || mv->y) && neidif < 3 && bs_abs(refmv->x - mv->x) < 3 && bs_abs(refmv->y - mv->y) < 3) {
        return;
    }
    if (sbw == 0 || sbh == 0) {
        return;
    }
    psyscale = orc_spatial_psy_factor(c->p.blk_w, c->p.blk_h, c->p.nblocks_h, c->p.nblocks_v, -1);
    detail_src += detail_src / ORC_MAX(neidif, 1);
    for (k = 0; k < 4; k++) {
        int f = (k & 1) ? sbw : 0, g = (k & 2) ? sbh : 0;
        const uint8_t *sd = srcd + f + g * ss, *md = refd + f + g * rs;
        unsigned avg_local, avg_sub, local_detail, dcd, sub_err, src_err, inter_err;
        int dc, lo, hi, lerp;
        if (mv->submask & (1 << k)) {
            continue;
        }
        avg_sub = (unsigned) bs_block_avg(md, rs, sbw, sbh);
        local_detail = (unsigned) bs_block_detail(sd, ss, sbw, sbh, &avg_local);
        dcd = (unsigned) bs_abs((int) avg_local - (int) avg_sub) + 2;
        if (local_detail > (unsigned) (SQR(dcd) * (unsigned) bw * (unsigned) bh * ratio >> 5)) {
            continue;
        }
        dc = (int) (avg_local + (unsigned) avg_src * 3 + 2) >> 2;
        P_err_intra(sd, ss, md, rs, (int) avg_sub, dc, sbw, sbh, &sub_err, &src_err, &inter_err, &psy, (int) ratio);
        lo = AVG2(detail_src, (int) local_detail);
        hi = detail_src;
        lerp = (lo * (32 - psyscale) + hi * psyscale) >> 5;
        local_detail = (unsigned) ORC_MAX(lerp, lo);
        if ((sub_err + local_detail) < inter_err || (src_err + local_detail) < inter_err) {
            mv->submask |= (uint8_t) (1 << k);
            err_src += src_err;
            err_sub += sub_err;
            avg_tot += sub_err < src_err ? avg_sub : (unsigned) dc;
            nsub++;
            detail_src = detail_src * 4 / 5;
        }
    }
    if (mv->submask) {
        mv->flags |= ORC_MV_INTRA;
        mv->dc = err_src < err_sub ? (uint16_t) ((avg_tot / (unsigned) nsub) | 0x100) : 0;
    }
}

static void
test_subblock_intra_c(const hme_ctx *c, orc_mv *mv, unsigned mad, unsigned detail_src, unsigned avg_src, int cbx, int cby, int cbmx,
                      int cbmy, int cbw, int cbh) /* hme.c:987 */
{
    int k, sbw = cbw / 2, sbh = cbh / 2;
    unsigned thr, avg_ramp;
    if (c->p.effort < 6) {
        return;
    }
    thr = (mv->flags & ORC_MV_INTRA) ? detail_src : SQR(detail_src);
    if (sbw == 0 || sbh == 0 || mad <= thr || thr > 64 || (bs_abs(mv->x) < 4 && bs_abs(mv->y) < 4)) {
        return;
    }
    avg_ramp = avg_src * avg_src >> 8;
    for (k = 0; k < 4; k++) {
        int f = (k & 1) ? sbw : 0, g = (k & 2) ? sbh : 0;
        int us, vs, um, vm;
        unsigned dif;
        if (mv->submask & (1 << k)) {
            continue;
        }
        us = P_plane_avg(&c->srcc[0], cbx + f, cby + g, sbw, sbh);
        vs = P_plane_avg(&c->srcc[1], cbx + f, cby + g, sbw, sbh);
        um = P_plane_avg(&c->refc[0], cbmx + f, cbmy + g, sbw, sbh);
        vm = P_plane_avg(&c->refc[1], cbmx + f, cbmy + g, sbw, sbh);
        dif = (unsigned) (SQR(us - um) + SQR(vs - vm)) * avg_ramp >> 8;
        if (dif > thr) {
            mv->submask |= (uint8_t) (1 << k);
        }
    }
    if (mv->submask) {
        mv->flags |= ORC_MV_INTRA;
    }
}

/* ---------------- one block ---------------- */

static const int rectx[9] = { 0, 1, -1, 0, 0, -1, 1, -1, 1 };
static const int recty[9] = { 0, 0, 0, 1, -1, -1, -1, 1, 1 };

static int
add_cand(vec2 *cands, int n, int x, int y)
{
    cands[n].x = x;
    cands[n].y = y;
    return n + 1;
}

static void
hme_block(hme_ctx *c, int level, int i, int j, int gx, int gy)
{
    const orc_params *P = &c->p;
    int nxb = P->nblocks_h, nyb = P->nblocks_v, y_w = P->blk_w, y_h = P->blk_h;
    int step = 1 << level;
    const hplane *src = &c->src[level], *ref = &c->ref[level], *ogr = &c->ogr[level];
    orc_mv *mvf = c->mvf[level];
    const orc_mv *parent = level < c->pyr_levels ? c->mvf[level + 1] : NULL;
    orc_mv *mv = &mvf[i + j * nxb];
    vec2 cands[MAXC];
    int n = 0, k, m, bx, by, bw, bh, dx, dy, lax = 0, lay = 0, motion_bias, good_enough = 0;
    unsigned best, best_score, score_zero, score, qthresh, var_src = 0, avg_src = 0;
    psy_t psy = { 2, 1, 0 };
    const uint8_t *sblk;
    int best_k;

    bx = (i * y_w) >> level;
    by = (j * y_h) >> level;
    if (bx >= src->w || by >= src->h) {
        memset(mv, 0, sizeof(*mv));
        return;
    }
    bw = ORC_MIN(src->w - bx, y_w);
    bh = ORC_MIN(src->h - by, y_h);
    sblk = at(src, bx, by);
    n = add_cand(cands, n, 0, 0);
    motion_bias = y_w * y_h;
    if (level <= 1) {
        int tvar;
        var_src = (unsigned) bs_block_detail(sblk, src->stride, bw, bh, &avg_src);
        tvar = (int) (var_src + SQR(var_src >> 10));
        tvar = (8 * tvar * c->quant >> 9) / (bw * bh);
        if (tvar) {
            int hvar = (int) bs_hist_var(sblk, src->stride, bw, bh);
            int qtex = bs_quant_tex(sblk, src->stride, bw, bh);
            int npeaks = bs_peaks(sblk, src->stride, bw, bh, (int) avg_src);
            motion_bias += tvar * (hvar - qtex) * npeaks;
        }
        motion_bias = ORC_MAX(motion_bias, 0) / (2 + (bs_abs(gx) + bs_abs(gy)));
        if (var_src <= (unsigned) (8 * bw * bh * c->quant >> 9)) {
            psy.err_weight = 2, psy.tex_weight = 1, psy.avg_weight = 2;
            motion_bias = 0;
        } else {
            psy.err_weight = 1, psy.tex_weight = 2, psy.avg_weight = 1;
        }
        if (var_src > (unsigned) (24 * bw * bh)) {
            psy.avg_weight = 0;
        }
    }
    if (parent != NULL) {
        static const int pt[18] = { 0, 0, -2, 0, 2, 0, 0, -2, 0, 2, -2, -2, 2, 2, 2, -2, -2, 2 };
        unsigned parent_mask = ~(((unsigned) step << 1) - 1);
        int pi = (int) ((unsigned) i & parent_mask), pj = (int) ((unsigned) j & parent_mask);
        int sumx = 0, sumy = 0, npar = 0;
        vec2 lc[16], inl[16];
        for (m = 0; m < 9; m++) {
            int x = pi + pt[2 * m] * step, y = pj + pt[2 * m + 1] * step;
            if (x >= 0 && x < nxb && y >= 0 && y < nyb) {
                const orc_mv *pm = &parent[x + y * nxb];
                sumx += pm->x;
                sumy += pm->y;
                lc[npar].x = pm->x;
                lc[npar].y = pm->y;
                npar++;
            }
        }
        if (npar) {
            int nl;
            lax = sumx / npar;
            lay = sumy / npar;
            nl = find_inliers(lc, inl, npar, &lax, &lay);
            n = add_cand(cands, n, lax, lay);
            /* spatial predictions: every stored vector passes through the qpel->fpel rounding,
             * whatever unit it really is in (hme.c:1194-1227) */
            if (level == 0) {
                int px, py;
                movec_pred(mvf, nxb, i, j, &px, &py);
                n = add_cand(cands, n, qp2fp(px), qp2fp(py));
            }
            if (i > 0) {
                n = add_cand(cands, n, qp2fp(mvf[(i - step) + j * nxb].x), qp2fp(mvf[(i - step) + j * nxb].y));
            }
            if (j > 0) {
                n = add_cand(cands, n, qp2fp(mvf[i + (j - step) * nxb].x), qp2fp(mvf[i + (j - step) * nxb].y));
            }
            if (i > 0 && j > 0) {
                n = add_cand(cands, n, qp2fp(mvf[(i - step) + (j - step) * nxb].x), qp2fp(mvf[(i - step) + (j - step) * nxb].y));
            }
            if (c->ref_mvf != NULL) {
                for (k = 0; k < 9; k++) {
                    int rx = i + rectx[k] * step, ry = j + recty[k] * step;
                    if (rx < 0 || ry < 0 || rx >= nxb || ry >= nyb) {
                        continue;
                    }
                    n = add_cand(cands, n, qp2fp(c->ref_mvf[rx + ry * nxb].x), qp2fp(c->ref_mvf[rx + ry * nxb].y));
                }
            }
            n = add_cand(cands, n, gx, gy);
            for (m = 0; m < nl; m++) {
                n = add_cand(cands, n, inl[m].x, inl[m].y);
            }
        }
    }
    for (k = 0; k < n; k++) {
        /* the reference stores candidates as int16 pairs */
        cands[k].x = (int16_t) orc_sar((int16_t) cands[k].x, level);
        cands[k].y = (int16_t) orc_sar((int16_t) cands[k].y, level);
    }
    { /* keep the first occurrence of every distinct vector (hme.c:1166) */
        int newn = 1;
        for (k = 1; k < n; k++) {
            for (m = 0; m < newn; m++) {
                if (cands[k].x == cands[m].x && cands[k].y == cands[m].y) {
                    break;
                }
            }
            if (m == newn) {
                cands[newn++] = cands[k];
            }
        }
        n = newn;
    }
    best_k = 0;
    best_score = score_zero = UINT_MAX;
    for (k = 0; k < n; k++) {
        dx = cands[k].x;
        dy = cands[k].y;
        if (invalid_block(ref, bx + dx, by + dy, bw, bh, 0)) {
            continue;
        }
        score = P_hier_metr(level, sblk, src->stride, at(ref, bx + dx, by + dy), ref->stride, bw, bh, &psy);
        if (dx == 0 && dy == 0) {
            score_zero = score;
        }
        score += (unsigned) mv_cost(c, mvf, i, j, dx * step * 4, dy * step * 4, level);
        if (dx == lax && dy == lay) {
            score = (unsigned) ORC_MAX((int) score - (motion_bias >> level), 0);
        }
        if (best_score > score) {
            best_score = score;
            best_k = k;
        }
    }
    dx = cands[best_k].x;
    dy = cands[best_k].y;
    memset(mv, 0, sizeof(*mv));
    best = best_score;
    qthresh = (unsigned) (c->quant * bw * bh >> 11);
    {
        unsigned zoscore = P_metr(sblk, src->stride, at(ogr, bx, by), ogr->stride, bw, bh, &psy);
        if (bs_abs(dx) <= 1 && bs_abs(dy) <= 1) {
            qthresh *= 2;
        }
        if (zoscore < qthresh) {
            best = level == 0 ? score_zero : 0;
            dx = dy = 0;
            good_enough = 1;
        }
    }
    if (!good_enough) { /* refine_best_fpel_cand, hme.c:1300 */
        unsigned metr[4] = { UINT_MAX, UINT_MAX, UINT_MAX, UINT_MAX };
        int again = 1;
        while (again && !good_enough) {
            int tvx, tvy;
            again = 0;
            for (k = 0; k < 5; k++) {
                tvx = dx + rectx[k];
                tvy = dy + recty[k];
                if (invalid_block(ref, bx + tvx, by + tvy, bw, bh, 0)) {
                    continue;
                }
                score = P_hier_metr(level, sblk, src->stride, at(ref, bx + tvx, by + tvy), ref->stride, bw, bh, &psy);
                if (k >= 1) {
                    metr[k - 1] = score;
                }
                if (level == 0 && !tvx && !tvy && score <= qthresh) {
                    dx = tvx;
                    dy = tvy;
                    best = score;
                    good_enough = 1;
                    break;
                }
                score += (unsigned) mv_cost(c, mvf, i, j, tvx * step * 4, tvy * step * 4, level);
                if (best > score) {
                    best = score;
                    dx = tvx;
                    dy = tvy;
                    again = 1;
                    break;
                }
            }
            if (again || good_enough) {
                continue;
            }
            tvx = dx + (metr[0] <= metr[1] ? 1 : -1);
            tvy = dy + (metr[2] <= metr[3] ? 1 : -1);
            if (invalid_block(ref, bx + tvx, by + tvy, bw, bh, 0)) {
                break;
            }
            score = P_hier_metr(level, sblk, src->stride, at(ref, bx + tvx, by + tvy), ref->stride, bw, bh, &psy);
            score += (unsigned) mv_cost(c, mvf, i, j, tvx * step * 4, tvy * step * 4, level);
            if (best > score) {
                best = score;
                dx = tvx;
                dy = tvy;
                again = 1;
            }
        }
    }
    mv->x = (int16_t) (dx * step);
    mv->y = (int16_t) (dy * step);
    if (level != 0) {
        return;
    }

    { /* sub-pel refinement + mode decision (hme.c:1598-1821) */
        int fpelx = mv->x, fpely = mv->y, sx = 0, sy = 0, found_sub = 0;
        unsigned yarea = (unsigned) (bw * bh), best_fp;
        const hplane *ref0 = &c->ref[0];
        const uint8_t *refd, *ogrd;
        unsigned var_ref, avg_ref, mad, ogrerr, ogrmad, avg_y_dif, avg_c_dif, ratio = 32, chroma_ratio;
        int uavg_src, vavg_src, uavg_ref, vavg_ref, cbx, cby, cbw, cbh, cbmx, cbmy;
        int eprmi, eprmd, eprmr, neidif, oob, ipolvar, dv, skipped = 0;
        unsigned skipt = ((unsigned) c->quant * (unsigned) c->quant) >> 19;
        bs_chroma_psy cpsy;
        const orc_mv *refmv = c->ref_mvf ? &c->ref_mvf[i + j * nxb] : NULL;
        int hs = P->hshift, vs = P->vshift;

        if (fpelx == lax && fpely == lay) {
            best += (unsigned) motion_bias;
        }
        best_fp = best;
        mv->x = mv->y = 0;
        if (P->effort >= 4) {
            if (!invalid_block(ref0, bx + lax, by + lay, bw, bh, 4)) {
                best = subpixel_me(c, mvf, &sx, &sy, lax, lay, i, j, best_fp, bx, by, bw, bh, &psy);
                if (sx || sy) {
                    fpelx = lax;
                    fpely = lay;
                    found_sub = 1;
                }
            }
            if (!found_sub && !good_enough && !invalid_block(ref0, bx + fpelx, by + fpely, bw, bh, 4)) {
                best = subpixel_me(c, mvf, &sx, &sy, fpelx, fpely, i, j, best_fp, bx, by, bw, bh, &psy);
            }
        }
        mv->x = (int16_t) (fpelx * 4 + sx);
        mv->y = (int16_t) (fpely * 4 + sy);

        if ((mv->x | mv->y) & 3) {
            ratio = (best << 5) / (best_fp + !best_fp);
        }
        ogrd = at(&c->ogr[0], bx + fpelx, by + fpely);
        refd = at(ref0, bx + fpelx, by + fpely);
        ogrerr = P_metr(sblk, src->stride, ogrd, c->ogr[0].stride, bw, bh, &psy);
        ogrmad = (ogrerr + yarea / 2) / yarea;
        ogrmad = ogrmad * ratio >> 5;
        mad = (best + yarea / 2) / yarea;
        var_ref = (unsigned) bs_block_detail(refd, ref0->stride, bw, bh, &avg_ref);
        dv = (int) ORC_MIN(ratio, 32u);
        ipolvar = (int) ((var_src * (unsigned) dv + var_ref * (unsigned) (32 - dv)) >> 5);
        dv = bs_abs((int) var_src - ipolvar);
        if (var_src > 16 * yarea && var_src < 32 * yarea) {
            mv->flags |= ORC_MV_MAINTAIN;
        }
        cbx = i * (y_w >> hs);
        cby = j * (y_h >> vs);
        cbmx = cbx + orc_sar(fpelx, hs);
        cbmy = cby + orc_sar(fpely, vs);
        cbw = bw >> hs;
        cbh = bh >> vs;
        chroma_ratio = (unsigned) ((cbw * cbh) << 4) / yarea;
        uavg_src = P_plane_avg(&c->srcc[0], cbx, cby, cbw, cbh);
        vavg_src = P_plane_avg(&c->srcc[1], cbx, cby, cbw, cbh);
        uavg_ref = P_plane_avg(&c->refc[0], cbmx, cbmy, cbw, cbh);
        vavg_ref = P_plane_avg(&c->refc[1], cbmx, cbmy, cbw, cbh);
        bs_chroma_analysis(&cpsy, (int) avg_src, uavg_src, vavg_src);
        avg_y_dif = (unsigned) bs_abs((int) avg_src - (int) avg_ref);
        avg_c_dif = (unsigned) AVG2(bs_abs(uavg_src - uavg_ref), bs_abs(vavg_src - vavg_ref));
        P_calc_eprm(sblk, src->stride, refd, ref0->stride, (int) avg_src, (int) avg_ref, bw, bh, &eprmi, &eprmd, &eprmr);
        {
            int px = i * y_w + orc_sar(mv->x, 2), py = j * y_h + orc_sar(mv->y, 2);
            oob = px < 0 || py < 0 || px >= ((nxb - 1) * y_w) - 1 || py >= ((nyb - 1) * y_h) - 1;
        }
        {
            int a, b;
            neighbordif2(mvf, nxb, i, j, &a, &b);
            neidif = (a + b) / 3;
        }
        if ((good_enough || (mv->x == 0 && mv->y == 0)) && c->skip_block_thresh >= 0 && !P->lossless) { /* skip test */
            unsigned cth, sth = skipt * yarea, zsub[3];
            sth += 4 * var_src;
            sth += yarea * (unsigned) c->skip_block_thresh;
            if (c->quant < (1 << 10)) {
                sth = sth * (unsigned) c->quant >> 10;
            }
            if (avg_y_dif <= 2) {
                sth = ORC_MAX(sth, 3 * (yarea + var_src));
            }
            sth = ORC_MAX(sth, yarea);
            if (good_enough) {
                sth *= 2;
            }
            P_yuv_max_subblock_err(zsub, c, bx, by, bx, by, bw, bh, cbx, cby, cbx, cby, cbw, cbh, &psy);
            cth = chroma_ratio * sth * ORC_MAX(skipt, 1u) >> 5;
            zsub[0] = zsub[0] * ratio >> 5;
            zsub[1] = zsub[1] * ratio >> 5;
            zsub[2] = zsub[2] * ratio >> 5;
            zsub[0] += (unsigned) SQR((int) avg_src - (int) avg_ref) * yarea;
            if (zsub[0] <= sth && zsub[1] <= cth && zsub[2] <= cth) {
                mv->flags |= ORC_MV_SKIP;
                mv->x = mv->y = 0;
                mv->err = 0;
                skipped = 1;
            }
        }
        if (!skipped) {
            if (!oob && !P->lossless) {
                int y_prereq = avg_y_dif <= 2, c_prereq = !cpsy.greyish && avg_c_dif <= 2;
                if (y_prereq || c_prereq) {
                    unsigned bsub[3], xth = skipt * yarea;
                    int utex, vtex, carea = 4 * cbw * cbh;
                    P_yuv_max_subblock_err(bsub, c, bx, by, bx + fpelx, by + fpely, bw, bh, cbx, cby, cbmx, cbmy, cbw, cbh, &psy);
                    xth += (unsigned) ipolvar;
                    xth = (unsigned) ORC_MAX((int) xth - ((int) yarea * neidif * 2), 0);
                    xth = xth * (unsigned) c->quant >> 12;
                    xth = ORC_CLAMP(xth, 32u, yarea * 4);
                    bsub[0] = bsub[0] * ratio >> 5;
                    bsub[1] = bsub[1] * ratio >> 5;
                    bsub[2] = bsub[2] * ratio >> 5;
                    if (y_prereq && bsub[0] < 4 * xth) {
                        mv->flags |= ORC_MV_NOXMITY;
                    }
                    utex = (int) bs_block_tex(at(&c->srcc[0], cbx, cby), c->srcc[0].stride, cbw, cbh);
                    vtex = (int) bs_block_tex(at(&c->srcc[1], cbx, cby), c->srcc[1].stride, cbw, cbh);
                    c_prereq &= (utex > carea || vtex > carea);
                    xth = chroma_ratio * xth >> 4;
                    if (c_prereq && bsub[1] < xth && bsub[2] < xth) {
                        mv->flags |= ORC_MV_NOXMITC;
                    }
                }
                if ((unsigned) dv < var_src / 4) {
                    mv->flags |= ORC_MV_SIMCMPLX;
                }
            }
            test_subblock_intra_y(c, refmv, mv, sblk, src->stride, refd, ref0->stride, ipolvar, (int) avg_src, neidif, ratio, bw, bh);
            test_subblock_intra_c(c, mv, mad, (unsigned) (ipolvar / (bw * bh)), avg_src, cbx, cby, cbmx, cbmy, cbw, cbh);
            if (!(mv->flags & ORC_MV_NOXMITY)) {
                mv->err = (uint16_t) mad;
                c->total_err += mad;
            }
            c->ndiff += (ogrmad > 11) + (avg_c_dif >= 32);
        }
        if (best > 0) {
            c->eligible++;
        }
        if (mv->flags & ORC_MV_INTRA) {
            int merged = (mv->dc & 0x100) ? eprmd : eprmi;
            if (mv->submask != 0xF) {
                merged |= eprmr;
            }
            mv->flags = (mv->flags & ~(uint32_t) ORC_MV_EPRM) | (merged ? ORC_MV_EPRM : 0);
            c->nintra++;
            mv->x = (int16_t) (fpelx * 4);
            mv->y = (int16_t) (fpely * 4);
        } else {
            int merged = eprmr;
            if (mv->submask) {
                merged |= eprmi;
            }
            mv->flags = (mv->flags & ~(uint32_t) ORC_MV_EPRM) | (merged ? ORC_MV_EPRM : 0);
        }
        if (mv->flags & (ORC_MV_INTRA | ORC_MV_EPRM)) {
            mv->flags &= ~(uint32_t) ORC_MV_SIMCMPLX;
        }
    }
}

/*
 * dsv_hme (hme.c:2001): all levels, coarse to fine.  planes[] carry luma of each level for
 * src / reconstructed ref / original ref; mvf[level] must each hold nblocks entries.
 * Returns the intra percentage; *scb = scene-change block %, *avg_err = mean block error.
 */
int
orc_hme(hme_ctx *c, int *scb, int *avg_err)
{
    int level, gx = 0, gy = 0;
    int nxb = c->p.nblocks_h, nyb = c->p.nblocks_v;

    for (level = c->pyr_levels; level >= 0; level--) {
        int step = 1 << level;
        int nbx = (nxb + step - 1) / step, nby = (nyb + step - 1) / step;
        int t, bj;
        memset(c->mvf[level], 0, sizeof(orc_mv) * (size_t) nxb * (size_t) nyb);
        c->nintra = c->ndiff = c->eligible = 0;
        c->total_err = 0;
        /* anti-diagonal fronts; inside a front the order is reversed relative to raster */
        for (t = 0; t <= nbx + nby - 2; t++) {
            for (bj = ORC_MIN(nby - 1, t); bj >= 0; bj--) {
                int bi = t - bj;
                if (bi >= nbx) {
                    break;
                }
                hme_block(c, level, bi * step, bj * step, gx, gy);
            }
        }
        if (level != 0) { /* global_motion, hme.c:1973 */
            int i, j, ax = 0, ay = 0, nblk = 0;
            for (j = 0; j < nyb; j += step) {
                for (i = 0; i < nxb; i += step) {
                    ax += c->mvf[level][i + j * nxb].x;
                    ay += c->mvf[level][i + j * nxb].y;
                    nblk++;
                }
            }
            gx = nblk ? ax * 2 / nblk : 0;
            gy = nblk ? ay * 2 / nblk : 0;
        }
    }
    *scb = c->ndiff * 100 / (c->eligible ? c->eligible : 1);
    *avg_err = (int) (c->total_err / (unsigned) (nxb * nyb));
    return (c->nintra * 100) / (nxb * nyb);
}
