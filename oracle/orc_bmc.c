/*
 * oracle/orc_bmc.c -- TEST INFRASTRUCTURE (see orc_common.h).
 *
 * Block motion compensation, residual formation / reconstruction and the in-loop
 * 4x4 smoothing filters, restated from reference src/bmc.c in the decomposition used
 * by csrc/bmc.hip:
 *   - prediction, subtraction and reconstruction are independent per block;
 *   - the in-place, raster-order-dependent filters (luma_filter bmc.c:460,
 *     chroma_filter :605, dsv_intra_filter :391) are evaluated in WAVEFRONT order:
 *     cell (i, j) only depends on cells (i-1,j), (i-2,j), (i-1,j-1), (i,j-1), (i+1,j-1),
 *     so all cells with equal i + 2j are processed "simultaneously" here (in
 *     arbitrary order inside a front) and the result must equal the raster order.
 */
#include "orc_common.h"

typedef struct {
    uint8_t *data;
    int stride, w, h;
} oplane;

extern int orc_spatial_psy_factor(int blk_w, int blk_h, int nbh, int nbv, int sub);

static uint8_t clamp_u8(int v) { return (uint8_t) (v > 255 ? 255 : (v < 0 ? 0 : v)); }
static int iabs(int v) { return v < 0 ? -v : v; }

/* ------------------------------------------------------------------ */
/* prediction (bmc.c:815-923)                                           */

static int
block_mean(const uint8_t *p, int stride, int w, int h) /* bmc.c:25 */
{
    int x, y, s = 0;
    for (y = 0; y < h; y++) {
        for (x = 0; x < w; x++) {
            s += p[y * stride + x];
        }
    }
    return s / (w * h);
}

/* two-pass quarter-pel luma interpolation, 4-tap (19,-3) or (20,-4): bmc.c:661-769 */
static void
luma_subpel(uint8_t *dst, int ds, const uint8_t *ref, int rs, int bw, int bh, int dx, int dy, int tmc)
{
    int16_t tmp[(32 + 3) * 32];
    int large = iabs(dx) >= 8 || iabs(dy) >= 8;
    int fx = dx & 3, fy = dy & 3;
    int soft_x = large || !(fx & 1) || (tmc & 1);
    int soft_y = large || !(fy & 1) || (tmc & 1);
    int x, y;

    for (y = 0; y < bh + 3; y++) {
        for (x = 0; x < bw; x++) {
            int a = ref[y * rs + x], b = ref[y * rs + x + 1], c = ref[y * rs + x + 2], d = ref[y * rs + x + 3];
            int f = soft_x ? (19 * (b + c) - 3 * (a + d)) : (20 * (b + c) - 4 * (a + d));
            int v;
            switch (fx) {
                case 0: v = (64 * b + 32) >> 6; break;
                case 1: v = (f + 32 * b + 32) >> 6; break;
                case 2: v = (2 * f + 32) >> 6; break;
                default: v = (f + 32 * c + 32) >> 6; break;
            }
            tmp[y * 32 + x] = (int16_t) v;
        }
    }
    for (y = 0; y < bh; y++) {
        for (x = 0; x < bw; x++) {
            int a = tmp[y * 32 + x], b = tmp[(y + 1) * 32 + x], c = tmp[(y + 2) * 32 + x], d = tmp[(y + 3) * 32 + x];
            int f = soft_y ? (19 * (b + c) - 3 * (a + d)) : (20 * (b + c) - 4 * (a + d));
            int v;
            switch (fy) {
                case 0: v = (64 * b + 32) >> 6; break;
                case 1: v = (f + 32 * b + 32) >> 6; break;
                case 2: v = (2 * f + 32) >> 6; break;
                default: v = (f + 32 * c + 32) >> 6; break;
            }
            dst[y * ds + x] = clamp_u8(v);
        }
    }
}

static void
chroma_subpel(uint8_t *dst, int ds, const uint8_t *ref, int rs, int w, int h, int dx, int dy, int sh, int sv) /* bmc.c:772 */
{
    int hb = 2 + sh, vb = 2 + sv, hf = 1 << hb, vf = 1 << vb;
    int x, y;
    dx &= hf - 1;
    dy &= vf - 1;
    if (dx | dy) {
        int f0 = (hf - dx) * (vf - dy), f1 = dx * (vf - dy), f2 = (hf - dx) * dy, f3 = dx * dy;
        int sf = hb + vb, af = 1 << (sf - 1);
        for (y = 0; y < h; y++) {
            for (x = 0; x < w; x++) {
                dst[y * ds + x] = (uint8_t) ((f0 * ref[y * rs + x] + f1 * ref[y * rs + x + 1] + f2 * ref[(y + 1) * rs + x] +
                                              f3 * ref[(y + 1) * rs + x + 1] + af) >> sf);
            }
        }
    } else {
        for (y = 0; y < h; y++) {
            memcpy(dst + y * ds, ref + y * rs, (size_t) w);
        }
    }
}

static void
predict_block(const orc_mv *mv, const orc_params *p, int c, const oplane *rp, const oplane *dp, int i, int j)
{
    int sh = c ? p->hshift : 0, sv = c ? p->vshift : 0;
    int bw = p->blk_w >> sh, bh = p->blk_h >> sv;
    int limx = (dp->w - bw) + ORC_BORDER - 1, limy = (dp->h - bh) + ORC_BORDER - 1;
    int x = i * bw, y = j * bh;
    int px = x + orc_sar(mv->x, 2 + sh), py = y + orc_sar(mv->y, 2 + sv);
    uint8_t *dst = dp->data + y * dp->stride + x;
    int r;

    if (mv->flags & ORC_MV_INTRA) {
        px = ORC_CLAMP(px, -ORC_BORDER, limx);
        py = ORC_CLAMP(py, -ORC_BORDER, limy);
        if (mv->submask == 0xF) {
            int dc = (c == 0 && mv->dc) ? mv->dc : block_mean(rp->data + py * rp->stride + px, rp->stride, bw, bh);
            for (r = 0; r < bh; r++) {
                memset(dst + r * dp->stride, dc, (size_t) bw);
            }
        } else {
            int sbw = bw / 2, sbh = bh / 2, k;
            for (k = 0; k < 4; k++) { /* quadrant order 00, 01, 10, 11 (bmc.c:866-899) */
                int f = (k & 1) ? sbw : 0, g = (k & 2) ? sbh : 0;
                const uint8_t *src = rp->data + (py + g) * rp->stride + (px + f);
                uint8_t *d = dst + g * dp->stride + f;
                if (mv->submask & (1 << k)) {
                    int dc = (c == 0 && mv->dc) ? mv->dc : block_mean(src, rp->stride, sbw, sbh);
                    for (r = 0; r < sbh; r++) {
                        memset(d + r * dp->stride, dc, (size_t) sbw);
                    }
                } else {
                    for (r = 0; r < sbh; r++) {
                        memcpy(d + r * dp->stride, src + r * rp->stride, (size_t) sbw);
                    }
                }
            }
        }
        return;
    }
    if (c == 0) {
        if (!((mv->x | mv->y) & 3)) {
            px = ORC_CLAMP(px, -ORC_BORDER, limx);
            py = ORC_CLAMP(py, -ORC_BORDER, limy);
            for (r = 0; r < bh; r++) {
                memcpy(dst + r * dp->stride, rp->data + (py + r) * rp->stride + px, (size_t) bw);
            }
        } else {
            px = ORC_CLAMP(px - 1, -ORC_BORDER, limx);
            py = ORC_CLAMP(py - 1, -ORC_BORDER, limy);
            luma_subpel(dst, dp->stride, rp->data + py * rp->stride + px, rp->stride, bw, bh, mv->x, mv->y, p->temporal_mc);
        }
    } else {
        px = ORC_CLAMP(px, -ORC_BORDER, limx);
        py = ORC_CLAMP(py, -ORC_BORDER, limy);
        chroma_subpel(dst, dp->stride, rp->data + py * rp->stride + px, rp->stride, bw, bh, mv->x, mv->y, sh, sv);
    }
}

static void
predict_plane(const orc_mv *mvs, const orc_params *p, int c, const oplane *rp, const oplane *dp)
{
    int i, j;
    for (j = 0; j < p->nblocks_v; j++) {
        for (i = 0; i < p->nblocks_h; i++) {
            predict_block(&mvs[i + j * p->nblocks_h], p, c, rp, dp, i, j);
        }
    }
}

/* residual = source - prediction (bmc.c:989-1055); res holds the source on entry */
static void
subtract_plane(const orc_mv *mvs, const orc_params *p, int c, const oplane *res, const oplane *pred)
{
    int sh = c ? p->hshift : 0, sv = c ? p->vshift : 0;
    int bw = p->blk_w >> sh, bh = p->blk_h >> sv;
    int i, j, m, n;
    for (j = 0; j < p->nblocks_v; j++) {
        for (i = 0; i < p->nblocks_h; i++) {
            const orc_mv *mv = &mvs[i + j * p->nblocks_h];
            uint8_t *r = res->data + (j * bh) * res->stride + i * bw;
            const uint8_t *q = pred->data + (j * bh) * pred->stride + i * bw;
            int intra = mv->flags & ORC_MV_INTRA, skip = mv->flags & ORC_MV_SKIP;
            int noxmit = c == 0 ? (mv->flags & ORC_MV_NOXMITY) : (mv->flags & ORC_MV_NOXMITC);
            for (n = 0; n < bh; n++) {
                for (m = 0; m < bw; m++) {
                    int s = r[n * res->stride + m], pv = q[n * pred->stride + m], v;
                    if (p->lossless) {
                        v = (uint8_t) (s - pv + 128);
                    } else if (!intra && (skip || noxmit)) {
                        v = 128;
                    } else if (mv->flags & ORC_MV_EPRM) {
                        v = clamp_u8((s - pv + 256) >> 1);
                    } else {
                        v = clamp_u8(s - pv + 128);
                    }
                    r[n * res->stride + m] = (uint8_t) v;
                }
            }
        }
    }
}

/* out = prediction + residual (bmc.c:925-987) */
static void
reconstruct_plane(const orc_mv *mvs, const orc_params *p, int c, const oplane *res, const oplane *pred, const oplane *out)
{
    int sh = c ? p->hshift : 0, sv = c ? p->vshift : 0;
    int bw = p->blk_w >> sh, bh = p->blk_h >> sv;
    int i, j, m, n;
    for (j = 0; j < p->nblocks_v; j++) {
        for (i = 0; i < p->nblocks_h; i++) {
            const orc_mv *mv = &mvs[i + j * p->nblocks_h];
            const uint8_t *r = res->data + (j * bh) * res->stride + i * bw;
            const uint8_t *q = pred->data + (j * bh) * pred->stride + i * bw;
            uint8_t *o = out->data + (j * bh) * out->stride + i * bw;
            int plain = !(mv->flags & ORC_MV_EPRM) || (!(mv->flags & ORC_MV_INTRA) && (mv->flags & ORC_MV_SKIP));
            for (n = 0; n < bh; n++) {
                for (m = 0; m < bw; m++) {
                    int rv = r[n * res->stride + m], pv = q[n * pred->stride + m];
                    if (p->lossless) {
                        o[n * out->stride + m] = (uint8_t) (pv + rv - 128);
                    } else if (plain) {
                        o[n * out->stride + m] = clamp_u8(pv + rv - 128);
                    } else {
                        o[n * out->stride + m] = clamp_u8(pv + (rv - 128) * 2);
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* 4x4 edge smoothing primitives (bmc.c:53-191)                         */

/* one 6-sample line: e2 e1 e0 | i0 i1 i2 straddling an edge; returns 1 and the 4 new values */
static int
smooth6(int e2, int e1, int e0, int i0, int i1, int i2, int t, int out[4])
{
    int avg = (5 * (e0 + i0) + 3 * (e1 + i1) + 8) >> 4;
    if (iabs(e0 - avg) < t && iabs(i0 - avg) < t && iabs(e1 - avg) < t && iabs(i1 - avg) < t && iabs(e2 - avg) < t &&
        iabs(i2 - avg) < t) {
        int a5 = avg * 5;
        out[0] = (3 * (avg + e1) + 2 * e2 + 4) >> 3; /* new e1 */
        out[1] = (a5 + 2 * e1 + e2 + 4) >> 3;        /* new e0 */
        out[2] = avg;                                /* new i0 */
        out[3] = (a5 + 2 * i1 + i2 + 4) >> 3;        /* new i1 */
        return 1;
    }
    return 0;
}

/* generic: filter across the edges at position (x,y) along direction (sx: step across the edge, sl: step along it) */
static void
edge_filter(uint8_t *b, int across, int along, int in_edge, int tE, int tM)
{
    int n, o[4];
    for (n = 0; n < 4; n++) {
        uint8_t *p = b + n * along;
        if (smooth6(p[-3 * across], p[-2 * across], p[-across], p[0], p[across], p[2 * across], tE, o)) {
            p[-2 * across] = (uint8_t) o[0];
            p[0] = (uint8_t) o[2];
            p[-across] = (uint8_t) o[1];
            p[across] = (uint8_t) o[3];
        }
        if (in_edge) {
            uint8_t *k = p + 4 * across;
            /* mirrored roles: inner side is the block being closed (bmc.c:109-126) */
            if (smooth6(k[3 * across], k[2 * across], k[across], k[0], k[-across], k[-2 * across], tM, o)) {
                k[0] = (uint8_t) o[2];
                k[2 * across] = (uint8_t) o[0];
                k[-across] = (uint8_t) o[3];
                k[across] = (uint8_t) o[1];
            }
        }
    }
}

static void
hfilter(const oplane *dp, int x, int y, int edge, int tE, int tM) /* ihfilter4x4 bmc.c:70 */
{
    if (x < 4 || x > dp->w - 4 || (edge && tE <= 0) || tM <= 0) {
        return;
    }
    if (!edge) {
        tE = tM;
    }
    edge_filter(dp->data + y * dp->stride + x, 1, dp->stride, x < dp->w - 8, tE, tM);
}

static void
vfilter(const oplane *dp, int x, int y, int edge, int tE, int tM) /* ivfilter4x4 bmc.c:130 */
{
    if (y < 4 || y > dp->h - 4 || (edge && tE <= 0) || tM <= 0) {
        return;
    }
    if (!edge) {
        tE = tM;
    }
    edge_filter(dp->data + y * dp->stride + x, dp->stride, 1, y < dp->h - 8, tE, tM);
}

static void
ds2x2(const uint8_t *a, int as, int d[4])
{
    d[0] = (a[0] + a[1] + a[as] + a[as + 1] + 2) >> 2;
    d[1] = (a[2] + a[3] + a[as + 2] + a[as + 3] + 2) >> 2;
    a += 2 * as;
    d[2] = (a[0] + a[1] + a[as] + a[as + 1] + 2) >> 2;
    d[3] = (a[2] + a[3] + a[as + 2] + a[as + 3] + 2) >> 2;
}

static unsigned
dsff(const uint8_t *a, int as) /* bmc.c:194 */
{
    int d[4];
    unsigned sh, sv;
    ds2x2(a, as, d);
    sh = (unsigned) iabs((d[0] + d[1]) - (d[3] + d[2]));
    sv = (unsigned) iabs((d[2] + d[1]) - (d[3] + d[0]));
    if (ORC_MAX(sh, sv) < 8) {
        return 0;
    }
    d[2] = 255 - d[2];
    d[3] = 255 - d[3];
    sh = (unsigned) iabs(d[0] - d[1] + d[2] - d[3]);
    sv = (unsigned) iabs(d[0] + d[1] - d[2] - d[3]) >> 2;
    return sh > sv ? (3 * sh + sv + 2) >> 2 : (3 * sv + sh + 2) >> 2;
}

static void
artf(const uint8_t *a, int as, int *psh, int *psv, int *pslh, int *pslv) /* bmc.c:224-270 */
{
    int x, y, sh = 0, sv = 0, d[4], hh;
    for (y = 0; y < 4; y += 2) {
        for (x = 0; x < 4; x += 2) {
            int x0 = a[y * as + x], x1 = a[y * as + x + 1], x2 = a[(y + 1) * as + x], x3 = a[(y + 1) * as + x + 1];
            hh = iabs(x0 - x1 - x2 + x3) >> 1;
            sh += iabs(x0 - x1 + x2 - x3) + hh;
            sv += iabs(x0 + x1 - x2 - x3) + hh;
        }
    }
    *psh = sh;
    *psv = sv;
    ds2x2(a, as, d);
    hh = iabs(d[0] - d[1] - d[2] + d[3]) >> 1;
    *pslh = iabs(d[0] - d[1] + d[2] - d[3]) + hh;
    *pslv = iabs(d[0] + d[1] - d[2] - d[3]) + hh;
}

static void
degrad(uint8_t *a, int as) /* bmc.c:276 */
{
    int hist[16], sums[16];
    int x, y, lo = -1, hi = -1, alo, ahi, t;
    memset(hist, 0, sizeof(hist));
    memset(sums, 0, sizeof(sums));
    for (y = 0; y < 4; y++) {
        for (x = 0; x < 4; x++) {
            int v = a[y * as + x];
            hist[v >> 4]++;
            sums[v >> 4] += v;
        }
    }
    for (x = 0; x < 16; x++) {
        if (hist[x]) {
            if (lo < 0) {
                lo = x;
            }
            hi = x;
        }
    }
    if (lo >= hi) {
        return;
    }
    alo = sums[lo] / hist[lo];
    ahi = sums[hi] / hist[hi];
    if (alo == 0) {
        alo = 1;
    }
    if (ahi == 0) {
        ahi = 1;
    }
    t = (alo + ahi + 1) >> 1;
    for (y = 0; y < 4; y++) {
        for (x = 0; x < 4; x++) {
            int os = a[y * as + x];
            if (os < t) {
                a[y * as + x] = (uint8_t) (os + (hist[lo] * (alo - os)) / 16);
            } else if (os > t) {
                a[y * as + x] = (uint8_t) (os + (hist[hi] * (ahi - os)) / 16);
            }
        }
    }
}

static int
filter_q(const orc_params *p, int q) /* bmc.c:376 */
{
    int psyf = orc_spatial_psy_factor(p->blk_w, p->blk_h, p->nblocks_h, p->nblocks_v, -1);
    if (q > 1536) {
        q = 1536;
    }
    q += q * psyf >> 10;
    if (q < 1024) {
        q = 512 + q / 2;
    }
    return q;
}

static int
curve_tex(int tt) /* bmc.c:364 */
{
    if (tt < 8) {
        return (8 - tt) * 8;
    }
    if (tt > 192) {
        return 0;
    }
    return tt - 7;
}

/* motion-vector neighbour difference (dsv.c:403) */
static void
neighbordif2(const orc_mv *v, int nbh, int x, int y, int *dx, int *dy)
{
    const orc_mv *c = &v[x + y * nbh];
    int cx = c->x, cy = c->y, lx = cx, ly = cy, tx = cx, ty = cy;
    if (iabs(cx) < 2 && iabs(cy) < 2) {
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
    *dx = iabs(lx - cx) + iabs(ly - cy);
    *dy = iabs(tx - cx) + iabs(ty - cy);
}

/* ---- one 4x4 cell of each filter ---- */

static void
intra_cell(const oplane *dp, const orc_params *p, const uint8_t *bd, int q, int fthresh, int i, int j, int nsbx, int nsby)
{
    int x = i * 4, y = j * 4;
    int flags, sh, sv, shl, svl, tt = 32, mx;
    uint8_t *a = dp->data + y * dp->stride + x;
    if (y + 4 >= dp->h || x + 4 >= dp->w) {
        return;
    }
    flags = bd[(i * p->nblocks_h / nsbx) + (j * p->nblocks_v / nsby) * p->nblocks_h];
    if (flags & ORC_BD_RINGING) {
        return;
    }
    artf(a, dp->stride, &sh, &sv, &shl, &svl);
    mx = ORC_MAX(sh, sv);
    if (!(mx < 256 && mx > 8)) {
        return;
    }
    if (flags & (ORC_BD_MAINTAIN | ORC_BD_STABLE)) {
        tt = (int) dsff(a, dp->stride);
        if (flags & ORC_BD_STABLE) {
            tt = tt * 5 >> 2;
        }
    } else {
        tt >>= 2;
    }
    tt = tt * 2 / 3;
    tt = (tt * q) >> 12;
    tt = ORC_CLAMP(tt, 0, fthresh);
    hfilter(dp, x, y, 0, tt, tt);
    vfilter(dp, x, y, 0, tt, tt);
    tt = sh > sv ? (3 * sh + sv) : (3 * sv + sh);
    tt = curve_tex(tt);
    tt = 16 + ((tt + 2) >> 2);
    tt = (tt * q) >> 12;
    tt = ORC_CLAMP(tt, 0, fthresh);
    hfilter(dp, x, y, 0, tt, tt);
    vfilter(dp, x, y, 0, tt, tt);
}

static void
luma_cell(const oplane *dp, const orc_params *p, const orc_mv *vecs, int q, int fthresh, int do_filter, int sharpen,
          int i, int j, int nsbx, int nsby)
{
    int x = i * 4, y = j * 4;
    int fx = i * p->nblocks_h / nsbx, fy = j * p->nblocks_v / nsby;
    int edgeh = (x % p->blk_w) == 0, edgehs = (x % (p->blk_w / 2)) == 0;
    int edgev = (y % p->blk_h) == 0, edgevs = (y % (p->blk_h / 2)) == 0;
    const orc_mv *mv = &vecs[fx + fy * p->nblocks_h];
    int amx, amy, ndx = -1, ndy = -1;
    uint8_t *a = dp->data + y * dp->stride + x;

    if (y + 4 >= dp->h || (mv->flags & ORC_MV_SKIP) || x + 4 >= dp->w) {
        return;
    }
    amx = iabs(mv->x);
    amy = iabs(mv->y);
    if (do_filter) {
        neighbordif2(vecs, p->nblocks_h, fx, fy, &ndx, &ndy);
    }
    if (mv->flags & ORC_MV_INTRA) {
        int tH = ORC_CLAMP((64 * q) >> 12, 2, 32), tL = ORC_CLAMP((32 * q) >> 12, 2, 32);
        int eh = edgeh, ev = edgev;
        if (mv->submask != 0xF) {
            eh |= edgehs;
            ev |= edgevs;
        }
        hfilter(dp, x, y, eh, tH, tL);
        vfilter(dp, x, y, ev, tH, tL);
        return;
    }
    if (do_filter && (ndx || ndy)) {
        int tt, addx, addy, sh, sv, shl, svl;
        int eprm = (mv->flags & ORC_MV_EPRM) != 0;
        int eh = edgeh || eprm, ev = edgev || eprm;
        int tndc = (ndx + ndy + 1) >> 1;
        artf(a, dp->stride, &sh, &sv, &shl, &svl);
        if (sh < 2 * sv && sv < 2 * sh) {
            int ix, iy;
            if (ndx < amx) {
                ndx >>= 1;
            }
            if (ndy < amy) {
                ndy >>= 1;
            }
            shl = shl > 128 ? 0 : 128 - shl;
            svl = svl > 128 ? 0 : 128 - svl;
            ix = ORC_MIN(amx, 32);
            iy = ORC_MIN(amy, 32);
            tt = ((sh * (32 - iy) + shl * iy) + 16) >> 5;
            tt += ((sv * (32 - ix) + svl * ix) + 16) >> 5;
            tt = (tt + 1) >> 1;
            if (ndx < amy && ndy < amx) {
                tt = 0;
            }
        } else {
            tt = (sh + sv + 1) >> 1;
        }
        tt = (tt * tndc + 4) >> 3;
        tt = (ORC_MIN(tt, fthresh) * q) >> 12;
        addx = (ORC_MIN(ndy, fthresh) * q) >> 12;
        addy = (ORC_MIN(ndx, fthresh) * q) >> 12;
        if (sh > 2 * sv || amy > 2 * amx) {
            vfilter(dp, x, y, ev, tt + addy, tt);
        } else if (sv > 2 * sh || amx > 2 * amy) {
            hfilter(dp, x, y, eh, tt + addx, tt);
        } else {
            hfilter(dp, x, y, eh, tt + addx, tt);
            vfilter(dp, x, y, ev, tt + addy, tt);
        }
    }
    if (sharpen && (mv->x & 3) && (mv->y & 3) && ((mv->x | mv->y) & 1) && amx < 8 && amy < 8) {
        degrad(a, dp->stride);
    }
}

static void
chroma_block(const oplane *dp, const orc_params *p, const orc_mv *vecs, int q, int i, int j)
{
    int bw = p->blk_w >> p->hshift, bh = p->blk_h >> p->vshift;
    int x = i * bw, y = j * bh, z;
    const orc_mv *mv = &vecs[i + j * p->nblocks_h];
    int it = ORC_CLAMP((64 * q) >> 12, 2, 32);
    int tx = it, ty = it;

    if (mv->flags & ORC_MV_SKIP) {
        return;
    }
    if (!(mv->flags & ORC_MV_INTRA)) {
        int ndx, ndy, amx = iabs(mv->x), amy = iabs(mv->y);
        neighbordif2(vecs, p->nblocks_h, i, j, &ndx, &ndy);
        if (ndx < amy && ndy < amx) {
            tx = ty = 0;
        } else {
            tx = (ORC_MIN(ndy, 64) * q) >> 12;
            ty = (ORC_MIN(ndx, 64) * q) >> 12;
        }
    }
    for (z = 0; z < bh; z += 4) {
        if (y + z + 4 < dp->h) {
            hfilter(dp, x, y + z, 0, tx, tx);
        }
    }
    for (z = 0; z < bw; z += 4) {
        if (x + z + 4 < dp->w) {
            vfilter(dp, x + z, y, 0, ty, ty);
        }
    }
}

/* wavefront sweep: all (i,j) with i + 2j == t, t ascending; inside a front the order is
 * deliberately scrambled (descending j) to demonstrate independence */
#define WAVEFRONT(NX, NY, BODY)                                   \
    do {                                                          \
        int t_, i, j;                                             \
        for (t_ = 0; t_ <= ((NX) - 1) + 2 * ((NY) - 1); t_++) {   \
            for (j = ORC_MIN((NY) - 1, t_ / 2); j >= 0; j--) {    \
                i = t_ - 2 * j;                                   \
                if (i >= (NX)) {                                  \
                    break;                                        \
                }                                                 \
                BODY;                                             \
            }                                                     \
        }                                                         \
    } while (0)

void
orc_intra_filter(uint8_t *data, int stride, int w, int h, const orc_params *p, const uint8_t *bd, int q, int do_filter)
{
    oplane dp = {data, stride, w, h};
    int nsbx = w / 4, nsby = h / 4, fthresh;
    if (p->lossless || !do_filter) {
        return;
    }
    q = filter_q(p, q);
    fthresh = 32 * (14 - orc_lb2((unsigned) q));
    WAVEFRONT(nsbx, nsby, intra_cell(&dp, p, bd, q, fthresh, i, j, nsbx, nsby));
}

static void
luma_filter(const oplane *dp, const orc_params *p, const orc_mv *vecs, int q, int do_filter)
{
    int nsbx = dp->w / 4, nsby = dp->h / 4, fthresh;
    int sharpen = p->inter_sharpen ? p->temporal_mc : 0;
    if (p->lossless) {
        return;
    }
    q = filter_q(p, q);
    fthresh = 32 * (14 - orc_lb2((unsigned) q));
    WAVEFRONT(nsbx, nsby, luma_cell(dp, p, vecs, q, fthresh, do_filter, sharpen, i, j, nsbx, nsby));
}

static void
chroma_filter(const oplane *dp, const orc_params *p, const orc_mv *vecs, int q)
{
    if (p->lossless) {
        return;
    }
    WAVEFRONT(p->nblocks_h, p->nblocks_v, chroma_block(dp, p, vecs, q, i, j));
}

/* ------------------------------------------------------------------ */
/* frame-level entry points; planes are passed as arrays of 3 (data at pixel (0,0), bordered) */

typedef struct {
    uint8_t *data[3];
    int stride[3], w[3], h[3];
} oframe;

static oplane
pl(const oframe *f, int c)
{
    oplane p;
    p.data = f->data[c];
    p.stride = f->stride[c];
    p.w = f->w[c];
    p.h = f->h[c];
    return p;
}

void
orc_sub_pred(const orc_mv *mvs, const orc_params *p, const oframe *pred, const oframe *resd, const oframe *ref) /* bmc.c:1057 */
{
    int c;
    for (c = 0; c < 3; c++) {
        oplane pp = pl(pred, c), rp = pl(resd, c), fp = pl(ref, c);
        predict_plane(mvs, p, c, &fp, &pp);
        subtract_plane(mvs, p, c, &rp, &pp);
    }
}

void
orc_add_res(const orc_mv *mvs, const orc_params *p, int q, const oframe *resd, const oframe *pred, int do_filter) /* bmc.c:1072 */
{
    int c;
    for (c = 0; c < 3; c++) {
        oplane pp = pl(pred, c), rp = pl(resd, c);
        reconstruct_plane(mvs, p, c, &rp, &pp, &rp);
        if (c == 0) {
            luma_filter(&rp, p, mvs, q, do_filter);
        } else {
            chroma_filter(&rp, p, mvs, q);
        }
    }
}

void
orc_add_pred(const orc_mv *mvs, const orc_params *p, int q, const oframe *resd, const oframe *out, const oframe *ref,
             int do_filter) /* bmc.c:1093 */
{
    int c;
    for (c = 0; c < 3; c++) {
        oplane rp = pl(resd, c), op = pl(out, c), fp = pl(ref, c);
        predict_plane(mvs, p, c, &fp, &op);
        reconstruct_plane(mvs, p, c, &rp, &op, &op);
        if (c == 0) {
            luma_filter(&op, p, mvs, q, do_filter);
        } else {
            chroma_filter(&op, p, mvs, q);
        }
    }
}
