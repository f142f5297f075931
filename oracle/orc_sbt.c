/*
 * oracle/orc_sbt.c -- TEST INFRASTRUCTURE (see orc_common.h).
 *
 * Multi-level integer subband transform, forward and inverse, restated from
 * reference src/sbt.c in direct (gather) form:
 *   - every 1-D output sample is computed from the *unmodified* input vector
 *     (the reference lifts in place; each lifting step only reads the other
 *     parity, so the closed forms below are exactly equivalent),
 *   - the shrinking LL band ping-pongs between two scratch images while the
 *     high bands of every level are written once, straight to their final
 *     Mallat position in the coefficient plane.
 * This is the decomposition the HIP kernels in csrc/sbt.hip implement.
 *
 * Reference entry points restated: dsv_fwd_sbt (sbt.c:848), dsv_inv_sbt (sbt.c:890).
 */
#include "orc_common.h"

enum { F_HAAR = 0, F_LLI, F_LLP, F_CC, F_L2A, F_L1, F_LOSSLESS };

typedef struct {
    int plane, isP, lossless;
    const uint8_t *bd; /* per-block flag bytes */
    int nbh, nbv;
} sbt_ctx;

/* filter selection per (plane, frame type, level): sbt.c:22-29, 862-885 */
static int
pick_filter(const sbt_ctx *c, int l, int lvls)
{
    if (c->lossless) {
        return (l >= 1 && l <= lvls - 2) ? F_LOSSLESS : F_HAAR;
    }
    if (c->plane == 0) {
        if (l == 4) {
            return c->isP ? F_LLP : F_LLI;
        }
        if (!c->isP && l == 2) {
            return F_L2A;
        }
        if (!c->isP && l == 1) {
            return F_L1;
        }
        return F_HAAR;
    }
    if (!c->isP && l >= 1 && l <= lvls - 2) {
        return F_CC;
    }
    return F_HAAR;
}

static int
ovf_safety(const sbt_ctx *c, int l, int lvls) /* sbt.c:29 */
{
    return (l >= 6 && l >= (lvls - 3) && !c->lossless);
}

/* ------------------------------------------------------------------ */
/* 1-D analysis in closed form.  x(i) = p[i * s], n samples.           */

typedef struct {
    const int32_t *p;
    int s, n;
    /* adaptive (ringing) tap selection walk: flag byte for pair k is
     * sb[((k * delta2) >> 14) * sbs]  (sbt.c:227-238, 392-405) */
    const uint8_t *sb;
    int delta2, sbs;
} vec1d;

#define XV(v, i) ((v)->p[(i) * (v)->s])

static int
reflect_idx(int i, int nm1) /* sbt.c:106 with n := n-1 */
{
    if (i < 0) {
        i = -i;
    }
    if (i >= nm1) {
        i = nm1 + nm1 - i;
    }
    return i;
}

/* predict step on odd position i: sbt.c:191-197 DO_SIMPLE_HI with -= */
static int
hi3(const vec1d *v, int i)
{
    if (i < v->n - 1) {
        return XV(v, i) - ((XV(v, i - 1) + XV(v, i + 1) + 1) >> 1);
    }
    return XV(v, i) - XV(v, i - 1); /* last sample of an even-length vector */
}

/* 3-tap update on even position i: sbt.c:199-203 DO_SIMPLE_LO with += */
static int
lo3(const vec1d *v, int i)
{
    int even_n = v->n & ~1;
    if (i == 0) {
        return XV(v, 0) + (hi3(v, 1) >> 1);
    }
    if (i >= even_n) {
        return XV(v, i); /* trailing even sample of an odd-length vector is not lifted */
    }
    return XV(v, i) + ((hi3(v, i - 1) + hi3(v, i + 1) + 2) >> 2);
}

/* 5-tap update on even position i: sbt.c:216-225 */
static int
lo5(const vec1d *v, int i, int c0, int ca, int cs)
{
    int even_n = v->n & ~1;
    int nm1 = v->n - 1;
    if (i == 0) {
        return XV(v, 0) + (hi3(v, 1) >> 1);
    }
    if (i >= even_n) {
        return XV(v, i);
    }
    return XV(v, i) + ((-hi3(v, reflect_idx(i - 3, nm1)) +
                        c0 * (hi3(v, i - 1) + hi3(v, i + 1)) -
                        hi3(v, reflect_idx(i + 3, nm1)) + ca) >> cs);
}

static int
ringing_at(const vec1d *v, int walk_index)
{
    int sbp = walk_index * v->delta2;
    return v->sb[(sbp >> ORC_BLOCK_P) * v->sbs] & ORC_BD_RINGING;
}

/* adaptive 5-tap update (level-2 luma intra): sbt.c:227-238; pair k=i/2 uses walk index k-1 */
static int
lo5a(const vec1d *v, int i)
{
    if (i >= 2 && i < (v->n & ~1) && ringing_at(v, i / 2 - 1)) {
        return lo5(v, i, 3, 4, 3);
    }
    return lo5(v, i, 9, 16, 5);
}

static int
rgx(const vec1d *v, int i)
{
    return XV(v, reflect_idx(i, v->n - 1));
}

/* ASF93 9-tap low / 3-tap high FIR (sbt.c:243-276), even n only */
static void
l1_pair(const vec1d *v, int k, int *L, int *H)
{
    int n = v->n;
    int i = 2 * k; /* even centre */
    if (k == 0) {
        int h = hi3(v, 1);
        *L = 2 * (XV(v, 0) + (h >> 1));
        *H = 4 * h;
        return;
    }
    if (i == n - 2) {
        int h0 = hi3(v, n - 3);
        int h1 = hi3(v, n - 1);
        *L = 2 * (XV(v, n - 2) + ((h0 + h1 + 2) >> 2));
        *H = 4 * h1;
        return;
    }
    {
        int lo, hi;
        if (ringing_at(v, k)) {
            lo = 46 * rgx(v, i) + 20 * (rgx(v, i - 1) + rgx(v, i + 1)) - 9 * (rgx(v, i - 2) + rgx(v, i + 2)) -
                 4 * (rgx(v, i - 3) + rgx(v, i + 3)) + 2 * (rgx(v, i - 4) + rgx(v, i + 4));
        } else {
            lo = 46 * rgx(v, i) + 19 * (rgx(v, i - 1) + rgx(v, i + 1)) - 8 * (rgx(v, i - 2) + rgx(v, i + 2)) -
                 3 * (rgx(v, i - 3) + rgx(v, i + 3)) + 1 * (rgx(v, i - 4) + rgx(v, i + 4));
        }
        hi = 32 * rgx(v, i + 1) - 16 * (rgx(v, i) + rgx(v, i + 2));
        *L = (lo + 16) >> 5;
        *H = (hi + 4) >> 3;
    }
}

/* one analysis pair k -> (low, high); has_hi = 0 for the unpaired last sample of odd n */
static void
analysis_pair(int filter, const vec1d *v, int k, int *L, int *H)
{
    int i = 2 * k;
    int has_hi = (i + 1) < v->n;
    int lo = 0, hi = 0;

    if (filter == F_L1) {
        l1_pair(v, k, L, H);
        return;
    }
    if (has_hi) {
        hi = hi3(v, i + 1);
    }
    switch (filter) {
        case F_LLI:
            lo = lo3(v, i) * 5 / 2;
            hi = hi * 4;
            break;
        case F_LLP:
            lo = lo3(v, i) * 5 / 2;
            hi = hi * 2;
            break;
        case F_CC:
            lo = lo5(v, i, 3, 8, 4) * 2;
            break;
        case F_L2A:
            lo = lo5a(v, i) * 2;
            hi = hi * 3;
            hi = hi - orc_sar(hi, 3); /* SHREX, sbt.c:170-178 */
            break;
        default: /* F_LOSSLESS */
            lo = lo3(v, i);
            break;
    }
    *L = lo;
    *H = hi;
}

/* ------------------------------------------------------------------ */
/* 1-D synthesis in closed form.  The packed input is read through a
 * callback-free pair of accessors: low(k), high(k).                   */

typedef struct {
    const int32_t *lowp;  /* low(k)  = lowp[k * s]  */
    const int32_t *highp; /* high(k) = highp[k * s] */
    int s, n;
    const uint8_t *sb;
    int delta2, sbs;
} syn1d;

static int
syn_even_raw(int filter, const syn1d *v, int k)
{
    int a = v->lowp[k * v->s];
    switch (filter) {
        case F_LLI:
        case F_LLP:
            return a * 2 / 5;
        case F_CC:
        case F_L2A:
        case F_L1:
            return a / 2;
        default:
            return a;
    }
}

static int
syn_odd_raw(int filter, const syn1d *v, int k)
{
    int a = v->highp[k * v->s];
    switch (filter) {
        case F_LLI:
        case F_L1:
            return a / 4;
        case F_LLP:
            return a / 2;
        case F_L2A:
            a = a / 3;
            return a + orc_sar(a, 3);
        default:
            return a;
    }
}

/* odd raw sample by *position* (for the 5-tap reflected reads) */
static int
odd_at(int filter, const syn1d *v, int pos)
{
    return syn_odd_raw(filter, v, pos >> 1);
}

/* un-lifted even sample at position i = 2k */
static int
syn_even(int filter, const syn1d *v, int k)
{
    int i = 2 * k;
    int n = v->n, even_n = n & ~1, nm1 = n - 1;
    int e = syn_even_raw(filter, v, k);

    if (i == 0) {
        return e - (odd_at(filter, v, 1) >> 1);
    }
    if (i >= even_n) {
        return e;
    }
    if (filter == F_CC || filter == F_L2A) {
        int c0 = 3, ca = 8, cs = 4;
        if (filter == F_L2A) {
            int sbp = (k - 1) * v->delta2;
            if (v->sb[(sbp >> ORC_BLOCK_P) * v->sbs] & ORC_BD_RINGING) {
                c0 = 3, ca = 4, cs = 3;
            } else {
                c0 = 9, ca = 16, cs = 5;
            }
        }
        return e - ((-odd_at(filter, v, reflect_idx(i - 3, nm1)) +
                     c0 * (odd_at(filter, v, i - 1) + odd_at(filter, v, i + 1)) -
                     odd_at(filter, v, reflect_idx(i + 3, nm1)) + ca) >> cs);
    }
    return e - ((odd_at(filter, v, i - 1) + odd_at(filter, v, i + 1) + 2) >> 2);
}

/* reconstructed odd sample at position i = 2k + 1 */
static int
syn_odd(int filter, const syn1d *v, int k)
{
    int i = 2 * k + 1;
    int n = v->n;
    int o = syn_odd_raw(filter, v, k);

    if (i < n - 1) {
        if (filter == F_L1 && (n & 1) && i == n - 2) {
            return o; /* DO_SIMPLE_INV (sbt.c:205-213) never updates it for odd n */
        }
        return o + ((syn_even(filter, v, k) + syn_even(filter, v, k + 1) + 1) >> 1);
    }
    return o + syn_even(filter, v, k); /* i == n-1, even n */
}

/* ------------------------------------------------------------------ */
/* 2-D levels                                                          */

static int32_t *scratch[3];
static int scratch_len;

static void
need_scratch(int len)
{
    int i;
    if (scratch_len >= len) {
        return;
    }
    for (i = 0; i < 3; i++) {
        free(scratch[i]);
        scratch[i] = (int32_t *) calloc((size_t) len, sizeof(int32_t));
    }
    scratch_len = len;
}

/* separable forward level: S (LL of previous level) -> rows -> R -> columns ->
 * LL to D, high bands to C.  All images have row stride w.  (sbt.c:449-521) */
static void
fwd_separable(int filter, const sbt_ctx *c, const int32_t *S, int32_t *R, int32_t *D, int32_t *C,
              int w, int sw, int sh)
{
    int i, j, k;
    int hw = (sw + 1) / 2, hh = (sh + 1) / 2;
    int dbx = 0, dby = 0;

    if (filter == F_L2A || filter == F_L1) {
        dbx = (c->nbh << ORC_BLOCK_P) / sw;
        dby = (c->nbv << ORC_BLOCK_P) / sh;
    }
    for (j = 0; j < sh; j++) {
        vec1d v;
        v.p = S + j * w;
        v.s = 1;
        v.n = sw;
        v.sb = c->bd ? c->bd + ((j * dby) >> ORC_BLOCK_P) * c->nbh : NULL;
        v.delta2 = 2 * dbx;
        v.sbs = 1;
        for (k = 0; k < hw; k++) {
            int L, H;
            analysis_pair(filter, &v, k, &L, &H);
            R[j * w + k] = L;
            if (2 * k + 1 < sw) {
                R[j * w + hw + k] = H;
            }
        }
    }
    for (i = 0; i < sw; i++) {
        vec1d v;
        v.p = R + i;
        v.s = w;
        v.n = sh;
        v.sb = c->bd ? c->bd + ((i * dbx) >> ORC_BLOCK_P) : NULL;
        v.delta2 = 2 * dby;
        v.sbs = c->nbh;
        for (k = 0; k < hh; k++) {
            int L, H;
            analysis_pair(filter, &v, k, &L, &H);
            if (i < hw) {
                D[k * w + i] = L;
            } else {
                C[k * w + i] = L;
            }
            if (2 * k + 1 < sh) {
                C[(hh + k) * w + i] = H;
            }
        }
    }
}

/* Haar forward level (sbt.c:547-612) */
static void
fwd_haar(const int32_t *S, int32_t *D, int32_t *C, int w, int sw, int sh, int ovf)
{
    int hw = (sw + 1) / 2, hh = (sh + 1) / 2;
    int idx, jy;
    int dv = ovf ? 2 : 1;

    for (jy = 0; jy < hh; jy++) {
        for (idx = 0; idx < hw; idx++) {
            int x = 2 * idx, y = 2 * jy;
            int hasx = (x + 1) < sw, hasy = (y + 1) < sh;
            int x0 = S[y * w + x];
            if (hasx && hasy) {
                int x1 = S[y * w + x + 1], x2 = S[(y + 1) * w + x], x3 = S[(y + 1) * w + x + 1];
                D[jy * w + idx] = (x0 + x1 + x2 + x3) / dv;
                C[jy * w + hw + idx] = (x0 - x1 + x2 - x3);
                C[(hh + jy) * w + idx] = (x0 + x1 - x2 - x3);
                C[(hh + jy) * w + hw + idx] = (x0 - x1 - x2 + x3);
            } else if (hasy) { /* odd column */
                int x2 = S[(y + 1) * w + x];
                D[jy * w + idx] = 2 * (x0 + x2) / dv;
                C[(hh + jy) * w + idx] = 2 * (x0 - x2);
            } else if (hasx) { /* odd row */
                int x1 = S[y * w + x + 1];
                D[jy * w + idx] = 2 * (x0 + x1) / dv;
                C[jy * w + hw + idx] = 2 * (x0 - x1);
            } else {
                D[jy * w + idx] = (x0 * 4) / dv;
            }
        }
    }
}

/* number of levels: sbt.c:835-845 (the lb2++ there never fires: dsv_lb2 is a ceil-log2) */
static int
nlevels(int w, int h)
{
    return orc_lb2((unsigned) ORC_MAX(w, h));
}

/*
 * Forward transform of one plane.  plane points at pixel (0,0) of a (bordered)
 * u8 image of pw x ph pixels; coefs is cw x ch int32 (cw >= pw: the reference reads
 * cw columns from each of the ph rows, sbt.c:799-813; rows >= ph are zero).
 */
void
orc_fwd_sbt(const uint8_t *plane, int stride, int pw, int ph, int32_t *coefs, int cw, int ch,
            int plane_idx, int isP, int lossless, const uint8_t *blockdata, int nbh, int nbv)
{
    sbt_ctx c;
    int lvls, l, x, y;
    int32_t *S, *D, *R;

    (void) pw;
    c.plane = plane_idx;
    c.isP = isP;
    c.lossless = lossless;
    c.bd = blockdata;
    c.nbh = nbh;
    c.nbv = nbv;
    need_scratch(cw * ch);
    S = scratch[0];
    D = scratch[1];
    R = scratch[2];
    for (y = 0; y < ch; y++) {
        for (x = 0; x < cw; x++) {
            S[y * cw + x] = (y < ph) ? plane[y * stride + x] - 128 : 0;
        }
    }
    lvls = nlevels(cw, ch);
    for (l = 1; l <= lvls; l++) {
        int sw = ORC_RSHIFT_UP(cw, l - 1), sh = ORC_RSHIFT_UP(ch, l - 1);
        int filter = pick_filter(&c, l, lvls);
        int32_t *dst = (l == lvls) ? coefs : D;
        if (filter == F_HAAR) {
            fwd_haar(S, dst, coefs, cw, sw, sh, ovf_safety(&c, l, lvls));
        } else {
            fwd_separable(filter, &c, S, R, dst, coefs, cw, sw, sh);
        }
        if (l != lvls) {
            int32_t *t = S;
            S = D;
            D = t;
        }
    }
}

/* fetch from the Mallat image whose LL quadrant (hw x hh) lives in LLp and the rest in C */
static int
mallat_get(const int32_t *LLp, const int32_t *C, int w, int hw, int hh, int x, int y)
{
    return (x < hw && y < hh) ? LLp[y * w + x] : C[y * w + x];
}

static int round2(int v) { return (v + (v < 0 ? -1 : 1)) / 2; } /* sbt.c:93 */
static int round4(int v) { return (v + (v < 0 ? -2 : 2)) / 4; } /* sbt.c:99 */

static int
nudge(int LL, int lp, int ln, int band, int hqp) /* sbt.c:723-741 */
{
    int mx = LL - ln, mn = lp - LL, t;
    if (mn > mx) {
        t = mn;
        mn = mx;
        mx = t;
    }
    mx = ORC_MIN(mx, 0);
    mn = ORC_MAX(mn, 0);
    if (mx != mn) {
        int n;
        t = round4(lp - ln);
        n = round2(ORC_CLAMP(t, mx, mn) - band * 2);
        band += ORC_CLAMP(n, -hqp, hqp);
    }
    return band;
}

/* Haar inverse level; filtered != 0 selects the LL-gradient nudge variant (sbt.c:616-795) */
static void
inv_haar(const int32_t *LLp, const int32_t *C, int32_t *D, int w, int sw, int sh, int ovf, int filtered, int hqp)
{
    int hw = (sw + 1) / 2, hh = (sh + 1) / 2;
    int idx, jy;

    for (jy = 0; jy < hh; jy++) {
        for (idx = 0; idx < hw; idx++) {
            int x = 2 * idx, y = 2 * jy;
            int hasx = (x + 1) < sw, hasy = (y + 1) < sh;
            int LL = LLp[jy * w + idx] * (1 << ovf);
            if (hasx && hasy) {
                int LH = C[jy * w + hw + idx];
                int HL = C[(hh + jy) * w + idx];
                int HH = C[(hh + jy) * w + hw + idx];
                if (filtered) {
                    if (idx > 0) {
                        int lp = LLp[jy * w + idx - 1] * (1 << ovf);
                        int ln = mallat_get(LLp, C, w, hw, hh, idx + 1, jy) * (1 << ovf);
                        LH = nudge(LL, lp, ln, LH, hqp);
                    }
                    if (jy > 0) {
                        int lp = LLp[(jy - 1) * w + idx] * (1 << ovf);
                        int ln = mallat_get(LLp, C, w, hw, hh, idx, jy + 1) * (1 << ovf);
                        HL = nudge(LL, lp, ln, HL, hqp);
                    }
                }
                D[y * w + x] = (LL + LH + HL + HH) / 4;
                D[y * w + x + 1] = (LL - LH + HL - HH) / 4;
                D[(y + 1) * w + x] = (LL + LH - HL - HH) / 4;
                D[(y + 1) * w + x + 1] = (LL - LH - HL + HH) / 4;
            } else if (hasy) {
                int HL = C[(hh + jy) * w + idx];
                D[y * w + x] = (LL + HL) / 4;
                D[(y + 1) * w + x] = (LL - HL) / 4;
            } else if (hasx) {
                int LH = C[jy * w + hw + idx];
                D[y * w + x] = (LL + LH) / 4;
                D[y * w + x + 1] = (LL - LH) / 4;
            } else {
                D[y * w + x] = LL / 4;
            }
        }
    }
}

/* separable inverse level: columns first, then rows (sbt.c:462-473, 524-544) */
static void
inv_separable(int filter, const sbt_ctx *c, const int32_t *LLp, const int32_t *C, int32_t *R, int32_t *D,
              int w, int sw, int sh)
{
    int i, j, k;
    int hw = (sw + 1) / 2, hh = (sh + 1) / 2;
    int dbx = 0, dby = 0;

    if (filter == F_L2A) {
        dbx = (c->nbh << ORC_BLOCK_P) / sw;
        dby = (c->nbv << ORC_BLOCK_P) / sh;
    }
    for (i = 0; i < sw; i++) {
        syn1d v;
        v.lowp = (i < hw) ? LLp + i : C + i;
        v.highp = C + hh * w + i;
        v.s = w;
        v.n = sh;
        v.sb = c->bd ? c->bd + ((i * dbx) >> ORC_BLOCK_P) : NULL;
        v.delta2 = 2 * dby;
        v.sbs = c->nbh;
        for (k = 0; k < hh; k++) {
            R[(2 * k) * w + i] = syn_even(filter, &v, k);
            if (2 * k + 1 < sh) {
                R[(2 * k + 1) * w + i] = syn_odd(filter, &v, k);
            }
        }
    }
    for (j = 0; j < sh; j++) {
        syn1d v;
        v.lowp = R + j * w;
        v.highp = R + j * w + hw;
        v.s = 1;
        v.n = sw;
        v.sb = c->bd ? c->bd + ((j * dby) >> ORC_BLOCK_P) * c->nbh : NULL;
        v.delta2 = 2 * dbx;
        v.sbs = 1;
        for (k = 0; k < hw; k++) {
            D[j * w + 2 * k] = syn_even(filter, &v, k);
            if (2 * k + 1 < sw) {
                D[j * w + 2 * k + 1] = syn_odd(filter, &v, k);
            }
        }
    }
}

/*
 * Inverse transform of one plane: coefs (cw x ch) -> u8 plane (pw x ph written,
 * clamp(v + 128), sbt.c:817-831).  coefs is only read.
 */
void
orc_inv_sbt(uint8_t *plane, int stride, int pw, int ph, const int32_t *coefs, int cw, int ch, int q,
            int plane_idx, int isP, int lossless, const uint8_t *blockdata, int nbh, int nbv)
{
    sbt_ctx c;
    int lvls, l, x, y;
    const int32_t *LLp;
    int32_t *D, *R, *other;

    c.plane = plane_idx;
    c.isP = isP;
    c.lossless = lossless;
    c.bd = blockdata;
    c.nbh = nbh;
    c.nbv = nbv;
    need_scratch(cw * ch);
    D = scratch[0];
    other = scratch[1];
    R = scratch[2];
    lvls = nlevels(cw, ch);
    LLp = coefs;
    for (l = lvls; l > 0; l--) {
        int sw = ORC_RSHIFT_UP(cw, l - 1), sh = ORC_RSHIFT_UP(ch, l - 1);
        int filter = pick_filter(&c, l, lvls);
        int ovf = ovf_safety(&c, l, lvls);
        if (filter == F_HAAR) {
            int hqp = (plane_idx == 0) ? (q / (isP ? 14 : (l > 4 ? 2 : 8))) : (q / 2); /* sbt.c:903 */
            int filtered = !lossless && (plane_idx == 0 || !isP);                     /* sbt.c:925 */
            inv_haar(LLp, coefs, D, cw, sw, sh, ovf, filtered, hqp);
        } else {
            inv_separable(filter, &c, LLp, coefs, R, D, cw, sw, sh);
        }
        LLp = D;
        {
            int32_t *t = D;
            D = other;
            other = t;
        }
    }
    for (y = 0; y < ph; y++) {
        for (x = 0; x < pw; x++) {
            int v = LLp[y * cw + x] + 128;
            plane[y * stride + x] = (uint8_t) ORC_CLAMP(v, 0, 255);
        }
    }
}
