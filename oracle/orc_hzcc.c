/*
 * oracle/orc_hzcc.c -- TEST INFRASTRUCTURE (see orc_common.h).
 *
 * Per-subband adaptive quantisation ("HZCC", reference src/hzcc.c) restated in the
 * decomposition used by csrc/quant.hip:
 *   1. the plane is quantised + dequantised in place by four dependent passes
 *      (LL region, then detail levels l = 0, 1, 2); inside a pass every coefficient
 *      is independent, except that with odd subband sizes the scanned regions of
 *      adjacent levels overlap by one column / row (hzcc.c:40-57 rounds every size
 *      up), which makes the last column / row of a level depend on its first one:
 *      those "dependent" cells form a second phase of the pass;
 *   2. each pass writes the quantised value of every scanned coefficient into a
 *      dense array indexed by scan position (LL raster, then level-major,
 *      subband-major rasters: hzcc.c:264-342);
 *   3. entropy coding (zero-run UEG + NEG / adaptive Rice, hzcc.c:230-232, bs.c)
 *      is a serial walk over the nonzero entries of that array.
 *
 * Reference entry point restated: dsv_encode_plane (hzcc.c:586), hzcc_enc (:235),
 * dsv_decode_plane (:617), hzcc_dec (:451).
 */
#include "orc_common.h"

#define MAXLVL 3
#define S_LH 1
#define S_HL 2
#define S_HH 3
#define MINQUANT 8

typedef struct {
    int plane, isP, lossless, do_psy;
    int hshift, vshift;
    int blk_w, blk_h, nbh, nbv;
    const uint8_t *bd;
    const orc_mv *mvs;
} hz_ctx;

static int dimat(int level, int v) { return ORC_RSHIFT_UP(v, MAXLVL - level); }

static int
subband_off(int level, int sub, int w, int h) /* hzcc.c:40 */
{
    int o = 0;
    if (sub & 1) {
        o += ORC_RSHIFT_UP(w, MAXLVL - level);
    }
    if (sub & 2) {
        o += ORC_RSHIFT_UP(h, MAXLVL - level) * w;
    }
    return o;
}

static int udiv_up(int a, int b) { return (a + b - 1) / b; }

int
orc_spatial_psy_factor(int blk_w, int blk_h, int nbh, int nbv, int sub) /* hzcc.c:67 */
{
    int scale, lo, hi;
    if (sub == S_LH) {
        lo = udiv_up(352, blk_w);
        hi = udiv_up(1920, blk_w);
        scale = nbh;
    } else if (sub == S_HL) {
        lo = udiv_up(288, blk_h);
        hi = udiv_up(1080, blk_h);
        scale = nbv;
    } else {
        lo = udiv_up(352, blk_w) * udiv_up(288, blk_h);
        hi = udiv_up(1920, blk_w) * udiv_up(1080, blk_h);
        scale = nbh * nbv;
    }
    scale = ORC_MAX(0, scale - lo);
    return (scale << 7) / (hi - lo);
}

static int
psy(const hz_ctx *c, int sub)
{
    return orc_spatial_psy_factor(c->blk_w, c->blk_h, c->nbh, c->nbv, sub);
}

static int
lfquant(const hz_ctx *c, int q) /* hzcc.c:89 */
{
    int pf = psy(c, S_HH);
    q -= (q * pf >> 10);
    q = ORC_MAX(q, MINQUANT);
    if (c->plane) {
        if (q > 256) {
            q = 256 + q / 4;
        }
        return ORC_MIN(q, 768);
    }
    return ORC_MIN(q, 3072);
}

static int
hfquant(const hz_ctx *c, int q, int s, int l) /* hzcc.c:108 */
{
    int chroma = c->plane != 0;
    int pf = psy(c, s);
    q /= 2;
    pf = q * pf >> (7 + (c->isP ? 0 : 1));
    if (chroma) {
        int tl = l - 2;
        if (s == S_LH) {
            tl += c->hshift;
        } else if (s == S_HL) {
            tl += c->vshift;
        }
        q = (q * 6) / (4 - tl);
    } else {
        if (l == 1) {
            q += pf / 2;
        } else if (l == 2) {
            q += pf;
        }
    }
    if (c->isP) {
        if (l != 2) {
            if (l == 0) {
                q *= 2;
                q -= pf;
            } else {
                q -= pf / 2;
            }
        }
        return ORC_MAX(q / 4, MINQUANT);
    }
    q = q * (15 + 3 * l) / 16;
    if (!chroma) {
        if (l == 0) {
            q = (q * 3) / 8;
        } else if (s == S_HH) {
            q *= 2;
        }
    } else {
        q /= 4;
        if (s == S_HH) {
            q *= 2;
        }
    }
    return ORC_MAX(q, MINQUANT);
}

static int
tmq_for_P(int tmq, int flags, int parc) /* hzcc.c:164 */
{
    if (parc || (flags & (ORC_BD_STABLE | ORC_BD_EPRM))) {
        return tmq * 7 >> 3;
    }
    if (flags & ORC_BD_INTRA) {
        return tmq * 6 >> 3;
    }
    return tmq;
}

static int
tmq_for_I(int tmq, int flags, int parc, int l) /* hzcc.c:171 */
{
    int sm = flags & (ORC_BD_STABLE | ORC_BD_MAINTAIN);
    if (l == 0) {
        return tmq;
    }
    if (sm == ORC_BD_STABLE) {
        return (l == 2) ? (tmq >> 2) : (tmq / 3);
    }
    if (sm == ORC_BD_MAINTAIN) {
        return tmq >> ((flags & ORC_BD_RINGING) ? 2 : !parc);
    }
    if (sm == (ORC_BD_STABLE | ORC_BD_MAINTAIN)) {
        return (l == 2) ? (tmq >> (2 + !parc)) : (tmq >> 2);
    }
    return tmq;
}

static int
quant_sub(int v, int q, int sub) /* hzcc.c:209 */
{
    return (v >= 0 ? v - sub : v + sub) / q;
}

static int32_t
dequant_S(int v, unsigned q) /* hzcc.c:217 */
{
    return (int32_t) ((unsigned) v * q + ((v < 0) ? 0u - (q * 2 / 3) : (q * 2 / 3)));
}

static int32_t
dequant_D(int v, unsigned q) /* hzcc.c:224 */
{
    return (int32_t) ((unsigned) v * q + ((v < 0) ? 0u - (q / 2) : (q / 2)));
}

static int sgn(int x) { return x < 0 ? -1 : (x > 0 ? 1 : 0); }
static int iabs(int x) { return x < 0 ? -x : x; }

/* quantise one detail coefficient; returns the quantised value and stores the tmq used */
static int
quant_detail(const hz_ctx *c, int val, int qp, int l, int flags, const orc_mv *mv, int parc, int gparc, int *tmq_out)
{
    int tmq = qp, v;
    int texture = !parc, gtexture = !gparc;

    if (c->isP) {
        tmq = tmq_for_P(tmq, flags, parc);
        if ((c->do_psy & 8) && c->plane == 0) { /* DSV_PSY_P_VISUAL_MASKING, hzcc.c:371-380 */
            int small_mv = iabs(mv->x) < 32 && iabs(mv->y) < 32;
            if ((gtexture && texture) || (mv->flags & ORC_MV_EPRM) || ((mv->flags & ORC_MV_MAINTAIN) && small_mv)) {
                v = quant_sub(val, tmq, tmq >> 3);
            } else if (texture || !(flags & ORC_BD_SIMCMPLX)) {
                v = quant_sub(val, tmq, tmq / 6);
            } else {
                v = quant_sub(val, tmq, tmq >> 2);
            }
        } else {
            v = val / tmq;
        }
    } else {
        tmq = tmq_for_I(tmq, flags, parc, l);
        if ((c->do_psy & 4) && c->plane == 0) { /* DSV_PSY_I_VISUAL_MASKING, hzcc.c:387-414 */
            int smf = flags & (ORC_BD_MAINTAIN | ORC_BD_STABLE);
            if (flags & ORC_BD_RINGING) {
                v = quant_sub(val, tmq, -(tmq / 6));
            } else if (l == 0) {
                v = quant_sub(val, tmq, -(tmq >> 3));
            } else {
                int edge = sgn(parc) == sgn(val);
                int stp;
                if (smf == 0) {
                    stp = -tmq / 3;
                } else if (edge && smf == ORC_BD_STABLE) {
                    stp = tmq >> 3;
                } else {
                    stp = -tmq / 6;
                }
                v = quant_sub(val, tmq, stp);
            }
        } else if (c->plane) {
            v = quant_sub(val, tmq, -(tmq >> 3));
        } else {
            v = val / tmq;
        }
    }
    *tmq_out = tmq;
    return v;
}

/* scan geometry: 10 segments (LL + 3 levels x 3 subbands) */
typedef struct {
    int off[10], sw[10], sh[10], base[11];
} scan_geom;

static void
make_scan(scan_geom *g, int w, int h)
{
    int l, s, k = 1;
    g->off[0] = 0;
    g->sw[0] = dimat(0, w);
    g->sh[0] = dimat(0, h);
    for (l = 0; l < MAXLVL; l++) {
        for (s = 1; s <= 3; s++, k++) {
            g->off[k] = subband_off(l, s, w, h);
            g->sw[k] = dimat(l, w);
            g->sh[k] = dimat(l, h);
        }
    }
    g->base[0] = 0;
    for (k = 0; k < 10; k++) {
        g->base[k + 1] = g->base[k] + g->sw[k] * g->sh[k];
    }
}

int
orc_scan_length(int w, int h)
{
    scan_geom g;
    make_scan(&g, w, h);
    return g.base[10];
}

/*
 * Quantise + dequantise the plane in place (coefs[0], the global DC, is left
 * untouched and reported as 0: hzcc.c:265,599-602) and fill qv[scan position].
 */
void
orc_quant_plane(int32_t *coefs, int w, int h, int q, int plane, int isP, int lossless, int do_psy,
                int hshift, int vshift, int blk_w, int blk_h, int nbh, int nbv,
                const uint8_t *blockdata, const orc_mv *mvs, int32_t *qv)
{
    hz_ctx c;
    scan_geom g;
    int l, s, x, y, phase;
    int qf = q * 3 / 2; /* fix_quant, hzcc.c:59 */

    c.plane = plane;
    c.isP = isP;
    c.lossless = lossless;
    c.do_psy = do_psy;
    c.hshift = hshift;
    c.vshift = vshift;
    c.blk_w = blk_w;
    c.blk_h = blk_h;
    c.nbh = nbh;
    c.nbv = nbv;
    c.bd = blockdata;
    c.mvs = mvs;
    make_scan(&g, w, h);

    /* LL region: one uniform step (hzcc.c:308-328) */
    {
        int qp = lfquant(&c, qf);
        for (y = 0; y < g.sh[0]; y++) {
            for (x = 0; x < g.sw[0]; x++) {
                int32_t *cell = coefs + y * w + x;
                int v;
                if (x == 0 && y == 0) {
                    qv[0] = 0;
                    continue;
                }
                if (lossless) {
                    v = *cell;
                } else {
                    v = isP ? (*cell / qp) : quant_sub(*cell, qp, -(qp / 6));
                    *cell = v ? (isP ? dequant_D(v, (unsigned) qp) : dequant_S(v, (unsigned) qp)) : 0;
                }
                qv[g.base[0] + y * g.sw[0] + x] = v;
            }
        }
    }
    for (l = 0; l < MAXLVL; l++) {
        int sw = dimat(l, w), sh = dimat(l, h);
        int dbx = (nbh << ORC_BLOCK_P) / sw, dby = (nbv << ORC_BLOCK_P) / sh;
        /* does the scanned region of level l-1 spill one column / row into level l? */
        int xdep = 2 * dimat(l - 1, w) > sw, ydep = 2 * dimat(l - 1, h) > sh;
        for (phase = 0; phase < 2; phase++) {
            for (s = 1; s <= 3; s++) {
                int seg = 1 + l * 3 + (s - 1);
                int par = subband_off(l - 1, s, w, h), gpar = subband_off(l - 2, s, w, h);
                int qp = lossless ? 1 : hfquant(&c, qf, s, l);
                for (y = 0; y < sh; y++) {
                    for (x = 0; x < sw; x++) {
                        int dependent = ((s & 1) && xdep && x == sw - 1) || ((s & 2) && ydep && y == sh - 1);
                        int32_t *cell = coefs + g.off[seg] + y * w + x;
                        int v;
                        if (dependent != phase) {
                            continue;
                        }
                        if (lossless) {
                            v = *cell;
                        } else {
                            int bi = ((y * dby) >> ORC_BLOCK_P) * nbh + ((x * dbx) >> ORC_BLOCK_P);
                            int parc = coefs[par + (y >> 1) * w + (x >> 1)];
                            int gparc = coefs[gpar + (y >> 2) * w + (x >> 2)];
                            int tmq;
                            v = quant_detail(&c, *cell, qp, l, blockdata[bi], mvs ? &mvs[bi] : NULL, parc, gparc, &tmq);
                            *cell = v ? dequant_D(v, (unsigned) tmq) : 0;
                        }
                        qv[g.base[seg] + y * sw + x] = v;
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* bit writer / reader and the codes of reference src/bs.c             */

typedef struct {
    uint8_t *start;
    unsigned pos; /* in bits; the buffer is assumed zero-filled (bs.c:143) */
} obs;

static void obs_align(obs *b) { b->pos = (b->pos + 7) & ~7u; }
static void
obs_put_bit(obs *b, int v)
{
    if (v) {
        b->start[b->pos >> 3] |= (uint8_t) (0x80 >> (b->pos & 7));
    }
    b->pos++;
}
static void
obs_put_bits(obs *b, unsigned n, unsigned v)
{
    while (n--) {
        obs_put_bit(b, (v >> n) & 1);
    }
}
static void
obs_put_ueg(obs *b, unsigned v) /* bs.c:132 interleaved exp-Golomb */
{
    int nb = -1, i;
    unsigned x;
    v++;
    for (x = v; x; x >>= 1) {
        nb++;
    }
    for (i = nb - 1; i >= 0; i--) {
        b->pos++;
        obs_put_bit(b, (v >> i) & 1);
    }
    obs_put_bit(b, 1);
}
static void
obs_put_seg(obs *b, int v) /* bs.c:175 */
{
    int s = v < 0;
    unsigned a = (unsigned) (s ? -v : v);
    obs_put_ueg(b, a);
    if (a) {
        obs_put_bit(b, s);
    }
}
static void
obs_put_neg(obs *b, int v) /* bs.c:206 */
{
    int s = v < 0;
    unsigned a = (unsigned) (s ? -v : v);
    obs_put_ueg(b, a - 1);
    if (a) {
        obs_put_bit(b, s);
    }
}
static void
obs_put_nrice(obs *b, int v, int *rk, int damp) /* bs.c:237,272 */
{
    unsigned u = ((unsigned) (2 * v) ^ (v < 0 ? ~0u : 0u)) - 1;
    unsigned k = (unsigned) (*rk >> damp), qq = u >> k;
    if (qq) {
        (*rk)++;
    } else if (*rk > 0) {
        (*rk)--;
    }
    b->pos += qq;
    obs_put_bit(b, 1);
    obs_put_bits(b, k, u & ((1u << k) - 1));
}

/* serial entropy pass over the dense quantised array: hzcc.c:249-252, 264-328, 423-447 */
static void
entropy_plane(obs *b, const int32_t *qv, int w, int h)
{
    scan_geom g;
    int seg, i, run = 0, nruns = 0, vk = 0;
    unsigned startp;

    make_scan(&g, w, h);
    obs_align(b);
    startp = b->pos;
    b->pos += 24;
    obs_align(b);
    for (seg = 0; seg < 10; seg++) {
        int l = seg == 0 ? 0 : (seg - 1) / 3;
        for (i = g.base[seg]; i < g.base[seg + 1]; i++) {
            int v = qv[i];
            if (v) {
                obs_put_ueg(b, (unsigned) run);
                if (seg == 0) {
                    obs_put_neg(b, v);
                } else {
                    obs_put_nrice(b, v, &vk, 3 + l);
                }
                run = -1;
                nruns++;
            }
            run++;
        }
    }
    obs_align(b);
    {
        unsigned endp = b->pos;
        b->pos = startp;
        obs_put_bits(b, 24, (unsigned) nruns);
        b->pos = endp;
    }
}

/*
 * Whole dsv_encode_plane (hzcc.c:586): returns the number of bytes appended at
 * byte offset `byte_pos` of the zero-filled buffer `out`.
 */
int
orc_encode_plane(uint8_t *out, int byte_pos, int32_t *coefs, int w, int h, int q, int plane, int isP, int lossless,
                 int do_psy, int hshift, int vshift, int blk_w, int blk_h, int nbh, int nbv,
                 const uint8_t *blockdata, const orc_mv *mvs)
{
    obs b;
    int32_t LL = coefs[0];
    int32_t *qv = (int32_t *) calloc((size_t) orc_scan_length(w, h), sizeof(int32_t));
    unsigned startp, endp;

    b.start = out;
    b.pos = (unsigned) byte_pos * 8;
    startp = b.pos;
    b.pos += 32;
    obs_put_seg(&b, LL);
    orc_quant_plane(coefs, w, h, q, plane, isP, lossless, do_psy, hshift, vshift, blk_w, blk_h, nbh, nbv, blockdata, mvs, qv);
    entropy_plane(&b, qv, w, h);
    obs_put_bits(&b, 8, 0x55);
    obs_align(&b);
    endp = b.pos;
    b.pos = startp;
    obs_put_bits(&b, 32, (endp - startp) / 8 - 4);
    free(qv);
    return (int) (endp / 8) - byte_pos;
}
