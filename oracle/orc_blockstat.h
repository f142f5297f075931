/*
 * oracle/orc_blockstat.h -- TEST INFRASTRUCTURE (see orc_common.h).
 *
 * Per-block statistics shared by intra analysis and motion estimation, restated from
 * reference src/hme.c (iisqrt :99, block_avg :436, block_tex :492, block_var :518,
 * block_detail :546, quant_tex :586, block_peaks :624, block_hist_var :711,
 * c_average :751, chroma_analysis :69).  Each is a pure function of a pixel block.
 */
#ifndef ORC_BLOCKSTAT_H
#define ORC_BLOCKSTAT_H

#include "orc_common.h"

static inline int bs_abs(int v) { return v < 0 ? -v : v; }

static inline unsigned
bs_isqrt(unsigned n) /* hme.c:99 */
{
    unsigned pos = 1u << 30, res = 0, rem = n;
    if (n == 0) {
        return 0;
    }
    while (pos > rem) {
        pos >>= 2;
    }
    while (pos) {
        unsigned dif = res + pos;
        res >>= 1;
        if (rem >= dif) {
            rem -= dif;
            res += pos;
        }
        pos >>= 2;
    }
    return res;
}

/* horizontal / vertical first-difference sums and the pixel sum of a block */
static inline void
bs_gradients(const uint8_t *a, int as, int w, int h, unsigned *sh, unsigned *sv, int *sum)
{
    int x, y;
    unsigned h_ = 0, v_ = 0;
    int s = 0;
    for (y = 0; y < h; y++) {
        const uint8_t *row = a + y * as, *up = y ? row - as : row;
        for (x = 0; x < w; x++) {
            s += row[x];
            v_ += (unsigned) bs_abs(row[x] - up[x]);
            if (x) {
                h_ += (unsigned) bs_abs(row[x] - row[x - 1]);
            }
        }
    }
    *sh = h_;
    *sv = v_;
    *sum = s;
}

static inline int
bs_abs_dev(const uint8_t *a, int as, int w, int h, int mean)
{
    int x, y, v = 0;
    for (y = 0; y < h; y++) {
        for (x = 0; x < w; x++) {
            v += bs_abs(a[y * as + x] - mean);
        }
    }
    return v;
}

static inline int
bs_block_avg(const uint8_t *a, int as, int w, int h)
{
    unsigned sh, sv;
    int s;
    bs_gradients(a, as, w, h, &sh, &sv, &s);
    return s / (w * h);
}

static inline unsigned
bs_block_tex(const uint8_t *a, int as, int w, int h)
{
    unsigned sh, sv;
    int s;
    bs_gradients(a, as, w, h, &sh, &sv, &s);
    return ORC_MAX(sh, sv);
}

static inline int
bs_block_var(const uint8_t *a, int as, int w, int h, unsigned *avg)
{
    int s = bs_block_avg(a, as, w, h);
    *avg = (unsigned) s;
    return bs_abs_dev(a, as, w, h, s);
}

static inline int
bs_block_detail(const uint8_t *a, int as, int w, int h, unsigned *avg) /* hme.c:546 */
{
    unsigned sh, sv;
    int s, var, tex;
    bs_gradients(a, as, w, h, &sh, &sv, &s);
    s /= (w * h);
    *avg = (unsigned) s;
    var = bs_abs_dev(a, as, w, h, s) >> 1;
    tex = (int) (ORC_MAX(sh, sv) - (unsigned) var);
    return var + ORC_MAX(tex, 0);
}

static inline int
bs_quant_tex(const uint8_t *a, int as, int w, int h) /* hme.c:586: texture of the 4-bit image, squared differences */
{
    int x, y;
    unsigned sh = 0, sv = 0;
    for (y = 0; y < h; y++) {
        const uint8_t *row = a + y * as, *up = y ? row - as : row;
        for (x = 0; x < w; x++) {
            int px = row[x] >> 4;
            int right = (x + 1 < w) ? (row[x + 1] >> 4) : px; /* the scan runs right-to-left from a copy of the last pixel */
            int d = px - right;
            sh += (unsigned) (d * d);
            d = px - (up[x] >> 4);
            sv += (unsigned) (d * d);
        }
    }
    return (int) (bs_isqrt(ORC_MAX(sh, sv)) / (unsigned) ((w + h + 1) >> 1));
}

/* 16-bin histogram of the block normalised by its mean; returns the scaled variance of the bins (hme.c:711) */
static inline unsigned
bs_hist_var(const uint8_t *a, int as, int w, int h)
{
    unsigned hist[16];
    unsigned avg, q16, var = 0;
    int x, y;
    memset(hist, 0, sizeof(hist));
    avg = (unsigned) bs_block_avg(a, as, w, h);
    if (avg == 0) {
        avg = 1;
    }
    q16 = (8u << 16) / avg;
    for (y = 0; y < h; y++) {
        for (x = 0; x < w; x++) {
            int hi = (int) (a[y * as + x] * q16 >> 16);
            hist[ORC_CLAMP(hi, 0, 15)]++;
        }
    }
    avg = (unsigned) (w * h) / 16;
    for (x = 0; x < 16; x++) {
        var += (hist[x] - avg) * (hist[x] - avg);
    }
    return (var * 16 * 16) / (16u * (unsigned) (w * h * w * h));
}

/* number of peaks in the 16-bin histogram of the 2x-decimated block (hme.c:624); bavg >= 0 */
static inline int
bs_peaks(const uint8_t *a, int as, int w, int h, int bavg)
{
    int hist[16];
    int x, y, avg = bavg ? bavg : 1, q16, maxv = 0, total = 0, np = 0;
    memset(hist, 0, sizeof(hist));
    q16 = (8 << 16) / avg;
    w /= 2;
    h /= 2;
    for (y = 0; y < h; y++) {
        for (x = 0; x < w; x++) {
            const uint8_t *p = a + (2 * y) * as + 2 * x;
            int ds = (int) ((unsigned) (p[0] + p[1] + p[as] + p[as + 1] + 2) >> 2);
            int hi = ds * q16 >> 16;
            hist[ORC_MIN(hi, 15)]++;
        }
    }
    for (x = 0; x < 16; x++) {
        maxv = ORC_MAX(maxv, hist[x]);
        total += hist[x];
    }
    avg = total / 16;
    maxv >>= 2;
    for (x = 0; x < 16; x++) {
        int c = hist[x], pk = 1;
        if (x > 0) {
            pk &= c > hist[x - 1];
        }
        if (x < 15) {
            pk &= c > hist[x + 1];
        }
        pk &= (c > maxv) || (c > avg);
        np += pk;
    }
    return np;
}

typedef struct {
    int nature, hifreq, greyish, skinnish;
} bs_chroma_psy;

static inline void
bs_chroma_analysis(bs_chroma_psy *c, int y, int u, int v) /* hme.c:69 */
{
    c->nature = u < 128 && v < 160;
    c->greyish = bs_abs(u - 128) < 8 && bs_abs(v - 128) < 8;
    c->skinnish = y > 80 && y < 230 && bs_abs(u - 108) < 24 && bs_abs(v - 148) < 24;
    c->hifreq = u > 160 && !c->greyish && !c->skinnish;
}

#endif
