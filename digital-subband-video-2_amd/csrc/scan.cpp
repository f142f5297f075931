// scan.cpp -- scan geometry of a coefficient plane (hzcc.c:40-57, 264-342): pure host arithmetic, kept apart from the kernels so
// that the decoder's parser can be built and fuzzed without a device (tests/parser_fuzz.cpp).
#include "quant.h"

namespace dsv2 {

static inline int rshift_up(int x, int s) { return (x + (1 << s) - 1) >> s; }
static inline int h_dimat(int level, int v) { return rshift_up(v, 3 - level); }
static inline int h_subband_off(int level, int sub, int w, int h)
{
    int o = 0;
    if (sub & 1) {
        o += rshift_up(w, 3 - level);
    }
    if (sub & 2) {
        o += rshift_up(h, 3 - level) * w;
    }
    return o;
}

void make_scan(ScanGeom *g, int w, int h) // scan order of hzcc.c:264-342
{
    int k = 1;
    g->w = w;
    g->h = h;
    g->off[0] = 0;
    g->sw[0] = h_dimat(0, w);
    g->sh[0] = h_dimat(0, h);
    for (int l = 0; l < 3; l++) {
        for (int s = 1; s <= 3; s++, k++) {
            g->off[k] = h_subband_off(l, s, w, h);
            g->sw[k] = h_dimat(l, w);
            g->sh[k] = h_dimat(l, h);
        }
    }
    g->base[0] = 0;
    for (k = 0; k < 10; k++) {
        g->base[k + 1] = g->base[k] + g->sw[k] * g->sh[k];
    }
}

} // namespace dsv2
