/*
 * oracle/orc_fmt.c -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
 *
 * Frame ingest / egress format conversions of the reference, restated per OUTPUT sample (the form the
 * kernels k_ingest_uyvy and k_to420 of digital-subband-video-2_amd/csrc/frame.hip use):
 *   orc_uyvy_to_planar   the UYVY branch of dsv_yuv_read            (reference src/dsv.c:177-205)
 *   orc_to420            conv444to422 + conv422to420, conv422to420,
 *                        conv411to420, conv410to420                 (reference src/util.c:79-153, used at
 *                                                                    src/dsv_main.c:1030-1048 for -out420p)
 * Parity status: PINNED -- tests/test_oracle_fmt.py compares both with the reference's own functions
 * (oracle/_ref/libdsv2refutil.so, compiled from the reference sources where they lie) on seeded planes,
 * including odd sizes.
 */
#include "orc_common.h"

/* interleaved U0 Y0 V0 Y1 ... rows of 2*w bytes -> planar Y (w x h), U, V (w/2 x h each), packed back to back */
void orc_uyvy_to_planar(const uint8_t *src, uint8_t *dst, int w, int h)
{
    uint8_t *Y = dst, *U = dst + (size_t) w * h, *V = U + (size_t) (w / 2) * h;
    int x, y;
    for (y = 0; y < h; y++) {
        for (x = 0; x < w; x++) {
            Y[(size_t) y * w + x] = src[(size_t) y * 2 * w + 2 * x + 1];
        }
        for (x = 0; x < w / 2; x++) {
            U[(size_t) y * (w / 2) + x] = src[(size_t) y * 2 * w + 4 * x];
            V[(size_t) y * (w / 2) + x] = src[(size_t) y * 2 * w + 4 * x + 2];
        }
    }
}

/* mode: 1 from 4:4:4, 2 from 4:2:2, 3 from 4:1:1, 4 from "4:1:0"; writes the dw x dh samples of the 4:2:0 chroma plane */
void orc_to420(const uint8_t *src, int ss, int sw, int sh, uint8_t *dst, int ds, int dw, int dh, int mode)
{
    int x, y;
    for (y = 0; y < dh; y++) {
        for (x = 0; x < dw; x++) {
            int v, x0, x1, y0, y1, a, b, sx, sy;
            switch (mode) {
                case 1:
                    x0 = 2 * x;
                    x1 = x0 < sw - 1 ? x0 + 1 : sw - 1;
                    y0 = 2 * y;
                    y1 = y0 < sh - 1 ? y0 + 1 : sh - 1;
                    a = (src[y0 * ss + x0] + src[y0 * ss + x1] + 1) >> 1;
                    b = (src[y1 * ss + x0] + src[y1 * ss + x1] + 1) >> 1;
                    v = (a + b + 1) >> 1;
                    break;
                case 2:
                    y0 = 2 * y;
                    y1 = y0 < sh - 1 ? y0 + 1 : sh - 1;
                    v = (src[y0 * ss + x] + src[y1 * ss + x] + 1) >> 1;
                    break;
                case 3:
                    y0 = 2 * y;
                    y1 = y0 < sh - 1 ? y0 + 1 : sh - 1;
                    sx = ORC_MIN(x >> 1, sw - 1);
                    v = (src[y0 * ss + sx] + src[y1 * ss + sx] + 1) >> 1;
                    break;
                default:
                    sx = ORC_MIN(x >> 1, sw - 1);
                    sy = ORC_MIN(y >> 1, sh - 1);
                    v = src[sy * ss + sx];
                    break;
            }
            dst[y * ds + x] = (uint8_t) v;
        }
    }
}
