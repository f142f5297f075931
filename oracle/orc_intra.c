/*
 * oracle/orc_intra.c -- TEST INFRASTRUCTURE (see orc_common.h).
 *
 * I-frame block analysis restated from reference src/hme.c:1835-1971
 * (dsv_intra_analysis): per block, a pure function of the source picture that
 * yields three flags -- RINGING (use the ringing subband filter / quantiser offsets),
 * MAINTAIN (protect low-detail blocks) and SKIP (reused as "keep high frequencies").
 * One independent work item per block: the HIP kernel is csrc/intra.hip.
 */
#include "orc_blockstat.h"

extern int orc_spatial_psy_factor(int blk_w, int blk_h, int nbh, int nbv, int sub);

void
orc_intra_analysis(const uint8_t *const planes[3], const int strides[3], const orc_params *p, orc_mv *out)
{
    int i, j;
    int scale = 2 * orc_spatial_psy_factor(p->blk_w, p->blk_h, p->nblocks_h, p->nblocks_v, -1);
    for (j = 0; j < p->nblocks_v; j++) {
        for (i = 0; i < p->nblocks_h; i++) {
            orc_mv *mv = &out[i + j * p->nblocks_h];
            int bx = i * p->blk_w, by = j * p->blk_h;
            int bw, bh, cbx, cby, cbw, cbh;
            unsigned luma_detail, luma_avg, var_t;
            int maintain = 1, keep_hf = 1, foliage = 0, is_text = 0, ringing = 0;
            const uint8_t *a;

            memset(mv, 0, sizeof(*mv));
            if (bx >= p->width || by >= p->height) {
                continue;
            }
            bw = ORC_MIN(p->width - bx, p->blk_w);
            bh = ORC_MIN(p->height - by, p->blk_h);
            cbx = i * (p->blk_w >> p->hshift);
            cby = j * (p->blk_h >> p->vshift);
            cbw = bw >> p->hshift;
            cbh = bh >> p->vshift;
            a = planes[0] + by * strides[0] + bx;
            luma_detail = (unsigned) bs_block_detail(a, strides[0], bw, bh, &luma_avg);

            if (p->do_psy & (16 | 2)) { /* ADAPTIVE_RINGING | CONTENT_ANALYSIS */
                bs_chroma_psy cp;
                int tf = 0, tf2 = 0, x, y, su = 0, sv = 0;
                int hvar = (int) bs_hist_var(a, strides[0], bw, bh);
                int qtex = bs_quant_tex(a, strides[0], bw, bh);
                int luma_var = bs_block_var(a, strides[0], bw, bh, &luma_avg) / (bw * bh);
                int luma_tex = (int) (bs_block_tex(a, strides[0], bw, bh) / (unsigned) (bw * bh));
                int npeaks = bs_peaks(a, strides[0], bw, bh, (int) luma_avg);

                is_text = bs_abs(npeaks - 2) <= 1;
                if (qtex == 1 || qtex == 2) {
                    tf2 = hvar <= 3 && (luma_tex >= 10 && luma_var >= luma_tex);
                }
                if (qtex == 2 || qtex == 3) {
                    tf = luma_tex >= 8 && luma_var >= 2 * luma_tex;
                    tf &= bs_abs(hvar - 5) <= 3;
                }
                is_text &= (tf || tf2);
                for (y = 0; y < cbh; y++) {
                    for (x = 0; x < cbw; x++) {
                        su += planes[1][(cby + y) * strides[1] + cbx + x];
                        sv += planes[2][(cby + y) * strides[2] + cbx + x];
                    }
                }
                bs_chroma_analysis(&cp, (int) luma_avg, su / (cbw * cbh), sv / (cbw * cbh));
                foliage = cp.nature && luma_avg < 160;
                foliage &= luma_detail > (unsigned) ((36 * bw * bh) / ORC_MAX(scale, 1));
                if (foliage) {
                    is_text = 0;
                }
                if ((p->do_psy & 16) && !cp.hifreq && (foliage || (hvar <= (ORC_MIN(qtex - 3, 2) * 16) && qtex > 1))) {
                    ringing = 1;
                }
                var_t = 8;
                if (cp.nature || cp.greyish || cp.skinnish) {
                    var_t += 12;
                } else if (!cp.hifreq) {
                    var_t += 8;
                }
            } else {
                var_t = 16;
            }
            if (p->do_psy & (2 | 1)) { /* CONTENT_ANALYSIS | ADAPTIVE_QUANT */
                luma_detail /= (unsigned) (bw * bh);
                keep_hf &= luma_detail < 48;
                maintain = luma_detail < var_t * 4;
            }
            if (p->do_psy & 2) {
                if (foliage) {
                    keep_hf = 0;
                    maintain = 1;
                } else if (is_text) {
                    keep_hf = 1;
                    maintain = 0;
                }
            }
            if ((p->do_psy & 16) && luma_avg < 24) {
                ringing = 1;
            }
            mv->flags = (ringing ? ORC_MV_RINGING : 0) | (maintain ? ORC_MV_MAINTAIN : 0) | (keep_hf ? ORC_MV_SKIP : 0);
        }
    }
}
