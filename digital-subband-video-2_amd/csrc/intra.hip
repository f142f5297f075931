// intra.hip -- I-frame block analysis on gfx950.
//
// Replaces reference src/hme.c:1835-1971 (dsv_intra_analysis): per block, a pure function of
// the source picture producing the RINGING / MAINTAIN / SKIP("keep high frequencies") flags that
// steer the adaptive subband filters, the quantiser and the intra filter.  One wavefront per
// block (one workgroup of 64 lanes), statistics via the cooperative reductions of blockstat.h.
// CPU proof of the restatement: oracle/orc_intra.c.
#include "blockstat.h"
#include "bmc.h"
#include "hme.h"
#include "prio.h"

namespace dsv2 {

// tab == nullptr: the single picture `one`; otherwise blockIdx.z indexes a device table of pictures
__global__ __launch_bounds__(64) void k_intra_analysis(const IntraJob *__restrict__ tab, IntraJob one, AnalysisParams p)
{
    DSV2_CENSUS_SCOPE();
    __shared__ int hist[16];
    const IntraJob &job = tab ? tab[blockIdx.z] : one;
    const Planes3 &src = job.src;
    DSV_MV *out = job.out;
    int i = blockIdx.x, j = blockIdx.y;
    int lane = threadIdx.x;
    DSV_MV *mv = &out[i + j * p.nbh];
    int bx = i * p.blk_w, by = j * p.blk_h;
    if (bx >= p.width || by >= p.height) {
        if (lane == 0) {
            DSV_MV z = {};
            *mv = z;
        }
        return;
    }
    int bw = min(p.width - bx, p.blk_w), bh = min(p.height - by, p.blk_h);
    int cbx = i * (p.blk_w >> p.hshift), cby = j * (p.blk_h >> p.vshift);
    int cbw = bw >> p.hshift, cbh = bh >> p.vshift;
    const uint8_t *a = src.p[0].data + (ptrdiff_t) by * src.p[0].stride + bx;
    int as = src.p[0].stride;
    unsigned luma_avg, var_t;
    unsigned luma_detail = (unsigned) ws_block_detail(a, as, bw, bh, luma_avg);
    bool maintain = true, keep_hf = true, foliage = false, is_text = false, ringing = false;

    if (p.do_psy & (DSV_PSY_ADAPTIVE_RINGING | DSV_PSY_CONTENT_ANALYSIS)) {
        int hvar = (int) ws_hist_var(a, as, bw, bh, hist);
        int qtex = ws_quant_tex(a, as, bw, bh);
        int luma_var = ws_block_var(a, as, bw, bh, luma_avg) / (bw * bh);
        int luma_tex = (int) (ws_block_tex(a, as, bw, bh) / (unsigned) (bw * bh));
        int npeaks = ws_peaks(a, as, bw, bh, (int) luma_avg, hist);
        bool tf = false, tf2 = false;
        is_text = abs(npeaks - 2) <= 1;
        if (qtex == 1 || qtex == 2) {
            tf2 = hvar <= 3 && (luma_tex >= 10 && luma_var >= luma_tex);
        }
        if (qtex == 2 || qtex == 3) {
            tf = luma_tex >= 8 && luma_var >= 2 * luma_tex;
            tf = tf && (abs(hvar - 5) <= 3);
        }
        is_text = is_text && (tf || tf2);
        int uavg = ws_block_sum(src.p[1].data + (ptrdiff_t) cby * src.p[1].stride + cbx, src.p[1].stride, cbw, cbh) / (cbw * cbh);
        int vavg = ws_block_sum(src.p[2].data + (ptrdiff_t) cby * src.p[2].stride + cbx, src.p[2].stride, cbw, cbh) / (cbw * cbh);
        ChromaPsy cp = chroma_analysis((int) luma_avg, uavg, vavg);
        foliage = cp.nature && luma_avg < 160;
        foliage = foliage && (luma_detail > (unsigned) ((36 * bw * bh) / max(p.scale, 1)));
        if (foliage) {
            is_text = false;
        }
        if ((p.do_psy & DSV_PSY_ADAPTIVE_RINGING) && !cp.hifreq && (foliage || (hvar <= (min(qtex - 3, 2) * 16) && qtex > 1))) {
            ringing = true;
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
    if (p.do_psy & (DSV_PSY_CONTENT_ANALYSIS | DSV_PSY_ADAPTIVE_QUANT)) {
        luma_detail /= (unsigned) (bw * bh);
        keep_hf = keep_hf && luma_detail < 48;
        maintain = luma_detail < var_t * 4;
    }
    if (p.do_psy & DSV_PSY_CONTENT_ANALYSIS) {
        if (foliage) {
            keep_hf = false;
            maintain = true;
        } else if (is_text) {
            keep_hf = true;
            maintain = false;
        }
    }
    if ((p.do_psy & DSV_PSY_ADAPTIVE_RINGING) && luma_avg < 24) {
        ringing = true;
    }
    if (lane == 0) {
        DSV_MV o = {};
        o.flags = (ringing ? (1u << DSV_MV_BIT_RINGING) : 0u) | (maintain ? (1u << DSV_MV_BIT_MAINTAIN) : 0u) |
                  (keep_hf ? (1u << DSV_MV_BIT_SKIP) : 0u);
        *mv = o;
    }
}

void intra_analysis(hipStream_t s, const DFrame &src, const AnalysisParams &p, DSV_MV *d_out)
{
    Planes3 pl;
    for (int c = 0; c < 3; c++) {
        pl.p[c] = src.p[c];
    }
    DSV2_LAUNCH(k_intra_analysis, dim3(p.nbh, p.nbv), dim3(64), 0, s, nullptr, IntraJob{pl, d_out}, p);
    HIPCHK(hipGetLastError());
}

void intra_analysis_batch(hipStream_t s, const IntraJob *d_jobs, int n, const AnalysisParams &p)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_intra_analysis, dim3(p.nbh, p.nbv, n), dim3(64), 0, s, d_jobs, IntraJob{}, p);
    HIPCHK(hipGetLastError());
}

} // namespace dsv2
