// hme_fast32.h -- level 0 of the fast block routine for 32 x 32 blocks (dsv_encoder.c:1203-1211: every picture of 2160p and up) and
// for 32 x 16 blocks (the same lines: pictures wider than 1280 and at least twice as wide as high -- 1920 x 800, 2560 x 1080), 4:2:0.
// Included by hme.hip behind hme_fast.h, whose per-quad primitives it uses unchanged.
//
// Layout (SrcBlk<NQ>): the block is FOUR 16 x 16 quadrants (NQ = 4), quadrant k at (16 (k & 1), 16 (k >> 1)), or the upper TWO of them
// (NQ = 2: 32 x 16); lane (qi, qj) owns quad (qi, qj) of each.  A block sum is the sum of the quadrants' sums before the metric's square root; first-difference sums add the differences
// ACROSS the quadrant seams (blk_grad_partials); the mode decision's four luma sub-blocks (hme.c:370, :891) are the quadrants
// themselves for a whole 32 x 32 block; for a clipped one -- widths are multiples of 16 here, heights of 16 (NQ = 4) or 8 (NQ = 2) --
// and for a 32 x 16 block (sub-blocks of 16 x 8: the halves of a quadrant) they are found per quad (sub_of).  The chroma block is
// 16 x 16 (16 x 8): a quad per lane, its four sub-blocks by lane as in the 4:4:4 form of the 16 x 16 routine.
// The sub-pel search works on the centred 16 x 16 window whatever the block size (hme.c:1100-1108): subpel_probes' LDS image, with
// the four neighbours' squared errors summed over the quadrants.
#pragma once

// pixel sum and horizontal / vertical first-difference sums of the lane's quads of a 32 x 32 (or clipped) block: the quadrants'
// own partials (neighbours inside a quadrant come from the neighbouring lanes) + the differences across the seams, where the
// left / upper neighbour is the last quad column / row of the quadrant next door
template <int NQ> __device__ __forceinline__ void blk_grad_partials(const Quad (&q)[NQ], const bool (&act)[NQ], int qi, int qj, int &sum, int &sh, int &sv)
{
    const int lane = hme_lane();
    sum = sh = sv = 0;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        int s, h, v;
        quad_grad_partials(q[k], act[k], qi, qj, 0, 0, s, h, v);
        sum += s;
        sh += h;
        sv += v;
        if (k & 1) { // seam to the quadrant on the left: its lane (7, qj)
            const int l2 = __shfl(q[k - 1].p2(), lane + 7, 64), l4 = __shfl(q[k - 1].p4(), lane + 7, 64);
            if (act[k] && qi == 0) {
                sh += abs(q[k].p1() - l2) + abs(q[k].p3() - l4);
            }
        }
        if (k & 2) { // seam to the quadrant above: its lane (qi, 7)
            const int u3 = __shfl(q[k - 2].p3(), lane + 56, 64), u4 = __shfl(q[k - 2].p4(), lane + 56, 64);
            if (act[k] && qj == 0) {
                sv += abs(q[k].p1() - u3) + abs(q[k].p2() - u4);
            }
        }
    }
}

// the four neighbour errors + the half-pel image of the centred window: subpel_probes (hme_fast.h) with the errors over four quadrants
template <int NQ, class Ctx>
__device__ __forceinline__ unsigned subpel_probes32(const Ctx &c, FastLds &S, int fpelx, int fpely, const SrcBlk<NQ> &B, const Psy &psy, unsigned &dirs)
{
    const int lane = hme_lane();
    const int qi = B.qi, qj = B.qj, bx = B.bx, by = B.by, bw = B.bw, bh = B.bh;
    const DPlane &src = c.src[0], &ref = c.ref[0];
    int v4[4] = {0, 0, 0, 0};
    Quad aw;
    {
        const int dxs[4] = {1, -1, 0, 0}, dys[4] = {0, 0, 1, -1};
        QuadRaw b4[4][NQ];
#pragma unroll
        for (int n = 0; n < 4; n++) {
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                b4[n][k] = ldq_raw(at(ref, bx + fpelx + dxs[n] + 16 * (k & 1), by + fpely + dys[n] + 16 * (k >> 1)), ref.stride, qi, qj, B.act[k]);
            }
        }
        const int xx = bx + ((bw >> 1) - 8), yy = by + ((bh >> 1) - 8);
        const QuadRaw awr = ldq_raw(at(src, xx, yy), src.stride, qi, qj, true); // the centred 16x16 source window
        const HpelWin hw = load_hpel_window(at(ref, xx + fpelx - 1, yy + fpely - 1), ref.stride);
        __builtin_amdgcn_sched_barrier(0);
        aw = ldq_finish(awr, true);
#pragma unroll
        for (int n = 0; n < 4; n++) {
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                v4[n] += B.act[k] ? (int) qsse(B.a[k], ldq_finish(b4[n][k], B.act[k])) : 0;
            }
        }
        uint32_t *win32 = (uint32_t *) S.sp.win;
        win32[lane] = hw.d0;
        if (lane + 64 < 100) {
            win32[lane + 64] = hw.d1;
        }
    }
    int r4 = reduceN<4>(v4);
    unsigned quad0 = (unsigned) bcastN<4>(r4, 0), quad1 = (unsigned) bcastN<4>(r4, 1), quad2 = (unsigned) bcastN<4>(r4, 2),
             quad3 = (unsigned) bcastN<4>(r4, 3);
    __syncthreads();
    build_hpel_at<20>(S.sp, S.sp.win);
    int pri0 = 0, pri1 = -1, sec0 = -1, sec1 = 0;
    unsigned ms1 = quad1, ms2 = quad3;
    if (quad3 >= quad2) {
        pri1 = 1;
        ms2 = quad2;
    }
    if (quad1 >= quad0) {
        sec0 = 1;
        ms1 = quad0;
    }
    if (ms2 > ms1) {
        int t0 = sec0, t1 = sec1;
        sec0 = pri0, sec1 = pri1;
        pri0 = t0, pri1 = t1;
    }
    dirs = (unsigned) (pri0 + 1) | ((unsigned) (pri1 + 1) << 2) | ((unsigned) (sec0 + 1) << 4) | ((unsigned) (sec1 + 1) << 6);
    int v8[8];
#pragma unroll
    for (int n = 0; n < 8; n++) {
        int t0, t1;
        subpel_probe_offset(dirs, n, t0, t1);
        int X = 4 + 8 * qi + t0, Y = 4 + 8 * qj + t1;
        const int ph = __builtin_amdgcn_readfirstlane((t0 & 1) | ((t1 & 1) << 1));
        Quad qs;
        qs.w = n < 7 ? qquad_ph(S.sp.h, X, Y, ph) : 0u;
        v8[n] = n < 7 ? (int) qmetric(aw, qs, psy) : 0;
    }
    int r8 = reduceN<8>(v8);
    unsigned acc = (unsigned) bcastL<8>(r8, lane & 7);
    __syncthreads(); // (the LDS image may be rebuilt by a second search)
    return metric_return(acc, 16, 16);
}

// sub-pel refinement + mode decision of a 32 x 32 / 32 x 16 block, 4:2:0 (hme.c:1598-1821): hme_l0_tail's arithmetic on four / two quadrants
template <bool FULL, int NQ, class Ctx>
__device__ __forceinline__ void hme_l0_tail32(const Ctx &c_in, int i, int j, FastLds &S, RowAcc &acc, DSV_MV *out, DSV_MV mv, const CostCtx &cc, const SrcBlk<NQ> &B,
                                              int lax, int lay, int motion_bias, bool good_enough, unsigned best, unsigned var_src, unsigned avg_src,
                                              const Psy &psy, const NbPre &pre)
{
    Ctx c = fenced(c_in, 0);
    const int lane = hme_lane();
    const int qi = B.qi, qj = B.qj, bx = B.bx, by = B.by, bw = B.bw, bh = B.bh;
    const int nxb = c.a.nbh, nyb = c.a.nbv, y_w = BlkDim<NQ>::W, y_h = BlkDim<NQ>::H;
    DPlane ref0 = c.ref[0];
    const int qw = bw >> 1, qh = bh >> 1;
    int fpelx = mv.u.mv.x, fpely = mv.u.mv.y, sx = 0, sy = 0;
    bool found_sub = false;
    const unsigned yarea = (unsigned) (bw * bh);
    if (fpelx == lax && fpely == lay) {
        best += (unsigned) motion_bias;
    }
    const unsigned best_fp = best;
    if (c.effort >= 4) {
        bool searched_lax = false;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) { // (see hme_l0_tail)
            const int ccx = pass == 0 ? lax : fpelx, ccy = pass == 0 ? lay : fpely;
            const bool same_centre = searched_lax && fpelx == lax && fpely == lay;
            const bool run = (pass == 0 || (!found_sub && !good_enough && !same_centre)) && !invalid_block(ref0, bx + ccx, by + ccy, bw, bh, 4);
            if (!run) {
                continue;
            }
            if (best_fp != 0) {
                unsigned dirs;
                const unsigned mr = subpel_probes32<NQ>(c, S, ccx, ccy, B, psy, dirs);
                best = subpel_decide(cc, c.effort, mr, dirs, sx, sy, ccx, ccy, best_fp, bw, bh);
            }
            if (pass == 0) {
                searched_lax = true;
                if (sx || sy) {
                    fpelx = lax;
                    fpely = lay;
                    found_sub = true;
                }
            }
        }
    }
    mv.u.mv.x = (int16_t) (fpelx * 4 + sx);
    mv.u.mv.y = (int16_t) (fpely * 4 + sy);
    unsigned ratio = 32;
    if ((mv.u.mv.x | mv.u.mv.y) & 3) {
        ratio = udiv_fast(best << 5, best_fp + !best_fp);
    }
    HME_MARK(S, 5);
    c = fenced(c_in, 0);
    ref0 = c.ref[0];
    // ---- operands of the mode decision, one load round ----
    const int cbx = (i * y_w) >> 1, cby = (j * y_h) >> 1;
    const int cbmx = cbx + sarx(fpelx, 1), cbmy = cby + sarx(fpely, 1);
    const int cbw = bw >> 1, cbh = bh >> 1;
    const bool actc = qi < (cbw >> 1) && qj < (cbh >> 1); // the lane's quad of the (up to) 16 x 16 (NQ = 2: 16 x 8) chroma blocks
    const bool skip_test = (good_enough || (fpelx | fpely | sx | sy) == 0) && c.skip_block_thresh >= 0 && !c.lossless;
    Quad r[NQ], o[NQ], rz[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        const int ox = 16 * (k & 1), oy = 16 * (k >> 1);
        r[k] = ldq(at(ref0, bx + fpelx + ox, by + fpely + oy), ref0.stride, qi, qj, B.act[k]);
        o[k] = ldq(at(c.ogr[0], bx + fpelx + ox, by + fpely + oy), c.ogr[0].stride, qi, qj, B.act[k]);
        rz[k].w = 0;
        if (skip_test) {
            rz[k] = ldq(at(ref0, bx + ox, by + oy), ref0.stride, qi, qj, B.act[k]);
        }
    }
    Quad usq, vsq, umq, vmq, uzq, vzq;
    uzq.w = vzq.w = 0;
    usq = ldq(at(c.srcc[0], cbx, cby), c.srcc[0].stride, qi, qj, actc);
    vsq = ldq(at(c.srcc[1], cbx, cby), c.srcc[1].stride, qi, qj, actc);
    umq = ldq(at(c.refc[0], cbmx, cbmy), c.refc[0].stride, qi, qj, actc);
    vmq = ldq(at(c.refc[1], cbmx, cbmy), c.refc[1].stride, qi, qj, actc);
    if (skip_test) {
        uzq = ldq(at(c.refc[0], cbx, cby), c.refc[0].stride, qi, qj, actc);
        vzq = ldq(at(c.refc[1], cbx, cby), c.refc[1].stride, qi, qj, actc);
    }
    // which of the block's four sub-blocks (bw / 2 x bh / 2 each) a quad belongs to; the sub-block's first quad column / row in
    // lane coordinates (widths are multiples of 16, and so are the heights of 32-high blocks: a sub-block never straddles a
    // quadrant seam; a whole 32 x 32 block's sub-blocks ARE its quadrants)
    constexpr bool kSubIsQuadrant = FULL && NQ == 4;
    int sub_of[NQ], sqi0[NQ], sqj0[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        const int xq = 8 * (k & 1) + qi, yq = 8 * (k >> 1) + qj;
        sub_of[k] = kSubIsQuadrant ? k : ((xq >= (qw >> 1) ? 1 : 0) | (yq >= (qh >> 1) ? 2 : 0));
        sqi0[k] = kSubIsQuadrant ? 0 : max(((sub_of[k] & 1) ? (qw >> 1) : 0) - 8 * (k & 1), 0);
        sqj0[k] = kSubIsQuadrant ? 0 : max(((sub_of[k] & 2) ? (qh >> 1) : 0) - 8 * (k >> 1), 0);
    }
    const int kqc = (qi >= (cbw >> 2) ? 1 : 0) | (qj >= (cbh >> 2) ? 2 : 0); // chroma sub-block of this lane's chroma quad
    // sum over the lane's quads of f(k) that lie in luma sub-block kk
    auto lum_sub = [&](int kk, auto f) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            t += (B.act[k] && sub_of[k] == kk) ? (int) f(k) : 0;
        }
        return t;
    };

    // round 1: block sums
    int v[16];
    {
        int rs, rh, rv;
        blk_grad_partials<NQ>(r, B.act, qi, qj, rs, rh, rv);
        v[0] = 0;
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            v[0] += B.act[k] ? (int) qmetric(B.a[k], o[k], psy) : 0;
        }
        v[1] = rs;
        v[2] = rh;
        v[3] = rv;
        quad_grad_partials(usq, actc, qi, qj, 0, 0, v[4], v[8], v[9]);
        quad_grad_partials(vsq, actc, qi, qj, 0, 0, v[5], v[10], v[11]);
        v[6] = actc ? umq.p1() + umq.p2() + umq.p3() + umq.p4() : 0;
        v[7] = actc ? vmq.p1() + vmq.p2() + vmq.p3() + vmq.p4() : 0;
        v[12] = v[13] = v[14] = v[15] = 0;
    }
    int R = reduceN<16>(v);
    unsigned ogrerr = metric_return((unsigned) bcastN<16>(R, 0), bw, bh);
    int ref_sum = bcastN<16>(R, 1);
    unsigned ref_sh = (unsigned) bcastN<16>(R, 2), ref_sv = (unsigned) bcastN<16>(R, 3);
    int uavg_src = div_nn(bcastN<16>(R, 4), cbw * cbh), vavg_src = div_nn(bcastN<16>(R, 5), cbw * cbh);
    int uavg_ref = div_nn(bcastN<16>(R, 6), cbw * cbh), vavg_ref = div_nn(bcastN<16>(R, 7), cbw * cbh);
    int utex = (int) max((unsigned) bcastN<16>(R, 8), (unsigned) bcastN<16>(R, 9));
    int vtex = (int) max((unsigned) bcastN<16>(R, 10), (unsigned) bcastN<16>(R, 11));
    unsigned avg_ref = (unsigned) div_nn(ref_sum, bw * bh);

    // round 2: reference deviation + -- for the skip test -- the zero-motion sub-block metrics
    int ref_dev;
    unsigned zsub[3] = {0u, 0u, 0u};
    {
        int dev = 0;
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            dev += quad_absdev(r[k], B.act[k], (int) avg_ref);
        }
        if (skip_test) {
            v[0] = dev;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                v[1 + kk] = lum_sub(kk, [&](int k) { return qmetric(B.a[k], rz[k], psy); });
                v[5 + kk] = (actc && kqc == kk) ? (int) qmetric(usq, uzq, psy) : 0;
                v[9 + kk] = (actc && kqc == kk) ? (int) qmetric(vsq, vzq, psy) : 0;
            }
            v[13] = v[14] = v[15] = 0;
            R = reduceN<16>(v);
            ref_dev = bcastN<16>(R, 0) >> 1;
#pragma unroll
            for (int z = 0; z < 3; z++) {
                unsigned m0 = (unsigned) bcastN<16>(R, 1 + 4 * z), m1 = (unsigned) bcastN<16>(R, 2 + 4 * z);
                unsigned m2 = (unsigned) bcastN<16>(R, 3 + 4 * z), m3 = (unsigned) bcastN<16>(R, 4 + 4 * z);
                zsub[z] = max(max(m0, m1), max(m2, m3));
            }
        } else {
            ref_dev = wave_sum(dev) >> 1;
        }
    }
    int tex_ref = (int) (max(ref_sh, ref_sv) - (unsigned) ref_dev);
    unsigned var_ref = (unsigned) (ref_dev + max(tex_ref, 0));

    unsigned ogrmad = div_nn(ogrerr + yarea / 2, yarea);
    ogrmad = ogrmad * ratio >> 5;
    unsigned mad = div_nn(best + yarea / 2, yarea);
    int dv = (int) min(ratio, 32u);
    int ipolvar = (int) ((var_src * (unsigned) dv + var_ref * (unsigned) (32 - dv)) >> 5);
    dv = abs((int) var_src - ipolvar);
    if (var_src > 16 * yarea && var_src < 32 * yarea) {
        mv.flags |= 1u << DSV_MV_BIT_MAINTAIN;
    }
    unsigned chroma_ratio = div_nn((unsigned) ((cbw * cbh) << 4), yarea);
    ChromaPsy cpsy = chroma_analysis((int) avg_src, uavg_src, vavg_src);
    unsigned avg_y_dif = (unsigned) abs((int) avg_src - (int) avg_ref);
    unsigned avg_c_dif = (unsigned) AVG2(abs(uavg_src - uavg_ref), abs(vavg_src - vavg_ref));
    int eprmi, eprmd, eprmr;
    {
        int as128 = (int) avg_src - 128, ar128 = (int) avg_ref - 128;
        int ci = 0, cd = 0, cr = 0;
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            if (B.act[k]) {
                const Quad &a = B.a[k];
                cr |= (((a.p1() - r[k].p1()) + 128) | ((a.p2() - r[k].p2()) + 128) | ((a.p3() - r[k].p3()) + 128) | ((a.p4() - r[k].p4()) + 128)) & ~0xff;
                ci |= ((a.p1() - ar128) | (a.p2() - ar128) | (a.p3() - ar128) | (a.p4() - ar128)) & ~0xff;
                cd |= ((a.p1() - as128) | (a.p2() - as128) | (a.p3() - as128) | (a.p4() - as128)) & ~0xff;
            }
        }
        eprmi = __any(ci != 0) ? 1 : 0;
        eprmd = __any(cd != 0) ? 1 : 0;
        eprmr = __any(cr != 0) ? 1 : 0;
    }
    bool oob;
    {
        int px = i * y_w + sarx(mv.u.mv.x, 2), py = j * y_h + sarx(mv.u.mv.y, 2);
        oob = px < 0 || py < 0 || px >= ((nxb - 1) * y_w) - 1 || py >= ((nyb - 1) * y_h) - 1;
    }
    int neidif;
    {
        int na, nb_;
        neighbordif2_pre(pre, i, j, mv.u.mv.x, mv.u.mv.y, na, nb_);
        neidif = (na + nb_) / 3;
    }
    unsigned skipt = ((unsigned) c.quant * (unsigned) c.quant) >> 19;
    bool skipped = false;
    if (skip_test) {
        unsigned sth = skipt * yarea;
        sth += 4 * var_src;
        sth += yarea * (unsigned) c.skip_block_thresh;
        if (c.quant < (1 << 10)) {
            sth = sth * (unsigned) c.quant >> 10;
        }
        if (avg_y_dif <= 2) {
            sth = max(sth, 3 * (yarea + var_src));
        }
        sth = max(sth, yarea);
        if (good_enough) {
            sth *= 2;
        }
        unsigned cth = chroma_ratio * sth * max(skipt, 1u) >> 5;
        unsigned z0 = zsub[0] * ratio >> 5, z1 = zsub[1] * ratio >> 5, z2 = zsub[2] * ratio >> 5;
        z0 += (unsigned) SQR((int) avg_src - (int) avg_ref) * yarea;
        if (z0 <= sth && z1 <= cth && z2 <= cth) {
            mv.flags |= 1u << DSV_MV_BIT_SKIP;
            mv.u.all = 0;
            mv.err = 0;
            skipped = true;
        }
    }
    int add_err = 0, add_ndiff = 0;
    if (!skipped) {
        if (!oob && !c.lossless) {
            bool y_prereq = avg_y_dif <= 2, c_prereq = !cpsy.greyish && avg_c_dif <= 2;
            if (y_prereq || c_prereq) {
                // round 3: sub-block metrics at the chosen full-pel motion (hme.c:1741)
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    v[kk] = lum_sub(kk, [&](int k) { return qmetric(B.a[k], r[k], psy); });
                    v[4 + kk] = (actc && kqc == kk) ? (int) qmetric(usq, umq, psy) : 0;
                    v[8 + kk] = (actc && kqc == kk) ? (int) qmetric(vsq, vmq, psy) : 0;
                    v[12 + kk] = 0;
                }
                R = reduceN<16>(v);
                unsigned bsub[3];
#pragma unroll
                for (int z = 0; z < 3; z++) {
                    unsigned m0 = (unsigned) bcastN<16>(R, 4 * z), m1 = (unsigned) bcastN<16>(R, 4 * z + 1);
                    unsigned m2 = (unsigned) bcastN<16>(R, 4 * z + 2), m3 = (unsigned) bcastN<16>(R, 4 * z + 3);
                    bsub[z] = max(max(m0, m1), max(m2, m3)) * ratio >> 5;
                }
                unsigned xth = skipt * yarea;
                int carea = 4 * cbw * cbh;
                xth += (unsigned) ipolvar;
                xth = (unsigned) max((int) xth - ((int) yarea * neidif * 2), 0);
                xth = xth * (unsigned) c.quant >> 12;
                xth = min(max(xth, 32u), yarea * 4);
                if (y_prereq && bsub[0] < 4 * xth) {
                    mv.flags |= 1u << DSV_MV_BIT_NOXMITY;
                }
                c_prereq = c_prereq && (utex > carea || vtex > carea);
                xth = chroma_ratio * xth >> 4;
                if (c_prereq && bsub[1] < xth && bsub[2] < xth) {
                    mv.flags |= 1u << DSV_MV_BIT_NOXMITC;
                }
            }
            if ((unsigned) dv < var_src / 4) {
                mv.flags |= 1u << DSV_MV_BIT_SIMCMPLX;
            }
        }
        HME_MARK(S, 6);
        c = fenced(c_in, 0);
        ref0 = c.ref[0];
        // ---- test_subblock_intra_y (hme.c:891), all four sub-blocks evaluated together ----
        {
            int rx = mv.u.mv.x, ry = mv.u.mv.y;
            if (c.ref_mvf != nullptr) {
                uint32_t colo = pre.colo;
                if (!pre.colo_ok) {
                    colo = (uint32_t) __builtin_amdgcn_readfirstlane((int) *(const uint32_t *) &c.ref_mvf[i + j * nxb]);
                }
                rx = (int) (int16_t) (colo & 0xffffu);
                ry = (int) (int16_t) (colo >> 16);
            }
            int sbw = bw / 2, sbh = bh / 2;
            bool run = !(mv.u.all && neidif < 3 && abs(rx - mv.u.mv.x) < 3 && abs(ry - mv.u.mv.y) < 3) && sbw != 0 && sbh != 0;
            if (run) {
                // per quad: the source's sum / first differences inside its sub-block, the reference's sum
                int ss[NQ], sh[NQ], sv2[NQ], rsum[NQ];
#pragma unroll
                for (int k = 0; k < NQ; k++) {
                    quad_grad_partials(B.a[k], B.act[k], qi, qj, sqi0[k], sqj0[k], ss[k], sh[k], sv2[k]);
                    rsum[k] = B.act[k] ? r[k].p1() + r[k].p2() + r[k].p3() + r[k].p4() : 0;
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    v[4 * kk + 0] = lum_sub(kk, [&](int k) { return ss[k]; });
                    v[4 * kk + 1] = lum_sub(kk, [&](int k) { return rsum[k]; });
                    v[4 * kk + 2] = lum_sub(kk, [&](int k) { return sh[k]; });
                    v[4 * kk + 3] = lum_sub(kk, [&](int k) { return sv2[k]; });
                }
                R = reduceN<16>(v);
                int w16[16];
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    w16[t] = 0;
                }
#pragma unroll
                for (int k = 0; k < NQ; k++) {
                    // the quad's own sub-block's averages (per lane: a clipped block's sub-blocks are not the quadrants)
                    const int my_avg_local = div_nn(bcastL<16>(R, 4 * sub_of[k] + 0), sbw * sbh);
                    const int my_avg_sub = div_nn(bcastL<16>(R, 4 * sub_of[k] + 1), sbw * sbh);
                    const int my_dc = (int) ((unsigned) my_avg_local + (unsigned) avg_src * 3 + 2) >> 2;
                    unsigned e_inter = 0, e_sb = 0, e_src = 0;
                    int dev = 0;
                    if (B.act[k]) {
                        quad_err_intra(B.a[k], r[k], my_avg_sub, my_dc, (int) ratio, e_inter, e_sb, e_src);
                        dev = quad_absdev(B.a[k], true, my_avg_local);
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        const bool in = B.act[k] && sub_of[k] == kk;
                        w16[kk] += in ? dev : 0;
                        w16[4 + 3 * kk + 0] += in ? (int) e_inter : 0;
                        w16[4 + 3 * kk + 1] += in ? (int) e_sb : 0;
                        w16[4 + 3 * kk + 2] += in ? (int) e_src : 0;
                    }
                }
                int R2 = reduceN<16>(w16);
                int detail_src = ipolvar, nsub = 0;
                unsigned avg_tot = 0, err_sub = 0, err_src = 0;
                detail_src += sdiv_fast(detail_src, max(neidif, 1));
                for (int k = 0; k < 4; k++) {
                    if (mv.submask & (1 << k)) {
                        continue;
                    }
                    unsigned avg_local = (unsigned) div_nn(bcastN<16>(R, 4 * k + 0), sbw * sbh);
                    unsigned avg_sub = (unsigned) div_nn(bcastN<16>(R, 4 * k + 1), sbw * sbh);
                    unsigned g_sh = (unsigned) bcastN<16>(R, 4 * k + 2), g_sv = (unsigned) bcastN<16>(R, 4 * k + 3);
                    int var = bcastN<16>(R2, k) >> 1;
                    int tex = (int) (max(g_sh, g_sv) - (unsigned) var);
                    unsigned local_detail = (unsigned) (var + max(tex, 0));
                    unsigned dcd = (unsigned) abs((int) avg_local - (int) avg_sub) + 2;
                    if (local_detail > (unsigned) (SQR(dcd) * (unsigned) bw * (unsigned) bh * ratio >> 5)) {
                        continue;
                    }
                    int dc = (int) (avg_local + (unsigned) avg_src * 3 + 2) >> 2;
                    unsigned inter_err = (unsigned) bcastN<16>(R2, 4 + 3 * k + 0) * ratio >> 5;
                    unsigned sub_err = (unsigned) bcastN<16>(R2, 4 + 3 * k + 1), src_err = (unsigned) bcastN<16>(R2, 4 + 3 * k + 2);
                    int lo = AVG2(detail_src, (int) local_detail), hi = detail_src;
                    int lerp = (lo * (32 - c.psyscale) + hi * c.psyscale) >> 5;
                    local_detail = (unsigned) max(lerp, lo);
                    if ((sub_err + local_detail) < inter_err || (src_err + local_detail) < inter_err) {
                        mv.submask |= (uint8_t) (1 << k);
                        err_src += src_err;
                        err_sub += sub_err;
                        avg_tot += sub_err < src_err ? avg_sub : (unsigned) dc;
                        nsub++;
                        detail_src = detail_src * 4 / 5;
                    }
                }
                if (mv.submask) {
                    mv.flags |= 1u << DSV_MV_BIT_INTRA;
                    mv.dc = err_src < err_sub ? (uint16_t) (udiv_fast(avg_tot, (unsigned) nsub) | DSV_SRC_DC_PRED) : 0;
                }
            }
        }
        HME_MARK(S, 7);
        c = fenced(c_in, 0);
        ref0 = c.ref[0];
        // ---- test_subblock_intra_c (hme.c:987) ----
        if (c.effort >= 6) {
            unsigned detail_c = (unsigned) div_nn(ipolvar, bw * bh);
            unsigned thr = (mv.flags & (1u << DSV_MV_BIT_INTRA)) ? detail_c : SQR(detail_c);
            int sbw = cbw / 2, sbh = cbh / 2;
            if (!(sbw == 0 || sbh == 0 || mad <= thr || thr > 64 || (abs((int) mv.u.mv.x) < 4 && abs((int) mv.u.mv.y) < 4))) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bool in = actc && kqc == k;
                    v[4 * k + 0] = in ? usq.p1() + usq.p2() + usq.p3() + usq.p4() : 0;
                    v[4 * k + 1] = in ? vsq.p1() + vsq.p2() + vsq.p3() + vsq.p4() : 0;
                    v[4 * k + 2] = in ? umq.p1() + umq.p2() + umq.p3() + umq.p4() : 0;
                    v[4 * k + 3] = in ? vmq.p1() + vmq.p2() + vmq.p3() + vmq.p4() : 0;
                }
                R = reduceN<16>(v);
                unsigned avg_ramp = avg_src * avg_src >> 8;
                for (int k = 0; k < 4; k++) {
                    if (mv.submask & (1 << k)) {
                        continue;
                    }
                    int a_us = div_nn(bcastN<16>(R, 4 * k + 0), sbw * sbh), a_vs = div_nn(bcastN<16>(R, 4 * k + 1), sbw * sbh);
                    int a_um = div_nn(bcastN<16>(R, 4 * k + 2), sbw * sbh), a_vm = div_nn(bcastN<16>(R, 4 * k + 3), sbw * sbh);
                    unsigned dif = (unsigned) (SQR(a_us - a_um) + SQR(a_vs - a_vm)) * avg_ramp >> 8;
                    if (dif > thr) {
                        mv.submask |= (uint8_t) (1 << k);
                    }
                }
                if (mv.submask) {
                    mv.flags |= 1u << DSV_MV_BIT_INTRA;
                }
            }
        }
        if (!(mv.flags & (1u << DSV_MV_BIT_NOXMITY))) {
            mv.err = (uint16_t) mad;
            add_err = (int) mad;
        }
        add_ndiff = (ogrmad > 11) + (avg_c_dif >= 32);
    }
    int is_intra = 0;
    if (mv.flags & (1u << DSV_MV_BIT_INTRA)) {
        int merged = (mv.dc & DSV_SRC_DC_PRED) ? eprmd : eprmi;
        if (mv.submask != DSV_MASK_ALL_INTRA) {
            merged |= eprmr;
        }
        mv.flags = (mv.flags & ~(1u << DSV_MV_BIT_EPRM)) | (merged ? (1u << DSV_MV_BIT_EPRM) : 0u);
        is_intra = 1;
        mv.u.mv.x = (int16_t) (fpelx * 4);
        mv.u.mv.y = (int16_t) (fpely * 4);
    } else {
        int merged = eprmr;
        if (mv.submask) {
            merged |= eprmi;
        }
        mv.flags = (mv.flags & ~(1u << DSV_MV_BIT_EPRM)) | (merged ? (1u << DSV_MV_BIT_EPRM) : 0u);
    }
    if (mv.flags & ((1u << DSV_MV_BIT_INTRA) | (1u << DSV_MV_BIT_EPRM))) {
        mv.flags &= ~(1u << DSV_MV_BIT_SIMCMPLX);
    }
    HME_MARK(S, 8);
    if (lane == 0) {
        st_mv_final(c, out, mv);
    }
    acc.left_head = (unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) mv.u.all) |
                    ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) mv.flags) << 32);
    acc.have_left = true;
    acc.intra += is_intra;
    acc.ndiff += add_ndiff;
    acc.elig += best > 0 ? 1 : 0;
    acc.err += add_err;
}

// the block of the row pipeline (hme_block_l0_t's in-place form) for a 32 x 32 (NQ = 4) / 32 x 16 (NQ = 2) block
template <bool FULL, int NQ, class Ctx> __device__ __forceinline__ void hme_block_l0_32_t(const Ctx &c_in, int i, int j, int gx, int gy, FastLds &S, RowAcc &acc)
{
    Ctx c = fenced(c_in, 0);
    const int lane = hme_lane();
    const int nxb = c.a.nbh, nyb = c.a.nbv;
    const DPlane src = c.src[0];
    DPlane ref = c.ref[0];
    DSV_MV *mvf = c.mvf[0];
    DSV_MV *out = &mvf[i + j * nxb];
    DSV_MV mv = {};
    HME_COUNT(S, 10, 1);
    constexpr int BW = BlkDim<NQ>::W, BH = BlkDim<NQ>::H;
    const int bx = i * BW, by = j * BH;
    const int bw = FULL ? BW : min(src.w - bx, BW), bh = FULL ? BH : min(src.h - by, BH);
    const SrcBlk<NQ> B = load_src_blk<FULL, NQ>(src, bx, by, bw, bh, 0);
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) v4i_t *cv4i_t;
    typedef const __attribute__((address_space(1))) uint32_t *gu32p_t;
    const v4i_t pre_words = *(cv4i_t) &c.stats[i + j * nxb];
    // lanes 6..14 fetch the co-located vectors of the previous frame, lanes 16..24 the parent level's (hme.c:1443-1528)
    bool pvalid = false, tvalid = false;
    const DSV_MV *parent = c.pyr_levels > 0 ? c.mvf[1] : nullptr;
    uint32_t ov;
    {
        const DSV_MV *op = out;
        if (parent != nullptr) {
            const int pi = i & ~1, pj = j & ~1;
            if (lane >= 16 && lane < 25) {
                int m = lane - 16;
                int x = pi + 2 * tab9(kParX, m), y = pj + 2 * tab9(kParY, m);
                if (x >= 0 && x < nxb && y >= 0 && y < nyb) {
                    op = &parent[x + y * nxb];
                    pvalid = true;
                }
            } else if (lane >= 6 && lane <= 14 && c.ref_mvf != nullptr) {
                int k = lane - 6;
                int rx = i + tab9(kRectX, k), ry = j + tab9(kRectY, k);
                if (rx >= 0 && ry >= 0 && rx < nxb && ry < nyb) {
                    op = &c.ref_mvf[rx + ry * nxb];
                    tvalid = true;
                }
            }
        }
        ov = *(gu32p_t) op;
    }
    bool nb_ok = false;
    const MvHead nbv = load_neighbour_heads(mvf, out, i, j, 1, nxb, c.counters, acc, nb_ok);
    const unsigned var_src = (unsigned) pre_words.y, avg_src = (unsigned) pre_words.z;
    int motion_bias = (int) udiv_fast((unsigned) max(pre_words.x, 0), (unsigned) (2 + (abs(gx) + abs(gy))));
    if (var_src <= (unsigned) (8 * bw * bh * c.quant >> 9)) {
        motion_bias = 0;
    }
    const Psy psy = psy_of_source(var_src, bw, bh, c.quant);
    HME_MARK(S, 1);
    CostCtx cc;
    {
        int v0 = __builtin_amdgcn_readlane((int) nbv.all, 3), v1 = __builtin_amdgcn_readlane((int) nbv.all, 4),
            v2 = __builtin_amdgcn_readlane((int) nbv.all, 5);
        v0 = i > 0 ? v0 : 0;
        v1 = j > 0 ? v1 : 0;
        v2 = i > 0 && j > 0 ? v2 : 0;
        cc.px = pred1((int) (int16_t) (v0 & 0xffff), (int) (int16_t) (v1 & 0xffff), (int) (int16_t) (v2 & 0xffff));
        cc.py = pred1(v0 >> 16, v1 >> 16, v2 >> 16);
    }
    cc.q = c.quant;
    cc.b2sr = b2sr_of(c);
    // the list by canonical position (see hme_block_l0_t)
    const int pvx = (int) (int16_t) (ov & 0xffffu), pvy = (int) (int16_t) (ov >> 16);
    const uint32_t colo = (uint32_t) __builtin_amdgcn_readlane((int) ov, 6);
    const bool colo_ok = parent != nullptr && c.ref_mvf != nullptr;
    bool inl = false;
    int nin = 0, lax = 0, lay = 0;
    const bool open = parent != nullptr && parent_average_cached(acc, i & ~1, pvalid, pvx, pvy, lax, lay, inl, nin);
    bool exist = lane == 0;
    int cxv = 0, cyv = 0;
    if (open) {
        if (lane == 1) {
            exist = true;
            cxv = qp2fp((int16_t) (lax * 4));
            cyv = qp2fp((int16_t) (lay * 4));
        } else if (lane == 2) {
            exist = true;
            cxv = qp2fp((int16_t) cc.px);
            cyv = qp2fp((int16_t) cc.py);
        } else if (nb_ok) {
            exist = true;
            cxv = qp2fp(nbv.x);
            cyv = qp2fp(nbv.y);
        } else if (tvalid) {
            exist = true;
            cxv = qp2fp(pvx);
            cyv = qp2fp(pvy);
        } else if (lane == 15) {
            exist = true;
            cxv = qp2fp((int16_t) (gx * 4));
            cyv = qp2fp((int16_t) (gy * 4));
        } else if (lane >= 16 && lane < 25 && nin && inl) {
            exist = true;
            cxv = qp2fp((int16_t) (pvx * 4));
            cyv = qp2fp((int16_t) (pvy * 4));
        }
    }
    const int key = ((int) (int16_t) cxv & 0xffff) | (int) ((unsigned) (int) (int16_t) cyv << 16);
    const bool keep = exist && !dedup_lanes(exist, key);
    const int mx = (int) (int16_t) (key & 0xffff), my = key >> 16;
    HME_COUNT(S, 14, __popcll(__ballot(keep)));
    const unsigned raw0 = metric_return(score_lanes<NQ>(__ballot(keep), key, ref, B, 0, psy), bw, bh);
    HME_MARK(S, 2);
    int dx, dy;
    unsigned best, score_zero;
    {
        const bool valid = keep && !invalid_block(ref, bx + mx, by + my, bw, bh, 0);
        unsigned sc = raw0 + (unsigned) mv_cost(cc, mx * 4, my * 4, 0);
        if (mx == lax && my == lay) {
            sc = (unsigned) max((int) sc - motion_bias, 0);
        }
        if (!valid) {
            sc = 0xffffffffu;
        }
        unsigned mn = wave_min_u(sc);
        unsigned long long hit = __ballot(valid && sc == mn);
        int best_k = (mn != 0xffffffffu && hit) ? (int) __ffsll((long long) hit) - 1 : 0;
        best = mn;
        bool z_valid = __builtin_amdgcn_readlane((int) valid, 0) != 0;
        unsigned z_raw = (unsigned) __builtin_amdgcn_readlane((int) raw0, 0);
        score_zero = z_valid ? z_raw : 0xffffffffu;
        dx = __builtin_amdgcn_readlane(mx, best_k);
        dy = __builtin_amdgcn_readlane(my, best_k);
    }
    unsigned qthresh = (unsigned) (c.quant * bw * bh >> 11);
    bool good_enough = false;
    {
        const unsigned zoscore = (unsigned) pre_words.w;
        if (abs(dx) <= 1 && abs(dy) <= 1) {
            qthresh *= 2;
        }
        if (zoscore < qthresh) {
            best = score_zero;
            dx = dy = 0;
            good_enough = true;
        }
    }
    HME_MARK(S, 3);
    c = fenced(c_in, 0);
    ref = c.ref[0];
    if (!good_enough) {
        refine_fpel<true, NQ>(ref, B, 0, psy, cc, qthresh, dx, dy, best, good_enough, S);
    }
    HME_MARK(S, 4);
    mv.u.mv.x = (int16_t) dx;
    mv.u.mv.y = (int16_t) dy;
    NbPre pre;
    pre.l_all = (uint32_t) __builtin_amdgcn_readlane((int) nbv.all, 3);
    pre.l_flags = (uint32_t) __builtin_amdgcn_readlane((int) nbv.flags, 3);
    pre.t_all = (uint32_t) __builtin_amdgcn_readlane((int) nbv.all, 4);
    pre.t_flags = (uint32_t) __builtin_amdgcn_readlane((int) nbv.flags, 4);
    pre.colo = colo;
    pre.colo_ok = colo_ok;
    hme_l0_tail32<FULL, NQ>(c, i, j, S, acc, out, mv, cc, B, lax, lay, motion_bias, good_enough, best, var_src, avg_src, psy, pre);
}

template <int NQ, class Ctx> __device__ __forceinline__ void hme_block_l0_32(const Ctx &c, int i, int j, int gx, int gy, FastLds &S, RowAcc &acc)
{
    constexpr int BW = BlkDim<NQ>::W, BH = BlkDim<NQ>::H;
    const DPlane &src = c.src[0];
    if (src.w - i * BW >= BW && src.h - j * BH >= BH) {
        hme_block_l0_32_t<true, NQ>(c, i, j, gx, gy, S, acc);
    } else {
        hme_block_l0_32_t<false, NQ>(c, i, j, gx, gy, S, acc);
    }
}
