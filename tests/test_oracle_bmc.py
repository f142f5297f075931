"""oracle/orc_bmc.c (per-block MC + wavefront-ordered in-loop filters) vs the reference's bmc.c entry points."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
import orcabi as O
from test_oracle_sbt import rand_frame

pytestmark = pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built")

CASES = [(352, 288, A.SUBSAMP_420), (354, 290, A.SUBSAMP_420), (640, 360, A.SUBSAMP_444), (1000, 562, A.SUBSAMP_422)]


def rand_motion(rng, params, big=False):
    nb = params.nblocks_h * params.nblocks_v
    mv = np.zeros(nb, dtype=A.MV_DTYPE)
    amp = 130 if big else 24
    mv["x"] = rng.randint(-amp, amp + 1, size=nb)
    mv["y"] = rng.randint(-amp, amp + 1, size=nb)
    kind = rng.randint(0, 10, size=nb)
    flags = np.zeros(nb, dtype=np.uint32)
    flags[kind == 0] |= 1 << 3                                   # skip
    flags[kind == 1] |= 1 << 0                                   # intra
    flags[kind == 2] |= (1 << 0) | (1 << 1)                      # intra + eprm
    flags[kind == 3] |= 1 << 1                                   # eprm
    flags[kind == 4] |= 1 << 5                                   # noxmit luma
    flags[kind == 5] |= 1 << 6                                   # noxmit chroma
    mv["flags"] = flags
    intra = (flags & 1) != 0
    mv["submask"][intra] = rng.choice([15, 15, 1, 6, 9, 8, 7], size=int(intra.sum()))
    mv["dc"][intra] = rng.choice([0, 0, 0x100 | 77, 0x100 | 200], size=int(intra.sum()))
    mv["x"][intra] &= ~3
    mv["y"][intra] &= ~3
    skip = (flags & 8) != 0
    mv["x"][skip] = 0
    mv["y"][skip] = 0
    # a smooth region so that neighbour differences are small in places
    mv["x"][: nb // 3] = 5
    mv["y"][: nb // 3] = -3
    return mv


def clone(hf):
    g = A.HostFrame(hf.subsamp, hf.w, hf.h, border=True)
    g.buf[:] = hf.buf
    return g


@pytest.mark.parametrize("w,h,subsamp", CASES)
@pytest.mark.parametrize("lossless,tmc,do_filter,q", [(0, 0, 1, 700), (0, 1, 1, 172), (0, 1, 0, 2500), (1, 0, 1, 1)])
def test_motion_compensation_and_filters(w, h, subsamp, lossless, tmc, do_filter, q):
    ref, orc = A.load_ref(), A.load_oracle()
    meta = A.mk_meta(w, h, subsamp, inter_sharpen=1)
    params = A.mk_params(meta, w, h, 1, lossless, temporal_mc=tmc)
    rng = np.random.RandomState(w + h + q + tmc)
    mvs = rand_motion(rng, params, big=(q == 700))
    mvp = C.cast(mvs.ctypes.data, C.POINTER(A.MV))
    refframe = rand_frame(subsamp, w, h, seed=3)
    ref.dsv_extend_frame(refframe.ptr())
    src = rand_frame(subsamp, w, h, seed=4)
    ref.dsv_extend_frame(src.ptr())
    op = O.orc_params(params, meta)

    # encoder side: prediction + residual
    pred_r, resd_r = A.HostFrame(subsamp, w, h), clone(src)
    pred_o, resd_o = A.HostFrame(subsamp, w, h), clone(src)
    ref.dsv_sub_pred(mvp, C.byref(params), pred_r.ptr(), resd_r.ptr(), refframe.ptr())
    orc.orc_sub_pred(C.c_void_p(mvs.ctypes.data), C.byref(op), C.byref(O.oframe(pred_o)), C.byref(O.oframe(resd_o)),
                     C.byref(O.oframe(refframe)))
    for c in range(3):
        assert np.array_equal(pred_r.full[c], pred_o.full[c]), "prediction plane %d" % c
        assert np.array_equal(resd_r.full[c], resd_o.full[c]), "residual plane %d" % c

    # encoder-side reconstruction + in-loop filters
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.isP = 1
    ref.dsv_add_res(mvp, C.byref(fm), q, resd_r.ptr(), pred_r.ptr(), do_filter)
    orc.orc_add_res(C.c_void_p(mvs.ctypes.data), C.byref(op), q, C.byref(O.oframe(resd_o)), C.byref(O.oframe(pred_o)), do_filter)
    for c in range(3):
        assert np.array_equal(resd_r.full[c], resd_o.full[c]), "add_res plane %d" % c

    # decoder side
    resd = rand_frame(subsamp, w, h, seed=9)
    out_r, out_o = A.HostFrame(subsamp, w, h), A.HostFrame(subsamp, w, h)
    ref.dsv_add_pred(mvp, C.byref(fm), q, resd.ptr(), out_r.ptr(), refframe.ptr(), do_filter)
    orc.orc_add_pred(C.c_void_p(mvs.ctypes.data), C.byref(op), q, C.byref(O.oframe(resd)), C.byref(O.oframe(out_o)),
                     C.byref(O.oframe(refframe)), do_filter)
    for c in range(3):
        assert np.array_equal(out_r.full[c], out_o.full[c]), "add_pred plane %d" % c


@pytest.mark.parametrize("w,h,subsamp", CASES)
@pytest.mark.parametrize("q", [60, 400, 3000])
def test_intra_filter(w, h, subsamp, q):
    ref, orc = A.load_ref(), A.load_oracle()
    meta = A.mk_meta(w, h, subsamp)
    params = A.mk_params(meta, w, h, 0, 0)
    nb = params.nblocks_h * params.nblocks_v
    rng = np.random.RandomState(q + w)
    bd = rng.choice([0, 1, 2, 3, 8, 9, 10], size=nb).astype(np.uint8)
    a = rand_frame(subsamp, w, h, seed=21)
    # smoother content so that the texture window (8 < max(sh,sv) < 256) is hit often
    a.plane(0)[:, :] = (a.plane(0).astype(np.int32) // 8 + 100).astype(np.uint8)
    b = clone(a)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(bd, C.c_uint8)
    ref.dsv_intra_filter(q, C.byref(params), C.byref(fm), 0, a.plane_ptr(0), 1)
    op = O.orc_params(params, meta)
    orc.orc_intra_filter(b.c.planes[0].data, b.strides[0], w, h, C.byref(op), A.np_ptr(bd, C.c_uint8), q, 1)
    assert np.array_equal(a.full[0], b.full[0])
    assert not np.array_equal(a.plane(0), rand_frame(subsamp, w, h, seed=21).plane(0) // 8 + 100) or q == 60
