"""HIP motion compensation / reconstruction / in-loop filters (wavefront kernels) vs the real reference."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from test_oracle_bmc import CASES, clone, rand_motion
from test_oracle_sbt import rand_frame

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


# (2560 x 1440: 32 x 32 blocks; 1920 x 800, 16384 x 64: 32 x 16 -- in 4:2:0 those are predicted a 16 x 16 piece per wavefront, intra blocks whole)
@pytest.mark.parametrize("w,h,subsamp", CASES + [(1920, 1080, A.SUBSAMP_420), (2560, 1440, A.SUBSAMP_420), (16384, 64, A.SUBSAMP_420), (1920, 800, A.SUBSAMP_420),
                                             (2560, 1440, A.SUBSAMP_444)])
@pytest.mark.parametrize("lossless,tmc,do_filter,q", [(0, 0, 1, 700), (0, 1, 1, 172), (0, 1, 0, 2500), (1, 0, 1, 1)])
def test_motion_compensation_and_filters(w, h, subsamp, lossless, tmc, do_filter, q):
    ref, hip = A.load_ref(), A.load_hip()
    meta = A.mk_meta(w, h, subsamp, inter_sharpen=1)
    params = A.mk_params(meta, w, h, 1, lossless, temporal_mc=tmc)
    rng = np.random.RandomState(w + h + q + tmc)
    mvs = rand_motion(rng, params, big=(q == 700))
    mvp = C.cast(mvs.ctypes.data, C.POINTER(A.MV))
    refframe = rand_frame(subsamp, w, h, seed=3)
    ref.dsv_extend_frame(refframe.ptr())
    src = rand_frame(subsamp, w, h, seed=4)
    ref.dsv_extend_frame(src.ptr())

    pred_r, resd_r = A.HostFrame(subsamp, w, h), clone(src)
    pred_h, resd_h = A.HostFrame(subsamp, w, h), clone(src)
    ref.dsv_sub_pred(mvp, C.byref(params), pred_r.ptr(), resd_r.ptr(), refframe.ptr())
    hip.dsv_sub_pred(mvp, C.byref(params), pred_h.ptr(), resd_h.ptr(), refframe.ptr())
    for c in range(3):
        assert np.array_equal(pred_r.full[c], pred_h.full[c]), "prediction plane %d" % c
        assert np.array_equal(resd_r.full[c], resd_h.full[c]), "residual plane %d" % c

    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.isP = 1
    ref.dsv_add_res(mvp, C.byref(fm), q, resd_r.ptr(), pred_r.ptr(), do_filter)
    hip.dsv_add_res(mvp, C.byref(fm), q, resd_h.ptr(), pred_h.ptr(), do_filter)
    for c in range(3):
        assert np.array_equal(resd_r.full[c], resd_h.full[c]), "add_res plane %d" % c

    resd = rand_frame(subsamp, w, h, seed=9)
    out_r, out_h = A.HostFrame(subsamp, w, h), A.HostFrame(subsamp, w, h)
    ref.dsv_add_pred(mvp, C.byref(fm), q, resd.ptr(), out_r.ptr(), refframe.ptr(), do_filter)
    hip.dsv_add_pred(mvp, C.byref(fm), q, resd.ptr(), out_h.ptr(), refframe.ptr(), do_filter)
    for c in range(3):
        assert np.array_equal(out_r.full[c], out_h.full[c]), "add_pred plane %d" % c


# (16384 wide: cell index x block count reaches 2^21, the exact-divide branch of the filters' cell -> block mapping)
@pytest.mark.parametrize("w,h,subsamp", CASES + [(1920, 1080, A.SUBSAMP_420), (2560, 1440, A.SUBSAMP_420), (16384, 64, A.SUBSAMP_420)])
@pytest.mark.parametrize("q", [60, 400, 3000])
def test_intra_filter(w, h, subsamp, q):
    ref, hip = A.load_ref(), A.load_hip()
    meta = A.mk_meta(w, h, subsamp)
    params = A.mk_params(meta, w, h, 0, 0)
    nb = params.nblocks_h * params.nblocks_v
    rng = np.random.RandomState(q + w)
    bd = rng.choice([0, 1, 2, 3, 8, 9, 10], size=nb).astype(np.uint8)
    a = rand_frame(subsamp, w, h, seed=21)
    a.plane(0)[:, :] = (a.plane(0).astype(np.int32) // 8 + 100).astype(np.uint8)
    b = clone(a)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(bd, C.c_uint8)
    ref.dsv_intra_filter(q, C.byref(params), C.byref(fm), 0, a.plane_ptr(0), 1)
    hip.dsv_intra_filter(q, C.byref(params), C.byref(fm), 0, b.plane_ptr(0), 1)
    assert np.array_equal(a.plane(0), b.plane(0))
