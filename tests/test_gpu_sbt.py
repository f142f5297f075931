"""HIP subband transform + picture helpers vs the real reference (oracle/_ref) through the C ABI seam."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from test_oracle_sbt import CASES, rand_frame, ref_fwd, ref_inv

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def hip_fwd(hip, frame, plane, isP, blockdata, params, cdims):
    cw, ch = cdims[plane]
    coefs = np.zeros(cw * ch, dtype=np.int32)
    cs = A.COEFS(A.np_ptr(coefs, C.c_int32), cw, ch)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
    fm.cur_plane, fm.isP = plane, isP
    hip.dsv_fwd_sbt(frame.plane_ptr(plane), C.byref(cs), C.byref(fm))
    return coefs


def hip_inv(hip, coefs, plane, isP, q, blockdata, params, cdims, subsamp, w, h):
    cw, ch = cdims[plane]
    out = A.HostFrame(subsamp, w, h, border=True)
    cc = coefs.copy()
    cs = A.COEFS(A.np_ptr(cc, C.c_int32), cw, ch)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
    fm.cur_plane, fm.isP = plane, isP
    hip.dsv_inv_sbt(out.plane_ptr(plane), C.byref(cs), q, C.byref(fm))
    return out.plane(plane).copy()


@pytest.mark.parametrize("w,h,subsamp", CASES + [(1920, 1080, A.SUBSAMP_420)])
@pytest.mark.parametrize("isP,lossless", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_sbt_matches_reference(w, h, subsamp, isP, lossless):
    ref, hip = A.load_ref(), A.load_hip()
    meta = A.mk_meta(w, h, subsamp)
    params = A.mk_params(meta, w, h, isP, lossless)
    nb = params.nblocks_h * params.nblocks_v
    rng = np.random.RandomState(w * 7 + h + isP)
    blockdata = rng.randint(0, 128, size=nb).astype(np.uint8)
    frame = rand_frame(subsamp, w, h, seed=w + h)
    cdims = A.coef_dims(subsamp, w, h)
    for plane in range(3):
        want = ref_fwd(ref, frame, plane, isP, lossless, blockdata, params, cdims)
        got = hip_fwd(hip, frame, plane, isP, blockdata, params, cdims)
        assert np.array_equal(want, got), "fwd plane %d" % plane
        q = 1 if lossless else 200
        coefs = want.copy()
        if not lossless:
            coefs = (coefs // 24) * 24
        want_px = ref_inv(ref, coefs, plane, isP, lossless, q, blockdata, params, cdims, subsamp, w, h)
        got_px = hip_inv(hip, coefs, plane, isP, q, blockdata, params, cdims, subsamp, w, h)
        assert np.array_equal(want_px, got_px), "inv plane %d" % plane
        if lossless:
            pw, ph = frame.dims[plane]
            assert np.array_equal(got_px, frame.plane(plane)[:ph, :pw])


@pytest.mark.parametrize("w,h,subsamp", [(352, 288, A.SUBSAMP_420), (354, 290, A.SUBSAMP_420),
                                          (1920, 1080, A.SUBSAMP_420), (642, 362, A.SUBSAMP_444)])
def test_extend_and_ds2x_match_reference(w, h, subsamp):
    ref, hip = A.load_ref(), A.load_hip()
    a = rand_frame(subsamp, w, h, seed=5)
    b = A.HostFrame(subsamp, w, h, border=True)
    b.buf[:] = a.buf
    ref.dsv_extend_frame(a.ptr())
    hip.dsv_extend_frame(b.ptr())
    for c in range(3):
        assert np.array_equal(a.full[c], b.full[c]), "extend plane %d" % c
    # 2x decimation pyramid, three levels deep (dsv_encoder.c:494-516)
    pa, pb = a, b
    for lvl in range(1, 4):
        dw, dh = (w + (1 << lvl) - 1) >> lvl, (h + (1 << lvl) - 1) >> lvl
        na, nb_ = A.HostFrame(subsamp, dw, dh, border=True), A.HostFrame(subsamp, dw, dh, border=True)
        ref.dsv_ds2x_frame_luma(na.ptr(), pa.ptr())
        hip.dsv_ds2x_frame_luma(nb_.ptr(), pb.ptr())
        assert np.array_equal(na.plane(0), nb_.plane(0)), "ds2x level %d" % lvl
        ref.dsv_extend_frame_luma(na.ptr())
        hip.dsv_extend_frame_luma(nb_.ptr())
        assert np.array_equal(na.full[0], nb_.full[0]), "extend luma level %d" % lvl
        pa, pb = na, nb_
