"""oracle/orc_sbt.c (direct-form restatement) vs the real reference's dsv_fwd_sbt / dsv_inv_sbt."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A

pytestmark = pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built")

CASES = [
    # (w, h, subsamp)
    (352, 288, A.SUBSAMP_420),
    (354, 290, A.SUBSAMP_420),   # odd chroma dims -> coef planes rounded up to even
    (1000, 562, A.SUBSAMP_420),
    (640, 360, A.SUBSAMP_444),
    (96, 80, A.SUBSAMP_422),
]


def rand_frame(subsamp, w, h, seed):
    rng = np.random.RandomState(seed)
    f = A.HostFrame(subsamp, w, h, border=True)
    f.buf[:] = rng.randint(0, 256, size=f.buf.shape, dtype=np.uint8)
    # smooth-ish content so the transform sees structure, plus noise
    for i in range(3):
        pw, ph = f.dims[i]
        yy, xx = np.mgrid[0:ph, 0:pw]
        img = 128 + 60 * np.sin(xx / 17.0 + seed) + 50 * np.cos(yy / 11.0) + rng.randint(-20, 21, size=(ph, pw))
        f.plane(i)[:, :] = np.clip(img, 0, 255).astype(np.uint8)
    return f


def ref_fwd(ref, frame, plane, isP, lossless, blockdata, params, cdims):
    cw, ch = cdims[plane]
    coefs = np.zeros(cw * ch, dtype=np.int32)
    cs = A.COEFS(A.np_ptr(coefs, C.c_int32), cw, ch)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
    fm.cur_plane, fm.isP = plane, isP
    ref.dsv_fwd_sbt(frame.plane_ptr(plane), C.byref(cs), C.byref(fm))
    return coefs


def ref_inv(ref, coefs, plane, isP, lossless, q, blockdata, params, cdims, subsamp, w, h):
    cw, ch = cdims[plane]
    out = A.HostFrame(subsamp, w, h, border=True)
    cc = coefs.copy()
    cs = A.COEFS(A.np_ptr(cc, C.c_int32), cw, ch)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
    fm.cur_plane, fm.isP = plane, isP
    ref.dsv_inv_sbt(out.plane_ptr(plane), C.byref(cs), q, C.byref(fm))
    return out.plane(plane).copy()


@pytest.mark.parametrize("w,h,subsamp", CASES)
@pytest.mark.parametrize("isP,lossless", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_fwd_inv_match_reference(w, h, subsamp, isP, lossless):
    ref = A.load_ref()
    orc = A.load_oracle()
    meta = A.mk_meta(w, h, subsamp)
    params = A.mk_params(meta, w, h, isP, lossless)
    nb = params.nblocks_h * params.nblocks_v
    rng = np.random.RandomState(w * 7 + h + isP)
    blockdata = rng.randint(0, 128, size=nb).astype(np.uint8)
    frame = rand_frame(subsamp, w, h, seed=w + h)
    cdims = A.coef_dims(subsamp, w, h)
    for plane in range(3):
        cw, ch = cdims[plane]
        pw, ph = frame.dims[plane]
        want = ref_fwd(ref, frame, plane, isP, lossless, blockdata, params, cdims)
        got = np.zeros(cw * ch, dtype=np.int32)
        orc.orc_fwd_sbt(frame.c.planes[plane].data, frame.strides[plane], pw, ph,
                        A.np_ptr(got, C.c_int32), cw, ch, plane, isP, lossless,
                        A.np_ptr(blockdata, C.c_uint8), params.nblocks_h, params.nblocks_v)
        assert np.array_equal(want, got), "fwd plane %d" % plane
        # inverse on a perturbed (pseudo-quantised) coefficient set
        q = 1 if lossless else 200
        coefs = want.copy()
        if not lossless:
            step = 24
            coefs = (coefs // step) * step
        want_px = ref_inv(ref, coefs, plane, isP, lossless, q, blockdata, params, cdims, subsamp, w, h)
        out = A.HostFrame(subsamp, w, h, border=True)
        orc.orc_inv_sbt(out.c.planes[plane].data, out.strides[plane], pw, ph,
                        A.np_ptr(coefs, C.c_int32), cw, ch, q, plane, isP, lossless,
                        A.np_ptr(blockdata, C.c_uint8), params.nblocks_h, params.nblocks_v)
        assert np.array_equal(want_px, out.plane(plane)), "inv plane %d" % plane
        if lossless:
            assert np.array_equal(out.plane(plane), frame.plane(plane)[:ph, :pw])
