"""oracle/orc_hzcc.c (phase-decomposed quantiser + entropy restatement) vs the reference's dsv_encode_plane."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from test_oracle_sbt import rand_frame, ref_fwd

pytestmark = pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built")

CASES = [
    (352, 288, A.SUBSAMP_420),
    (354, 290, A.SUBSAMP_420),
    (1000, 562, A.SUBSAMP_420),   # odd subband sizes: scanned regions of adjacent levels overlap
    (960, 540, A.SUBSAMP_444),    # the 1080p chroma geometry as a luma plane (135 is odd)
]


def rand_mvs(rng, nb):
    mv = np.zeros(nb, dtype=A.MV_DTYPE)
    mv["x"] = rng.randint(-80, 81, size=nb)
    mv["y"] = rng.randint(-80, 81, size=nb)
    mv["flags"] = rng.randint(0, 256, size=nb)
    return mv


def ref_encode_plane(ref, coefs, cw, ch, q, plane, isP, params, blockdata, mvs):
    cc = coefs.copy()
    out = np.zeros(cw * ch * 8 + 1024, dtype=np.uint8)
    bs = A.BS(A.np_ptr(out, C.c_uint8), 0)
    cs = A.COEFS(A.np_ptr(cc, C.c_int32), cw, ch)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
    fm.mvs = C.cast(mvs.ctypes.data, C.POINTER(A.MV))
    fm.cur_plane, fm.isP = plane, isP
    ref.dsv_encode_plane(C.byref(bs), C.byref(cs), q, C.byref(fm))
    assert bs.pos % 8 == 0
    return out[:bs.pos // 8].copy(), cc


def orc_encode_plane(orc, coefs, cw, ch, q, plane, isP, params, subsamp, blockdata, mvs):
    cc = coefs.copy()
    out = np.zeros(cw * ch * 8 + 1024, dtype=np.uint8)
    n = orc.orc_encode_plane(A.np_ptr(out, C.c_uint8), 0, A.np_ptr(cc, C.c_int32), cw, ch, q, plane, isP,
                             params.lossless, params.do_psy, (subsamp >> 2) & 3, subsamp & 3,
                             params.blk_w, params.blk_h, params.nblocks_h, params.nblocks_v,
                             A.np_ptr(blockdata, C.c_uint8), C.c_void_p(mvs.ctypes.data))
    return out[:n].copy(), cc


@pytest.mark.parametrize("w,h,subsamp", CASES)
@pytest.mark.parametrize("isP,lossless,q,do_psy", [(0, 0, 180, 0xff), (1, 0, 172, 0xff), (0, 0, 40, 0),
                                                    (1, 0, 900, 0x1), (0, 1, 1, 0xff), (1, 1, 1, 0xff)])
def test_encode_plane_matches_reference(w, h, subsamp, isP, lossless, q, do_psy):
    ref, orc = A.load_ref(), A.load_oracle()
    meta = A.mk_meta(w, h, subsamp)
    params = A.mk_params(meta, w, h, isP, lossless, do_psy=do_psy)
    nb = params.nblocks_h * params.nblocks_v
    rng = np.random.RandomState(w + 3 * h + isP + q)
    blockdata = rng.randint(0, 128, size=nb).astype(np.uint8)
    mvs = rand_mvs(rng, nb)
    frame = rand_frame(subsamp, w, h, seed=w + h + 1)
    cdims = A.coef_dims(subsamp, w, h)
    for plane in range(3):
        cw, ch = cdims[plane]
        coefs = ref_fwd(ref, frame, plane, isP, lossless, blockdata, params, cdims)
        want_bytes, want_coefs = ref_encode_plane(ref, coefs, cw, ch, q, plane, isP, params, blockdata, mvs)
        got_bytes, got_coefs = orc_encode_plane(orc, coefs, cw, ch, q, plane, isP, params, subsamp, blockdata, mvs)
        assert np.array_equal(want_coefs, got_coefs), "dequantised coefficients, plane %d" % plane
        assert np.array_equal(want_bytes, got_bytes), "plane bitstream, plane %d" % plane
