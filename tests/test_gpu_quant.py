"""HIP quantiser + host entropy coder (dsv_encode_plane / dsv_decode_plane seam) vs the real reference."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from test_oracle_hzcc import CASES, rand_mvs, ref_encode_plane
from test_oracle_sbt import rand_frame, ref_fwd

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def decode_plane(lib, data, cw, ch, q, plane, isP, params, blockdata):
    buf = np.concatenate([data, np.zeros(64, dtype=np.uint8)])
    coefs = np.zeros(cw * ch, dtype=np.int32)
    bs = A.BS(A.np_ptr(buf, C.c_uint8), 0)
    cs = A.COEFS(A.np_ptr(coefs, C.c_int32), cw, ch)
    fm = A.FMETA()
    fm.params = C.pointer(params)
    fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
    fm.cur_plane, fm.isP = plane, isP
    ok = lib.dsv_decode_plane(C.byref(bs), C.byref(cs), q, C.byref(fm))
    return ok, coefs, bs.pos


@pytest.mark.parametrize("w,h,subsamp", CASES + [(1920, 1080, A.SUBSAMP_420), (1920, 800, A.SUBSAMP_420)])  # (1920 x 800: 32 x 16 blocks)
@pytest.mark.parametrize("isP,lossless,q,do_psy", [(0, 0, 180, 0xff), (1, 0, 172, 0xff), (0, 0, 40, 0),
                                                    (1, 0, 900, 0x1), (0, 1, 1, 0xff), (1, 1, 1, 0xff)])
def test_encode_decode_plane_match_reference(w, h, subsamp, isP, lossless, q, do_psy):
    ref, hip = A.load_ref(), A.load_hip()
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
        got_bytes, got_coefs = ref_encode_plane(hip, coefs, cw, ch, q, plane, isP, params, blockdata, mvs)
        assert np.array_equal(want_coefs, got_coefs), "dequantised coefficients, plane %d" % plane
        assert np.array_equal(want_bytes, got_bytes), "plane bitstream, plane %d" % plane
        ok_r, dec_r, pos_r = decode_plane(ref, want_bytes, cw, ch, q, plane, isP, params, blockdata)
        ok_h, dec_h, pos_h = decode_plane(hip, want_bytes, cw, ch, q, plane, isP, params, blockdata)
        assert ok_r == 1 and ok_h == 1 and pos_r == pos_h
        assert np.array_equal(dec_r, dec_h), "decoded coefficients, plane %d" % plane
