"""oracle/orc_fmt.c (UYVY de-interleave, -out420p chroma conversions) vs the reference's own functions:
dsv_yuv_read's UYVY branch (dsv.c:177-205, through a temporary file) and util.c:79-153 compiled into
oracle/_ref/libdsv2refutil.so."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A

REFUTIL_SO = os.path.join(A.ROOT, "oracle", "_ref", "libdsv2refutil.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REFUTIL_SO), reason="oracle/_ref not built")

MODES = {A.SUBSAMP_444: 1, A.SUBSAMP_422: 2, 0x8: 3, 0xA: 4}
CONV = {1: None, 2: "conv422to420", 3: "conv411to420", 4: "conv410to420"}


def chroma_dims(subsamp, w, h):
    hs, vs = (subsamp >> 2) & 3, subsamp & 3
    return (w + (1 << hs) - 1) >> hs, (h + (1 << vs) - 1) >> vs


def mk_plane(arr):
    p = A.PLANE()
    p.data = arr.ctypes.data_as(C.POINTER(C.c_uint8))
    p.stride = arr.shape[1]
    p.h = arr.shape[0]
    p.w = arr.shape[1]
    return p


def ref_to420(util, src, subsamp, w, h):
    """the reference's chain for one chroma plane (dsv_main.c:1030-1048); src: 2-D array of the decoded chroma plane"""
    dw, dh = chroma_dims(A.SUBSAMP_420, w, h)
    pad = 8  # conv411 / conv410 may write a column / row past the 4:2:0 plane (into the frame's stride slack)
    dst = np.zeros((dh + pad, dw + pad), dtype=np.uint8)
    sp, dp = mk_plane(src), mk_plane(dst)
    sp.w, sp.h = src.shape[1], src.shape[0]
    dp.w, dp.h = dw, dh
    mode = MODES[subsamp]
    if mode == 1:
        mw, mh = chroma_dims(A.SUBSAMP_422, w, h)
        mid = np.zeros((mh + pad, mw + pad), dtype=np.uint8)
        mp = mk_plane(mid)
        mp.w, mp.h = mw, mh
        util.conv444to422(C.byref(sp), C.byref(mp))
        util.conv422to420(C.byref(mp), C.byref(dp))
    else:
        getattr(util, CONV[mode])(C.byref(sp), C.byref(dp))
    return dst[:dh, :dw].copy()


def orc_to420(orc, src, subsamp, w, h):
    dw, dh = chroma_dims(A.SUBSAMP_420, w, h)
    dst = np.zeros((dh, dw), dtype=np.uint8)
    orc.orc_to420(src.ctypes.data_as(C.POINTER(C.c_uint8)), src.shape[1], src.shape[1], src.shape[0],
                  dst.ctypes.data_as(C.POINTER(C.c_uint8)), dw, dw, dh, MODES[subsamp])
    return dst


@pytest.mark.parametrize("subsamp", [A.SUBSAMP_444, A.SUBSAMP_422, 0x8, 0xA])
@pytest.mark.parametrize("w,h", [(352, 288), (354, 290), (358, 294), (64, 36), (1920, 1080)])
def test_to420_matches_reference(subsamp, w, h):
    util, orc = C.CDLL(REFUTIL_SO), A.load_oracle()
    cw, ch = chroma_dims(subsamp, w, h)
    rng = np.random.RandomState(w * 7 + h + subsamp)
    src = rng.randint(0, 256, size=(ch, cw)).astype(np.uint8)
    assert np.array_equal(ref_to420(util, src, subsamp, w, h), orc_to420(orc, src, subsamp, w, h))


@pytest.mark.parametrize("w,h", [(352, 288), (64, 36), (1920, 1080)])
def test_uyvy_deinterleave_matches_reference(w, h, tmp_path):
    ref, orc = A.load_ref(), A.load_oracle()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    rng = np.random.RandomState(w + h)
    raw = rng.randint(0, 256, size=2 * w * h * 2).astype(np.uint8)  # two pictures
    path = tmp_path / "in.uyvy"
    raw.tofile(path)
    ref.dsv_yuv_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    ref.dsv_yuv_read.restype = C.c_int
    f = libc.fopen(str(path).encode(), b"rb")
    for fno in (0, 1):
        want = np.zeros(2 * w * h, dtype=np.uint8)
        assert ref.dsv_yuv_read(f, fno, want.ctypes.data, w, h, 0x14) == 0
        got = np.zeros(2 * w * h, dtype=np.uint8)
        pic = raw[fno * 2 * w * h:(fno + 1) * 2 * w * h]
        orc.orc_uyvy_to_planar(pic.ctypes.data_as(C.POINTER(C.c_uint8)), got.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)
        assert np.array_equal(want, got)
    libc.fclose(f)
