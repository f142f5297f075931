"""The decoder's output frames INCLUDING their 32-pixel borders (frame.c:358-404 extend_plane; dsv_decoder.c:552: a picture that
serves as a reference is extended before it is handed out).  Every other test compares visible pixels only; the borders are
what the NEXT picture's motion compensation reads, and what a caller that keeps the frame sees."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from codec_run import encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)

BORDER = 32


def decode_bordered(lib, packets):
    """[(fnum, is_ref, [bordered plane arrays])] -- every plane with its border rows and columns"""
    dec = A.DECODER()
    out = []
    for pk in packets:
        is_ref = bool(pk[5] & 0x04) and bool(pk[5] & 0x02)
        buf = A.BUF()
        lib.dsv_mk_buf(C.byref(buf), len(pk) + 64)
        C.memmove(buf.data, pk, len(pk))
        fp = C.POINTER(A.FRAME)()
        fn = C.c_uint32(0)
        code = lib.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
        if code == A.DEC_OK and fp:
            f = fp.contents
            assert f.border, "decoder output frames are bordered frames (dsv_decoder.c:467)"
            planes = []
            for c in range(3):
                p = f.planes[c]
                base = C.cast(p.data, C.c_void_p).value - BORDER * p.stride - BORDER
                a = np.ctypeslib.as_array((C.c_uint8 * ((p.h + 2 * BORDER) * p.stride)).from_address(base))
                planes.append(a.reshape(p.h + 2 * BORDER, p.stride)[:, :p.w + 2 * BORDER].copy())
            out.append((fn.value, is_ref, planes))
            lib.dsv_frame_ref_dec(fp)
        elif code == A.DEC_EOS:
            break
        else:
            assert code != A.DEC_ERROR
    lib.dsv_dec_free(C.byref(dec))
    return out


@pytest.mark.parametrize("w,h,subsamp,fmt,n,gop", [(352, 288, A.SUBSAMP_420, "420", 7, 4), (354, 290, A.SUBSAMP_420, "420", 5, 3),
                                                    (1280, 720, A.SUBSAMP_420, "420", 4, 48), (320, 240, A.SUBSAMP_444, "444", 5, 4)])
def test_reference_pictures_equal_including_borders(w, h, subsamp, fmt, n, gop):
    ref, hip = A.load_ref(), A.load_hip()
    v = load_pkg().synth.SynthVideo(w, h, fmt, seed=91)
    packets = encode_stream(ref, [v.frame_bytes(t) for t in range(n)], w, h, subsamp, eos=True, qp=60, gop=gop)[0]
    a, b = decode_bordered(ref, packets), decode_bordered(hip, packets)
    assert len(a) == len(b) == n
    checked = 0
    for (fa, ra, pa), (fb, rb, pb) in zip(a, b):
        assert fa == fb and ra == rb
        for c in range(3):
            vis_a, vis_b = pa[c][BORDER:-BORDER, BORDER:-BORDER], pb[c][BORDER:-BORDER, BORDER:-BORDER]
            assert np.array_equal(vis_a, vis_b), "frame %d plane %d: visible pixels differ" % (fa, c)
            if ra:  # a reference picture: extended before it is returned -- the whole bordered plane must agree
                assert np.array_equal(pa[c], pb[c]), "frame %d plane %d: border differs" % (fa, c)
                checked += 1
    assert checked == 3 * n  # (gop != 0: every picture is a reference picture, dsv_encoder.c:1247-1271)
