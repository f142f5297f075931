"""Chroma format coverage of the drop-in encoder / decoder: 4:4:4, 4:2:2, 4:2:0, 4:1:1 and the reference's "4:1:0"
(quarter horizontal, quarter vertical, dsv.h:87-95) give bit-identical packets and pictures."""
import os

import numpy as np
import pytest

import dsvabi as A
from codec_run import decode_stream, encode_stream

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)

FMT = {"444": (0x0, 0, 0), "422": (0x4, 1, 0), "420": (0x5, 1, 1), "411": (0x8, 2, 0), "410": (0xA, 2, 2)}


def frames(w, h, hs, vs, n, seed):
    rng = np.random.default_rng(seed)
    cw, ch = (w + (1 << hs) - 1) >> hs, (h + (1 << vs) - 1) >> vs
    yy, xx = np.mgrid[0:h, 0:w]
    cy, cx = np.mgrid[0:ch, 0:cw]
    out = []
    for t in range(n):
        y = ((np.sin((xx + 3 * t) / 17.0) + np.cos((yy + 2 * t) / 11.0)) * 50 + 128 + rng.integers(-3, 4, (h, w))).clip(0, 255).astype(np.uint8)
        u = ((cx * 2 + t * 3) % 200 + 20).astype(np.uint8)
        v = ((cy * 3 + t * 5) % 180 + 30).astype(np.uint8)
        out.append(y.tobytes() + u.tobytes() + v.tobytes())
    return out


@pytest.mark.parametrize("name", sorted(FMT))
@pytest.mark.parametrize("w,h", [(352, 288), (176, 144)])
def test_format_bit_exact(name, w, h):
    code, hs, vs = FMT[name]
    ref, hip = A.load_ref(), A.load_hip()
    fr = frames(w, h, hs, vs, 4, 5)
    pr, _ = encode_stream(ref, fr, w, h, code, qp=60, gop=12)
    ph, _ = encode_stream(hip, fr, w, h, code, qp=60, gop=12)
    assert pr == ph
    dr, dh = decode_stream(ref, pr), decode_stream(hip, pr)
    assert len(dr) == len(dh)
    for a, b in zip(dr, dh):
        assert a[0] == b[0]
        for c in (1, 2, 3):
            assert np.array_equal(a[c], b[c])
