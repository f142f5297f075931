"""Damaged picture packets: the GPU decoder returns the same codes and the same pictures as the reference
decoder (planes that fail to parse stay zero, dsv_decoder.c:516-523; truncated symbol lists, hzcc.c:525-529)."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from codec_run import encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built")]


def decode_all(lib, packets):
    """Like codec_run.decode_stream but keeps going after DSV_DEC_ERROR and records every return code."""
    dec = A.DECODER()
    out = []
    for pk in packets:
        buf = A.BUF()
        lib.dsv_mk_buf(C.byref(buf), len(pk) + 64)
        C.memmove(buf.data, pk, len(pk))
        fp = C.POINTER(A.FRAME)()
        fn = C.c_uint32(0)
        code = lib.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
        planes = None
        if code == A.DEC_OK and fp:
            f = fp.contents
            planes = []
            for c in range(3):
                p = f.planes[c]
                a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
                planes.append(a.reshape(-1, p.stride)[:p.h, :p.w].copy())
            lib.dsv_frame_ref_dec(fp)
        out.append((code, fn.value if code == A.DEC_OK else None, planes))
        if code == A.DEC_EOS:
            break
    lib.dsv_dec_free(C.byref(dec))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_damaged_plane_data(seed):
    ref, hip = A.load_ref(), A.load_hip()
    ref.dsv_set_log_level(0)  # the reference reports every damaged plane on stderr
    pkg = load_pkg()
    w, h = 352, 288
    v = pkg.synth.SynthVideo(w, h, "420", seed=11)
    frames = [v.frame_bytes(t) for t in range(5)]
    packets, _ = encode_stream(ref, frames, w, h, A.SUBSAMP_420, eos=True, qp=60, gop=4)
    rng = np.random.default_rng(seed)
    damaged = []
    for pk in packets:
        b = bytearray(pk)
        if len(b) > 400:  # a picture packet: damage bytes in the second half (coefficient data of some plane)
            for _ in range(1 + seed % 3):
                i = int(rng.integers(len(b) // 2, len(b) - 8))
                b[i] ^= int(rng.integers(1, 256))
        damaged.append(bytes(b))
    want, got = decode_all(ref, damaged), decode_all(hip, damaged)
    assert [x[0] for x in want] == [x[0] for x in got]
    for (cw, fw, pw), (cg, fg, pg) in zip(want, got):
        assert fw == fg
        if pw is not None:
            for c in range(3):
                assert np.array_equal(pw[c], pg[c])
