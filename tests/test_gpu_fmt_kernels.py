"""Frame ingest / egress kernels through the C ABI:
  * UYVY pictures handed over interleaved (dsv2hip_enc_set_uyvy_input): the packets must equal the reference encoder's
    on the planar pictures dsv_yuv_read (dsv.c:177-205) would have produced;
  * pictures delivered as 4:2:0 by the decoder (dsv2hip_dec_set_out420p): must equal the reference decoder's picture put
    through the reference CLI's conversions (util.c:79-153), here via their pinned restatement oracle/orc_fmt.c."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import dsvabi as A
from codec_run import configure_encoder, decode_stream, encode_stream
from test_oracle_fmt import MODES, chroma_dims, orc_to420

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def rand_video(w, h, subsamp, n, seed):
    """smooth-ish random planar pictures (so that P frames have something to predict)"""
    rng = np.random.RandomState(seed)
    cw, ch = chroma_dims(subsamp, w, h)
    base = [rng.randint(0, 256, size=(hh // 8 + 2, ww // 8 + 2)).astype(np.uint8) for (ww, hh) in ((w, h), (cw, ch), (cw, ch))]
    out = []
    for t in range(n):
        planes = []
        for b, (ww, hh) in zip(base, ((w, h), (cw, ch), (cw, ch))):
            big = np.kron(np.roll(b, t, axis=1), np.ones((8, 8), dtype=np.uint8))[:hh, :ww]
            noise = rng.randint(0, 5, size=(hh, ww)).astype(np.uint8)
            planes.append(np.clip(big.astype(np.int32) + noise, 0, 255).astype(np.uint8))
        out.append(b"".join(p.tobytes() for p in planes))
    return out


def interleave_uyvy(planar, w, h):
    a = np.frombuffer(planar, dtype=np.uint8)
    y = a[:w * h].reshape(h, w)
    u = a[w * h:w * h + (w // 2) * h].reshape(h, w // 2)
    v = a[w * h + (w // 2) * h:].reshape(h, w // 2)
    out = np.zeros((h, 2 * w), dtype=np.uint8)
    out[:, 0::4] = u
    out[:, 1::2] = y
    out[:, 2::4] = v
    return out.tobytes()


@pytest.mark.parametrize("w,h", [(352, 288), (1280, 720), (330, 250)])
@pytest.mark.parametrize("via_host", [False, True])
def test_uyvy_ingest_equals_reference(w, h, via_host):
    ref, hip = A.load_ref(), A.load_hip()
    hip.dsv2hip_enc_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch_host.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(A.BUF),
                                           C.POINTER(C.c_int)]
    hip.dsv2hip_enc_set_uyvy_input.argtypes = [C.POINTER(A.ENCODER), C.c_int]
    n, UYVY = 5, 0x14
    planar = rand_video(w, h, A.SUBSAMP_422, n, seed=w + h)
    want, _ = encode_stream(ref, planar, w, h, UYVY, eos=False, qp=60, gop=48)
    meta = A.mk_meta(w, h, UYVY)
    enc = A.ENCODER()
    configure_encoder(hip, enc, meta, qp=60, gop=48)
    assert hip.dsv2hip_enc_set_uyvy_input(C.byref(enc), 1) == 0
    encp = (C.POINTER(A.ENCODER) * 1)(C.pointer(enc))
    bufs = (A.BUF * 4)()
    nb = (C.c_int * 1)()
    got = []
    inter = [np.frombuffer(interleave_uyvy(p, w, h), dtype=np.uint8).copy() for p in planar]
    for t in range(n):
        if via_host:
            cur = (C.c_void_p * 1)(inter[t].ctypes.data)
            nxt = (C.c_void_p * 1)(inter[t + 1].ctypes.data if t + 1 < n else None)
            assert hip.dsv2hip_enc_batch_host(1, encp, cur, nxt, bufs, nb) == 0
        else:
            d = torch.from_numpy(inter[t]).cuda()
            torch.cuda.synchronize()
            ptr = (C.c_void_p * 1)(d.data_ptr())
            assert hip.dsv2hip_enc_batch(1, encp, ptr, bufs, nb) == 0
        for i in range(nb[0]):
            got.append(bytes(C.string_at(bufs[i].data, bufs[i].len)))
            hip.dsv_buf_free(C.byref(bufs[i]))
    hip.dsv_enc_free(C.byref(enc))
    assert got == want


@pytest.mark.parametrize("subsamp", [A.SUBSAMP_444, A.SUBSAMP_422, 0x8, 0xA, A.SUBSAMP_420])
@pytest.mark.parametrize("w,h", [(352, 288), (330, 250)])
def test_out420p_equals_reference_conversions(subsamp, w, h):
    ref, hip, orc = A.load_ref(), A.load_hip(), A.load_oracle()
    hip.dsv2hip_dec_set_out420p.argtypes = [C.POINTER(A.DECODER), C.c_int]
    frames = rand_video(w, h, subsamp, 4, seed=subsamp + w)
    packets, _ = encode_stream(ref, frames, w, h, subsamp, eos=False, qp=70, gop=48)
    want = decode_stream(ref, packets)  # [(fnum, Y, U, V)] in the stream's own format
    dec = A.DECODER()
    assert hip.dsv2hip_dec_set_out420p(C.byref(dec), 1) == 0
    got = []
    for pk in packets:
        buf = A.BUF()
        hip.dsv_mk_buf(C.byref(buf), len(pk) + 64)
        C.memmove(buf.data, pk, len(pk))
        fp = C.POINTER(A.FRAME)()
        fn = C.c_uint32(0)
        code = hip.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
        if code == A.DEC_OK and fp:
            f = fp.contents
            assert f.format == A.SUBSAMP_420
            planes = []
            for c in range(3):
                p = f.planes[c]
                a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
                planes.append(a.reshape(-1, p.stride)[:p.h, :p.w].copy())
            got.append(planes)
            hip.dsv_frame_ref_dec(fp)
    hip.dsv_dec_free(C.byref(dec))
    assert len(got) == len(want)
    for (fnum, y, u, v), g in zip(want, got):
        assert np.array_equal(y, g[0])
        if subsamp == A.SUBSAMP_420:
            assert np.array_equal(u, g[1]) and np.array_equal(v, g[2])
        else:
            assert np.array_equal(orc_to420(orc, np.ascontiguousarray(u), subsamp, w, h), g[1])
            assert np.array_equal(orc_to420(orc, np.ascontiguousarray(v), subsamp, w, h), g[2])
