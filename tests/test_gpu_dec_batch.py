"""Lockstep batch decoding (dsv2hip_dec_batch): n packets per step == n independent reference decodes."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from codec_run import decode_stream, encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def bind(hip):
    hip.dsv2hip_dec_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.DECODER)), C.POINTER(A.BUF), C.POINTER(C.POINTER(A.FRAME)),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    hip.dsv2hip_dec_batch.restype = C.c_int


def planes_of(fp):
    f = fp.contents
    out = []
    for c in range(3):
        p = f.planes[c]
        a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
        out.append(a.reshape(-1, p.stride)[:p.h, :p.w].copy())
    return out


def batch_decode(hip, streams):
    """streams: list of packet lists.  Step t feeds packet t of every stream that still has one."""
    n = len(streams)
    decs = [A.DECODER() for _ in range(n)]
    got = [[] for _ in range(n)]
    for t in range(max(len(s) for s in streams)):
        live = [k for k in range(n) if t < len(streams[k])]
        m = len(live)
        decp = (C.POINTER(A.DECODER) * m)(*[C.pointer(decs[k]) for k in live])
        bufs = (A.BUF * m)()
        for i, k in enumerate(live):
            pk = streams[k][t]
            hip.dsv_mk_buf(C.byref(bufs[i]), len(pk) + 64)
            C.memmove(bufs[i].data, pk, len(pk))
        outs = (C.POINTER(A.FRAME) * m)()
        fns = (C.c_uint32 * m)()
        rets = (C.c_int * m)()
        assert hip.dsv2hip_dec_batch(m, decp, bufs, outs, fns, rets) == m
        for i, k in enumerate(live):
            assert rets[i] != A.DEC_ERROR
            if rets[i] == A.DEC_OK and outs[i]:
                got[k].append((fns[i], *planes_of(outs[i])))
                hip.dsv_frame_ref_dec(outs[i])
    for d in decs:
        hip.dsv_dec_free(C.byref(d))
    return got


def check(want, got):
    assert len(want) == len(got)
    for (fa, *pa), (fb, *pb) in zip(want, got):
        assert fa == fb
        for c in range(3):
            assert np.array_equal(pa[c], pb[c]), "frame %d plane %d differs" % (fa, c)


def test_dec_batch_mixed_streams():
    """Four streams in lockstep: two CIF IP streams of different length, one intra-only CIF stream, and a
    720p stream (second geometry in the same step); metadata / picture / end-of-stream packets interleave."""
    ref, hip = A.load_ref(), A.load_hip()
    bind(hip)
    pkg = load_pkg()
    spec = [(352, 288, 9, dict(qp=60, gop=4)), (352, 288, 5, dict(qp=40, gop=12)), (352, 288, 4, dict(qp=85, gop=0)),
            (1280, 720, 3, dict(qp=60, gop=48))]
    streams = []
    for s, (w, h, nfr, cfg) in enumerate(spec):
        v = pkg.synth.SynthVideo(w, h, "420", seed=70 + s)
        frames = [v.frame_bytes(t) for t in range(nfr)]
        streams.append(encode_stream(ref, frames, w, h, A.SUBSAMP_420, eos=True, **cfg)[0])
    want = [decode_stream(ref, pk) for pk in streams]
    got = batch_decode(hip, streams)
    for s in range(len(spec)):
        check(want[s], got[s])


def test_dec_batch_444_lossless_and_single_call_agree():
    ref, hip = A.load_ref(), A.load_hip()
    bind(hip)
    pkg = load_pkg()
    streams = []
    for s in range(2):
        v = pkg.synth.SynthVideo(354, 290, "444", seed=90 + s)
        frames = [v.frame_bytes(t) for t in range(3)]
        streams.append(encode_stream(ref, frames, 354, 290, A.SUBSAMP_444, eos=True, qp=100 if s else 70, gop=48)[0])
    want = [decode_stream(ref, pk) for pk in streams]
    got = batch_decode(hip, streams)
    single = [decode_stream(hip, pk) for pk in streams]
    for s in range(2):
        check(want[s], got[s])
        check(want[s], single[s])


def test_lossless_round_trip_1080p_through_both_batch_engines():
    """Size-independent property at the full BASELINE picture size: lossless (-qp=100) batch encode followed by
    batch decode returns the input pictures bit for bit (no reference needed)."""
    import torch
    from codec_run import configure_encoder
    hip = A.load_hip()
    bind(hip)
    hip.dsv2hip_enc_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch.restype = C.c_int
    pkg = load_pkg()
    w, h, ns, nf = 1920, 1080, 2, 3
    vids = [pkg.synth.SynthVideo(w, h, "420", seed=40 + s) for s in range(ns)]
    frames = [[v.frame_bytes(t) for t in range(nf)] for v in vids]
    meta = A.mk_meta(w, h, A.SUBSAMP_420)
    encs = [A.ENCODER() for _ in range(ns)]
    for e in encs:
        configure_encoder(hip, e, meta, qp=100, gop=48)
    encp = (C.POINTER(A.ENCODER) * ns)(*[C.pointer(e) for e in encs])
    bufs = (A.BUF * (4 * ns))()
    nbufs = (C.c_int * ns)()
    streams = [[] for _ in range(ns)]
    for t in range(nf):
        dev = [torch.from_numpy(np.frombuffer(frames[s][t], dtype=np.uint8).copy()).cuda() for s in range(ns)]
        torch.cuda.synchronize()
        ptrs = (C.c_void_p * ns)(*[d.data_ptr() for d in dev])
        assert hip.dsv2hip_enc_batch(ns, encp, ptrs, bufs, nbufs) == 0
        for s in range(ns):
            for i in range(nbufs[s]):
                b = bufs[4 * s + i]
                streams[s].append(bytes(C.string_at(b.data, b.len)))
                hip.dsv_buf_free(C.byref(b))
    for e in encs:
        hip.dsv_enc_free(C.byref(e))
    got = batch_decode(hip, streams)
    for s in range(ns):
        assert len(got[s]) == nf
        for t, (fn, y, u, v) in enumerate(got[s]):
            raw = np.frombuffer(frames[s][t], dtype=np.uint8)
            assert np.array_equal(y.ravel(), raw[:w * h])
            assert np.array_equal(u.ravel(), raw[w * h:w * h + w * h // 4])
            assert np.array_equal(v.ravel(), raw[w * h + w * h // 4:])
