"""Compaction lists sized by need (round 4: CodecDev::init's list_symbols, encoder.cpp's redo of an overflowing picture).

A stream's symbol lists start at an eighth of its coefficient count; a picture with more symbols is seen in the count that
comes back, the lists are enlarged to the worst case and the picture's symbols are worked out a second time (predict +
subtract or the source copy into a spare working picture, forward transform, quantiser, compaction) and coded on the host.
The packets must not change: every case here is compared packet by packet with the reference encoder.

  DSV2_COMPACT_CAP=<symbols>   start the lists that short (the first intra picture overflows, whatever its size)
  DSV2_COMPACT_REDO=1          take the redo path on EVERY picture (intra and P, the two ways a working picture is made)"""
import ctypes as C
import hashlib
import os

import pytest

import dsvabi as A
from codec_run import configure_encoder, encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def frames_of(w, h, fmt, n, seed):
    v = load_pkg().synth.SynthVideo(w, h, fmt, seed=seed)
    return [v.frame_bytes(t) for t in range(n)]


def md5s(packets):
    return [hashlib.md5(p).hexdigest() for p in packets]


@pytest.mark.parametrize("fmt,sub", [("420", A.SUBSAMP_420), ("444", A.SUBSAMP_444)])
@pytest.mark.parametrize("switch", [dict(DSV2_COMPACT_CAP=1000), dict(DSV2_COMPACT_REDO=1), dict(DSV2_COMPACT_CAP=1000, DSV2_GPU_ENTROPY=0)])
def test_short_lists_and_forced_redo_equal_reference(fmt, sub, switch):
    ref, hip = A.load_ref(), A.load_hip()
    w, h, n = 640, 360, 9
    fr = frames_of(w, h, fmt, n, seed=71)
    want = encode_stream(ref, fr, w, h, sub, qp=70, gop=4, effort=10)[0]
    with env(**switch):
        got = encode_stream(hip, fr, w, h, sub, qp=70, gop=4, effort=10)[0]
    assert md5s(got) == md5s(want)


def test_lists_overflow_by_themselves_at_high_quality():
    """No switch: 1080p noise-rich content near the top of the quality range has more symbols than an eighth of the
    coefficients (the default lists) -- the first picture overflows on its own, later ones run in the enlarged lists."""
    ref, hip = A.load_ref(), A.load_hip()
    import numpy as np
    w, h, n = 1920, 1080, 3
    rng = np.random.default_rng(5)
    base = frames_of(w, h, "420", n, seed=72)
    fr = []
    for b in base:
        a = np.frombuffer(b, np.uint8).astype(np.int16) + rng.integers(-24, 25, len(b), dtype=np.int16)
        fr.append(np.clip(a, 0, 255).astype(np.uint8).tobytes())
    want = encode_stream(ref, fr, w, h, A.SUBSAMP_420, qp=97, gop=2, effort=10)[0]
    hip.dsv2hip_enc_list_growths.restype = C.c_long
    before = hip.dsv2hip_enc_list_growths()
    got = encode_stream(hip, fr, w, h, A.SUBSAMP_420, qp=97, gop=2, effort=10)[0]
    assert md5s(got) == md5s(want)
    assert hip.dsv2hip_enc_list_growths() > before, "the case did not overflow the default lists: it tests nothing"


def test_lossless_streams_start_with_full_lists():
    ref, hip = A.load_ref(), A.load_hip()
    w, h, n = 640, 360, 3
    fr = frames_of(w, h, "420", n, seed=73)
    want = encode_stream(ref, fr, w, h, A.SUBSAMP_420, qp=100, gop=2, effort=10)[0]
    hip.dsv2hip_enc_list_growths.restype = C.c_long
    before = hip.dsv2hip_enc_list_growths()
    got = encode_stream(hip, fr, w, h, A.SUBSAMP_420, qp=100, gop=2, effort=10)[0]
    assert md5s(got) == md5s(want)
    assert hip.dsv2hip_enc_list_growths() == before


def test_batch_of_streams_some_overflowing():
    """One lockstep launch over streams of two qualities and lists between their symbol counts: the pictures of the high-quality
    streams are redone beside those of the others, which are left alone."""
    ref, hip = A.load_ref(), A.load_hip()
    hip.dsv2hip_enc_batch_host.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch_host.restype = C.c_int
    hip.dsv2hip_host_alloc.argtypes = [C.c_size_t]
    hip.dsv2hip_host_alloc.restype = C.c_void_p
    hip.dsv2hip_host_free.argtypes = [C.c_void_p]
    hip.dsv2hip_enc_list_growths.restype = C.c_long
    w, h, n, S = 640, 360, 6, 6
    qp = [92 if s % 2 else 25 for s in range(S)]
    inputs = [frames_of(w, h, "420", n, seed=80 + s) for s in range(S)]
    want = [md5s(encode_stream(ref, inputs[s], w, h, A.SUBSAMP_420, eos=False, qp=qp[s], gop=3, effort=10)[0]) for s in range(S)]
    P = len(inputs[0][0])
    pinned = []
    for s in range(S):
        p = hip.dsv2hip_host_alloc(P * n)
        for t in range(n):
            C.memmove(p + t * P, inputs[s][t], P)
        pinned.append(p)
    meta = A.mk_meta(w, h, A.SUBSAMP_420)
    encs = [A.ENCODER() for _ in range(S)]
    for s, e in enumerate(encs):
        configure_encoder(hip, e, meta, qp=qp[s], gop=3, effort=10)
    got = [[] for _ in range(S)]
    before = hip.dsv2hip_enc_list_growths()
    with env(DSV2_COMPACT_CAP=100000):
        for t in range(n):
            gp = (C.POINTER(A.ENCODER) * S)(*[C.pointer(e) for e in encs])
            gb = (A.BUF * (4 * S))()
            gn = (C.c_int * S)()
            cur = (C.c_void_p * S)(*[pinned[s] + t * P for s in range(S)])
            nxt = (C.c_void_p * S)(*[(pinned[s] + (t + 1) * P) if t + 1 < n else None for s in range(S)])
            assert hip.dsv2hip_enc_batch_host(S, gp, cur, nxt, gb, gn) == 0
            for s in range(S):
                for b in range(gn[s]):
                    buf = gb[4 * s + b]
                    got[s].append(hashlib.md5(C.string_at(buf.data, buf.len)).hexdigest())
                    hip.dsv_buf_free(C.byref(buf))
            if t == 0:
                assert hip.dsv2hip_enc_list_growths() - before == S // 2, "lists of 100 000 symbols do not separate the two qualities"
    for e in encs:
        hip.dsv_enc_free(C.byref(e))
    for p in pinned:
        hip.dsv2hip_host_free(p)
    assert got == want
