"""1080p streams of the bench's content classes -- panning, a scene cut in mid-stream, a static picture, fast motion -- through
the lockstep batch engine (dsv2hip_enc_batch_host), in ONE launch together and at two GOP phases each, every stream against
the reference encode of its own input (until round 4 these classes were only checked inside bench.py).

What the classes exercise (dsv_encoder.c:546 scene_change_detection, hme.c:1559 "good enough", bmc.c:990 skip blocks):
  cut     a P picture that the scene-change test flips to intra in mid-batch, beside pictures that stay P
  static  every block of every P picture skipped
  fast    4.5 / 3 pixels a frame and squares up to 12: long vectors, many intra blocks
The short GOP (6) makes every stream cross GOP boundaries inside the run, at a phase of its own."""
import ctypes as C
import hashlib
import os

import pytest

import dsvabi as A
from codec_run import configure_encoder, encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)

W, H, GOP, QP, NFRAMES = 1920, 1080, 6, 60, 14
CLASSES = ("pan", "cut", "static", "fast")
PHASES = (0, 3)  # global step at which a stream codes its first picture


def stream_frames(klass, va, vb):
    if klass == "static":
        return [va[0]] * NFRAMES
    if klass == "fast":
        return [va[(3 * t) % len(va)] for t in range(NFRAMES)]
    if klass == "cut":
        return [va[t] if t < 7 else vb[t] for t in range(NFRAMES)]
    return [va[t] for t in range(NFRAMES)]


def test_content_classes_in_one_launch_equal_reference():
    ref, hip = A.load_ref(), A.load_hip()
    hip.dsv2hip_enc_batch_host.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch_host.restype = C.c_int
    hip.dsv2hip_host_alloc.argtypes = [C.c_size_t]
    hip.dsv2hip_host_alloc.restype = C.c_void_p
    hip.dsv2hip_host_free.argtypes = [C.c_void_p]
    pkg = load_pkg()
    nf = 3 * NFRAMES
    va = [pkg.synth.SynthVideo(W, H, "420", seed=61).frame_bytes(t) for t in range(nf)]
    vb = [pkg.synth.SynthVideo(W, H, "420", seed=62).frame_bytes(t) for t in range(NFRAMES)]
    inputs = {k: stream_frames(k, va, vb) for k in CLASSES}
    want = {k: [hashlib.md5(p).hexdigest() for p in encode_stream(ref, inputs[k], W, H, A.SUBSAMP_420, eos=False, qp=QP, gop=GOP, effort=10)[0]]
            for k in CLASSES}
    # the cut really flips a P picture: an intra picture (packet type bit 0 clear) that is not at a GOP start
    pk_cut = encode_stream(ref, inputs["cut"], W, H, A.SUBSAMP_420, eos=False, qp=QP, gop=GOP, effort=10)[0]
    pics = [p for p in pk_cut if p[5] & 0x04]  # DSV_PT_PIC | is_ref << 1 | has_ref (dsv.h:41-45)
    assert len(pics) == NFRAMES
    assert not (pics[7][5] & 1) and 7 % GOP, "the scene cut at frame 7 did not flip that P picture to intra: the class tests nothing"

    P = len(va[0])
    streams = [(k, r0) for r0 in PHASES for k in CLASSES]  # ordered by phase: the started streams are always a prefix
    pinned = {}
    for k in CLASSES:
        p = hip.dsv2hip_host_alloc(P * NFRAMES)
        assert p
        for t in range(NFRAMES):
            C.memmove(p + t * P, inputs[k][t], P)
        pinned[k] = p
    meta = A.mk_meta(W, H, A.SUBSAMP_420)
    encs = [A.ENCODER() for _ in streams]
    for e in encs:
        configure_encoder(hip, e, meta, qp=QP, gop=GOP, effort=10)
    got = [[] for _ in streams]
    for step in range(NFRAMES + max(PHASES)):
        ids = [s for s, (k, r0) in enumerate(streams) if r0 <= step < r0 + NFRAMES]
        m = len(ids)
        gp = (C.POINTER(A.ENCODER) * m)(*[C.pointer(encs[s]) for s in ids])
        gb = (A.BUF * (4 * m))()
        gn = (C.c_int * m)()
        cur = (C.c_void_p * m)(*[pinned[streams[s][0]] + (step - streams[s][1]) * P for s in ids])
        nxt = (C.c_void_p * m)(*[(pinned[streams[s][0]] + (step + 1 - streams[s][1]) * P) if step + 1 - streams[s][1] < NFRAMES else None for s in ids])
        assert hip.dsv2hip_enc_batch_host(m, gp, cur, nxt, gb, gn) == 0
        for i, s in enumerate(ids):
            for b in range(gn[i]):
                buf = gb[4 * i + b]
                got[s].append(hashlib.md5(C.string_at(buf.data, buf.len)).hexdigest())
                hip.dsv_buf_free(C.byref(buf))
    for e in encs:
        hip.dsv_enc_free(C.byref(e))
    for p in pinned.values():
        hip.dsv2hip_host_free(p)
    for s, (k, r0) in enumerate(streams):
        assert got[s] == want[k], "class %s at GOP phase %d: first differing packet %d" % (
            k, r0, next((i for i, (a, b) in enumerate(zip(got[s], want[k])) if a != b), min(len(got[s]), len(want[k]))))
