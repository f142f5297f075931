"""The reference's OWN entry points under the reference's own kind of parallelism: T host threads, each looping plain
dsv_enc (dsv_encoder.h:190-199) / dsv_dec (dsv_decoder.h:54-61) on an instance of its own with ordinary (pageable)
DSV_FRAMEs.  The library runs concurrent callers as shared lockstep steps (csrc/batch.h: Coalescer); every thread must
still get, call by call, exactly the packets / pictures the reference produces for its stream alone."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import dsvabi as A
from codec_run import configure_encoder, decode_stream, encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def _bind_stats(hip):
    for name in ("dsv2hip_enc_queue_stats", "dsv2hip_dec_queue_stats"):
        fn = getattr(hip, name)
        fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        fn.restype = None


def _enc_thread(hip, frames, w, h, subsamp, cfg, out, idx, start):
    meta = A.mk_meta(w, h, subsamp)
    enc = A.ENCODER()
    configure_encoder(hip, enc, meta, **cfg)
    bufs = (A.BUF * 4)()
    pk = []
    start.wait()
    for fb in frames:
        arr = np.frombuffer(fb, dtype=np.uint8).copy()  # pageable memory, as any caller of the reference would hand over
        fr = hip.dsv_load_planar_frame(subsamp, arr.ctypes.data, w, h)
        n = hip.dsv_enc(C.byref(enc), fr, bufs)
        for i in range(n):
            pk.append(bytes(C.string_at(bufs[i].data, bufs[i].len)))
            hip.dsv_buf_free(C.byref(bufs[i]))
    hip.dsv_enc_end_of_stream(C.byref(enc), bufs)
    pk.append(bytes(C.string_at(bufs[0].data, bufs[0].len)))
    hip.dsv_buf_free(C.byref(bufs[0]))
    hip.dsv_enc_free(C.byref(enc))
    out[idx] = pk


@pytest.mark.parametrize("w,h,threads,nframes,gop", [(352, 288, 6, 9, 4), (1280, 720, 8, 5, 48), (1920, 1080, 3, 4, 48)])
def test_threads_of_plain_dsv_enc_equal_reference(w, h, threads, nframes, gop):
    ref, hip = A.load_ref(), A.load_hip()
    _bind_stats(hip)
    pkg = load_pkg()
    vids = [pkg.synth.SynthVideo(w, h, "420", seed=300 + s) for s in range(threads)]
    frames = [[v.frame_bytes(t) for t in range(nframes)] for v in vids]
    # one thread's stream gets a scene cut (P -> I flip inside a shared step), another one is two frames shorter (it leaves
    # the crowd early: the others must not wait for it beyond the window)
    frames[1][nframes - 2] = bytes(255 - b for b in frames[1][nframes - 2])
    frames[2] = frames[2][:nframes - 2]
    cfg = dict(qp=60, gop=gop)
    want = [encode_stream(ref, frames[s], w, h, A.SUBSAMP_420, eos=True, **cfg)[0] for s in range(threads)]
    hip.dsv2hip_enc_queue_stats(None, 1)
    got = [None] * threads
    start = threading.Barrier(threads)
    ths = [threading.Thread(target=_enc_thread, args=(hip, frames[s], w, h, A.SUBSAMP_420, cfg, got, s, start)) for s in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for s in range(threads):
        assert got[s] is not None and len(got[s]) == len(want[s]), "stream %d: %s packets" % (s, None if got[s] is None else len(got[s]))
        for i, (a, b) in enumerate(zip(want[s], got[s])):
            assert a == b, "stream %d packet %d differs" % (s, i)
    st = (C.c_ulonglong * 4)()
    hip.dsv2hip_enc_queue_stats(st, 0)
    calls = sum(len(f) for f in frames)
    assert st[0] == calls
    # the callers really shared steps (the first call of a run may go alone; after that the crowd is expected)
    assert st[1] < calls and st[2] >= 2, "queue did not merge concurrent callers: %s" % list(st)


def _dec_thread(hip, packets, out, idx, start):
    dec = A.DECODER()
    pics = []
    start.wait()
    for pk in packets:
        buf = A.BUF()
        hip.dsv_mk_buf(C.byref(buf), len(pk) + 64)
        C.memmove(buf.data, pk, len(pk))
        fp = C.POINTER(A.FRAME)()
        fn = C.c_uint32(0)
        code = hip.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
        if code == A.DEC_OK and fp:
            f = fp.contents
            planes = []
            for c in range(3):
                p = f.planes[c]
                a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
                planes.append(a.reshape(-1, p.stride)[:p.h, :p.w].copy())
            pics.append((fn.value, planes))
            hip.dsv_frame_ref_dec(fp)
        elif code == A.DEC_EOS:
            break
        else:
            assert code != A.DEC_ERROR
    hip.dsv_dec_free(C.byref(dec))
    out[idx] = pics


@pytest.mark.parametrize("w,h,threads,nframes", [(352, 288, 5, 8), (1280, 720, 4, 4)])
def test_threads_of_plain_dsv_dec_equal_reference(w, h, threads, nframes):
    ref, hip = A.load_ref(), A.load_hip()
    _bind_stats(hip)
    pkg = load_pkg()
    streams = []
    for s in range(threads):
        v = pkg.synth.SynthVideo(w, h, "420", seed=340 + s)
        n = nframes - (2 if s == 1 else 0)
        streams.append(encode_stream(ref, [v.frame_bytes(t) for t in range(n)], w, h, A.SUBSAMP_420, eos=True, qp=60, gop=4)[0])
    want = [decode_stream(ref, pk) for pk in streams]
    hip.dsv2hip_dec_queue_stats(None, 1)
    got = [None] * threads
    start = threading.Barrier(threads)
    ths = [threading.Thread(target=_dec_thread, args=(hip, streams[s], got, s, start)) for s in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for s in range(threads):
        assert got[s] is not None and len(got[s]) == len(want[s])
        for (fa, ya, ua, va), (fb, pb) in zip(want[s], got[s]):
            assert fa == fb
            assert np.array_equal(ya, pb[0]) and np.array_equal(ua, pb[1]) and np.array_equal(va, pb[2]), "stream %d frame %d differs" % (s, fa)
    st = (C.c_ulonglong * 4)()
    hip.dsv2hip_dec_queue_stats(st, 0)
    assert st[0] == sum(len(p) for p in streams)
    assert st[1] < st[0] and st[2] >= 2, "queue did not merge concurrent dsv_dec callers: %s" % list(st)


def test_queue_off_is_the_direct_path():
    """DSV2_COALESCE=0 (read once per process: checked in a child) still equals the reference"""
    import subprocess
    import sys
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "import dsvabi as A\n"
        "from codec_run import encode_stream\n"
        "from conftest import load_pkg\n"
        "pkg = load_pkg(); v = pkg.synth.SynthVideo(352, 288, '420', seed=77)\n"
        "fr = [v.frame_bytes(t) for t in range(4)]\n"
        "a = encode_stream(A.load_ref(), fr, 352, 288, A.SUBSAMP_420, qp=60, gop=3)[0]\n"
        "b = encode_stream(A.load_hip(), fr, 352, 288, A.SUBSAMP_420, qp=60, gop=3)[0]\n"
        "assert a == b\n" % os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSV2_COALESCE="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
