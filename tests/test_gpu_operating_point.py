"""Bit-exactness AT THE BENCHMARKED OPERATING POINT: 1080p 4:2:0 qp=60 gop=48, many streams per lockstep step,
several lockstep groups running concurrently on one GPU (so the row-pipelined motion search of one group shares
the chip with the other groups' kernels), repeated in one process.  Every stream must equal the reference
encode of its input; further processes repeat the run with a stage in its other form (general block routine, few
persistent workers, the filters' other kernels) as a cross-check.

Reference: hme.c:1373-1833 (search order dependencies), parallel_encode_yuv.sh:36-50 (independent streams)."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import dsvabi as A
from codec_run import configure_encoder, encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)

W, H, GOP, QP = 1920, 1080, 48, 60
NSEED, NFRAMES = 4, 8


def bind(hip):
    hip.dsv2hip_enc_batch_host.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch_host.restype = C.c_int
    hip.dsv2hip_host_alloc.argtypes = [C.c_size_t]
    hip.dsv2hip_host_alloc.restype = C.c_void_p
    hip.dsv2hip_host_free.argtypes = [C.c_void_p]


def gen_inputs():
    pkg = load_pkg()
    return [[pkg.synth.SynthVideo(W, H, "420", seed=40 + k).frame_bytes(t) for t in range(NFRAMES)] for k in range(NSEED)]


def run_once(hip, frames, nstreams, ngroups, pinned):
    """nstreams encoders (stream s codes video s % NSEED) in ngroups concurrent lockstep groups; returns per-stream md5"""
    P = len(frames[0][0])
    meta = A.mk_meta(W, H, A.SUBSAMP_420)
    encs = [A.ENCODER() for _ in range(nstreams)]
    for e in encs:
        configure_encoder(hip, e, meta, qp=QP, gop=GOP, effort=10)
    digests = [hashlib.md5() for _ in range(nstreams)]
    groups = [list(range(g, nstreams, ngroups)) for g in range(ngroups)]
    errs = []

    def worker(ids):
        try:
            m = len(ids)
            gp = (C.POINTER(A.ENCODER) * m)(*[C.pointer(encs[s]) for s in ids])
            gb = (A.BUF * (4 * m))()
            gn = (C.c_int * m)()
            for t in range(NFRAMES):
                cur = (C.c_void_p * m)(*[pinned[s % NSEED] + t * P for s in ids])
                nxt = (C.c_void_p * m)(*[(pinned[s % NSEED] + (t + 1) * P) if t + 1 < NFRAMES else None for s in ids])
                assert hip.dsv2hip_enc_batch_host(m, gp, cur, nxt, gb, gn) == 0
                for k, s in enumerate(ids):
                    for i in range(gn[k]):
                        b = gb[4 * k + i]
                        digests[s].update(C.string_at(b.data, b.len))
                        hip.dsv_buf_free(C.byref(b))
        except BaseException as e:  # noqa: BLE001  (re-raised in the main thread)
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(ids,)) for ids in groups]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for e in encs:
        hip.dsv_enc_free(C.byref(e))
    if errs:
        raise errs[0]
    return [d.hexdigest() for d in digests]


def reference_digests(frames):
    ref = A.load_ref()
    out = []
    for k in range(NSEED):
        pk, _ = encode_stream(ref, frames[k], W, H, A.SUBSAMP_420, eos=False, qp=QP, gop=GOP, effort=10)
        out.append(hashlib.md5(b"".join(pk)).hexdigest())
    return out


def pin(hip, frames):
    P = len(frames[0][0])
    blocks = []
    for k in range(NSEED):
        p = hip.dsv2hip_host_alloc(P * NFRAMES)
        assert p
        for t in range(NFRAMES):
            C.memmove(p + t * P, frames[k][t], P)
        blocks.append(p)
    return blocks


def test_many_streams_concurrent_groups_repeated():
    hip = A.load_hip()
    bind(hip)
    frames = gen_inputs()
    want = reference_digests(frames)
    pinned = pin(hip, frames)
    nstreams, ngroups, repeats = 64, 2, 10
    for rep in range(repeats):
        got = run_once(hip, frames, nstreams, ngroups, pinned)
        bad = [s for s in range(nstreams) if got[s] != want[s % NSEED]]
        assert not bad, "repeat %d: streams %s differ from the reference" % (rep, bad[:8])
    # four groups, as the bench runs them
    got = run_once(hip, frames, 96, 4, pinned)
    assert all(got[s] == want[s % NSEED] for s in range(96))
    for p in pinned:
        hip.dsv2hip_host_free(p)


def test_the_headline_768_streams_in_four_groups():
    """the headline's own operating point (review, round 5: it was only checked inside bench.py): 768 encoder instances in four
    lockstep groups of 192, every launch carrying 192 pictures, the four groups passing the search token round -- every one of the
    768 streams against the reference encode of its input (63 GB of encoder instances: the test frees them before it returns)"""
    hip = A.load_hip()
    bind(hip)
    frames = gen_inputs()
    want = reference_digests(frames)
    pinned = pin(hip, frames)
    got = run_once(hip, frames, 768, 4, pinned)
    bad = [s for s in range(768) if got[s] != want[s % NSEED]]
    assert not bad, "768 streams / 4 groups: streams %s differ from the reference" % bad[:8]
    for p in pinned:
        hip.dsv2hip_host_free(p)


def test_192_pictures_per_launch_with_search_token():
    """the bench's launch size: 192 streams in ONE lockstep group (every kernel launch carries 192 pictures: 13 056 block rows
    in the level-0 search launch), then 384 streams in two groups of 192 that pass the search token back and
    forth; every stream against the reference"""
    hip = A.load_hip()
    bind(hip)
    frames = gen_inputs()
    want = reference_digests(frames)
    pinned = pin(hip, frames)
    got = run_once(hip, frames, 192, 1, pinned)
    bad = [s for s in range(192) if got[s] != want[s % NSEED]]
    assert not bad, "one group of 192: streams %s differ from the reference" % bad[:8]
    got = run_once(hip, frames, 384, 2, pinned)
    bad = [s for s in range(384) if got[s] != want[s % NSEED]]
    assert not bad, "two groups of 192: streams %s differ from the reference" % bad[:8]
    for p in pinned:
        hip.dsv2hip_host_free(p)


_CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import dsvabi as A
import test_gpu_operating_point as T
hip = A.load_hip()
T.bind(hip)
frames = T.gen_inputs()
pinned = T.pin(hip, frames)
print(json.dumps(T.run_once(hip, frames, 32, 2, pinned)))
"""


@pytest.mark.parametrize("env_extra", [{"DSV2_HME_FAST": "0"}, {"DSV2_HME_PERSIST": "64"}, {"DSV2_FILTER_PAIR_MAX": "0"}, {"DSV2_FILTER_PAIR_MAX": "100000"},
                                       {"DSV2_FILTER_RING": "0"}])
def test_alternative_forms_agree(env_extra):
    """the same streams in a process where a stage runs its other form: the search with the general block routine at every level
    (the reference's block loop with wave-cooperative primitives: no pre-passes, no fast routines); the row pipeline with 64
    persistent workers instead of 3 072 (rows queue for workers: the ticket order is all that keeps it moving); the in-loop
    filter's luma sweep with a lane per cell / a lane pair per cell whatever the batch; the in-loop filters' global-memory
    kernels (no LDS ring)"""
    frames = gen_inputs()
    want = reference_digests(frames)
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", _CHILD % os.path.dirname(os.path.abspath(__file__))], env=env, stdout=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(got[s] == want[s % NSEED] for s in range(len(got)))
