"""Lockstep batch encoding (dsv2hip_enc_batch): n streams per step == n independent reference encodes."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import dsvabi as A
from codec_run import configure_encoder, encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


@pytest.mark.parametrize("w,h,nstreams,nframes,gop", [(352, 288, 3, 7, 4), (1280, 720, 4, 3, 48)])
def test_batch_equals_reference_per_stream(w, h, nstreams, nframes, gop):
    ref, hip = A.load_ref(), A.load_hip()
    hip.dsv2hip_enc_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch.restype = C.c_int
    pkg = load_pkg()
    vids = [pkg.synth.SynthVideo(w, h, "420", seed=20 + s) for s in range(nstreams)]
    frames = [[v.frame_bytes(t) for t in range(nframes)] for v in vids]
    # stream 1 gets a hard scene cut so that one stream flips P->I while the others stay P
    frames[1][nframes - 2] = bytes(255 - b for b in frames[1][nframes - 2])
    want = [encode_stream(ref, frames[s], w, h, A.SUBSAMP_420, eos=False, qp=60, gop=gop)[0] for s in range(nstreams)]

    meta = A.mk_meta(w, h, A.SUBSAMP_420)
    encs = [A.ENCODER() for _ in range(nstreams)]
    for e in encs:
        configure_encoder(hip, e, meta, qp=60, gop=gop)
    encp = (C.POINTER(A.ENCODER) * nstreams)(*[C.pointer(e) for e in encs])
    bufs = (A.BUF * (4 * nstreams))()
    nbufs = (C.c_int * nstreams)()
    got = [[] for _ in range(nstreams)]
    for t in range(nframes):
        dev = [torch.from_numpy(np.frombuffer(frames[s][t], dtype=np.uint8).copy()).cuda() for s in range(nstreams)]
        torch.cuda.synchronize()
        ptrs = (C.c_void_p * nstreams)(*[d.data_ptr() for d in dev])
        assert hip.dsv2hip_enc_batch(nstreams, encp, ptrs, bufs, nbufs) == 0
        for s in range(nstreams):
            for i in range(nbufs[s]):
                b = bufs[4 * s + i]
                got[s].append(bytes(C.string_at(b.data, b.len)))
                hip.dsv_buf_free(C.byref(b))
    for e in encs:
        hip.dsv_enc_free(C.byref(e))
    for s in range(nstreams):
        assert len(want[s]) == len(got[s])
        for i, (a, b) in enumerate(zip(want[s], got[s])):
            assert a == b, "stream %d packet %d differs" % (s, i)
