"""Body of __graft_entry__.smoke(): a CIF frame through the HIP subband transform, checked against the oracle."""
import ctypes as C

import numpy as np

import dsvabi as A
from test_oracle_sbt import rand_frame


def run():
    hip = A.load_hip()
    assert hip.dsv2hip_device_ok() == 0, "no HIP device"
    orc = A.load_oracle()
    w, h, subsamp = 352, 288, A.SUBSAMP_420
    meta = A.mk_meta(w, h, subsamp)
    for isP in (0, 1):
        params = A.mk_params(meta, w, h, isP, 0)
        nb = params.nblocks_h * params.nblocks_v
        blockdata = (np.arange(nb) % 16).astype(np.uint8)
        frame = rand_frame(subsamp, w, h, seed=11)
        cdims = A.coef_dims(subsamp, w, h)
        for plane in range(3):
            cw, ch = cdims[plane]
            pw, ph = frame.dims[plane]
            want = np.zeros(cw * ch, dtype=np.int32)
            orc.orc_fwd_sbt(frame.c.planes[plane].data, frame.strides[plane], pw, ph,
                            A.np_ptr(want, C.c_int32), cw, ch, plane, isP, 0,
                            A.np_ptr(blockdata, C.c_uint8), params.nblocks_h, params.nblocks_v)
            got = np.zeros(cw * ch, dtype=np.int32)
            cs = A.COEFS(A.np_ptr(got, C.c_int32), cw, ch)
            fm = A.FMETA()
            fm.params = C.pointer(params)
            fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
            fm.cur_plane, fm.isP = plane, isP
            hip.dsv_fwd_sbt(frame.plane_ptr(plane), C.byref(cs), C.byref(fm))
            assert np.array_equal(want, got), "smoke: fwd sbt mismatch plane %d isP %d" % (plane, isP)
    print("smoke ok")
    run_codec()


def run_codec():
    """two CIF streams, 4 frames each, through the lockstep batch engine with host pictures (upload, every kernel group,
    GPU entropy coder) against the real reference where oracle/_ref travelled with the repo"""
    import os
    from codec_run import configure_encoder, encode_stream
    from conftest import load_pkg
    hip = A.load_hip()
    hip.dsv2hip_enc_batch_host.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(A.BUF),
                                           C.POINTER(C.c_int)]
    w, h, n, S = 352, 288, 4, 2
    vids = [load_pkg().synth.SynthVideo(w, h, "420", seed=70 + s) for s in range(S)]
    frames = [[np.frombuffer(v.frame_bytes(t), dtype=np.uint8).copy() for t in range(n)] for v in vids]
    meta = A.mk_meta(w, h, A.SUBSAMP_420)
    encs = [A.ENCODER() for _ in range(S)]
    for e in encs:
        configure_encoder(hip, e, meta, qp=60, gop=48)
    gp = (C.POINTER(A.ENCODER) * S)(*[C.pointer(e) for e in encs])
    gb = (A.BUF * (4 * S))()
    gn = (C.c_int * S)()
    got = [b"" for _ in range(S)]
    for t in range(n):
        cur = (C.c_void_p * S)(*[frames[s][t].ctypes.data for s in range(S)])
        nxt = (C.c_void_p * S)(*[(frames[s][t + 1].ctypes.data if t + 1 < n else None) for s in range(S)])
        assert hip.dsv2hip_enc_batch_host(S, gp, cur, nxt, gb, gn) == 0
        for s in range(S):
            for i in range(gn[s]):
                b = gb[4 * s + i]
                got[s] += C.string_at(b.data, b.len)
                hip.dsv_buf_free(C.byref(b))
    for e in encs:
        hip.dsv_enc_free(C.byref(e))
    if os.path.exists(A.REF_SO):
        ref = A.load_ref()
        for s in range(S):
            want, _ = encode_stream(ref, [f.tobytes() for f in frames[s]], w, h, A.SUBSAMP_420, eos=False, qp=60, gop=48)
            assert got[s] == b"".join(want), "smoke: stream %d differs from the reference" % s
        print("smoke codec ok (vs reference)")
    else:
        assert all(len(g) > 1000 for g in got)
        print("smoke codec ran (oracle/_ref absent: not compared)")
