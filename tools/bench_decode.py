#!/usr/bin/env python3
"""Decode throughput of the lockstep batch decoder, same workload family as bench.py:
1080p 4:2:0 -qp=60 -gop=48 streams, S decoder instances per GPU in G lockstep groups.  Prints one JSON line.
The packets are produced on the fly with the GPU encoder (bit-identical to the reference's); every decoded
picture is delivered to host memory as a DSV_FRAME exactly like dsv_dec does (that D2H copy is part of the
timed region)."""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
W_, H_, GOP, QP = 1920, 1080, 48, 60


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--groups", type=int, default=4)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import dsvabi as A
    from codec_run import decode_stream, encode_stream
    from conftest import load_pkg

    hip = A.load_hip()
    assert hip.dsv2hip_device_ok() == 0, "no HIP device: the product has no CPU path"
    hip.dsv2hip_dec_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.DECODER)), C.POINTER(A.BUF), C.POINTER(C.POINTER(A.FRAME)),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    hip.dsv2hip_dec_batch.restype = C.c_int
    pkg = load_pkg()
    S, G, K, Wm = args.streams, max(1, min(args.groups, args.streams)), args.steps, args.warmup
    nfr = Wm + K
    # a few distinct videos, encoded once; streams reuse them
    vids = []
    for k in range(min(S, 4)):
        v = pkg.synth.SynthVideo(W_, H_, "420", seed=1 + k)
        uniq = [v.frame_bytes(t) for t in range(min(nfr, 24))]
        frames = [uniq[t % len(uniq)] if (t // len(uniq)) % 2 == 0 else uniq[len(uniq) - 1 - t % len(uniq)] for t in range(nfr)]
        pk = encode_stream(hip, frames, W_, H_, A.SUBSAMP_420, eos=False, qp=QP, gop=GOP, effort=10)[0]
        vids.append(pk)  # [meta, pic0, pic1, ...] (one metadata packet per GOP start)
    decs = [A.DECODER() for _ in range(S)]
    group_of = [list(range(g, S, G)) for g in range(G)]

    def make_bufs(ids, t):
        m = len(ids)
        bufs = (A.BUF * m)()
        for i, s in enumerate(ids):
            pk = vids[s % len(vids)][t]
            hip.dsv_mk_buf(C.byref(bufs[i]), len(pk) + 64)
            C.memmove(bufs[i].data, pk, len(pk))
        return bufs

    npk = len(vids[0])
    assert all(len(v) == npk for v in vids)
    # packets [0, first) = warm-up (metadata + first pictures), then the timed ones
    first = npk - K if npk > K else 0
    plan = {g: [make_bufs(group_of[g], t) for t in range(npk)] for g in range(G)}
    decoded = [0] * G

    def worker(g, t0, t1, bar):
        ids = group_of[g]
        m = len(ids)
        decp = (C.POINTER(A.DECODER) * m)(*[C.pointer(decs[s]) for s in ids])
        outs = (C.POINTER(A.FRAME) * m)()
        fns = (C.c_uint32 * m)()
        rets = (C.c_int * m)()
        bar.wait()
        for t in range(t0, t1):
            hip.dsv2hip_dec_batch(m, decp, plan[g][t], outs, fns, rets)
            for i in range(m):
                if rets[i] == A.DEC_OK and outs[i]:
                    decoded[g] += 1
                    hip.dsv_frame_ref_dec(outs[i])
        bar.wait()

    def run(t0, t1):
        bar = threading.Barrier(G + 1)
        ths = [threading.Thread(target=worker, args=(g, t0, t1, bar)) for g in range(G)]
        for th in ths:
            th.start()
        ts = time.perf_counter()
        bar.wait()
        bar.wait()
        te = time.perf_counter()
        for th in ths:
            th.join()
        return te - ts

    run(0, first)
    before = sum(decoded)
    elapsed = run(first, npk)
    nframes = sum(decoded) - before
    for d in decs:
        hip.dsv_dec_free(C.byref(d))
    fps = nframes / elapsed
    result = {"metric": "decoded frames/s, 1080p 4:2:0 qp=60 gop=48 (pictures identical to the reference decoder's)", "value": round(fps, 2),
              "unit": "frames/s", "n_gpus": 1, "steps": npk - first, "ms_per_step": round(1e3 * elapsed / max(1, npk - first), 3),
              "higher_is_better": True, "dtype": "u8/int32", "data": "synthetic",
              "config": {"workload": "1920x1080 4:2:0 -qp=60 -gop=48, %d decoder instances in %d lockstep groups, frames delivered to host memory" % (S, G),
                         "streams_per_gpu": S, "groups": G, "frames": nframes, "mpix_per_s": round(fps * W_ * H_ / 1e6, 1)}}
    if not args.no_cpu_baseline and os.path.exists(A.REF_SO):
        ref = A.load_ref()
        pk = vids[0][:25]
        t0 = time.perf_counter()
        out = decode_stream(ref, pk)
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(len(out) / dt, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                                  "sample": "first %d pictures of stream 0, reference C decoder -O3, 1 thread" % len(out)}
    print(json.dumps(result))


if __name__ == "__main__":
    main()
