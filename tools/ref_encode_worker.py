#!/usr/bin/env python3
"""CPU-side worker of bench.py: encodes one synthetic stream with the REAL reference library
(oracle/_ref/libdsv2ref.so, the unmodified reference compiled by oracle/Makefile) on one thread.

It is test/measurement infrastructure (cpu_baseline, cpu_baseline_8proc and the in-bench parity
check); it never touches the GPU or the product library.  Protocol (line based, stdin/stdout):

    argv:   W H FMT(420|444) SEED QP GOP EFFORT OUT_PATH  i0,i1,i2,...   (frame indices of the synthetic video, in order)
    stdout: "ready"                      after the frames are generated
    stdin:  "go N"                       encode the first N frames with a fresh encoder
    stdout: {"frames": N, "t0": .., "t1": .., "cpu_s": ..}   (wall clock of the encode loop only)
            and OUT_PATH holds, per frame, a 4-byte little-endian length + that frame's packet bytes
    stdin:  "quit"
"""
import json
import os
import struct
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    w, h, fmt, seed, qp, gop, effort = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    out_path = sys.argv[8]
    idx = [int(x) for x in sys.argv[9].split(",")]
    import ctypes as C

    import numpy as np

    import dsvabi as A
    from codec_run import configure_encoder
    from conftest import load_pkg

    pkg = load_pkg()
    v = pkg.synth.SynthVideo(w, h, fmt, seed=seed)
    cache = {}
    for i in idx:
        if i not in cache:
            cache[i] = np.frombuffer(v.frame_bytes(i), dtype=np.uint8).copy()
    ref = A.load_ref()
    subsamp = A.SUBSAMP_420 if fmt == "420" else A.SUBSAMP_444
    print("ready", flush=True)
    for line in sys.stdin:
        cmd = line.split()
        if not cmd or cmd[0] == "quit":
            break
        n = min(int(cmd[1]), len(idx))
        meta = A.mk_meta(w, h, subsamp)
        enc = A.ENCODER()
        configure_encoder(ref, enc, meta, qp=qp, gop=gop, effort=effort)
        bufs = (A.BUF * 4)()
        per_frame = []
        c0 = time.process_time()
        t0 = time.time()
        for t in range(n):
            arr = cache[idx[t]]
            fr = ref.dsv_load_planar_frame(subsamp, arr.ctypes.data, w, h)
            nb = ref.dsv_enc(C.byref(enc), fr, bufs)
            pk = b""
            for i in range(nb):
                pk += C.string_at(bufs[i].data, bufs[i].len)
                ref.dsv_buf_free(C.byref(bufs[i]))
            per_frame.append(pk)
        t1 = time.time()
        c1 = time.process_time()
        ref.dsv_enc_free(C.byref(enc))
        with open(out_path, "wb") as f:
            for pk in per_frame:
                f.write(struct.pack("<I", len(pk)))
                f.write(pk)
        print(json.dumps({"frames": n, "t0": t0, "t1": t1, "cpu_s": c1 - c0}), flush=True)


if __name__ == "__main__":
    main()
