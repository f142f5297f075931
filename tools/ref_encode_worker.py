#!/usr/bin/env python3
"""CPU-side worker of bench.py: encodes one synthetic stream with the REAL reference library
(oracle/_ref/libdsv2ref.so, the unmodified reference compiled by oracle/Makefile) on one thread.

It is test/measurement infrastructure (cpu_baseline, cpu_baseline_8proc and the in-bench parity
check); it never touches the GPU or the product library.  Protocol (line based, stdin/stdout):

    argv:   W H FMT(420|444) SEED QP GOP EFFORT OUT_PATH  i0,i1,i2,...   (frame indices of the synthetic video, in order;
            an entry "seed:index" names a frame of ANOTHER synthetic video -- a stream with a scene cut)
    stdout: "ready"                      after the frames are generated
    stdin:  "go N"                       encode the first N frames with a fresh encoder
    stdout: {"frames": N, "t0": .., "t1": .., "cpu_s": ..}   (wall clock of the encode loop only)
            and OUT_PATH holds, per frame, a 4-byte little-endian length + that frame's packet bytes
    stdin:  "dec N"                      decode the packets of the last "go" (first N frames) with the reference DECODER
    stdout: {"frames": N, "t0": .., "t1": .., "md5": [hex, ...]}   md5 of every decoded picture (Y, U, V planes, tight)
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
    idx = [(int(x.split(":")[0]), int(x.split(":")[1])) if ":" in x else (seed, int(x)) for x in sys.argv[9].split(",")]
    import ctypes as C

    import numpy as np

    import dsvabi as A
    from codec_run import configure_encoder
    from conftest import load_pkg

    pkg = load_pkg()
    vids = {}
    cache = {}
    for i in idx:
        if i not in cache:
            if i[0] not in vids:
                vids[i[0]] = pkg.synth.SynthVideo(w, h, fmt, seed=i[0])
            cache[i] = np.frombuffer(vids[i[0]].frame_bytes(i[1]), dtype=np.uint8).copy()
    ref = A.load_ref()
    subsamp = A.SUBSAMP_420 if fmt == "420" else A.SUBSAMP_444
    print("ready", flush=True)
    packets = []  # per frame: list of packets of the last "go"
    for line in sys.stdin:
        cmd = line.split()
        if not cmd or cmd[0] == "quit":
            break
        if cmd[0] == "dec":
            print(json.dumps(decode(ref, A, C, np, packets[:int(cmd[1])])), flush=True)
            continue
        n = min(int(cmd[1]), len(idx))
        meta = A.mk_meta(w, h, subsamp)
        enc = A.ENCODER()
        configure_encoder(ref, enc, meta, qp=qp, gop=gop, effort=effort)
        bufs = (A.BUF * 4)()
        per_frame = []
        packets = []
        c0 = time.process_time()
        t0 = time.time()
        for t in range(n):
            arr = cache[idx[t]]
            fr = ref.dsv_load_planar_frame(subsamp, arr.ctypes.data, w, h)
            nb = ref.dsv_enc(C.byref(enc), fr, bufs)
            pks = []
            for i in range(nb):
                pks.append(C.string_at(bufs[i].data, bufs[i].len))
                ref.dsv_buf_free(C.byref(bufs[i]))
            packets.append(pks)
            per_frame.append(b"".join(pks))
        t1 = time.time()
        c1 = time.process_time()
        ref.dsv_enc_free(C.byref(enc))
        with open(out_path, "wb") as f:
            for pk in per_frame:
                f.write(struct.pack("<I", len(pk)))
                f.write(pk)
        print(json.dumps({"frames": n, "t0": t0, "t1": t1, "cpu_s": c1 - c0}), flush=True)


def decode(ref, A, C, np, packets):
    """the reference decoder (dsv_decoder.c:394) over the packets of the last encode; hashing happens outside the clock"""
    import hashlib
    dec = A.DECODER()
    pics = []
    t0 = time.time()
    for pks in packets:
        for pk in pks:
            buf = A.BUF()
            ref.dsv_mk_buf(C.byref(buf), len(pk) + 64)
            C.memmove(buf.data, pk, len(pk))
            buf.len = len(pk)
            fp = C.POINTER(A.FRAME)()
            fn = C.c_uint32(0)
            code = ref.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
            if code == A.DEC_OK and fp:
                pics.append(fp)
    t1 = time.time()
    md5 = []
    for fp in pics:
        h = hashlib.md5()
        for c in range(3):
            p = fp.contents.planes[c]
            a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,)).reshape(p.h, p.stride)[:, :p.w]
            h.update(np.ascontiguousarray(a).tobytes())
        md5.append(h.hexdigest())
        ref.dsv_frame_ref_dec(fp)
    ref.dsv_dec_free(C.byref(dec))
    return {"frames": len(md5), "t0": t0, "t1": t1, "md5": md5}


if __name__ == "__main__":
    main()
