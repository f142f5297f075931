#!/usr/bin/env python3
"""Debug aid: encode one golden entry repeatedly with the HIP library and report the first packet that
differs from the reference build (oracle/_ref), to chase timing-dependent mismatches.
usage: tools/repro_golden.py [entry] [runs]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import dsvabi as A  # noqa: E402
from codec_run import encode_stream  # noqa: E402
from conftest import load_pkg  # noqa: E402
from golden_common import GOLDEN, cli_equivalent_cfg  # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "c4_1080p_444_lossless").split(",")
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pkg = load_pkg()
hip = A.load_hip()
cases = []
for name in names:
    g = GOLDEN[name]
    v = pkg.synth.SynthVideo(g["w"], g["h"], g["fmt"], seed=g["seed"])
    frames = [v.frame_bytes(t) for t in range(g["n"])]
    cfg, eos = cli_equivalent_cfg(g["flags"])
    subsamp = A.SUBSAMP_420 if g["fmt"] == "420" else A.SUBSAMP_444
    ref, _ = encode_stream(A.load_ref(), frames, g["w"], g["h"], subsamp, eos=eos, **cfg)
    cases.append((name, g, frames, cfg, eos, subsamp, ref))
bad = {name: 0 for name in names}
for r in range(runs):
    for name, g, frames, cfg, eos, subsamp, ref in cases:  # entries alternate inside one process
        out, _ = encode_stream(hip, frames, g["w"], g["h"], subsamp, eos=eos, **cfg)
        diff = [i for i, (a, b) in enumerate(zip(ref, out)) if a != b]
        if diff or len(out) != len(ref):
            bad[name] += 1
            i = diff[0] if diff else min(len(out), len(ref))
            a, b = ref[i], out[i]
            first = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), min(len(a), len(b)))
            print(f"run {r} {name}: first differing packet {i} (type byte {a[4]:#x}), sizes {len(a)} vs {len(b)}, "
                  f"first byte {first}; {len(diff)} packets differ")
for name in names:
    print(f"{name}: {bad[name]}/{runs} runs differ")
