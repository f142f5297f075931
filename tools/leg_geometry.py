#!/usr/bin/env python3
"""One leg of the lockstep engine at another geometry, alone (experiments): tools/leg_geometry.py W H FMT STREAMS GROUPS [STEPS]
Prints frames/s; no reference check (bench.py's legs do that)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402

w, h, fmt, S, G = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
K = int(sys.argv[6]) if len(sys.argv) > 6 else 6
vids = bench.gen_videos([(w, h, fmt, 301 + k, 10) for k in range(2)], 2)
os.environ.setdefault("DSV2_HOST_THREADS", "16")
import torch  # noqa: E402
import dsvabi as A  # noqa: E402

torch.cuda.set_device(0)
hip = A.load_hip()
bench.bind_abi(hip, A)
run = bench.EncodeRun(hip, A, torch, w, h, fmt, 60, 48, 10, S, G, vids, False, seeds=[301, 302])
f, e, _ = bench.timed_leg(run, 2, K)
p, b = run.twins_equal()
print("%dx%d %s %d streams %d groups: %.1f frames/s, %.2f ms/step, twins %d/%d equal" % (w, h, fmt, S, G, f / e, 1e3 * e / K, p - b, p))
run.free()
