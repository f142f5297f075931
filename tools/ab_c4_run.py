#!/usr/bin/env python3
"""BASELINE config 4 alone (1080p 4:4:4 lossless, 128 streams in 4 lockstep groups, frames 0 .. 9 of each stream): one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402

vids = bench.gen_videos([(1920, 1080, "444", 201, 12)], 1)
import torch  # noqa: E402
import dsvabi as A  # noqa: E402

hip = A.load_hip()
assert hip.dsv2hip_device_ok() == 0
bench.bind_abi(hip, A)
run = bench.EncodeRun(hip, A, torch, 1920, 1080, "444", 100, 60, 10, 128, 4, vids, False, seeds=[201])
run.run(2)
k = 8
e = run.run(k)
print(json.dumps({"value": round(128 * k / e, 2), "unit": "frames/s", "frames": 128 * (k + 2), "ms_per_step": round(1e3 * e / k, 2)}))
run.free()
