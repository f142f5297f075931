#!/usr/bin/env python3
"""Frames/s of the batch encoder at one geometry (experiments): tools/probe/geometry_fps.py W H [streams] [steps]
Same engine and timed region as bench.py's legs (benchparts.common.EncodeRun); twins compared, no reference re-encode."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from benchparts.common import EncodeRun, bind_abi, gen_videos, timed_leg, under_profiler_  # noqa: E402


def main():
    w, h = int(sys.argv[1]), int(sys.argv[2])
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 192
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    # (before the GPU runtime starts: forked generators -- and none under the profiler, whose preloaded library has started it already)
    vids = gen_videos([(w, h, "420", 401 + i, 16) for i in range(4)], 1 if under_profiler_() else 4)
    import torch
    import dsvabi as A
    hip = A.load_hip()
    assert hip.dsv2hip_device_ok() == 0
    hip.dsv2hip_set_device(0)
    bind_abi(hip, A)
    run = EncodeRun(hip, A, torch, w, h, "420", 60, 48, 10, S, 4, vids, False, seeds=[401 + i for i in range(4)])
    f, e, _ = timed_leg(run, 2, k)
    p, b = run.twins_equal()
    print("%dx%d: %d streams, %.1f frames/s, %.1f Mpix/s, %.2f ms per step, twins %d/%d equal" % (w, h, S, f / e, f / e * w * h / 1e6, 1e3 * e / k, p - b, p))
    run.free()


if __name__ == "__main__":
    main()
