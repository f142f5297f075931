#!/usr/bin/env python3
"""Phase clocks of the in-loop filter's luma sweep (debugging build: make -C digital-subband-video-2_amd/csrc prof): what wave 0 of
every luma sweep spends per front at the column hand-over (vector-memory wait), issuing its global traffic, in the cell
routine, and at the LDS wait + barrier.  Default: one stream (the single-stream critical path)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_SO = os.path.join(ROOT, "digital-subband-video-2_amd", "libdsv2hip_prof.so")
os.environ["DSV2HIP_LIB"] = PROF_SO
import bench  # noqa: E402

args = sys.argv[1:] or ["--streams", "1", "--groups", "1", "--steps", "24", "--warmup", "2", "--no-stagger", "--no-mix"]
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-profile", "--no-extras"] + args
bench.main()
lib = ctypes.CDLL(PROF_SO)
out = (ctypes.c_ulonglong * 8)()
lib.dsv2hip_debug_filter_prof(out)
names = ["column hand-over (vmcnt wait + 4 LDS stores)", "retire + look-ahead + column fetch (issue)", "cell routine", "LDS wait + barrier"]
sweeps = float(out[4]) or 1.0
tot = float(sum(out[:4])) or 1.0
fronts = 1046.0  # nsbx + 2 + 2 (nsby - 1) + 13 at 1080p
for k, nm in enumerate(names):
    print("%d %-48s %6.2f %%  %9.0f ticks per sweep  %7.1f per front" % (k, nm, out[k] / tot * 100, out[k] / sweeps, out[k] / sweeps / fronts), file=sys.stderr)
print("sweeps %d, %.0f ticks per sweep (s_memtime ticks: the shader clock, ~2.2 GHz here)" % (out[4], tot / sweeps), file=sys.stderr)
cnt = (ctypes.c_ulonglong * 8)()
if hasattr(lib, "dsv2hip_debug_filter_counts"):
    lib.dsv2hip_debug_filter_counts(cnt)
    n = float(cnt[0]) or 1.0
    print("pair sweep, (wavefront, front) pairs with a cell: %d; any live cell %.3f, horizontal pass %.3f, vertical pass %.3f, sharpen %.3f; live cells per such pair %.1f of 32"
          % (cnt[0], cnt[1] / n, cnt[2] / n, cnt[3] / n, cnt[4] / n, cnt[5] / 2.0 / (float(cnt[1]) or 1.0)), file=sys.stderr)
