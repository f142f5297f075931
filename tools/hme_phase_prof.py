#!/usr/bin/env python3
"""Phase clocks of the level-0 motion search (debugging build: make -C digital-subband-video-2_amd/csrc prof).  Runs bench.py's loop in-process, then prints the share of each phase."""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_SO = os.path.join(ROOT, "digital-subband-video-2_amd", "libdsv2hip_prof.so")
os.environ["DSV2HIP_LIB"] = PROF_SO
import bench  # noqa: E402

sys.argv = ["bench.py", "--no-cpu-baseline", "--no-profile", "--no-extras"] + sys.argv[1:]
bench.main()
lib = ctypes.CDLL(PROF_SO)
out = (ctypes.c_ulonglong * 32)()
lib.dsv2hip_debug_hme_prof(out)
names = ["wait row above", "first load round", "list + scores", "best candidate + good-enough", "refinement",
         "sub-pel", "mode decision A", "intra sub-block luma", "intra sub-block chroma + rest", "store + publish"]
tot = float(sum(out[:10])) or 1.0
for k, nm in enumerate(names):
    print(f"{k} {nm:34s} {out[k] / tot * 100:6.2f} %  {out[k] / 1e9:9.3f} Gticks", file=sys.stderr)
print(f"total {tot / 1e9:.3f} Gticks", file=sys.stderr)
if out[15]:
    print(f"(inside candidate gather: parent average + inliers {out[15] / 1e9:.3f} Gticks = {out[15] / tot * 100:.2f} % of the above total)", file=sys.stderr)
blocks = float(out[10]) or 1.0
print(f"blocks {out[10]}  refined {out[11] / blocks:.3f}  refinement rounds/block {out[12] / blocks:.3f}  "
      f"sub-pel searches/block {out[13] / blocks:.3f}  vectors scored/block {out[14] / blocks:.2f}  ticks/block {tot / blocks:.0f}", file=sys.stderr)
print("per block: " + "  ".join(f"{nm} {out[k] / blocks:.3f}" for k, nm in [
    (23, "good-enough before the tail"), (21, "sub-pel pass 0"), (22, "sub-pel pass 1"), (27, "sub-pel vector"), (16, "skip test"), (17, "skipped"),
    (18, "sub-block metrics at the vector (round 3)"), (19, "intra test luma"), (20, "intra test chroma"), (24, "intra")]), file=sys.stderr)
