#!/bin/bash
# usage (GPU box): tools/trace_timeline.sh <tag> [streams] [ENV=VAL ...] -- kernel timeline of ONE lockstep step (one group, default one
# stream): every launch of a steady-state P step with its start offset, duration and the idle gap in front of it
tag=$1; shift
streams=${1:-1}; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tl
env "$@" timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$tag -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --no-stagger --no-mix --streams $streams --groups 1 --steps 12 --warmup 2 > gpurun_out/tl/$tag.json 2> gpurun_out/tl/$tag.err
cp gpurun_out/tl/$tag/*/*_kernel_trace.csv gpurun_out/tl/$tag.csv
rm -rf gpurun_out/tl/$tag
python3 - "$tag" <<'PY' | tee gpurun_out/tl/${tag}_timeline.txt
import csv, json, sys
t = sys.argv[1]
r = json.loads([l for l in open(f"gpurun_out/tl/{t}.json") if l.startswith("{")][-1])
print(t, r["value"], "fps", r["ms_per_step"], "ms/step")
rows = sorted(csv.DictReader(open(f"gpurun_out/tl/{t}.csv")), key=lambda x: int(x["Start_Timestamp"]))
def nm(x):
    return x["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dsv2::", "")[:52]
# steps are delimited by the ingest kernel; take the third from the end
starts = [i for i, x in enumerate(rows) if nm(x).startswith("k_ingest")]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for x in rows[a:b]:
    s, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
    print("  %9.1f us  +%7.1f gap  %8.1f us  %-52s grid %s,%s,%s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, nm(x), x["Grid_Size_X"], x["Grid_Size_Y"], x["Grid_Size_Z"]))
    busy += e - s
    prev_end = max(prev_end, e)
print("  step span %.1f us (next step's first kernel at %.1f), kernels busy %.1f us, %d launches" % ((prev_end - t0) / 1e3, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
PY
