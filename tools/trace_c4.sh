#!/bin/bash
# usage (GPU box): tools/trace_c4.sh -- kernel totals of BASELINE config 4 (1080p 4:4:4 lossless, 128 streams / 4 groups) per frame
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c4
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4/t -- python3 tools/ab_c4_run.py > gpurun_out/c4/run.json 2> gpurun_out/c4/run.err
cp gpurun_out/c4/t/*/*_kernel_stats.csv gpurun_out/c4/stats.csv; rm -rf gpurun_out/c4/t
python3 - <<'PY'
import csv, json
r = json.loads([l for l in open("gpurun_out/c4/run.json") if l.startswith("{")][-1])
print(r)
rows = list(csv.DictReader(open("gpurun_out/c4/stats.csv")))
frames = r["frames"]
tot = sum(float(x["TotalDurationNs"]) for x in rows)
print("sum of kernel durations %.1f us per frame" % (tot / 1e3 / frames))
for x in rows[:22]:
    print("  %-62s %6s calls %9.1f us avg %8.2f us/frame" % (x["Name"][:62], x["Calls"], float(x["AverageNs"]) / 1e3, float(x["TotalDurationNs"]) / 1e3 / frames))
PY
