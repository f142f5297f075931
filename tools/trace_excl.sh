#!/bin/bash
# usage (GPU box): tools/trace_excl.sh <tag> [ENV=VAL ...]  -- exclusive kernel costs: ONE group of 96 streams (nothing shares the
# GPU with a kernel while it runs), kernel trace, total duration of every kernel divided by the frames encoded = us per frame
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ts
env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts/$tag -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --no-stagger --no-mix --streams ${EXCL_STREAMS:-96} --groups 1 --steps 12 --warmup 2 > gpurun_out/ts/$tag.json 2> gpurun_out/ts/$tag.err
cp gpurun_out/ts/$tag/*/*_kernel_stats.csv gpurun_out/ts/$tag.csv
rm -rf gpurun_out/ts/$tag
python3 - "$tag" <<'PY'
import csv, json, sys
t = sys.argv[1]
r = json.loads([l for l in open(f"gpurun_out/ts/{t}.json") if l.startswith("{")][-1])
frames = r["config"]["streams_per_gpu"] * (r["steps"] + r["warmup"])
print(t, r["value"], "fps", r["ms_per_step"], "ms/step;", frames, "frames traced (1 intra + 13 inter per stream)")
rows = list(csv.DictReader(open(f"gpurun_out/ts/{t}.csv")))
tot = sum(float(x["TotalDurationNs"]) for x in rows)
print("  sum of kernel durations %.1f us per frame" % (tot / 1e3 / frames))
for x in rows[:30]:
    print("  %-62s %6s calls %8.1f us avg %7.2f us/frame" % (x["Name"][:62], x["Calls"], float(x["AverageNs"]) / 1e3, float(x["TotalDurationNs"]) / 1e3 / frames))
PY
