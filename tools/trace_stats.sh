#!/bin/bash
# usage (GPU box): tools/trace_stats.sh <tag> [ENV=VAL ...]  -- kernel-trace stats of the headline bench under the given environment
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ts
env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts/$tag -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --steps 24 > gpurun_out/ts/$tag.json 2> gpurun_out/ts/$tag.err
cp gpurun_out/ts/$tag/*/*_kernel_stats.csv gpurun_out/ts/$tag.csv
rm -rf gpurun_out/ts/$tag
python3 - "$tag" <<'PY'
import csv, json, sys
t = sys.argv[1]
r = json.loads([l for l in open(f"gpurun_out/ts/{t}.json") if l.startswith("{")][-1])
print(t, r["value"], "fps", r["ms_per_step"], "ms/step")
rows = list(csv.DictReader(open(f"gpurun_out/ts/{t}.csv")))
for x in rows[:22]:
    print("  %-60s %6s calls %9.1f ms total %9.1f us avg" % (x["Name"][:60], x["Calls"], float(x["TotalDurationNs"]) / 1e6, float(x["AverageNs"]) / 1e3))
PY
