#!/bin/bash
# kernel statistics of the batch encoder at one geometry: tools/probe/geometry_trace.sh W H [streams] [steps]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
rm -rf /tmp/geo_trace; mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/geo_trace -- python3 tools/probe/geometry_fps.py "$@" > gpurun_out/geo_trace.log 2>&1
tail -1 gpurun_out/geo_trace.log
f=$(find /tmp/geo_trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%6.2f%%  %6d calls  %9.1f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
