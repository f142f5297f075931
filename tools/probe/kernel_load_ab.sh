#!/bin/bash
# under-load durations of kernels matching a regex, for several builds of the library (kernel trace of the headline, un-serialised):
#   tools/probe/kernel_load_ab.sh "k_quant_level4|k_scatter" libdsv2hip_base.so libdsv2hip.so
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
re=$1; shift
for L in "$@"; do
    out=gpurun_out/klab; rm -rf $out; mkdir -p $out
    DSV2HIP_LIB=$PWD/digital-subband-video-2_amd/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --steps 32 > $out/b.json 2> $out/err.txt
    python3 - "$re" "$L" <<'PY'
import csv, glob, json, re, sys
rx = re.compile(sys.argv[1])
b = json.loads([l for l in open("gpurun_out/klab/b.json") if l.startswith("{")][-1])
print("%s: %.1f frames/s under the profiler" % (sys.argv[2], b["value"]))
for r in csv.DictReader(open(glob.glob("gpurun_out/klab/t/*/*_kernel_stats.csv")[0])):
    if rx.search(r["Name"]):
        print("   %-60s calls %6s avg %9.1f us  total %9.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
    rm -rf $out
done
