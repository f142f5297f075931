#!/bin/bash
# usage (GPU box): tools/idle_trace.sh <tag> [ENV=VAL ...] -- kernel trace of the headline (12 steps) + tools/idle_analysis.py over it + the
# host-side phase split of every lockstep group (DSV2_TRACE=2)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/idle
export DSV2_TRACE=${DSV2_TRACE:-2}
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/idle/$tag -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --steps 12 --warmup 4 > gpurun_out/idle/$tag.json 2> gpurun_out/idle/$tag.err
python3 tools/idle_analysis.py gpurun_out/idle/$tag/*/*_kernel_trace.csv 44 24 > gpurun_out/idle/${tag}_idle.txt
head -30 gpurun_out/idle/${tag}_idle.txt
grep "^{" gpurun_out/idle/$tag.json | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
grep "batch n=" gpurun_out/idle/$tag.err | tail -8
grep "^\[t " gpurun_out/idle/$tag.err | tail -400 > gpurun_out/idle/${tag}_marks.txt
gzip -c gpurun_out/idle/$tag/*/*_kernel_trace.csv > gpurun_out/idle/${tag}_kernel_trace.csv.gz
gzip -c gpurun_out/idle/$tag/*/*_memory_copy_trace.csv > gpurun_out/idle/${tag}_memory_copy_trace.csv.gz
rm -rf gpurun_out/idle/$tag
