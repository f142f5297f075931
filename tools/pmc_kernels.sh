#!/bin/bash
# usage (GPU box): tools/pmc_kernels.sh <tag> [streams]  -- HBM-side counter traffic of EVERY kernel: FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes (the TCC block cannot hold both) plus a plain kernel trace of the same
# command for the durations; one lockstep group (nothing shares the GPU with a kernel while it runs), 1 intra + 13 inter
# pictures per stream.  Summary: gpurun_out/pmc_kernels_<tag>.txt (tools/summarise_pmc_kernels.py).
set -u
tag=$1; streams=${2:-96}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pk_$tag; mkdir -p $out
args="--gen-procs 1 --no-cpu-baseline --no-extras --no-profile --no-stagger --no-mix --streams $streams --groups 1 --steps 12 --warmup 2"
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py $args > $out/bench.json 2> $out/trace.err
cp $out/trace/*/*_kernel_trace.csv $out/kernel_trace.csv; rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py $args > /dev/null 2> $out/pmc_$c.err
    cp $out/pmc_$c/*/*_counter_collection.csv $out/pmc_$c.csv; rm -rf $out/pmc_$c
done
python3 tools/summarise_pmc_kernels.py $out $streams 14 | tee gpurun_out/pmc_kernels_$tag.txt
rm -f $out/kernel_trace.csv $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv
