#!/bin/bash
# Produces the profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/bench.json              default bench line (with cpu baselines, parity, configs, decode)
#   gpurun_out/prof/kernel_stats.csv        rocprofv3 --kernel-trace --stats of the same command (headline part only)
#   gpurun_out/prof/pmc_{fetch,write}.csv   FETCH_SIZE / WRITE_SIZE of the dominant kernel (separate passes)
# Copy the summaries into profiles/ afterwards (tools/summarise_profiles.py <prefix>).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
mkdir -p $out
# (the default bench line itself is taken LAST, by a separate call, once the PMC passes below have been summarised into
# profiles/pmc_traffic.json: bench.py quotes that file as roofline.traffic)
if [ "${WITH_BENCH:-0}" = 1 ]; then python3 bench.py > $out/bench.json 2> $out/bench.err; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras > $out/bench_traced.json 2> $out/trace.err
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
# the trace itself is large: keep only the dominant kernel's rows for the duration cross-check
head -1 $out/trace/*/*_kernel_trace.csv > $out/kernel_trace_hme.csv
grep k_hme_rows_b_fast_l0 $out/trace/*/*_kernel_trace.csv >> $out/kernel_trace_hme.csv
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_hme_rows_b_fast_l0" --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 6 --warmup 3 --gen-procs 1 --no-cpu-baseline --no-profile --no-extras > /dev/null 2> $out/pmc_$c.err
    cp $out/pmc_$c/*/*_counter_collection.csv $out/pmc_$c.csv
    rm -rf $out/pmc_$c
done
ls -la $out
