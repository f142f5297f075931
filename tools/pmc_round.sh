#!/bin/bash
# the two PMC passes of tools/profile_round.sh on their own (FETCH_SIZE / WRITE_SIZE of the level-0 search kernel)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_hme_rows_b_fast_l0" --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 6 --warmup 3 --gen-procs 1 --no-cpu-baseline --no-profile --no-extras > /dev/null 2> $out/pmc_$c.err
    cp $out/pmc_$c/*/*_counter_collection.csv $out/pmc_$c.csv
    rm -rf $out/pmc_$c
done
ls -la $out
