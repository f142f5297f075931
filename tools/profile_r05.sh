#!/bin/bash
# Round-5 profile artefacts (GPU box, repo root; run through gpurun).  Raw output under gpurun_out/prof5/, summarised into
# profiles/r05_* by tools/summarise_r05.py.  Parts (pick with PARTS="trace pmc excl insts", default all):
#   trace  rocprofv3 --kernel-trace --stats of the headline command -> kernel_stats.csv + kernel_trace.csv.gz
#   pmc    FETCH_SIZE / WRITE_SIZE of the level-0 search launch in the bench's own layout (separate passes)
#   excl   one lockstep group of 96 streams alone: exclusive kernel durations
#   insts  SQ_INSTS_VALU / SQ_INSTS_SALU / LDS / VMEM of every kernel (32 streams, 1 intra + 5 inter)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof5
mkdir -p $out
parts=${PARTS:-trace pmc excl insts}
L0=k_hme_rows_l0
for p in $parts; do case $p in
trace)
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras > $out/bench_traced.json 2> $out/trace.err
    cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
    gzip -c $out/trace/*/*_kernel_trace.csv > $out/kernel_trace.csv.gz
    rm -rf $out/trace ;;
pmc)
    # (the figure is only valid for the search sources it was measured on: their hash is taken HERE, not when summarising)
    cat digital-subband-video-2_amd/csrc/{hme.hip,hme_fast.h,hme.h,blockstat.h,dev.h} | sha256sum | cut -c1-16 > $out/kernel_source_sha16.txt
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$L0" --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 6 --warmup 3 --gen-procs 1 --no-cpu-baseline --no-profile --no-extras > /dev/null 2> $out/pmc_$c.err
        cp $out/pmc_$c/*/*_counter_collection.csv $out/pmc_$c.csv
        rm -rf $out/pmc_$c
    done ;;
excl)
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/excl -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --no-stagger --no-mix --streams 96 --groups 1 --steps 12 --warmup 2 > $out/excl.json 2> $out/excl.err
    cp $out/excl/*/*_kernel_stats.csv $out/excl_kernel_stats.csv
    rm -rf $out/excl ;;
insts)
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/insts -- python3 bench.py --steps 4 --warmup 2 --streams 32 --groups 1 --gen-procs 1 --no-stagger --no-extras --no-cpu-baseline --no-profile --no-mix > /dev/null 2> $out/insts.err
    cp $out/insts/*/*_counter_collection.csv $out/insts.csv
    gzip -f $out/insts.csv
    rm -rf $out/insts ;;
esac; done
ls -la $out
