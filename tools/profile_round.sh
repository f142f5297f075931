#!/bin/bash
# A round's profile artefacts (GPU box, repo root; run through gpurun).  ROUND=<n> (default 6): raw output under gpurun_out/prof<n>/,
# summarised into profiles/r0<n>_* by tools/summarise_round.py <n>.  Parts (pick with PARTS="trace pmc excl insts occ census", default all):
#   trace  rocprofv3 --kernel-trace --stats of the headline command -> kernel_stats.csv + kernel_trace.csv.gz
#   pmc    FETCH_SIZE / WRITE_SIZE of the level-0 search launch in the bench's own layout (separate passes)
#   excl   one lockstep group of 96 streams alone: exclusive kernel durations
#   insts  SQ_INSTS_VALU / SQ_INSTS_SALU / LDS / VMEM of every kernel (32 streams, 1 intra + 5 inter)
#   occ    what the machine is short of: SQ wave / busy / issue counters of EVERY kernel in the headline's own four-group command
#          (rocprofv3 --pmc runs the dispatches one at a time: each kernel is seen in the headline's launch shape, ALONE on the chip),
#          joined by summarise_round.py with the un-serialised trace of part `trace` (the same kernels sharing the chip)
#   census resident wavefront-time per kernel measured INSIDE the kernels with all four groups running (the census build of the library)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof${ROUND:-6}
mkdir -p $out
parts=${PARTS:-trace pmc excl insts occ census}
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
occ)
    # (8 SQ slots, 2 GRBM slots; GRBM_GUI_ACTIVE = clocks summed over the 8 XCDs)
    timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/occ -- python3 bench.py --steps 4 --warmup 2 --gen-procs 1 --no-cpu-baseline --no-profile --no-extras > $out/occ.json 2> $out/occ.err
    cp $out/occ/*/*_counter_collection.csv $out/occ.csv; gzip -f $out/occ.csv
    gzip -c $out/occ/*/*_kernel_trace.csv > $out/occ_kernel_trace.csv.gz
    rm -rf $out/occ
    for o in digital-subband-video-2_amd/csrc/build/*.hip.o; do tools/codeobj.sh $o; done > $out/kernel_registers.txt 2>/dev/null ;;
census)
    # residency UNDER LOAD, measured inside the kernels (csrc/prio.h; `make -C digital-subband-video-2_amd/csrc census` beforehand): no profiler, nothing serialised
    DSV2HIP_LIB=$PWD/digital-subband-video-2_amd/libdsv2hip_census.so DSV2_CENSUS=1 timeout 900 python3 bench.py --no-cpu-baseline --no-extras --no-profile --steps 48 > $out/census.json 2> $out/census.err ;;
esac; done
ls -la $out
