#!/bin/bash
# usage (GPU box): tools/pmc_filter.sh [streams]  -- what a front of the in-loop filter sweep costs: SQ counters of k_inter_filters_b
# per launch (1 stream by default: the single-stream critical path), divided by the 1 034 luma fronts of a 1080p plane
s=${1:-1}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES"; do
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "k_inter_filters_b" --output-format csv -d gpurun_out/pf -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --no-stagger --no-mix --streams $s --groups 1 --steps 8 --warmup 2 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pf/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-22s per launch %14.0f   per front (1034) %10.1f   (%d launches)" % (k, sum(v) / len(v), sum(v) / len(v) / 1034, len(v)))
PY
  rm -rf gpurun_out/pf
done
