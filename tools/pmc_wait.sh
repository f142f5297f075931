#!/bin/bash
# usage (GPU box, via gpurun): tools/pmc_wait.sh [kernel-regex]  -- where the wavefronts of the search kernels spend their cycles
# (SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY), one lockstep group alone
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
re=${1:-k_hme}
out=gpurun_out/pmc_wait
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex "$re" --output-format csv -d $out/raw -- python3 bench.py --steps 4 --warmup 2 --streams ${STREAMS:-96} --groups 1 --gen-procs 1 --no-stagger --no-extras --no-cpu-baseline --no-profile --no-mix > /dev/null 2> $out/err.txt
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_wait/raw/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("dsv2::", "").replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            n[k] += 1
with open("gpurun_out/pmc_wait/summary.txt", "w") as o:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = v.get("SQ_WAVE_CYCLES", 1) or 1
        line = "%-28s launches %4d  wave-cycles %.3e  parked %.1f %%  issue-stall %.1f %% (of which LDS %.1f %%)  issuing %.1f %%  waves %.3e  smem %.3e  busy %.3e" % (
            k[:28], n[k], wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc,
            100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_WAVES", 0), v.get("SQ_INSTS_SMEM", 0), v.get("SQ_BUSY_CYCLES", 0))
        print(line); o.write(line + "\n")
PY
rm -rf $out/raw
