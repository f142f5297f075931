#!/bin/bash
# usage (on the GPU box, repo root): tools/pmc_hme.sh <tag> COUNTER [COUNTER...]   -- one pass, level-0 search kernel only
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 420 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_hme_rows_b_fast_l0" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 2 --streams ${PMC_STREAMS:-64} --groups 1 --gen-procs 1 --no-stagger --no-extras --no-cpu-baseline --no-profile > /dev/null 2>&1
python3 - "$tag" <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob(f"gpurun_out/pmc_{sys.argv[1]}/*/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(d)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in rows:
        k = (r["Kernel_Name"][:34], r["Grid_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
    for k, v in agg.items():
        nl = n[k] / len(v)
        streams = int(k[1]) // (64 * 68)  # grid = streams x 68 block rows x 64 lanes
        print(k, int(nl), "per block:", {a: round(b / nl / (streams * 8160), 1) for a, b in v.items()})
PY
rm -rf gpurun_out/pmc_$tag
