#!/bin/bash
# usage (GPU box): tools/pmc_all.sh <tag> COUNTER [COUNTER...]  -- the counters of EVERY kernel of a short run (one group of 32
# streams, one intra + a few inter pictures), summed per kernel and divided by the frames encoded
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 4 --warmup 2 --streams 32 --groups 1 --gen-procs 1 --no-stagger --no-extras --no-cpu-baseline --no-profile --no-mix > /dev/null 2>&1
python3 - "$tag" <<'PY' | tee gpurun_out/pmc_all_$tag.txt
import csv, glob, collections, sys
frames = 32 * 6
for d in sorted(glob.glob(f"gpurun_out/pmc_{sys.argv[1]}/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(d)):
        agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:]][r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for v in agg.values() for c in v})
    tot = collections.defaultdict(float)
    print("per frame (1 intra + 5 inter per stream), thousands:", names)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        print("  %-46s" % k, "  ".join("%9.1f" % (v.get(c, 0) / frames / 1e3) for c in names))
        for c in names:
            tot[c] += v.get(c, 0) / frames / 1e3
    print("  %-46s" % "TOTAL", "  ".join("%9.1f" % tot[c] for c in names))
    if "SQ_INSTS_VALU" in tot:  # the input of bench.py's roofline.issue
        import json
        json.dump({"vector_per_frame": round(tot["SQ_INSTS_VALU"] * 1e3), "scalar_per_frame": round(tot.get("SQ_INSTS_SALU", 0) * 1e3),
                   "mix": "1 intra + 5 inter pictures per stream, 32 streams in one lockstep group",
                   "source": "tools/pmc_all.sh %s SQ_INSTS_VALU SQ_INSTS_SALU (rocprofv3 --pmc, every kernel of the run summed)" % sys.argv[1]},
                  open("gpurun_out/instruction_volume_%s.json" % sys.argv[1], "w"))
PY
rm -rf gpurun_out/pmc_$tag
