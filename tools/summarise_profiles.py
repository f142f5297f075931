#!/usr/bin/env python3
"""Turns the raw output of tools/profile_round.sh (gpurun_out/prof) into the committed summaries under profiles/.

usage: tools/summarise_profiles.py r01   (prefix for the file names)
Writes: profiles/<p>_bench.json, <p>_rocprof_kernel_stats.txt, <p>_pmc_hme.txt and profiles/pmc_traffic.json
(the latter is what bench.py reports as roofline.traffic for the matching configuration)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
prefix = sys.argv[1] if len(sys.argv) > 1 else "r01"

if os.path.exists(os.path.join(src, "bench.json")):  # the default bench line, exactly as bench.py printed it
    bench = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
    json.dump(bench, open(os.path.join(dst, prefix + "_bench.json"), "w"), indent=1)
traced = json.loads([l for l in open(os.path.join(src, "bench_traced.json")) if l.startswith("{")][-1])

rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
total = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, prefix + "_rocprof_kernel_stats.txt"), "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras   (MI355X, %d streams / %d groups)\n"
            % (traced["config"]["streams_per_gpu"], traced["config"]["groups"]))
    f.write("bench line under the profiler: %.1f frames/s, %.2f ms/step; sum of kernel durations %.1f ms\n"
            % (traced["value"], traced["ms_per_step"], total / 1e6))
    f.write("(durations of concurrently running kernels overlap: the groups share the GPU)\n\n")
    f.write("%-78s %8s %12s %12s %7s\n" % ("kernel", "calls", "total ms", "avg us", "%"))
    for r in rows[:40]:
        f.write("%-78s %8s %12.2f %12.2f %7.2f\n" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    # cross-check of the bench's HIP-event figure for the dominant kernel
    tr = list(csv.DictReader(open(os.path.join(src, "kernel_trace_hme.csv"))))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
    gmax = max(int(r["Grid_Size_X"]) for r in tr)
    full = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr if int(r["Grid_Size_X"]) >= gmax - 6 * 64]
    f.write("\nk_hme_rows_b_fast_l0_pre_w2 (level-0 search, the dominant kernel): %d launches, mean %.1f us over all of them (the pre-roll steps "
            "launch it for fewer streams), mean %.1f us over the %d full-size launches (up to %d inter pictures of a group) -- bench.py HIP-event span of that "
            "launch in its profiled steps: %.1f us (the event span also holds the wait for a free queue slot between the group's "
            "launches while the other three groups' kernels are being dispatched)\n"
            % (len(durs), sum(durs) / max(1, len(durs)), sum(full) / max(1, len(full)), len(full), gmax // 64,
               traced.get("roofline", {}).get("avg_launch_us", float("nan"))))

# PMC: per-launch HBM-side bytes of the dominant kernel
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rs = [r for r in csv.DictReader(open(os.path.join(src, "pmc_%s.csv" % c))) if r["Counter_Name"] == c]
    by_grid = collections.defaultdict(list)
    for r in rs:
        by_grid[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    agg[c] = by_grid
with open(os.path.join(dst, prefix + "_pmc_hme.txt"), "w") as f:
    f.write("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-include-regex k_hme_rows_b_fast_l0 -- python3 bench.py "
            "--steps 6 --warmup 3 (the bench's own stream layout: staggered GOP phases, phase-aligned groups)\nunits: KiB per launch as reported; narrow (2-byte per lane) loads, so the gfx950 "
            "half-count correction for 16-byte streaming reads is NOT applied (uncalibrated width); only the launches of the "
            "level-0 search kernel are listed (one lockstep group per launch)\n\n")
    f.write("%12s %8s %16s %16s\n" % ("grid size", "launches", "FETCH_SIZE KiB", "WRITE_SIZE KiB"))
    tot_f = tot_w = nl = 0
    gmax = max(agg["FETCH_SIZE"])
    for g in sorted(agg["FETCH_SIZE"]):
        fv, wv = agg["FETCH_SIZE"][g], agg["WRITE_SIZE"].get(g, [0.0])
        f.write("%12d %8d %16.1f %16.1f\n" % (g, len(fv), sum(fv) / len(fv), sum(wv) / len(wv)))
        if g >= 0.9 * gmax:  # the full-size launches (every stream of the group started): what the bench's timed steps launch
            tot_f += sum(fv)
            tot_w += sum(wv) * len(fv) / len(wv)
            nl += len(fv)
    bytes_per_launch = (tot_f + tot_w) * 1024.0 / max(1, nl)
    f.write("\nmean over the %d full-size launches (grid >= 0.9 x %d: 176 - 192 inter pictures of a group): %.2f MB fetched + written per launch\n"
            % (nl, gmax, bytes_per_launch / 1e6))
json.dump({"stage": "hme_level0", "kernel": "k_hme_rows_b_fast_l0_pre_w2", "streams_per_gpu": traced["config"]["streams_per_gpu"],
           "groups": traced["config"]["groups"], "stagger": True, "phase_aligned": bool(traced["config"].get("phase_aligned_groups")),
           "bytes_per_launch": round(bytes_per_launch),
           "source": "profiles/%s_pmc_hme.txt (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)" % prefix},
          open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "decode.json")):  # (since round 2 the decode leg is part of the bench line itself)
    dec = [l for l in open(os.path.join(src, "decode.json")) if l.startswith("{")]
    if dec:
        json.dump(json.loads(dec[-1]), open(os.path.join(dst, prefix + "_decode.json"), "w"), indent=1)
# (the committed bench line is exactly what bench.py printed: its roofline.traffic comes from profiles/pmc_traffic.json of
# the passes above when bench.py runs AFTER they have been summarised and their configuration matches)
