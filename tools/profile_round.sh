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
# how the four lockstep groups share the GPU over the timed region: concurrency of kernels, time with / without a search
python3 - $out/trace/*/*_kernel_trace.csv > $out/concurrency.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# timed region of the headline: the last 60 % of the span between the first and the last level-0 search launch
l0 = [r for r in rows if "k_hme_rows_b_fast_l0" in r[2]]
t_a, t_b = l0[0][0], l0[-1][1]
lo, hi = t_a + (t_b - t_a) * 4 // 10, t_b
ev = []
for s, e, n in rows:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        k = 1 if "k_hme_rows" in n else 0
        ev.append((s, 1, k))
        ev.append((e, -1, -k))
ev.sort()
span = hi - lo
conc = {}
busy = srch = both = 0
cur = cs = 0
prev = lo
for t, d, k in ev:
    dt = t - prev
    if dt > 0:
        conc[min(cur, 8)] = conc.get(min(cur, 8), 0) + dt
        if cur:
            busy += dt
        if cs:
            srch += dt
        if cs and cur > cs:
            both += dt
    cur += d
    cs += k
    prev = t
print("kernel concurrency over %.1f ms of the headline's timed region (all four groups' streams):" % (span / 1e6))
print("  some kernel running %.1f %%, a search launch running %.1f %%, a search launch AND another kernel %.1f %%" % (100.0 * busy / span, 100.0 * srch / span, 100.0 * both / span))
print("  kernels running at once: " + "  ".join("%d%s: %.1f %%" % (c, "+" if c == 8 else "", 100.0 * conc.get(c, 0) / span) for c in range(0, 9)))
by = {}
for s, e, n in rows:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        nm = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dsv2::", "")[:40]
        by[nm] = by.get(nm, 0) + (e - s)
print("  kernel time summed over streams, %% of the region (can exceed 100 in total):")
for nm, v in sorted(by.items(), key=lambda x: -x[1])[:14]:
    print("    %-42s %6.1f %%" % (nm, 100.0 * v / span))
PY
cat $out/concurrency.txt
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
