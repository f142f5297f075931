#!/usr/bin/env python3
"""Where the GPU sits idle during the headline: reads a rocprofv3 kernel trace (csv) of `bench.py --no-extras` and, for every
interval of the timed region in which NO kernel runs, notes for each hardware queue which kernel it ran last and which it runs
next -- the (last, next) pair names the host phase that lockstep group is in.

usage: tools/idle_analysis.py <kernel_trace.csv> [first_fraction]   (timed region = the last 1 - first_fraction of the search span)"""
import collections
import csv
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dsv2::", "")[:36]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")))
    rows.sort()
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    l0 = [r for r in rows if "k_hme_rows_b_fast_l0" in r[2]]
    t_a, t_b = l0[0][0], l0[-1][1]
    lo, hi = t_a + int((t_b - t_a) * frac), t_b
    if frac >= 1:  # the last `frac` level-0 launches (timed steps x groups, when nothing runs behind the headline)
        lo = l0[-int(frac)][0]
    span = hi - lo
    queues = sorted({r[3] for r in rows})
    perq = {q: [r for r in rows if r[3] == q] for q in queues}
    # idle intervals
    ev = []
    for s, e, n, q in rows:
        s2, e2 = max(s, lo), min(e, hi)
        if e2 > s2:
            ev.append((s2, 1))
            ev.append((e2, -1))
    ev.sort(key=lambda x: (x[0], -x[1]))
    idle = []
    cur = 0
    prev = lo
    for t, d in ev:
        if cur == 0 and t > prev:
            idle.append((prev, t))
        cur += d
        prev = t
    tot_idle = sum(b - a for a, b in idle)
    print("timed region %.1f ms, %d queues, idle (no kernel running) %.1f ms = %.1f %%" % (span / 1e6, len(queues), tot_idle / 1e6, 100.0 * tot_idle / span))
    hist = collections.Counter()
    for a, b in idle:
        d = (b - a) / 1e3
        k = "<5us" if d < 5 else "<20us" if d < 20 else "<100us" if d < 100 else "<500us" if d < 500 else "<2ms" if d < 2000 else ">=2ms"
        hist[k] += b - a
    print("  idle time by gap length: " + "  ".join("%s %.1f%%" % (k, 100.0 * hist[k] / max(1, tot_idle)) for k in ("<5us", "<20us", "<100us", "<500us", "<2ms", ">=2ms")))
    # what every queue is between, during idle
    import bisect
    starts = {q: [r[0] for r in perq[q]] for q in queues}
    state = collections.Counter()
    combo = collections.Counter()
    for a, b in idle:
        mid = (a + b) // 2
        st = []
        for q in queues:
            i = bisect.bisect_right(starts[q], mid) - 1
            last = perq[q][i][2] if i >= 0 else "-"
            nxt = perq[q][i + 1][2] if i + 1 < len(perq[q]) else "-"
            # a queue whose last kernel is still running cannot be (idle = nothing runs); so it is between the two
            wait_ms = (perq[q][i + 1][0] - perq[q][i][1]) / 1e6 if 0 <= i and i + 1 < len(perq[q]) else 0
            if wait_ms < 0.05:
                key = "(dispatch gap)"
            else:
                key = "%s -> %s" % (last, nxt)
            st.append(key)
            state[key] += (b - a)
        combo[" | ".join(sorted(st))] += b - a
    print("  idle-time-weighted state of the queues (sums to #queues x 100 %):")
    for k, v in state.most_common(24):
        print("    %6.1f %%  %s" % (100.0 * v / max(1, tot_idle), k))
    print("  most common joint states during idle:")
    for k, v in combo.most_common(10):
        print("    %6.1f %%  %s" % (100.0 * v / max(1, tot_idle), k))
    # per-queue: how long each host phase (gap > 50 us between consecutive kernels of a queue) lasts on average in the region
    print("  gaps > 50 us between consecutive kernels of one queue (per queue and step these are the host phases), mean ms x count:")
    gaps = collections.defaultdict(list)
    for q in queues:
        rs = perq[q]
        for i in range(len(rs) - 1):
            if rs[i][1] >= lo and rs[i + 1][0] <= hi and rs[i + 1][0] - rs[i][1] > 50000:
                gaps["%s -> %s" % (rs[i][2], rs[i + 1][2])].append((rs[i + 1][0] - rs[i][1]) / 1e6)
    for k, v in sorted(gaps.items(), key=lambda x: -sum(x[1]))[:20]:
        print("    %8.3f ms x %4d = %8.1f ms   %s" % (sum(v) / len(v), len(v), sum(v), k))


main()


def timeline(path, frac=0.7, window_ms=230.0):
    """per queue: runs of back-to-back kernels (gaps < 50 us merged) inside a window of the timed region"""
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")))
    rows.sort()
    l0 = [r for r in rows if "k_hme_rows_b_fast_l0" in r[2]]
    t_a, t_b = l0[0][0], l0[-1][1]
    lo = t_a + int((t_b - t_a) * frac)
    if frac >= 1:
        lo = l0[-int(frac)][0]
    hi = lo + int(window_ms * 1e6)
    import glob, os
    for mc in glob.glob(os.path.join(os.path.dirname(path), "*_memory_copy_trace.csv")):
        for r in csv.DictReader(open(mc)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?"), "copy-" + r.get("Direction", "?")))
    rows.sort()
    print("\ntimeline of %.0f ms (t = 0 at %.0f %% of the search span); per queue: [start .. end ms] first kernel .. last kernel (count)" % (window_ms, 100 * frac))
    for q in sorted({r[3] for r in rows}):
        rs = [r for r in rows if r[3] == q and r[1] >= lo and r[0] <= hi]
        if not rs:
            continue
        print(" queue %s" % q)
        runs = []
        for s, e, n, _ in rs:
            if runs and s - runs[-1][1] < 50000:
                runs[-1][1] = max(runs[-1][1], e)
                runs[-1][3] = n
                runs[-1][4] += 1
                if e - s > runs[-1][6]:
                    runs[-1][5], runs[-1][6] = n, e - s
            else:
                runs.append([s, e, n, n, 1, n, e - s])
        for s, e, a, b, c, big, bigd in runs:
            print("   [%8.2f .. %8.2f] %7.2f ms  %s .. %s (%d; longest %s %.2f ms)" % ((s - lo) / 1e6, (e - lo) / 1e6, (e - s) / 1e6, a, b, c, big, bigd / 1e6))


if len(sys.argv) > 3:
    timeline(sys.argv[1], float(sys.argv[3]))
