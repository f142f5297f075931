#!/usr/bin/env python3
"""Where a lockstep group's step goes (wall clock): reads the absolute phase marks of DSV2_TRACE=6 (stderr of bench.py) and prints, per
group thread, the mean milliseconds per step between consecutive marks -- ingest + pyramids up to the drained stream, the wait for the
search token, the search (token held), the rest of G1, the host phase H1, G2 enqueued / waited for, H2 -- over the last `steps` steps.
usage: DSV2_TRACE=6 python3 bench.py --no-extras --no-cpu-baseline --no-profile --steps 24 2> marks.txt; python3 tools/step_phases.py marks.txt [steps]"""
import collections
import re
import sys

path, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
marks = collections.defaultdict(list)
for ln in open(path):
    m = re.match(r"\[t (0x[0-9a-f]+)\] ([0-9.]+) (\S+) n=(\d+)", ln)
    if m:
        marks[m.group(1)].append((float(m.group(2)), m.group(3)))
order = ["enter", "p0", "pre-search-drained", "token", "g1-enqueued", "token-released", "g1-done", "h1-done", "g2-enqueued", "h1b-done", "g2-done", "syms", "h2-done"]
tot = collections.defaultdict(float)
nsteps = 0
for th, ev in marks.items():
    steps, cur = [], []
    for t, name in ev:
        if name == "enter" and cur:
            steps.append(cur)
            cur = []
        cur.append((t, name))
    if cur:
        steps.append(cur)
    steps = [s for s in steps if s[-1][1] == "h2-done"][-last:]
    for s in steps:
        nsteps += 1
        for (t0, a), (t1, b) in zip(s, s[1:]):
            tot[a + " -> " + b] += t1 - t0
        tot["STEP (enter -> h2-done)"] += s[-1][0] - s[0][0]
print("mean ms per step over %d group-steps of %d group threads" % (nsteps, len(marks)))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-46s %8.2f" % (k, v / max(1, nsteps)))
