#!/usr/bin/env python3
"""Filters DSV2_TRACE=10 lines (stdin): prints the steps whose device or host phases stand out."""
import re
import sys

for ln in sys.stdin:
    m = re.search(r"g1 enqueue ([\d.]+) wait ([\d.]+) \| h1 ([\d.]+) \| g2 enqueue ([\d.]+) wait ([\d.]+) \| syms ([\d.]+) \| h2 ([\d.]+)", ln)
    if not m:
        continue
    v = [float(x) for x in m.groups()]
    if v[4] > float(sys.argv[1]) or v[6] > float(sys.argv[2]):
        print(ln.strip()[:180])
