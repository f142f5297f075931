#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace CSV, divided by the number of steps (P-step launches of k_predict_w)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.defaultdict(float)
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsv2::", "")
    if "k_hme_rows" in k:
        k += " y=" + r.get("Grid_Size_Y", "?")
    tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[k] += 1
steps = max(1, max((c for k, c in cnt.items() if "k_predict_w" in k), default=1))
print("steps", steps, " total ms/step", round(sum(tot.values()) / steps / 1e3, 2))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{k[:60]:60s} {cnt[k]:6d} {v / steps:10.1f} us/step")
