#!/usr/bin/env python3
"""Reads bench.py output on stdin and prints one compact line: label, streams, groups, value, ms/step."""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
line = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
if not line:
    print(label, "no result")
    sys.exit(0)
d = json.loads(line[-1])
c = d["config"]
rf = d.get("roofline") or {}
print(label, "streams", c.get("streams_per_gpu"), "groups", c.get("groups"), "fps", d["value"], "ms/step", d["ms_per_step"],
      "roofline", rf.get("kernel"), rf.get("frac"))
