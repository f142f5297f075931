#!/usr/bin/env python3
import json, sys
d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", "fps", d["value"], "ms/step", d["ms_per_step"], "cores busy", d["config"].get("host_cpu_cores_busy"))
