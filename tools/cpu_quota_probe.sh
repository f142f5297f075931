#!/bin/bash
# usage: tools/cpu_quota_probe.sh <bench args...>  -- prints CPU time and CFS throttling of one bench.py run (GPU box)
s0=$(grep -E "^usage_usec|^nr_throttled|^throttled_usec|^nr_periods" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
t0=$(date +%s.%N)
python3 bench.py --no-cpu-baseline --no-profile --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('run', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', 'host cores busy', d['config'].get('host_cpu_cores_busy'))"
t1=$(date +%s.%N)
s1=$(grep -E "^usage_usec|^nr_throttled|^throttled_usec|^nr_periods" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
python3 - "$s0" "$s1" "$t0" "$t1" <<'PY'
import sys
a = [int(x) for x in sys.argv[1].split()]; b = [int(x) for x in sys.argv[2].split()]
wall = float(sys.argv[4]) - float(sys.argv[3])
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), "| wall %.1f s | cpu %.1f s (%.1f cores) | periods %d throttled %d | throttled thread-time %.1f s"
      % (wall, (b[0] - a[0]) / 1e6, (b[0] - a[0]) / 1e6 / wall, b[1] - a[1], b[2] - a[2], (b[3] - a[3]) / 1e6))
PY
