#!/bin/bash
# usage (GPU box): tools/ab_args.sh "--flag a" "--flag b" ...   -- one headline bench run per argument string
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for a in "$@"; do
  python3 bench.py --no-extras --no-cpu-baseline --steps 24 $a 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('[$a]', r['value'], 'fps', r['ms_per_step'], 'ms/step host cores', r['config']['host_cpu_cores_busy'], r['roofline']['stage_us_per_frame'])
"
done
