#!/bin/bash
# usage (GPU box): tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   -- one headline bench run per environment setting
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  env $e python3 bench.py --no-extras --no-cpu-baseline --steps 24 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('[$e]', r['value'], 'fps', r['ms_per_step'], 'ms/step host cores', r['config']['host_cpu_cores_busy'], r['roofline']['stage_us_per_frame'])
"
done
