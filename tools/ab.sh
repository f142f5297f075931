#!/bin/bash
# usage (GPU box): tools/ab.sh "ENV=a ENV2=b -- --flag x" "-- --groups 6" ...   -- one headline bench run per case (env settings, then bench arguments)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for c in "$@"; do
  e="${c%%--*}"; a="${c#*--}"; [ "$a" = "$c" ] && a=""
  env $e python3 bench.py --no-extras --no-cpu-baseline --no-profile --steps ${STEPS:-20} $a 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('[$c]', r['value'], 'fps', r['ms_per_step'], 'ms/step host cores', r['config']['host_cpu_cores_busy'])
"
done
