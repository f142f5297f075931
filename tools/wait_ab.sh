#!/bin/bash
# usage (GPU box): tools/wait_ab.sh  -- fine (20 us) against sparse host polling at 1 / 8 / 48 streams
cd "$GRAFT_REPO_ROOT"
for cfg in "1 1" "8 4" "48 4"; do
  set -- $cfg
  for e in DSV2_WAIT_FINE_MAX=0 DSV2_WAIT_FINE_MAX=16 DSV2_WAIT_FINE_MAX=0 DSV2_WAIT_FINE_MAX=16; do
    env $e timeout 300 python3 bench.py --streams $1 --groups $2 --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$e] streams $1 groups $2:', j['value'],'fps', j['ms_per_step'],'ms/step, host cores', j['config']['host_cpu_cores_busy'])"
  done
done
