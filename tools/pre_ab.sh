#!/bin/bash
# usage (GPU box): tools/pre_ab.sh VAR  -- VAR=0 against VAR=1: S = 1 / 8 / 48 and the headline
cd "$GRAFT_REPO_ROOT"
v=$1
for cfg in "1 1" "8 4" "48 4"; do
  set -- $cfg
  for e in $v=0 $v=1; do
    env $e timeout 300 python3 bench.py --streams $1 --groups $2 --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$e] streams $1 groups $2:', j['value'],'fps', j['ms_per_step'],'ms/step')"
  done
done
tools/ab_env.sh $v=0 $v=1 $v=0 $v=1 2>&1 | cut -c1-330
