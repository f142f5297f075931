#!/bin/bash
# usage (GPU box): tools/pair_ab.sh  -- lane-pair luma sweep against the lane-per-cell one: S = 1 / 8 / 48 and the headline
cd "$GRAFT_REPO_ROOT"
for cfg in "1 1" "8 4" "48 4"; do
  set -- $cfg
  for e in DSV2_FILTER_PAIR_MAX=0 DSV2_FILTER_PAIR_MAX=64; do
    env $e timeout 300 python3 bench.py --streams $1 --groups $2 --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$e] streams $1 groups $2:', j['value'],'fps', j['ms_per_step'],'ms/step')"
  done
done
tools/ab_env.sh DSV2_FILTER_PAIR_MAX=0 DSV2_FILTER_PAIR_MAX=64 DSV2_FILTER_PAIR_MAX=100000 2>&1 | cut -c1-330
