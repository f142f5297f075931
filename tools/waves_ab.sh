#!/bin/bash
# usage (GPU box): tools/waves_ab.sh  -- search wavefronts per SIMD (2 / 3 / 4) at 8, 48 and 192 streams in 4 groups
cd "$GRAFT_REPO_ROOT"
for cfg in "8 4" "48 4" "192 4"; do
  set -- $cfg
  for w in 2 3 4; do
    env DSV2_HME_WAVES_FAST=$w DSV2_HME_WAVES_FAST_LX=$w timeout 300 python3 bench.py --streams $1 --groups $2 --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[W=$w] streams $1 groups $2:', j['value'],'fps', j['ms_per_step'],'ms/step')"
  done
done
