#!/bin/bash
# usage (GPU box): tools/waves_probe.sh -- wavefronts per SIMD of the search kernels against the batch size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "48 4" "192 4" "8 4"; do
  set -- $cfg
  for w in 2 3 4; do
    DSV2_HME_WAVES_FAST=$w timeout 300 python3 bench.py --streams $1 --groups $2 --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('streams $1 groups $2 W=$w:', j['value'],'fps', j['ms_per_step'],'ms/step')"
  done
done
