#!/bin/bash
# usage (GPU box): tools/mid_ab.sh "ENV=..." ... -- 192 and 384 streams in 4 groups under different environments
cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  for s in 192 384; do
    env $e timeout 300 python3 bench.py --streams $s --groups 4 --steps 32 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$e] streams $s:', j['value'],'fps', j['ms_per_step'],'ms/step')"
  done
done
