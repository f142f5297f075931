#!/bin/bash
# usage (GPU box): tools/small_ab.sh "ENV=..." ... -- S=8 / G=4 and S=48 / G=4 under different environments
cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  for cfg in "8 4" "48 4"; do
    set -- $cfg
    env $e timeout 300 python3 bench.py --streams $1 --groups $2 --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$e] streams $1 groups $2:', j['value'],'fps', j['ms_per_step'],'ms/step')"
  done
done
