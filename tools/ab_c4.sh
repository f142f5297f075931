#!/bin/bash
# usage (GPU box): tools/ab_c4.sh "ENV=a" "ENV=b"  -- the other BASELINE configurations (C2 720p, C3 gop 60, C4 4:4:4 lossless) per environment
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  env $e python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --no-stagger 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('[$e]', r['value'], 'fps |', {k: v.get('value') for k, v in r.get('configs', {}).items()}, '| decode', r.get('decode', {}).get('value'))
"
done
