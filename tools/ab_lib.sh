#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <lib_a.so> <lib_b.so> [bench args...]  -- the headline (no extras) with two builds of the library,
# alternating A B A B on the same box
a=$1; b=$2; shift; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
  for lib in $a $b; do
    DSV2HIP_LIB=$PWD/$lib timeout 400 python3 bench.py --no-extras --no-cpu-baseline --no-profile --steps 24 --warmup 4 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$lib', j['value'],'fps', j['ms_per_step'],'ms/step cores',j['config']['host_cpu_cores_busy'], 'twins', j['parity_checked']['twin_pairs_equal'])"
  done
done
