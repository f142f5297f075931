#!/bin/bash
# usage (GPU box): tools/batch_probe.sh  -- small-batch operating points with the wall-clock phase split of a lockstep step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
out=gpurun_out/batch_probe.txt
: > $out
for cfg in ${PROBE_CFGS:-1:1 2:2 8:1 8:2 8:4 48:1 48:2 48:4 192:2 192:4}; do
  s=${cfg%:*}; g=${cfg#*:}
  echo "=== streams $s groups $g" >> $out
  DSV2_BATCH_TRACE=1 timeout 300 python3 bench.py --streams $s --groups $g --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile > /tmp/bp.log 2>&1
  grep "^\[batch n=" /tmp/bp.log | tail -2 >> $out
  grep -v "^\[batch" /tmp/bp.log | tail -2 | cut -c1-400 >> $out
done
cat $out
