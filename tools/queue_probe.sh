#!/bin/bash
# usage (GPU box): tools/queue_probe.sh -- do more lockstep groups than four help small batches once they have hardware queues of their own?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/queue_probe.txt; : > $out
run() { # streams groups [ENV=VAL...]
  s=$1; g=$2; shift; shift
  echo "=== streams $s groups $g $*" >> $out
  env "$@" timeout 300 python3 bench.py --streams $s --groups $g --steps 48 --warmup 4 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'],'fps', j['ms_per_step'],'ms/step p50',j['config'].get('ms_per_frame_p50'),'cores',j['config']['host_cpu_cores_busy'])" >> $out
}
run 8 8 GPU_MAX_HW_QUEUES=8
run 8 8 GPU_MAX_HW_QUEUES=16
run 8 8 GPU_MAX_HW_QUEUES=24
run 8 4 GPU_MAX_HW_QUEUES=16
run 48 8 GPU_MAX_HW_QUEUES=16
run 48 6 GPU_MAX_HW_QUEUES=16
run 48 4 GPU_MAX_HW_QUEUES=16
run 16 8 GPU_MAX_HW_QUEUES=16
cat $out
