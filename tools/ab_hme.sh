#!/bin/bash
# A/B of motion-search builds and occupancy caps on the GPU box: prints frames/s and the ME stage span per variant
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for lib in libdsv2hip.so libdsv2hip_winlds.so; do
  for w in 2 3 4; do
    for cfg in "--streams 384 --groups 4" "--streams 128 --groups 1"; do
      DSV2HIP_LIB=$PWD/digital-subband-video-2_amd/$lib DSV2_HME_WAVES_FAST=$w python3 bench.py --no-extras --no-cpu-baseline --steps 24 $cfg 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('$lib W=$w $cfg', r['value'], 'fps', r['ms_per_step'], 'ms/step  hme us/frame', r['roofline']['stage_us_per_frame']['hme'], 'launch us', r['roofline']['avg_launch_us'])
"
    done
  done
done
