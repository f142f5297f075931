#!/bin/bash
# usage (GPU box): tools/clock_probe.sh  -- shader clock, power and temperature sampled twice a second while the headline bench runs
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( for i in $(seq 1 90); do echo "t=$i $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor junction\)|fclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';')"; sleep 0.5; done ) > gpurun_out/clock_probe.txt &
SMI=$!
python3 bench.py --no-extras --no-cpu-baseline --steps 48 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print(r['value'], 'fps', r['ms_per_step'], 'ms/step')"
kill $SMI 2>/dev/null
awk 'NR%6==1' gpurun_out/clock_probe.txt | head -30
