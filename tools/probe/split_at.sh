#!/bin/bash
# level 0 in place (DSV2_HME_SPLIT=0) against the split form (1) at a given number of streams (4 groups)
S=${1:-192}; N=${N:-2}
for r in $(seq 1 $N); do
    for V in 0 1; do
        DSV2_HME_SPLIT=$V python bench.py --no-extras --no-cpu-baseline --no-profile --streams $S --steps 96 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('streams $S split $V  %8.1f fps' % d['value'], flush=True)"
    done
done
