#!/bin/bash
# frames/s of the headline leg against the number of lockstep groups (host threads) per GPU, interleaved rounds
N=${N:-2}
for r in $(seq 1 $N); do
    for G in "$@"; do
        python bench.py --no-extras --no-cpu-baseline --no-profile --groups $G > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('groups $G  %8.1f fps  %s' % (d['value'], d['config'].get('streams_per_gpu', '')), flush=True)"
    done
done
