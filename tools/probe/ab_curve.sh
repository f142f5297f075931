#!/bin/bash
# the batch curve (1 / 8 / 16 / 48 / 192 streams) with several builds of the library, interleaved
N=${N:-1}
for r in $(seq 1 $N); do
    for L in "$@"; do
        DSV2HIP_LIB=$PWD/digital-subband-video-2_amd/$L python bench.py --only-batch-curve --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('%-22s' % '$L', ' '.join('%d:%.1f' % (p['streams'], p['value']) for p in d['batch_curve']), flush=True)"
    done
done
