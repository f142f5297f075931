#!/bin/bash
# frames/s of the headline leg against the number of persistent search workers (DSV2_HME_PERSIST), interleaved rounds
LIB=${LIB:-digital-subband-video-2_amd/libdsv2hip.so}; N=${N:-3}
for r in $(seq 1 $N); do
    for P in "$@"; do
        DSV2HIP_LIB=$PWD/$LIB DSV2_HME_PERSIST=$P python bench.py --no-extras --no-cpu-baseline --no-profile > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$LIB persist $P  %8.1f fps' % d['value'], flush=True)"
    done
done
