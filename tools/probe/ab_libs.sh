#!/bin/bash
# interleaved A/B/C... of several builds of the library on one box: N=<rounds> tools/probe/ab_libs.sh lib1.so lib2.so ... [-- bench args]
N=${N:-2}; libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
for r in $(seq 1 $N); do
    for L in "${libs[@]}"; do
        DSV2HIP_LIB=$PWD/digital-subband-video-2_amd/$L python bench.py --no-extras --no-cpu-baseline --no-profile "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('%-28s %8.1f fps' % ('$L', d['value']), flush=True)"
    done
done
