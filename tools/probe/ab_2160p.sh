#!/bin/bash
# the 2160p configuration leg (64 streams, 12 timed steps) with several builds of the library, interleaved
N=${N:-2}
for r in $(seq 1 $N); do
    for L in "$@"; do
        DSV2HIP_LIB=$PWD/digital-subband-video-2_amd/$L python bench.py --no-batch-curve --no-api-legs --no-host-share --no-multi-rank --no-cpu-baseline --no-profile --steps 8 --warmup 2 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); c=d['configs']; print('%-24s 2160p %8.1f fps   444 lossless %8.1f   720p %8.1f' % ('$L', c['c_2160p_420_qp60_gop48']['value'], c['c4_1080p_444_lossless']['value'], c['c2_720p_420_qp60_gop48']['value']), flush=True)"
    done
done
