#!/bin/bash
# interleaved A/B of environment settings on one box: N=<rounds> tools/probe/ab_env.sh "VAR=a" "VAR=b" ... [-- bench args]   ("-" = no setting)
N=${N:-2}; envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
mkdir -p gpurun_out
for r in $(seq 1 $N); do
    for E in "${envs[@]}"; do
        if [ "$E" = "-" ]; then S=""; else S="$E"; fi
        env $S timeout 600 python bench.py --no-extras --no-cpu-baseline --no-profile "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('%-28s %8.1f fps' % ('$E', d['value']), flush=True)"
    done
done
