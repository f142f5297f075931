#!/bin/bash
# Repeats the HME parity test under the row-pipeline switches to expose timing-dependent faults.
# usage: tools/stress_hme.sh [runs]
runs=${1:-12}
for cfg in "0 1" "1 1" "3 1" "0 0"; do
    set -- $cfg
    f=0
    for i in $(seq 1 $runs); do
        DSV2_HME_FENCE=$1 DSV2_HME_ROWS=$2 python -m pytest tests/test_gpu_hme.py -q -x 2>&1 | tail -1 | grep -q failed && f=$((f + 1))
    done
    echo "fence=$1 rows=$2 failures=$f/$runs"
done
