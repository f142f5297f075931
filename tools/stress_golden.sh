#!/bin/bash
# Repeats the golden-stream and batch parity tests to expose timing-dependent faults.
runs=${1:-8}
for cfg in "0 1 4" "0 1 1" "0 1 2" "1 1 1" "0 0 4"; do
    set -- $cfg
    f=0
    for i in $(seq 1 $runs); do
        DSV2_HME_FENCE=$1 DSV2_HME_ROWS=$2 DSV2_HME_WAVES=$3 python -m pytest tests/test_gpu_golden.py tests/test_gpu_hme.py tests/test_gpu_batch.py -q 2>&1 | tail -1 | grep -q failed && f=$((f + 1))
    done
    echo "fence=$1 rows=$2 waves=$3 failures=$f/$runs"
done
