#!/bin/bash
# few-stream operating points against the number of lockstep groups (host threads): "streams:groups" pairs
for sg in "$@"; do
    S=${sg%%:*}; G=${sg##*:}
    python bench.py --no-extras --no-cpu-baseline --no-profile --no-mix --streams $S --groups $G --steps 96 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('streams $S groups $G  %8.1f fps  %.3f ms/step' % (d['value'], d['ms_per_step']), flush=True)"
done
