#!/bin/bash
# A/B of two builds of the library on one box: bench.py (headline leg only) alternately with DSV2HIP_LIB = $1 and $2, $3 rounds;
# prints frames/s, the level-0 launch under load and the search stages per round.
A=$1; B=$2; N=${3:-2}; shift 3
for r in $(seq 1 $N); do
    for L in $A $B; do
        DSV2HIP_LIB=$PWD/$L python bench.py --no-extras --no-cpu-baseline --no-profile "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
        python - "$L" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
r = d.get("roofline", {})
print("%-48s %8.1f fps   l0 launch %8.1f us   frac %.4f" % (sys.argv[1], d["value"], r.get("avg_launch_us", 0), r.get("frac", 0)), flush=True)
PY
    done
done
