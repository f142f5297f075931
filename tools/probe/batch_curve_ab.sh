cd "$GRAFT_REPO_ROOT"
for e in "DSV2_HME_SPLIT=0" "DSV2_HME_SPLIT=1"; do
env $e python3 bench.py --only-batch-curve --no-cpu-baseline --no-profile --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$e', [(p['streams'],p['value']) for p in r.get('batch_curve',[])])
"
done
