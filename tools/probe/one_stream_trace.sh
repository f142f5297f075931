cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/one
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/one/t -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --no-profile --no-stagger --no-mix --streams 1 --groups 1 --steps 40 --warmup 8 > gpurun_out/one/b.json 2> gpurun_out/one/err.txt
f=$(find gpurun_out/one/t -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/one/kernel_stats.csv
python3 - <<'PY'
import csv,json
d=json.load(open('gpurun_out/one/b.json')); print(d['value'],'fps')
rows=list(csv.DictReader(open('gpurun_out/one/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print('%-60s %6s calls %9.1f us avg %6.2f %%'%(r['Name'][:60],r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
print('sum of kernel time per frame (48 frames): %.1f us'%(tot/1e3/48))
PY
