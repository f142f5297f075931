cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_ic; rm -rf $out; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_DCACHE[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|TCP_[A-Z_]*STALL[A-Z_]*\|TA_BUSY[A-Z_]*\|TA_TA_BUSY[A-Z_]*\|TCP_PENDING[A-Z_]*\|TCC_BUSY[A-Z_]*\|TCP_TA_[A-Z_]*" | sort -u | head -60 > $out/names.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex "k_hme_rows_l0|k_predict_w|k_quant_level4|k_inv_haar_u8x4" --output-format csv -d $out/raw -- python3 bench.py --steps 4 --warmup 2 --gen-procs 1 --no-extras --no-cpu-baseline --no-profile > /dev/null 2> $out/err.txt
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_ic/raw/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("dsv2::", "").replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k, {c: "%.3e" % x for c, x in sorted(v.items())})
PY
cat $out/names.txt | tr '\n' ' '
tail -3 $out/err.txt
rm -rf $out/raw
