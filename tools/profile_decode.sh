#!/bin/bash
# Decode-side kernel statistics (GPU box, via gpurun): rocprofv3 --kernel-trace --stats over the bench's decode leg alone.
# -> gpurun_out/prof<ROUND>/decode_kernel_trace.csv.gz + decode.json (plane sections parsed as the library chooses: on the host on a 16-core box)
# and decode_dev_kernel_trace.csv.gz + decode_dev.json (DSV2_DEC_DEVICE_PARSE=1: P pictures' sections parsed by k_dec_parse);
# summarised by tools/summarise_round.py into profiles/r0<ROUND>_decode_kernel_stats.txt
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof${ROUND:-6}
mkdir -p $out
# (the encode that produces the packets runs first, untraced ... it cannot be left out of the process: its kernels are in the
# trace too and are told apart by name -- the decoder's launches carry the decode-only kernels k_dequant_*, k_zero_linear and
# k_predict_w<MC_RECONSTRUCT>; the rest is attributed by the phase the summary cuts out: everything after the last k_ent_out launch)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dec -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --decode-too --no-profile --streams 256 --groups 4 --steps 24 --warmup 4 > $out/decode.json 2> $out/decode.err
gzip -c $out/dec/*/*_kernel_trace.csv > $out/decode_kernel_trace.csv.gz
rm -rf $out/dec
ls -la $out
DSV2_DEC_DEVICE_PARSE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dec -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --decode-too --no-profile --streams 256 --groups 4 --steps 24 --warmup 4 > $out/decode_dev.json 2> $out/decode_dev.err
gzip -c $out/dec/*/*_kernel_trace.csv > $out/decode_dev_kernel_trace.csv.gz
rm -rf $out/dec
# HBM-side counter traffic of the decode leg's kernels (host-parsed leg): FETCH_SIZE and WRITE_SIZE in separate passes (the TCC block cannot
# hold both), each with the kernel trace of the same run for the dispatch order and the grids -> profiles/pmc_traffic_decode.json
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/dec -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --decode-too --no-profile --streams 256 --groups 4 --steps 8 --warmup 2 > /dev/null 2> $out/decode_pmc_$c.err
    cp $out/dec/*/*_counter_collection.csv $out/decode_pmc_$c.csv; gzip -f $out/decode_pmc_$c.csv
    gzip -c $out/dec/*/*_kernel_trace.csv > $out/decode_pmc_${c}_trace.csv.gz
    rm -rf $out/dec
done
