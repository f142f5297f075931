#!/bin/bash
# Decode-side kernel statistics (GPU box, via gpurun): rocprofv3 --kernel-trace --stats over the bench's decode leg alone.
# -> gpurun_out/prof5/decode_kernel_stats.csv + decode.json; summarised by tools/summarise_r05.py into profiles/r05_decode_kernel_stats.txt
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof5
mkdir -p $out
# (the encode that produces the packets runs first, untraced ... it cannot be left out of the process: its kernels are in the
# trace too and are told apart by name -- the decoder's launches carry the decode-only kernels k_dequant_*, k_zero_linear and
# k_predict_w<MC_RECONSTRUCT>; the rest is attributed by the phase the summary cuts out: everything after the last k_ent_out launch)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dec -- python3 bench.py --gen-procs 1 --no-cpu-baseline --no-extras --decode-too --no-profile --streams 256 --groups 4 --steps 24 --warmup 4 > $out/decode.json 2> $out/decode.err
gzip -c $out/dec/*/*_kernel_trace.csv > $out/decode_kernel_trace.csv.gz
rm -rf $out/dec
ls -la $out
