#!/bin/bash
# usage (GPU box): tools/proc_probe.sh -- start-up and per-frame cost of the drop-in CLI, alone and 8 processes at once
cd "$GRAFT_REPO_ROOT"
D=/dev/shm/pp; mkdir -p $D
python3 - <<'PY'
import sys, os
sys.path.insert(0, "tests")
from conftest import load_pkg
pkg = load_pkg()
v = pkg.synth.SynthVideo(1920, 1080, "420", seed=5)
with open("/dev/shm/pp/in.yuv", "wb") as f:
    for p in range(8):
        for t in range(48):
            f.write(v.frame_bytes(t % 24))
PY
EXE=oracle/_ref/dsv2_dropin
ARGS="-y -inp=$D/in.yuv -w=1920 -h=1080 -fps_num=30 -fps_den=1 -gop=48 -qp=60 -rc_mode=0 -noeos=1"
t() { local s=$(date +%s.%N); "$@" > /dev/null 2>&1; local e=$(date +%s.%N); echo "$(echo "$e - $s" | bc -l 2>/dev/null || python3 -c "print($e-$s)")"; }
for n in 1 12 48; do echo "1 proc nfr=$n: $(t $EXE e $ARGS -out=$D/o.dsv -sfr=0 -nfr=$n) s"; done
for env in "" "DSV2_HOST_THREADS=2" "DSV2_HOST_THREADS=2 GPU_MAX_HW_QUEUES=2"; do
  s=$(date +%s.%N)
  for p in 0 1 2 3 4 5 6 7; do ( env $env $EXE e $ARGS -out=$D/o$p.dsv -sfr=$((p*48)) -nfr=48 > /dev/null 2>&1; echo "  proc $p done at $(python3 -c "import time; print(round(time.time()-$s,2))")" ) & done
  wait
  e=$(date +%s.%N)
  echo "8 procs [$env]: $(python3 -c "print(round($e-$s,2), 'ms' , round(384/($e-$s),1), 'fps')")"
done
s=$(date +%s.%N)
for p in 0 1 2 3; do ( $EXE e $ARGS -out=$D/o$p.dsv -sfr=$((p*48)) -nfr=48 > /dev/null 2>&1 ) & done; wait
e=$(date +%s.%N); echo "4 procs: $(python3 -c "print(round($e-$s,2), round(192/($e-$s),1), 'fps')")"
s=$(date +%s.%N)
for p in 0 1; do ( $EXE e $ARGS -out=$D/o$p.dsv -sfr=$((p*48)) -nfr=48 > /dev/null 2>&1 ) & done; wait
e=$(date +%s.%N); echo "2 procs: $(python3 -c "print(round($e-$s,2), round(96/($e-$s),1), 'fps')")"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
rm -rf $D
