cd "$GRAFT_REPO_ROOT"
D=/dev/shm/pp; mkdir -p $D
python3 - <<'PY'
import sys
sys.path.insert(0, "tests")
from conftest import load_pkg
pkg = load_pkg()
v = pkg.synth.SynthVideo(1920, 1080, "420", seed=5)
with open("/dev/shm/pp/in.yuv", "wb") as f:
    for t in range(48):
        f.write(v.frame_bytes(t % 24))
PY
EXE=oracle/_ref/dsv2_dropin
ARGS="-y -inp=$D/in.yuv -w=1920 -h=1080 -fps_num=30 -fps_den=1 -gop=48 -qp=60 -rc_mode=0 -noeos=1"
for i in 1 2; do DSV2_TRACE=1 $EXE e $ARGS -out=$D/o.dsv -sfr=0 -nfr=3 2>&1 | grep startup; echo; done
DSV2_TRACE=1 DSV2_HOST_THREADS=2 $EXE e $ARGS -out=$D/o.dsv -sfr=0 -nfr=3 2>&1 | grep startup
rm -rf $D
