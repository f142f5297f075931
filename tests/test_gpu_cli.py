"""Drop-in proof at the process level: the reference's own CLI linked against libdsv2hip.so
(oracle/_ref/dsv2_dropin) vs the pure reference build (oracle/_ref/dsv2_ref), plus the independent
single-header decoder (oracle/_ref/d28dec_ref) on the GPU-produced stream."""
import hashlib
import os
import subprocess

import pytest

import dsvabi as A
from conftest import load_pkg

DROPIN = os.path.join(A.ROOT, "oracle", "_ref", "dsv2_dropin")
pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


@pytest.mark.parametrize("w,h,n,flags", [
    (352, 288, 10, ["-qp=85", "-gop=0"]),                       # BASELINE config 1 (intra only)
    (352, 288, 14, ["-qp=60", "-gop=6", "-effort=10"]),
    (1280, 720, 5, ["-qp=60", "-gop=48", "-effort=10"]),       # BASELINE config 2 geometry
])
def test_cli_streams_identical(tmp_path, w, h, n, flags):
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(w, h, "420", seed=11)
    y4m = str(tmp_path / "in.y4m")
    pkg.synth.write_y4m(y4m, v, n)
    outs = {}
    for name, exe in (("ref", A.REF_CLI), ("hip", DROPIN)):
        dsv = str(tmp_path / (name + ".dsv"))
        run([exe, "e", "-inp=" + y4m, "-out=" + dsv, "-y4m=1", "-y", "-nfr=%d" % n] + flags)
        outs[name] = dsv
    assert md5(outs["ref"]) == md5(outs["hip"]), "encoded streams differ"
    # decode the GPU-made stream with: the reference decoder, the GPU decoder, the single-header decoder
    dec = {}
    for name, exe in (("ref", A.REF_CLI), ("hip", DROPIN)):
        yuv = str(tmp_path / (name + ".yuv"))
        run([exe, "d", "-inp=" + outs["hip"], "-out=" + yuv, "-y"])
        dec[name] = yuv
    assert md5(dec["ref"]) == md5(dec["hip"]), "decoded pictures differ"
    assert os.path.getsize(dec["hip"]) == n * v.frame_size()
    if os.path.exists(A.REF_D28):
        yuv = str(tmp_path / "d28.yuv")
        run([A.REF_D28, "-inp=" + outs["hip"], "-out=" + yuv, "-y"])
        assert md5(yuv) == md5(dec["hip"]), "single-header decoder disagrees"
