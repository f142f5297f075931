"""The GPU entropy coder (entropy_gpu.hip) across the quality range, plus its host fallback: every packet equals the
reference's.  High qualities push the adaptive Rice state up (large coefficients), low ones leave long zero runs; lossless
codes unquantised values.  DSV2_GPU_ENTROPY_FORCE_FALLBACK / DSV2_GPU_ENTROPY=0 take the host coder (separate processes:
the switches are read once when the library loads)."""
import json
import os
import subprocess
import sys

import pytest

import dsvabi as A
from codec_run import encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)

CASES = [(1920, 1080, "420", 20), (1920, 1080, "420", 85), (1920, 1080, "420", 97), (1280, 720, "420", 99), (640, 360, "444", 100),
         (352, 288, "420", 5), (640, 360, "444", 92)]


def encode_both(w, h, fmt, qp, nframes=3):
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(w, h, fmt, seed=31 + qp)
    frames = [v.frame_bytes(t) for t in range(nframes)]
    subsamp = A.SUBSAMP_420 if fmt == "420" else A.SUBSAMP_444
    want, _ = encode_stream(A.load_ref(), frames, w, h, subsamp, eos=False, qp=qp, gop=48)
    got, _ = encode_stream(A.load_hip(), frames, w, h, subsamp, eos=False, qp=qp, gop=48)
    return want, got


@pytest.mark.parametrize("w,h,fmt,qp", CASES)
def test_quality_range(w, h, fmt, qp):
    want, got = encode_both(w, h, fmt, qp)
    assert want == got


_CHILD = r"""
import sys
sys.path.insert(0, %r)
import test_gpu_entropy_paths as T
bad = 0
for (w, h, fmt, qp) in [(1280, 720, "420", 60), (640, 360, "444", 100), (352, 288, "420", 95)]:
    want, got = T.encode_both(w, h, fmt, qp)
    bad += want != got
print("BAD", bad)
"""


# DSV2_ENT_EMIT_WORDS shrinks the emit kernel's LDS image of a chunk: 8 words sends every chunk down the path that ORs its
# code words straight into global memory, 40 mixes both paths (short chunks in LDS, long ones not)
# DSV2_SIDE_FORCE_FALLBACK: the per-block side information of every P picture is coded by the host (the path a frame takes whose
# sub-streams do not fit the device coder's images), from the field the device finalised
@pytest.mark.parametrize("env", [{"DSV2_GPU_ENTROPY_FORCE_FALLBACK": "1"}, {"DSV2_GPU_ENTROPY": "0"}, {"DSV2_ENT_EMIT_WORDS": "8"},
                                 {"DSV2_ENT_EMIT_WORDS": "40"}, {"DSV2_SIDE_FORCE_FALLBACK": "1"}])
def test_host_coder_paths(env):
    r = subprocess.run([sys.executable, "-c", _CHILD % os.path.dirname(os.path.abspath(__file__))], env=dict(os.environ, **env),
                       stdout=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0
    assert r.stdout.strip().splitlines()[-1] == "BAD 0"
