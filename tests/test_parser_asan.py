"""The decoder's untrusted-input parser under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU.

csrc/dec_parse.h (packet header, metadata, per-block side information) and csrc/entropy.cpp (plane sections) hold every
bit the decoder reads from a packet before anything reaches the device; tests/parser_fuzz.cpp compiles them host-only with the
sanitizers and feeds them the packets of reference-encoded streams, intact and damaged (flipped bytes, huge length fields,
truncations, noise tails).  Needs no GPU: the bounds checks of the parser had only ever run where a GPU was present
(tests/test_gpu_robustness.py), i.e. never under a sanitizer.
"""
import json
import os
import struct
import subprocess

import pytest

import dsvabi as A
from codec_run import encode_stream
from conftest import load_pkg

HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(A.ROOT, "digital-subband-video-2_amd", "csrc")
pytestmark = [pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built"),
              pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")]


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fuzz") / "parser_fuzz")
    cmd = ["make", "-C", os.path.join(A.ROOT, "oracle"), "parser-fuzz", "FUZZ_OUT=" + out]  # (the sanitizer flags live in oracle/Makefile)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return out


def write_packets(path, streams):
    with open(path, "wb") as f:
        for packets in streams:
            for pk in packets:
                f.write(struct.pack("<I", len(pk)))
                f.write(pk)


@pytest.mark.parametrize("w,h,subsamp,fmt,qp,gop", [(352, 288, A.SUBSAMP_420, "420", 60, 4), (354, 290, A.SUBSAMP_420, "420", 85, 3),
                                                     (320, 240, A.SUBSAMP_444, "444", 100, 4), (640, 360, A.SUBSAMP_420, "420", 30, 5)])
def test_parser_survives_damaged_packets(tmp_path, fuzz_bin, w, h, subsamp, fmt, qp, gop):
    ref = A.load_ref()
    v = load_pkg().synth.SynthVideo(w, h, fmt, seed=7)
    packets = encode_stream(ref, [v.frame_bytes(t) for t in range(6)], w, h, subsamp, eos=True, qp=qp, gop=gop)[0]
    path = str(tmp_path / "packets.bin")
    write_packets(path, [packets])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_bin, path, "400", "11"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-4000:])
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["clean_pictures"] == 6 and rep["clean_planes_ok"] == 18 and rep["clean_planes_bad"] == 0
    # the damaged copies really went through the parser, and through both of its outcomes
    assert rep["damaged_parsed"] > 1000 and rep["damaged_refused"] > 0 and rep["damaged_planes_bad"] > 0
