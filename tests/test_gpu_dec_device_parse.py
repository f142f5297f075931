"""The decoder's plane-section parse on the DEVICE (csrc/dec_parse_dev.hip): the decoder tests -- clean streams, batches,
borders, damaged and hostile packets, golden hashes -- once more with every picture's sections parsed by the kernel
(DSV2_DEC_DEVICE_PARSE=2), in its lane-parallel form and in its serial form (DSV2_DEC_LANE_ROUNDS=0: the cross-check of the rounds).
The switch is read when the library loads, so each variant is a pytest process of its own; what those tests compare against is
the reference library, as always."""
import os
import subprocess
import sys

import pytest

import dsvabi as A

pytestmark = [pytest.mark.gpu]

FILES = ["tests/test_gpu_robustness.py", "tests/test_gpu_dec_batch.py", "tests/test_gpu_dec_borders.py", "tests/test_gpu_golden.py"]


@pytest.mark.parametrize("env", [{"DSV2_DEC_DEVICE_PARSE": "2"}, {"DSV2_DEC_DEVICE_PARSE": "2", "DSV2_DEC_LANE_ROUNDS": "0"},
                                 {"DSV2_DEC_DEVICE_PARSE": "1"}, {"DSV2_DEC_DEVICE_PARSE": "0"}],
                         ids=["all-on-device", "all-on-device-serial-step", "P-on-device", "all-on-host"])
def test_decoder_suite_with_the_device_parser(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + FILES, cwd=A.ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    assert " passed" in r.stdout


def test_parse_mode_is_reported():
    import ctypes as C
    hip = A.load_hip()
    hip.dsv2hip_dec_parse_mode.restype = C.c_int
    want = os.environ.get("DSV2_DEC_DEVICE_PARSE")
    got = hip.dsv2hip_dec_parse_mode()
    assert got in (0, 1, 2)
    if want is not None:
        assert got == int(want)
