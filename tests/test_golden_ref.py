"""The oracle is pinned: the reference library, driven through the same harness the GPU tests use, reproduces the
golden hashes that were captured from the reference CLI (and the generator is byte-stable)."""
import hashlib
import os

import pytest

import dsvabi as A
from golden_common import GOLDEN, run_entry

pytestmark = pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built")


@pytest.mark.parametrize("name", ["c1_cif_intra", "x_cif_ip_long", "c2_720p_ip", "x_720p_abr", "x_720p_effort3", "x_1920x800_ip"])
def test_reference_library_reproduces_golden(name):
    g = GOLDEN[name]
    stream, decoded = run_entry(A.load_ref(), g)
    assert len(stream) == g["dsv_bytes"]
    assert hashlib.md5(stream).hexdigest() == g["dsv_md5"]
    assert hashlib.md5(decoded).hexdigest() == g["decoded_md5"]
