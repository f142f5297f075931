"""The GPU library reproduces the committed golden hashes of all five BASELINE configs (scaled-down frame counts)
without the reference being present."""
import hashlib

import pytest

from golden_common import GOLDEN, run_entry
import dsvabi as A

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_gpu_reproduces_golden(name):
    g = GOLDEN[name]
    stream, decoded = run_entry(A.load_hip(), g)
    assert len(stream) == g["dsv_bytes"]
    assert hashlib.md5(stream).hexdigest() == g["dsv_md5"], "bitstream hash"
    assert hashlib.md5(decoded).hexdigest() == g["decoded_md5"], "decoded picture hash"
    assert len(decoded) == g["decoded_bytes"]
