"""HIP hierarchical motion estimation (wavefront-per-block kernels) vs the real reference's dsv_hme."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from hme_common import HME, Scene, assert_fields_equal

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def run_lib(lib, sc, quant, effort, skip_thresh=0):
    return sc.run_reference(lib, quant, effort, skip_thresh)


@pytest.mark.parametrize("w,h,subsamp,seed,quant,effort,prev", [
    (352, 288, A.SUBSAMP_420, 1, 172, 10, True),
    (352, 288, A.SUBSAMP_420, 2, 900, 10, False),
    (354, 290, A.SUBSAMP_420, 3, 61, 10, True),
    (640, 360, A.SUBSAMP_444, 4, 300, 7, True),
    (640, 360, A.SUBSAMP_420, 5, 172, 3, True),
    (1280, 720, A.SUBSAMP_420, 6, 172, 10, True),
    (1920, 1080, A.SUBSAMP_420, 7, 172, 10, True),
    (640, 360, A.SUBSAMP_422, 8, 172, 10, True),     # 4:2:2: the general block routine at level 0
    (3840, 2160, A.SUBSAMP_420, 9, 172, 10, True),   # 32 x 32 blocks (dsv_encoder.c:1203-1211)
    (1920, 800, A.SUBSAMP_420, 10, 172, 10, True),   # 32 x 16 blocks: wider than 1280, not "mostly square" (dsv_encoder.c:1203-1209)
    (2560, 1080, A.SUBSAMP_420, 11, 172, 10, True),  # 32 x 16 blocks
    (1920, 816, A.SUBSAMP_420, 12, 300, 7, False),   # 32 x 16, effort 7 (no quarter-pel), no previous field
    (1936, 808, A.SUBSAMP_420, 13, 172, 10, True),   # 32 x 16 with a clipped last column (16 wide) and row (8 high): still the fast 32 x 16 routine
    (1928, 804, A.SUBSAMP_420, 14, 61, 10, True),    # 32 x 16 clipped to 8 wide / 4 high: the general routine at level 0, the fast one above it
    (2048, 860, A.SUBSAMP_444, 15, 172, 10, True),   # 32 x 16 in 4:4:4: level 0 on the general routine (the 32-wide level-0 form is 4:2:0)
])
def test_hme_matches_reference(w, h, subsamp, seed, quant, effort, prev):
    ref, hip = A.load_ref(), A.load_hip()
    sc = Scene(ref, w, h, subsamp, seed, with_prev_mvs=prev)
    want, ipct_r, scb_r, err_r = run_lib(ref, sc, quant, effort)
    got, ipct_h, scb_h, err_h = run_lib(hip, sc, quant, effort)
    for l in range(sc.levels, -1, -1):
        assert_fields_equal(want[l], got[l], "level %d" % l)
    assert (ipct_r, scb_r, err_r) == (ipct_h, scb_h, err_h)
