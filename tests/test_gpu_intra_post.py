"""GPU seam tests of the two entry points that round 1 only covered end to end:
dsv_intra_analysis (hme.c:1836, kernel k_intra_analysis) and dsv_post_process (bmc.c:341, kernel k_post_process),
each against the real reference through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from test_oracle_intra import ref_intra_flags, synth_frame
from test_oracle_sbt import rand_frame

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


@pytest.mark.parametrize("w,h,subsamp", [(352, 288, A.SUBSAMP_420), (354, 290, A.SUBSAMP_420), (640, 360, A.SUBSAMP_444),
                                          (1280, 720, A.SUBSAMP_420), (1920, 1080, A.SUBSAMP_420), (1920, 1080, A.SUBSAMP_444)])
@pytest.mark.parametrize("do_psy", [0xff, 0x1, 0x10, 0x0])
def test_intra_analysis_matches_reference(w, h, subsamp, do_psy):
    ref, hip = A.load_ref(), A.load_hip()
    meta = A.mk_meta(w, h, subsamp)
    params = A.mk_params(meta, w, h, 0, 0, do_psy=do_psy)
    for frame in (synth_frame(w, h, subsamp, 5), synth_frame(w, h, subsamp, 6, t=3), rand_frame(subsamp, w, h, seed=2)):
        ref.dsv_extend_frame(frame.ptr())
        want = ref_intra_flags(ref, frame, params)
        got = ref_intra_flags(hip, frame, params)  # same call, the product library
        assert np.array_equal(want, got)


@pytest.mark.parametrize("w,h", [(352, 288), (354, 290), (1000, 562), (1280, 720), (1920, 1080), (960, 540)])
@pytest.mark.parametrize("kind", ["synthetic", "random", "ramp"])
def test_post_process_matches_reference(w, h, kind):
    ref, hip = A.load_ref(), A.load_hip()
    ref.dsv_post_process.argtypes = [C.POINTER(A.PLANE)]
    hip.dsv_post_process.argtypes = [C.POINTER(A.PLANE)]
    if kind == "synthetic":
        a = synth_frame(w & ~1, h & ~1, A.SUBSAMP_420, 7)
        w, h = w & ~1, h & ~1
    elif kind == "random":
        a = rand_frame(A.SUBSAMP_420, w, h, seed=11)
    else:  # ramps that cross histogram buckets inside every 4x4 cell: the de-gradient blend acts on all of them
        a = A.HostFrame(A.SUBSAMP_420, w, h)
        yy, xx = np.mgrid[0:h, 0:w]
        a.plane(0)[:, :] = ((xx * 9 + yy * 5) % 256).astype(np.uint8)
    b = A.HostFrame(A.SUBSAMP_420, w, h)
    for c in range(3):
        b.full[c][:, :] = a.full[c]
    before = a.full[0].copy()
    ref.dsv_post_process(a.plane_ptr(0))
    hip.dsv_post_process(b.plane_ptr(0))
    assert np.array_equal(a.full[0], b.full[0])
    if kind != "random":
        assert not np.array_equal(before, a.full[0]), "the case does not exercise the filter"
