"""oracle/orc_intra.c vs the reference's dsv_intra_analysis on synthetic and random pictures."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
import orcabi as O
from conftest import load_pkg
from test_oracle_sbt import rand_frame

pytestmark = pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built")


def synth_frame(w, h, subsamp, seed, t=0):
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(w, h, "420" if subsamp == A.SUBSAMP_420 else "444", seed=seed)
    f = A.HostFrame(subsamp, w, h, border=True)
    f.set_planes(*v.frame(t))
    return f


def ref_intra_flags(ref, frame, params):
    nb = params.nblocks_h * params.nblocks_v
    p = ref.dsv_intra_analysis(frame.ptr(), C.byref(params))
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb * 16,)).copy().view(A.MV_DTYPE)
    ref.dsv_free(C.cast(p, C.c_void_p))
    return arr["flags"].copy()


@pytest.mark.parametrize("w,h,subsamp", [(352, 288, A.SUBSAMP_420), (354, 290, A.SUBSAMP_420), (640, 360, A.SUBSAMP_444),
                                          (1920, 1080, A.SUBSAMP_420)])
@pytest.mark.parametrize("do_psy", [0xff, 0x1, 0x10, 0x0])
def test_intra_analysis(w, h, subsamp, do_psy):
    ref, orc = A.load_ref(), A.load_oracle()
    meta = A.mk_meta(w, h, subsamp)
    params = A.mk_params(meta, w, h, 0, 0, do_psy=do_psy)
    nb = params.nblocks_h * params.nblocks_v
    for frame in (synth_frame(w, h, subsamp, 5), rand_frame(subsamp, w, h, seed=2)):
        ref.dsv_extend_frame(frame.ptr())
        want = ref_intra_flags(ref, frame, params)
        got = np.zeros(nb, dtype=A.MV_DTYPE)
        planes = (C.POINTER(C.c_uint8) * 3)(*[frame.c.planes[c].data for c in range(3)])
        strides = (C.c_int * 3)(*frame.strides)
        orc.orc_intra_analysis(planes, strides, C.byref(O.orc_params(params, meta)), C.c_void_p(got.ctypes.data))
        assert np.array_equal(want, got["flags"])
        assert len(set(want.tolist())) > 1 or do_psy != 0xff
