"""Body of __graft_entry__.smoke(): a CIF frame through the HIP subband transform, checked against the oracle."""
import ctypes as C

import numpy as np

import dsvabi as A
from test_oracle_sbt import rand_frame


def run():
    hip = A.load_hip()
    assert hip.dsv2hip_device_ok() == 0, "no HIP device"
    orc = A.load_oracle()
    w, h, subsamp = 352, 288, A.SUBSAMP_420
    meta = A.mk_meta(w, h, subsamp)
    for isP in (0, 1):
        params = A.mk_params(meta, w, h, isP, 0)
        nb = params.nblocks_h * params.nblocks_v
        blockdata = (np.arange(nb) % 16).astype(np.uint8)
        frame = rand_frame(subsamp, w, h, seed=11)
        cdims = A.coef_dims(subsamp, w, h)
        for plane in range(3):
            cw, ch = cdims[plane]
            pw, ph = frame.dims[plane]
            want = np.zeros(cw * ch, dtype=np.int32)
            orc.orc_fwd_sbt(frame.c.planes[plane].data, frame.strides[plane], pw, ph,
                            A.np_ptr(want, C.c_int32), cw, ch, plane, isP, 0,
                            A.np_ptr(blockdata, C.c_uint8), params.nblocks_h, params.nblocks_v)
            got = np.zeros(cw * ch, dtype=np.int32)
            cs = A.COEFS(A.np_ptr(got, C.c_int32), cw, ch)
            fm = A.FMETA()
            fm.params = C.pointer(params)
            fm.blockdata = A.np_ptr(blockdata, C.c_uint8)
            fm.cur_plane, fm.isP = plane, isP
            hip.dsv_fwd_sbt(frame.plane_ptr(plane), C.byref(cs), C.byref(fm))
            assert np.array_equal(want, got), "smoke: fwd sbt mismatch plane %d isP %d" % (plane, isP)
    print("smoke ok")
