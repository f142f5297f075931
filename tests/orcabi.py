"""ctypes views of the oracle's own structs (oracle/orc_common.h, orc_bmc.c)."""
import ctypes as C

import numpy as np

import dsvabi as A


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("width", "height", "hshift", "vshift", "blk_w", "blk_h", "nblocks_h", "nblocks_v",
                 "isP", "lossless", "do_psy", "effort", "temporal_mc", "inter_sharpen")]


class OFrame(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8) * 3), ("stride", C.c_int * 3), ("w", C.c_int * 3), ("h", C.c_int * 3)]


def orc_params(params, meta):
    p = OrcParams()
    p.width, p.height = meta.width, meta.height
    p.hshift, p.vshift = (meta.subsamp >> 2) & 3, meta.subsamp & 3
    p.blk_w, p.blk_h, p.nblocks_h, p.nblocks_v = params.blk_w, params.blk_h, params.nblocks_h, params.nblocks_v
    p.isP, p.lossless, p.do_psy, p.effort = params.has_ref, params.lossless, params.do_psy, params.effort
    p.temporal_mc, p.inter_sharpen = params.temporal_mc, meta.inter_sharpen
    return p


def oframe(hf):
    f = OFrame()
    for c in range(3):
        f.data[c] = hf.c.planes[c].data
        f.stride[c] = hf.strides[c]
        f.w[c], f.h[c] = hf.dims[c]
    return f


class HPlane(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("stride", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class HmeCtx(C.Structure):
    _fields_ = [("p", OrcParams), ("quant", C.c_int), ("skip_block_thresh", C.c_int), ("pyr_levels", C.c_int),
                ("src", HPlane * 6), ("ref", HPlane * 6), ("ogr", HPlane * 6),
                ("srcc", HPlane * 2), ("refc", HPlane * 2),
                ("mvf", C.c_void_p * 6), ("ref_mvf", C.c_void_p),
                ("nintra", C.c_int), ("ndiff", C.c_int), ("eligible", C.c_int), ("total_err", C.c_uint)]


def hplane(hf, c):
    p = HPlane()
    p.data = hf.c.planes[c].data
    p.stride = hf.strides[c]
    p.w, p.h = hf.dims[c]
    return p
