"""Shared set-up for the motion-estimation parity tests: pictures, pyramids, reference invocation."""
import ctypes as C

import numpy as np

import dsvabi as A
import orcabi as O
from conftest import load_pkg


class HME(C.Structure):
    _fields_ = [("params", C.POINTER(A.PARAMS)),
                ("src", C.POINTER(A.FRAME) * 6), ("ref", C.POINTER(A.FRAME) * 6), ("ogr", C.POINTER(A.FRAME) * 6),
                ("mvf", C.POINTER(A.MV) * 6), ("ref_mvf", C.POINTER(A.MV)),
                ("mv_bank", A.MV * 128), ("n_mv_bank_used", C.c_int),
                ("enc", C.POINTER(A.ENCODER)), ("quant", C.c_int)]


def pyramid_levels(w, h, nbh, nbv):
    lv = 0
    i = 1
    while i < min(w, h):
        i <<= 1
        lv += 1
    while (1 << lv) > max(nbh, nbv):
        lv -= 1
    return max(3, min(5, lv))


def frame_from_planes(subsamp, w, h, planes):
    f = A.HostFrame(subsamp, w, h, border=True)
    f.set_planes(*planes)
    return f


def build_pyramid(lib, base, levels):
    """[base, level1, ...] with luma decimation + extension done by `lib` (dsv_encoder.c:494)."""
    out = [base]
    prev = base
    for l in range(1, levels + 1):
        w, h = (base.w + (1 << l) - 1) >> l, (base.h + (1 << l) - 1) >> l
        f = A.HostFrame(base.subsamp, w, h, border=True)
        lib.dsv_ds2x_frame_luma(f.ptr(), prev.ptr())
        lib.dsv_extend_frame_luma(f.ptr())
        out.append(f)
        prev = f
    return out


class Scene:
    """src = frame t, ogr = frame t-1 (original), ref = degraded frame t-1 (stands for the reconstruction)."""

    def __init__(self, ref_lib, w, h, subsamp, seed, t=3, with_prev_mvs=True):
        pkg = load_pkg()
        self.w, self.h, self.subsamp = w, h, subsamp
        v = pkg.synth.SynthVideo(w, h, "420" if subsamp == A.SUBSAMP_420 else "444", seed=seed)
        cur, prev = v.frame(t), v.frame(t - 1)
        if subsamp not in (A.SUBSAMP_420, A.SUBSAMP_444):  # 4:2:2, 4:1:1 ...: the 4:4:4 picture's chroma decimated to the format's grid
            hs, vs = A.format_shifts(subsamp)
            cur = (cur[0], cur[1][::1 << vs, ::1 << hs].copy(), cur[2][::1 << vs, ::1 << hs].copy())
            prev = (prev[0], prev[1][::1 << vs, ::1 << hs].copy(), prev[2][::1 << vs, ::1 << hs].copy())
        rng = np.random.RandomState(seed)
        deg = [np.clip((p.astype(np.int32) // 6) * 6 + 3 + rng.randint(-1, 2, size=p.shape), 0, 255).astype(np.uint8)
               for p in prev]
        self.src0 = frame_from_planes(subsamp, w, h, cur)
        self.ogr0 = frame_from_planes(subsamp, w, h, prev)
        self.ref0 = frame_from_planes(subsamp, w, h, deg)
        for f in (self.src0, self.ogr0, self.ref0):
            ref_lib.dsv_extend_frame(f.ptr())
        self.meta = A.mk_meta(w, h, subsamp)
        self.params = A.mk_params(self.meta, w, h, 1, 0, temporal_mc=t & 1)
        self.nb = self.params.nblocks_h * self.params.nblocks_v
        self.levels = pyramid_levels(w, h, self.params.nblocks_h, self.params.nblocks_v)
        self.src = build_pyramid(ref_lib, self.src0, self.levels)
        self.ogr = build_pyramid(ref_lib, self.ogr0, self.levels)
        self.ref = build_pyramid(ref_lib, self.ref0, self.levels)
        self.prev_mvs = None
        if with_prev_mvs:
            m = np.zeros(self.nb, dtype=A.MV_DTYPE)
            m["x"] = rng.randint(-14, 15, size=self.nb)
            m["y"] = rng.randint(-10, 11, size=self.nb)
            m["x"][: self.nb // 2] = 6
            m["y"][: self.nb // 2] = 4
            self.prev_mvs = m

    def run_reference(self, ref_lib, quant, effort=10, skip_thresh=0):
        enc = A.ENCODER()
        ref_lib.dsv_enc_init(C.byref(enc))
        enc.pyramid_levels = self.levels
        enc.skip_block_thresh = skip_thresh
        self.params.effort = effort
        hme = HME()
        hme.params = C.pointer(self.params)
        for l in range(self.levels + 1):
            hme.src[l] = C.pointer(self.src[l].c)
            hme.ref[l] = C.pointer(self.ref[l].c)
            hme.ogr[l] = C.pointer(self.ogr[l].c)
        if self.prev_mvs is not None:
            hme.ref_mvf = C.cast(self.prev_mvs.ctypes.data, C.POINTER(A.MV))
        hme.enc = C.pointer(enc)
        hme.quant = quant
        scb, aerr = C.c_int(0), C.c_int(0)
        ref_lib.dsv_hme.argtypes = [C.POINTER(HME), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        ref_lib.dsv_hme.restype = C.c_int
        ipct = ref_lib.dsv_hme(C.byref(hme), C.byref(scb), C.byref(aerr))
        fields = []
        for l in range(self.levels + 1):
            arr = np.ctypeslib.as_array(C.cast(hme.mvf[l], C.POINTER(C.c_uint8)), shape=(self.nb * 16,)).copy().view(A.MV_DTYPE)
            ref_lib.dsv_free(C.cast(hme.mvf[l], C.c_void_p))
            fields.append(arr)
        return fields, ipct, scb.value, aerr.value

    def run_oracle(self, orc, quant, effort=10, skip_thresh=0):
        self.params.effort = effort
        ctx = O.HmeCtx()
        ctx.p = O.orc_params(self.params, self.meta)
        ctx.quant, ctx.skip_block_thresh, ctx.pyr_levels = quant, skip_thresh, self.levels
        for l in range(self.levels + 1):
            ctx.src[l] = O.hplane(self.src[l], 0)
            ctx.ref[l] = O.hplane(self.ref[l], 0)
            ctx.ogr[l] = O.hplane(self.ogr[l], 0)
        for c in (1, 2):
            ctx.srcc[c - 1] = O.hplane(self.src0, c)
            ctx.refc[c - 1] = O.hplane(self.ref0, c)
        fields = [np.zeros(self.nb, dtype=A.MV_DTYPE) for _ in range(self.levels + 1)]
        for l in range(self.levels + 1):
            ctx.mvf[l] = fields[l].ctypes.data
        ctx.ref_mvf = self.prev_mvs.ctypes.data if self.prev_mvs is not None else None
        scb, aerr = C.c_int(0), C.c_int(0)
        orc.orc_hme.restype = C.c_int
        ipct = orc.orc_hme(C.byref(ctx), C.byref(scb), C.byref(aerr))
        return fields, ipct, scb.value, aerr.value


def assert_fields_equal(want, got, what=""):
    for name in ("x", "y", "flags", "err", "dc", "submask"):
        if not np.array_equal(want[name], got[name]):
            bad = np.nonzero(want[name] != got[name])[0]
            raise AssertionError("%s: field %s differs at %d blocks, first %d: want %r got %r" %
                                 (what, name, len(bad), bad[0], want[name][bad[0]], got[name][bad[0]]))
