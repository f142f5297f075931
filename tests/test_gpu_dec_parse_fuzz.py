"""Differential fuzz of the decoder's two parsers: the same damaged packets decoded with the plane sections parsed on the
HOST (csrc/entropy.cpp: entropy_decode_plane) and on the DEVICE (csrc/dec_parse_dev.hip: k_dec_parse, lane-parallel rounds + exact
serial step) must give the same return codes and the same pictures, byte for byte -- whatever the damage: flipped bits anywhere in a
P picture's plane sections, overwritten length and count fields, truncated tails, runs of 0x00 / 0xff.  The switch is flipped at run
time (dsv2hip_dec_set_parse_mode), so both parsers run in ONE process on the same decoder state history.  (The reference decoder reads
out of bounds on some of these inputs, so it cannot be the judge here; tests/test_gpu_robustness.py holds the cases it can judge.)"""
import ctypes as C

import numpy as np
import pytest

import dsvabi as A
from codec_run import encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]


def _decode(hip, packets):
    """decode a packet list on a fresh decoder; returns [(code, fnum, md5-ish bytes of the three planes)]"""
    dec = A.DECODER()
    out = []
    for pk in packets:
        buf = A.BUF()
        hip.dsv_mk_buf(C.byref(buf), len(pk))
        C.memmove(buf.data, pk, len(pk))
        fp = C.POINTER(A.FRAME)()
        fn = C.c_uint32(0)
        code = hip.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
        planes = None
        if code == A.DEC_OK and fp:
            f = fp.contents
            planes = []
            for c in range(3):
                p = f.planes[c]
                a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
                planes.append(a.reshape(-1, p.stride)[:p.h, :p.w].tobytes())
            hip.dsv_frame_ref_dec(fp)
        out.append((code, fn.value if code == A.DEC_OK else None, planes))
    hip.dsv_dec_free(C.byref(dec))
    return out


def _mutations(pk, rng, count):
    """damaged copies of a picture packet; the first 40 bytes (packet header, frame number, block sizes, quantiser) are left alone so
    that the damage lands in the side information and the plane sections"""
    n = len(pk)
    for k in range(count):
        b = bytearray(pk)
        kind = k % 6
        if kind == 0:  # a few flipped bits
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(40, n))
                b[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:  # a run of garbage
            i = int(rng.integers(40, n - 8))
            ln = int(rng.integers(1, min(64, n - i)))
            b[i:i + ln] = bytes(rng.integers(0, 256, size=ln, dtype=np.uint8))
        elif kind == 2:  # truncation
            b = b[:int(rng.integers(40, n))]
        elif kind == 3:  # a run of zeros (long runs / huge exp-Golomb prefixes)
            i = int(rng.integers(40, n - 8))
            b[i:i + int(rng.integers(4, 40))] = bytes(40)[:min(40, n - i)][:int(rng.integers(4, 40))]
        elif kind == 4:  # a run of ones (unary quotients without end)
            i = int(rng.integers(40, n - 8))
            ln = int(rng.integers(4, 40))
            b[i:i + ln] = b"\xff" * min(ln, n - i)
        else:  # a big-endian 32-bit field somewhere overwritten with a large or a tiny number (plane lengths, symbol counts)
            i = int(rng.integers(40, n - 4))
            b[i:i + 4] = int(rng.choice([0, 1, 2, 255, 65535, 1 << 20, (1 << 24) - 1, (1 << 31) - 1, (1 << 32) - 1])).to_bytes(4, "big")
        yield bytes(b[:n])


@pytest.mark.parametrize("w,h,seed", [(352, 288, 3), (640, 368, 4)])
def test_device_parser_equals_host_parser_on_damaged_packets(w, h, seed):
    ref, hip = A.load_ref(), A.load_hip()
    hip.dsv2hip_dec_set_parse_mode.argtypes = [C.c_int]
    hip.dsv2hip_dec_set_parse_mode.restype = C.c_int
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(w, h, "420", seed=seed)
    frames = [v.frame_bytes(t) for t in range(4)]
    packets, _ = encode_stream(ref, frames, w, h, A.SUBSAMP_420, eos=False, qp=70, gop=48)
    meta, pic_i, pics_p = packets[0], packets[1], packets[2:]
    rng = np.random.default_rng(seed)
    cases = []
    for pk in pics_p[:2]:
        for bad in _mutations(pk, rng, 90):
            cases.append([meta, pic_i, bad, pics_p[-1]])       # the damaged P picture, then a clean one on top of it
    for bad in _mutations(pic_i, rng, 30):
        cases.append([meta, bad, pics_p[0]])                   # a damaged intra picture (device parser forced: mode 2)
    try:
        differing = []
        for k, case in enumerate(cases):
            assert hip.dsv2hip_dec_set_parse_mode(0) == 0
            host = _decode(hip, case)
            assert hip.dsv2hip_dec_set_parse_mode(2) == 2
            dev = _decode(hip, case)
            if host != dev:
                differing.append(k)
        assert not differing, "cases %s of %d: device-parsed result differs from host-parsed" % (differing[:10], len(cases))
    finally:
        hip.dsv2hip_dec_set_parse_mode(-1)
