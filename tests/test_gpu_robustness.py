"""Damaged picture packets: the GPU decoder returns the same codes and the same pictures as the reference
decoder (planes that fail to parse stay zero, dsv_decoder.c:516-523; truncated symbol lists, hzcc.c:525-529)."""
import ctypes as C
import os

import numpy as np
import pytest

import dsvabi as A
from codec_run import encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(A.REF_SO), reason="oracle/_ref not built")]


def decode_all(lib, packets):
    """Like codec_run.decode_stream but keeps going after DSV_DEC_ERROR and records every return code."""
    dec = A.DECODER()
    out = []
    for pk in packets:
        buf = A.BUF()
        lib.dsv_mk_buf(C.byref(buf), len(pk) + 64)
        C.memmove(buf.data, pk, len(pk))
        fp = C.POINTER(A.FRAME)()
        fn = C.c_uint32(0)
        code = lib.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
        planes = None
        if code == A.DEC_OK and fp:
            f = fp.contents
            planes = []
            for c in range(3):
                p = f.planes[c]
                a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
                planes.append(a.reshape(-1, p.stride)[:p.h, :p.w].copy())
            lib.dsv_frame_ref_dec(fp)
        out.append((code, fn.value if code == A.DEC_OK else None, planes))
        if code == A.DEC_EOS:
            break
    lib.dsv_dec_free(C.byref(dec))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_damaged_plane_data(seed):
    ref, hip = A.load_ref(), A.load_hip()
    ref.dsv_set_log_level(0)  # the reference reports every damaged plane on stderr
    pkg = load_pkg()
    w, h = 352, 288
    v = pkg.synth.SynthVideo(w, h, "420", seed=11)
    frames = [v.frame_bytes(t) for t in range(5)]
    packets, _ = encode_stream(ref, frames, w, h, A.SUBSAMP_420, eos=True, qp=60, gop=4)
    rng = np.random.default_rng(seed)
    damaged = []
    for pk in packets:
        b = bytearray(pk)
        if len(b) > 400:  # a picture packet: damage bytes in the second half (coefficient data of some plane)
            for _ in range(1 + seed % 3):
                i = int(rng.integers(len(b) // 2, len(b) - 8))
                b[i] ^= int(rng.integers(1, 256))
        damaged.append(bytes(b))
    want, got = decode_all(ref, damaged), decode_all(hip, damaged)
    assert [x[0] for x in want] == [x[0] for x in got]
    for (cw, fw, pw), (cg, fg, pg) in zip(want, got):
        assert fw == fg
        if pw is not None:
            for c in range(3):
                assert np.array_equal(pw[c], pg[c])


def _feed(hip, dec, pk):
    buf = A.BUF()
    hip.dsv_mk_buf(C.byref(buf), len(pk))  # exactly as long as the packet: nothing readable behind it
    C.memmove(buf.data, pk, len(pk))
    fp = C.POINTER(A.FRAME)()
    fn = C.c_uint32(0)
    code = hip.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
    if code == A.DEC_OK and fp:
        hip.dsv_frame_ref_dec(fp)
    return code


def _ueg(v):
    """bit string of the interleaved exp-Golomb code (bs.c:132)"""
    v += 1
    nb = v.bit_length() - 1
    return "".join("0" + str((v >> i) & 1) for i in range(nb - 1, -1, -1)) + "1"


def _meta_packet(fields):
    bits = "".join(_ueg(f) for f in fields) + "0"
    bits += "0" * (-len(bits) % 8)
    body = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    return b"DSV2" + bytes([8, 0x00]) + bytes(8) + body


def test_hostile_packets_return_errors_not_crashes():
    """The product only (the reference reads out of bounds on such input): truncated packets, absurd sub-stream and
    plane lengths, metadata the device pipeline cannot allocate -- every call returns a decoder code, the process
    survives and a clean stream decodes correctly afterwards on the same library."""
    ref, hip = A.load_ref(), A.load_hip()
    pkg = load_pkg()
    w, h = 352, 288
    v = pkg.synth.SynthVideo(w, h, "420", seed=12)
    frames = [v.frame_bytes(t) for t in range(3)]
    packets, _ = encode_stream(ref, frames, w, h, A.SUBSAMP_420, eos=False, qp=60, gop=48)
    meta, pic_i, pic_p = packets[0], packets[1], packets[2]
    rng = np.random.default_rng(5)
    codes = set()
    # (1) truncation at every kind of place
    for pk in (pic_i, pic_p):
        for cut in [15, 18, 20, 24, 30, 40, 64, 100, 200, len(pk) // 3, len(pk) // 2, len(pk) - 1]:
            dec = A.DECODER()
            assert _feed(hip, dec, meta) == A.DEC_GOT_META
            if pk is pic_p:
                assert _feed(hip, dec, pic_i) == A.DEC_OK
            codes.add(_feed(hip, dec, pk[:cut]))
            hip.dsv_dec_free(C.byref(dec))
    # (2) random garbage behind a valid picture header, and 0xff runs (huge exp-Golomb lengths)
    for k in range(24):
        dec = A.DECODER()
        assert _feed(hip, dec, meta) == A.DEC_GOT_META
        junk = bytes(rng.integers(0, 256, size=int(rng.integers(1, 4000)), dtype=np.uint8)) if k % 3 else bytes([0]) * 64 + bytes([0xff]) * 64
        codes.add(_feed(hip, dec, pic_i[:22 + k] + junk))
        hip.dsv_dec_free(C.byref(dec))
    # (3) metadata that must be refused: zero / odd / enormous sizes, unknown subsampling
    for fields in [(0, 288, 5, 30, 1, 1, 1, 1), (352, 0, 5, 30, 1, 1, 1, 1), (353, 288, 5, 30, 1, 1, 1, 1), (1 << 20, 1 << 20, 5, 30, 1, 1, 1, 1),
                   (352, 288, 3, 30, 1, 1, 1, 1), (352, 288, 0x3f, 30, 1, 1, 1, 1)]:
        dec = A.DECODER()
        assert _feed(hip, dec, _meta_packet(fields)) == A.DEC_GOT_META
        assert _feed(hip, dec, pic_i) == A.DEC_ERROR
        hip.dsv_dec_free(C.byref(dec))
    assert codes <= {A.DEC_OK, A.DEC_ERROR}
    # the library is still healthy
    from codec_run import decode_stream
    good = decode_stream(hip, packets)
    want = decode_stream(ref, packets)
    assert len(good) == len(want) == 3
    for a, b in zip(good, want):
        for c in (1, 2, 3):
            assert np.array_equal(a[c], b[c])
