"""End-to-end: the drop-in encoder / decoder API on the GPU vs the real reference library.
Bit-identical packets, bit-identical decoded pictures, cross-decoding both ways."""
import os

import numpy as np
import pytest

import dsvabi as A
from codec_run import decode_stream, encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def synth_frames(w, h, subsamp, n, seed):
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(w, h, "420" if subsamp == A.SUBSAMP_420 else "444", seed=seed)
    return [v.frame_bytes(t) for t in range(n)]


CASES = [
    # id, w, h, subsamp, nframes, cfg
    ("cif_intra", 352, 288, A.SUBSAMP_420, 5, dict(qp=85, gop=0)),
    ("cif_ip", 352, 288, A.SUBSAMP_420, 8, dict(qp=60, gop=6)),
    ("cif_ip_loweffort", 352, 288, A.SUBSAMP_420, 5, dict(qp=40, gop=12, effort=5)),
    ("cif_ip_effort3", 352, 288, A.SUBSAMP_420, 5, dict(qp=55, gop=12, effort=3)),
    ("cif_ip_effort7", 352, 288, A.SUBSAMP_420, 5, dict(qp=55, gop=12, effort=7)),
    ("odd_ip", 354, 290, A.SUBSAMP_420, 4, dict(qp=70, gop=12)),
    ("444_lossless", 320, 240, A.SUBSAMP_444, 3, dict(qp=100, gop=12)),
    ("cif_cqp", 352, 288, A.SUBSAMP_420, 4, dict(qp=50, gop=12, rc_mode=2)),
    ("cif_abr", 352, 288, A.SUBSAMP_420, 6, dict(qp=50, gop=12, rc_mode=1, bitrate=600000)),
    ("720p_ip", 1280, 720, A.SUBSAMP_420, 4, dict(qp=60, gop=48)),
    ("1080p_ip", 1920, 1080, A.SUBSAMP_420, 3, dict(qp=60, gop=48)),
    ("2160p_ip", 3840, 2160, A.SUBSAMP_420, 2, dict(qp=60, gop=48)),  # 32x32 blocks: general ME routine (operands staged in LDS), 512-thread ring sweep of the filters
]


@pytest.mark.parametrize("name,w,h,subsamp,n,cfg", CASES, ids=[c[0] for c in CASES])
def test_encode_decode_bit_exact(name, w, h, subsamp, n, cfg):
    ref, hip = A.load_ref(), A.load_hip()
    frames = synth_frames(w, h, subsamp, n, seed=len(name))
    pk_r, st_r = encode_stream(ref, frames, w, h, subsamp, **cfg)
    pk_h, st_h = encode_stream(hip, frames, w, h, subsamp, **cfg)
    assert len(pk_r) == len(pk_h)
    for i, (a, b) in enumerate(zip(pk_r, pk_h)):
        assert len(a) == len(b), "packet %d length %d vs %d" % (i, len(a), len(b))
        if a != b:
            d = next(k for k in range(len(a)) if a[k] != b[k])
            raise AssertionError("packet %d differs at byte %d of %d" % (i, d, len(a)))
    assert st_r == st_h
    dec_rr = decode_stream(ref, pk_r)
    dec_hh = decode_stream(hip, pk_h)
    assert len(dec_rr) == len(dec_hh) == n
    for (fa, ya, ua, va), (fb, yb, ub, vb) in zip(dec_rr, dec_hh):
        assert fa == fb
        assert np.array_equal(ya, yb) and np.array_equal(ua, ub) and np.array_equal(va, vb), "decoded frame %d" % fa
    if cfg.get("qp") == 100:
        cw, ch = (w, h) if subsamp == A.SUBSAMP_444 else (w // 2, h // 2)
        for t, (fn, y, u, v) in enumerate(dec_hh):
            src = np.frombuffer(frames[t], dtype=np.uint8)
            assert np.array_equal(y.ravel(), src[:w * h]), "lossless luma frame %d" % t
            assert np.array_equal(u.ravel(), src[w * h:w * h + cw * ch])


def test_frame_callback_sees_reference_pictures():
    """DSV_ENCODER.frame_callback(meta, orig, recon) (dsv_encoder.h:170, called at dsv_encoder.c:1299): same
    source and reconstructed pictures as the reference hands out, frame by frame."""
    import ctypes as C
    ref, hip = A.load_ref(), A.load_hip()
    w, h = 352, 288
    frames = synth_frames(w, h, A.SUBSAMP_420, 5, seed=9)
    CB = C.CFUNCTYPE(None, C.POINTER(A.META), C.POINTER(A.FRAME), C.POINTER(A.FRAME))

    def run(lib):
        seen = []

        def cb(meta, orig, recon):
            rec = []
            for fp in (orig, recon):
                f = fp.contents
                for c in range(3):
                    p = f.planes[c]
                    a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
                    rec.append(a.reshape(-1, p.stride)[:p.h, :p.w].copy())
            seen.append(rec)

        keep = CB(cb)
        pk, _ = encode_stream(lib, frames, w, h, A.SUBSAMP_420, qp=60, gop=3, frame_callback=C.cast(keep, C.c_void_p))
        return pk, seen

    pk_r, seen_r = run(ref)
    pk_h, seen_h = run(hip)
    assert pk_r == pk_h
    assert len(seen_r) == len(seen_h) == len(frames)
    for a, b in zip(seen_r, seen_h):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


OPTION_CASES = [
    dict(do_psy=0),
    dict(do_psy=0x0f),
    dict(do_scd=0),
    dict(do_inter_filter=0),
    dict(do_inter_filter=1, do_intra_filter=0),
    dict(skip_block_thresh=-1),
    dict(skip_block_thresh=8),
    dict(block_size_override_x=1, block_size_override_y=1),   # 32x32 blocks on a small picture
    dict(block_size_override_x=1, block_size_override_y=0),   # 32x16 blocks (dsv_encoder.c:1213-1220)
    dict(block_size_override_x=0, block_size_override_y=1),   # 16x32 blocks
    dict(variable_i_interval=1, gop=4),
    dict(do_temporal_aq=0, do_dark_intra_boost=0),
    dict(pyramid_levels=3),
    dict(scene_change_pct=20, intra_pct_thresh=30),
]


@pytest.mark.parametrize("opts", OPTION_CASES, ids=[",".join("%s=%s" % kv for kv in o.items()) for o in OPTION_CASES])
def test_encoder_option_matrix(opts):
    """Public DSV_ENCODER switches (dsv_encoder.h:68-188) away from their defaults: same packets as the reference."""
    ref, hip = A.load_ref(), A.load_hip()
    w, h = 352, 288
    frames = synth_frames(w, h, A.SUBSAMP_420, 6, seed=31)
    # a scene cut in the middle so that the scene-change / intra-refresh switches matter
    frames[4] = bytes(255 - b for b in frames[4])
    cfg = dict(qp=55, gop=12)
    cfg.update(opts)
    pk_r, st_r = encode_stream(ref, frames, w, h, A.SUBSAMP_420, **cfg)
    pk_h, st_h = encode_stream(hip, frames, w, h, A.SUBSAMP_420, **cfg)
    assert pk_r == pk_h
    assert st_r == st_h


@pytest.mark.parametrize("sharpen,fps", [(0, (30, 1)), (1, (25, 1)), (1, (60000, 1001))])
def test_metadata_variants_and_forced_metadata(sharpen, fps):
    """Stream metadata other than the defaults (no inter sharpening, other frame rates -> other refresh interval),
    dsv_enc_force_metadata() mid-stream, and dsv_get_metadata() on the decoder side."""
    import ctypes as C
    ref, hip = A.load_ref(), A.load_hip()
    w, h = 352, 288
    frames = synth_frames(w, h, A.SUBSAMP_420, 5, seed=17)

    def run(lib):
        meta = A.mk_meta(w, h, A.SUBSAMP_420, fps=fps, inter_sharpen=sharpen)
        enc = A.ENCODER()
        from codec_run import configure_encoder
        configure_encoder(lib, enc, meta, qp=60, gop=-1)
        packets, bufs, keep = [], (A.BUF * 4)(), []
        for t, fb in enumerate(frames):
            if t == 3:
                lib.dsv_enc_force_metadata(C.byref(enc))
            arr = np.frombuffer(fb, dtype=np.uint8).copy()
            keep.append(arr)
            fr = lib.dsv_load_planar_frame(A.SUBSAMP_420, arr.ctypes.data, w, h)
            n = lib.dsv_enc(C.byref(enc), fr, bufs)
            for i in range(n):
                packets.append(bytes(C.string_at(bufs[i].data, bufs[i].len)))
                lib.dsv_buf_free(C.byref(bufs[i]))
        lib.dsv_enc_free(C.byref(enc))
        return packets

    pk_r, pk_h = run(ref), run(hip)
    assert pk_r == pk_h
    # decoder side: metadata as parsed
    dec = A.DECODER()
    buf = A.BUF()
    hip.dsv_mk_buf(C.byref(buf), len(pk_h[0]) + 64)
    C.memmove(buf.data, pk_h[0], len(pk_h[0]))
    fp, fn = C.POINTER(A.FRAME)(), C.c_uint32(0)
    assert hip.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn)) == A.DEC_GOT_META
    m = hip.dsv_get_metadata(C.byref(dec)).contents
    assert (m.width, m.height, m.subsamp, m.fps_num, m.fps_den, m.inter_sharpen) == (w, h, A.SUBSAMP_420, fps[0], fps[1], sharpen)
    hip.dsv_dec_free(C.byref(dec))
