"""Drive the public codec API (dsv_enc / dsv_dec) of either library on in-memory frames."""
import ctypes as C

import numpy as np

import dsvabi as A


def configure_encoder(lib, enc, meta, qp=60, gop=48, effort=10, rc_mode=0, **over):
    """Field set-up as the reference CLI does it (dsv_main.c:548-723) for the given flags."""
    lib.dsv_enc_init(C.byref(enc))
    lib.dsv_enc_set_metadata(C.byref(enc), C.byref(meta))
    fps = (meta.fps_num + meta.fps_den // 2) // meta.fps_den
    enc.gop = fps if gop < 0 else gop
    enc.quality = qp * 4
    enc.effort = effort
    enc.rc_mode = rc_mode
    if rc_mode == 0:
        enc.min_quality = enc.quality - 20
        enc.min_I_frame_quality = enc.quality - 8
    else:
        enc.min_quality = 0
        enc.min_I_frame_quality = 20
    enc.max_quality = 400
    enc.min_quality = min(max(enc.min_quality, 0), 400)
    enc.min_I_frame_quality = min(max(enc.min_I_frame_quality, 0), 400)
    enc.min_q_step, enc.max_q_step = 2, 1
    enc.stable_refresh = min(max(fps, 1), 60)
    enc.bitrate = over.pop("bitrate", 4000000)
    for k, v in over.items():
        setattr(enc, k, v)
    lib.dsv_enc_start(C.byref(enc))


def encode_stream(lib, frames, w, h, subsamp, eos=True, **cfg):
    """frames: list of bytes (planar YUV). Returns (list of packet bytes, encoder struct)."""
    meta = A.mk_meta(w, h, subsamp)
    enc = A.ENCODER()
    configure_encoder(lib, enc, meta, **cfg)
    packets = []
    bufs = (A.BUF * 4)()
    keep = []
    for fb in frames:
        arr = np.frombuffer(fb, dtype=np.uint8).copy()
        keep.append(arr)
        fr = lib.dsv_load_planar_frame(subsamp, arr.ctypes.data, w, h)
        n = lib.dsv_enc(C.byref(enc), fr, bufs)
        for i in range(n):
            packets.append(bytes(C.string_at(bufs[i].data, bufs[i].len)))
            lib.dsv_buf_free(C.byref(bufs[i]))
    if eos:
        lib.dsv_enc_end_of_stream(C.byref(enc), bufs)
        packets.append(bytes(C.string_at(bufs[0].data, bufs[0].len)))
        lib.dsv_buf_free(C.byref(bufs[0]))
    stats = {k: getattr(enc.stats, k) for k in ("inum", "pnum", "isize", "psize", "eprm", "skip", "mbI", "mbP", "qpx", "hpx")}
    lib.dsv_enc_free(C.byref(enc))
    return packets, stats


def decode_stream(lib, packets):
    """Returns list of (fnum, Y, U, V) numpy planes."""
    dec = A.DECODER()
    out = []
    for pk in packets:
        buf = A.BUF()
        lib.dsv_mk_buf(C.byref(buf), len(pk) + 64)
        C.memmove(buf.data, pk, len(pk))
        fp = C.POINTER(A.FRAME)()
        fn = C.c_uint32(0)
        code = lib.dsv_dec(C.byref(dec), C.byref(buf), C.byref(fp), C.byref(fn))
        if code == A.DEC_OK and fp:
            f = fp.contents
            planes = []
            for c in range(3):
                p = f.planes[c]
                a = np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,))
                planes.append(a.reshape(-1, p.stride)[:p.h, :p.w].copy() if p.stride * p.h == a.size else None)
            out.append((fn.value, planes[0], planes[1], planes[2]))
            lib.dsv_frame_ref_dec(fp)
        elif code == A.DEC_EOS:
            break
        elif code == A.DEC_ERROR:
            raise RuntimeError("decoder error")
    lib.dsv_dec_free(C.byref(dec))
    return out
