"""Deterministic synthetic 8-bit planar YUV video (SURVEY.md §8d "Synthetic inputs").

Integer-only numpy arithmetic (uint32 hashing, cumulative-sum box blur, LUT
sinusoid) so the bytes are identical on every machine and numpy version: the
golden fixtures under tests/golden/ were produced from exactly these frames.

Content per frame t:
  * a low-passed noise texture sampled from a 2x oversampled canvas at an offset
    of (3t, 2t) half-pels -> 1.5 px/frame horizontal, 1 px/frame vertical motion
    (exercises half/quarter-pel motion estimation),
  * an additive integer "sinusoid" (triangle-LUT) pattern,
  * six solid 96x96 squares (scaled with the frame) moving at distinct integer velocities,
  * +-2 LSB per-frame noise on the right half; the left quarter is static (skip blocks),
  * chroma = smooth drifting gradients (+ the squares' colours).
"""
import numpy as np

_M32 = np.uint32(0xFFFFFFFF)


def _hash2d(h, w, seed):
    """uint32 hash of (y, x, seed) -- a stateless stand-in for an LCG stream."""
    y = np.arange(h, dtype=np.uint32)[:, None]
    x = np.arange(w, dtype=np.uint32)[None, :]
    v = (x * np.uint32(73856093)) ^ (y * np.uint32(19349663)) ^ np.uint32((seed * 83492791) & 0xFFFFFFFF)
    v = v * np.uint32(1103515245) + np.uint32(12345)
    v ^= v >> np.uint32(15)
    v = v * np.uint32(2246822519)
    v ^= v >> np.uint32(13)
    v = v * np.uint32(3266489917)
    v ^= v >> np.uint32(16)
    return v


def _box_blur(a, r):
    """(2r+1)^2 box mean with edge clamp, integer, via cumulative sums."""
    a = a.astype(np.int64)
    p = np.pad(a, r, mode="edge")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    k = 2 * r + 1
    s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return (s // (k * k)).astype(np.int32)


_TRI = np.concatenate([np.arange(0, 32), np.arange(32, 0, -1)]).astype(np.int32) - 16  # 64-entry LUT


class SynthVideo:
    """Frame generator. subsamp: '420' or '444'."""

    def __init__(self, width, height, subsamp="420", seed=1):
        assert width % 2 == 0 and height % 2 == 0
        self.w, self.h, self.subsamp, self.seed = width, height, subsamp, seed
        self.cw = width // 2 if subsamp == "420" else width
        self.ch = height // 2 if subsamp == "420" else height
        pad = 256
        cw2, ch2 = 2 * width + pad * 2, 2 * height + pad * 2
        n = (_hash2d(ch2, cw2, seed) >> np.uint32(24)).astype(np.int32)  # 0..255
        lo = _box_blur(n, 6)
        mid = _box_blur(n, 1)
        # contrast-stretched low-pass + some mid-frequency detail
        self.canvas = np.clip(128 + (lo - 128) * 6 + (mid - 128) // 3, 16, 240).astype(np.int32)
        self.pad = pad
        sq = max(16, (96 * height) // 1080 // 2 * 2)
        self.sq = sq
        hs = _hash2d(6, 8, seed + 77)
        self.squares = []
        for k in range(6):
            x0 = int(hs[k, 0] % np.uint32(max(1, width - sq)))
            y0 = int(hs[k, 1] % np.uint32(max(1, height - sq)))
            vx = int(hs[k, 2] % np.uint32(9)) - 4
            vy = int(hs[k, 3] % np.uint32(7)) - 3
            if vx == 0 and vy == 0:
                vx = 1
            lum = (6, 250, 40 + int(hs[k, 4] % np.uint32(180)))[k % 3]  # extremes trigger EPRM
            cu = 64 + int(hs[k, 5] % np.uint32(128))
            cv = 64 + int(hs[k, 6] % np.uint32(128))
            self.squares.append((x0, y0, vx, vy, lum, cu, cv))

    def frame(self, t):
        """Returns (Y, U, V) uint8 arrays for frame number t."""
        w, h, pad = self.w, self.h, self.pad
        ox = pad + (3 * t) % pad
        oy = pad + (2 * t) % pad
        c = self.canvas[oy:oy + 2 * h, ox:ox + 2 * w]
        tex = (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + 2) >> 2
        # left quarter of the picture is static (skip blocks), the rest pans
        c0 = self.canvas[pad:pad + 2 * h, pad:pad + 2 * (w // 4)]
        tex[:, :w // 4] = (c0[0::2, 0::2] + c0[0::2, 1::2] + c0[1::2, 0::2] + c0[1::2, 1::2] + 2) >> 2
        yy = np.arange(h, dtype=np.int32)[:, None]
        xx = np.arange(w, dtype=np.int32)[None, :]
        sinus = _TRI[((xx * 3 + yy + 2 * t) >> 2) & 63]
        noise = (_hash2d(h, w, self.seed * 1000 + 17 + t) % np.uint32(5)).astype(np.int32) - 2
        noise[:, :w // 2] = 0  # only the right half carries temporal noise
        sinus = np.broadcast_to(sinus, (h, w)).copy()
        sinus[:, :w // 4] = _TRI[((xx[:, :w // 4] * 3 + yy) >> 2) & 63]
        Y = tex + sinus + noise
        cyy = np.arange(self.ch, dtype=np.int32)[:, None]
        cxx = np.arange(self.cw, dtype=np.int32)[None, :]
        sc = self.w // self.cw
        U = 128 + _TRI[((cxx * sc + 2 * t) >> 3) & 63] * 2 + _TRI[((cyy * sc) >> 4) & 63]
        V = 128 + _TRI[((cyy * sc + 3 * t) >> 3) & 63] * 2 - _TRI[((cxx * sc) >> 4) & 63]
        U = np.broadcast_to(U, (self.ch, self.cw)).copy()
        V = np.broadcast_to(V, (self.ch, self.cw)).copy()
        for (x0, y0, vx, vy, lum, cu, cv) in self.squares:
            span_x, span_y = w - self.sq, h - self.sq
            px = (x0 + vx * t) % (2 * span_x) if span_x > 0 else 0
            py = (y0 + vy * t) % (2 * span_y) if span_y > 0 else 0
            if px >= span_x:
                px = 2 * span_x - px
            if py >= span_y:
                py = 2 * span_y - py
            Y[py:py + self.sq, px:px + self.sq] = lum + (noise[py:py + self.sq, px:px + self.sq] >> 1)
            sy, sx = h // self.ch, w // self.cw
            U[py // sy:(py + self.sq) // sy, px // sx:(px + self.sq) // sx] = cu
            V[py // sy:(py + self.sq) // sy, px // sx:(px + self.sq) // sx] = cv
        return (np.clip(Y, 0, 255).astype(np.uint8),
                np.clip(U, 0, 255).astype(np.uint8),
                np.clip(V, 0, 255).astype(np.uint8))

    def frame_bytes(self, t):
        y, u, v = self.frame(t)
        return y.tobytes() + u.tobytes() + v.tobytes()

    def frame_size(self):
        return self.w * self.h + 2 * self.cw * self.ch


def write_y4m(path, video, nframes, fps=(30, 1)):
    tag = "C420jpeg" if video.subsamp == "420" else "C444"
    with open(path, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 %s\n" % (video.w, video.h, fps[0], fps[1], tag)).encode())
        for t in range(nframes):
            f.write(b"FRAME\n")
            f.write(video.frame_bytes(t))


def write_yuv(path, video, nframes):
    with open(path, "wb") as f:
        for t in range(nframes):
            f.write(video.frame_bytes(t))
