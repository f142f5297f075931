#!/usr/bin/env python3
"""Debug aid: where does the batch decoder differ from the reference for a small 4:4:4 case."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import dsvabi as A  # noqa: E402
from codec_run import decode_stream, encode_stream  # noqa: E402
from conftest import load_pkg  # noqa: E402
from test_gpu_dec_batch import batch_decode, bind  # noqa: E402

ref, hip = A.load_ref(), A.load_hip()
bind(hip)
pkg = load_pkg()
for (w, h, fmt, sub, qp) in [(354, 290, "444", A.SUBSAMP_444, 70), (354, 290, "444", A.SUBSAMP_444, 100), (354, 290, "420", A.SUBSAMP_420, 70),
                             (352, 288, "444", A.SUBSAMP_444, 70)]:
    v = pkg.synth.SynthVideo(w, h, fmt, seed=90)
    frames = [v.frame_bytes(t) for t in range(3)]
    pk = encode_stream(ref, frames, w, h, sub, eos=True, qp=qp, gop=48)[0]
    want = decode_stream(ref, pk)
    for name, got in (("single", decode_stream(hip, pk)), ("batch", batch_decode(hip, [pk])[0])):
        for (fa, *pa), (fb, *pb) in zip(want, got):
            for c in range(3):
                d = np.argwhere(pa[c] != pb[c])
                if len(d):
                    print(w, h, fmt, qp, name, "frame", fa, "plane", c, "ndiff", len(d), "first", d[0], "rows", d[:, 0].min(), d[:, 0].max(),
                          "cols", d[:, 1].min(), d[:, 1].max())
print("done")
