"""Golden stream hashes (tests/golden/streams.json, produced by tests/make_golden.py with the real reference CLI)."""
import hashlib
import json
import os

import dsvabi as A
from codec_run import decode_stream, encode_stream
from conftest import load_pkg

GOLDEN = json.load(open(os.path.join(A.ROOT, "tests", "golden", "streams.json")))


def cli_equivalent_cfg(flags):
    """Translate the CLI flags of a golden entry into the library configuration (dsv_main.c:548-723)."""
    cfg = dict(qp=85, gop=-1, effort=10)
    eos = True
    for f in flags:
        k, v = f.lstrip("-").split("=")
        if k == "noeos":
            eos = not int(v)
        elif k == "kbps":
            cfg["bitrate"] = int(v) * 1024  # dsv_main.c:77 to_bps
        else:
            cfg[k] = int(v)
    return cfg, eos


def run_entry(lib, g):
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(g["w"], g["h"], g["fmt"], seed=g["seed"])
    frames = [v.frame_bytes(t) for t in range(g["n"])]
    assert hashlib.md5(b"".join(frames)).hexdigest() == g["input_md5"], "generator drifted"
    cfg, eos = cli_equivalent_cfg(g["flags"])
    subsamp = A.SUBSAMP_420 if g["fmt"] == "420" else A.SUBSAMP_444
    packets, _ = encode_stream(lib, frames, g["w"], g["h"], subsamp, eos=eos, **cfg)
    stream = b"".join(packets)
    dec = decode_stream(lib, packets)
    decoded = b"".join(p.tobytes() for (_, y, u, vv) in dec for p in (y, u, vv))
    return stream, decoded
