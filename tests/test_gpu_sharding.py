"""The N>1 path with the product encoder on ONE GPU: two ranks (gloo, DSV2_FORCE_DEVICE=0) encode their closed-GOP
segments with libdsv2hip.so; the gathered bytes must equal the reference invoked per segment and concatenated
(parallel_encode_yuv.sh:36-50).  Also: `bench.py --gpus 2` starts its two ranks by itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

import dsvabi as A
import shard_worker as SW
from codec_run import encode_stream
from conftest import load_pkg

pytestmark = [pytest.mark.gpu]  # (a GPU box without oracle/_ref FAILS these tests: conftest.py)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_product_encoder_equals_reference_per_segment(tmp_path):
    world, port = 2, free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DSV2_FORCE_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py"), str(tmp_path)], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0] * world
    got = open(tmp_path / "gathered.dsv", "rb").read()
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(SW.W, SW.H, "420", seed=SW.SEED)
    ref = A.load_ref()
    want = b""
    nseg = (SW.NFRAMES + SW.GOP - 1) // SW.GOP
    for s in range(nseg):
        a, b = pkg.sharding.frame_range(s, SW.GOP, SW.NFRAMES)
        pk, _ = encode_stream(ref, [v.frame_bytes(t) for t in range(a, b)], SW.W, SW.H, A.SUBSAMP_420, eos=False, qp=SW.QP, gop=SW.GOP, effort=10)
        want += b"".join(pk)
    assert got == want


def test_bench_spawns_its_ranks():
    """python bench.py --gpus 2 (no launcher, no WORLD_SIZE) must start two ranks and report n_gpus = 2"""
    env = dict(os.environ, DSV2_FORCE_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(A.ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--streams", "8", "--groups", "2",
                        "--steps", "3", "--warmup", "1", "--no-extras", "--profile-steps", "2"], env=env, stdout=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2
    assert res["config"]["frames_timed"] == 2 * 8 * 3
    assert res["parity_checked"]["mismatches"] == 0 and res["parity_checked"]["twin_pairs_equal"] == 8
