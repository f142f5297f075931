"""One rank of the N>1 path with the PRODUCT encoder: used by tests/test_gpu_sharding.py (several ranks share one GPU
through DSV2_FORCE_DEVICE, torch.distributed over gloo).  Each rank encodes its closed-GOP segments with a fresh encoder
per segment (libdsv2hip.so, lockstep batch over its segments) and the bytes are gathered to rank 0 in segment order."""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

W, H, GOP, NFRAMES, QP, SEED = 352, 288, 4, 22, 60, 9


def main():
    outdir = sys.argv[1]
    import numpy as np
    import torch
    import torch.distributed as dist

    import dsvabi as A
    from codec_run import configure_encoder
    from conftest import load_pkg
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = int(os.environ.get("DSV2_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    hip = A.load_hip()
    assert hip.dsv2hip_device_ok() == 0
    hip.dsv2hip_set_device(dev)
    hip.dsv2hip_enc_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch.restype = C.c_int
    pkg = load_pkg()
    v = pkg.synth.SynthVideo(W, H, "420", seed=SEED)
    nseg = (NFRAMES + GOP - 1) // GOP
    mine = pkg.sharding.assign_segments(nseg, world)[rank]
    meta = A.mk_meta(W, H, A.SUBSAMP_420)
    encs = {s: A.ENCODER() for s in mine}
    for s in mine:
        configure_encoder(hip, encs[s], meta, qp=QP, gop=GOP, effort=10)
    out = {s: b"" for s in mine}
    for t in range(GOP):  # lockstep over this rank's segments: frame t of every segment that still has one
        live = [s for s in mine if pkg.sharding.frame_range(s, GOP, NFRAMES)[0] + t < pkg.sharding.frame_range(s, GOP, NFRAMES)[1]]
        if not live:
            break
        m = len(live)
        devf = [torch.from_numpy(np.frombuffer(v.frame_bytes(pkg.sharding.frame_range(s, GOP, NFRAMES)[0] + t), dtype=np.uint8).copy()).cuda() for s in live]
        torch.cuda.synchronize()
        gp = (C.POINTER(A.ENCODER) * m)(*[C.pointer(encs[s]) for s in live])
        ptrs = (C.c_void_p * m)(*[d.data_ptr() for d in devf])
        gb = (A.BUF * (4 * m))()
        gn = (C.c_int * m)()
        assert hip.dsv2hip_enc_batch(m, gp, ptrs, gb, gn) == 0
        for k, s in enumerate(live):
            for i in range(gn[k]):
                b = gb[4 * k + i]
                out[s] += C.string_at(b.data, b.len)
                hip.dsv_buf_free(C.byref(b))
    for s in mine:
        hip.dsv_enc_free(C.byref(encs[s]))
    whole = pkg.sharding.gather_segments(dist, rank, world, out)
    if rank == 0:
        open(os.path.join(outdir, "gathered.dsv"), "wb").write(whole)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
