#!/usr/bin/env python3
"""Benchmark of the DSV2 encode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json north-star headline): synthetic 1920x1080 4:2:0, -qp=60 -gop=48, CRF,
effort 10 (quarter-pel + EPRM + in-loop filters).  Every rank drives S independent closed-GOP
streams on its GPU -- the reference's own segment-parallel recipe (parallel_encode_yuv.sh) -- as
G lockstep groups: each group is one host thread calling dsv2hip_enc_batch() for its S/G encoder
instances on one HIP stream, so a kernel launch serves a whole group and the groups overlap each
other's host and device phases.  A "step" is one frame of every stream: S frames per rank per
step.  Frames are resident in HBM before the timed region starts.  A default run (48 timed steps)
covers one full GOP of every stream: 1 I frame + 47 P frames each.

One JSON line is printed by rank 0: frames/s aggregated over all ranks (weak scaling), the
roofline object of the dominant kernel (HIP-event stage spans measured in a short extra pass of
the same configuration) and the CPU baseline (the real reference, oracle/_ref, one thread, bounded
sample) at N=1.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

# the lockstep groups each drive their own HIP stream; give them hardware queues of their own
# (ROCm maps streams onto 4 by default).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W_, H_ = 1920, 1080
GOP, QP = 48, 60
STAGES = ["ingest_pyramid", "hme", "predict_subtract", "fwd_sbt", "quant_compact", "inv_sbt", "recon_filters", "extend"]
# dominant-kernel name per stage (rocprofv3 --kernel-trace name prefix)
STAGE_KERNEL = {"hme": "k_hme_rows_b_fast_w2", "fwd_sbt": "k_fwd_haar/k_fwd_rows/k_fwd_cols", "inv_sbt": "k_inv_haar/k_inv_cols/k_inv_rows",
                "quant_compact": "k_quant_level", "recon_filters": "k_inter_filters", "predict_subtract": "k_predict_w",
                "ingest_pyramid": "k_extend/k_ds2x", "extend": "k_extend"}
N_PIX = W_ * H_
P_BYTES = N_PIX * 3 // 2
# algorithmic bytes per frame and stage, SURVEY.md section 8(d) (P-frame column)
STAGE_BYTES = {"ingest_pyramid": 2 * P_BYTES + 2.67 * N_PIX / 2, "hme": 5 * N_PIX, "predict_subtract": 4 * P_BYTES,
               "fwd_sbt": 5 * P_BYTES, "quant_compact": 8 * P_BYTES, "inv_sbt": 5 * P_BYTES, "recon_filters": 3 * P_BYTES + 2 * N_PIX,
               "extend": P_BYTES}
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSV2_STREAMS", "384")),
                    help="independent closed-GOP streams (encoder instances) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke-testing the N>1 path "
                                                      "on a box with fewer GPUs than ranks, together with DSV2_FORCE_DEVICE)")
    ap.add_argument("--no-profile", action="store_true", help="skip the extra stage-timing pass that feeds the roofline object")
    ap.add_argument("--profile-steps", type=int, default=6)
    ap.add_argument("--groups", type=int, default=int(os.environ.get("DSV2_GROUPS", "4")),
                    help="batch mode: split the streams into this many lockstep groups, one host thread + HIP stream each")
    ap.add_argument("--mode", choices=["batch", "threads"], default=os.environ.get("DSV2_BENCH_MODE", "batch"),
                    help="batch: lockstep dsv2hip_enc_batch over all streams; threads: one host thread + HIP stream per stream")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import dsvabi as A
    from codec_run import configure_encoder
    from conftest import load_pkg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # host phases run on a worker pool inside the library: share the box's cores between the ranks
    os.environ.setdefault("DSV2_HOST_THREADS", str(min(48, max(16, (os.cpu_count() or 16) // max(1, world)))))
    local = int(os.environ.get("DSV2_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    hip = A.load_hip()
    assert hip.dsv2hip_device_ok() == 0, "no HIP device: the product has no CPU path"
    hip.dsv2hip_set_device(local)
    hip.dsv2hip_prof_enable.argtypes = [C.c_int]
    hip.dsv2hip_prof_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    hip.dsv2hip_prof_read_units.argtypes = [C.POINTER(C.c_longlong)]
    hip.dsv2hip_enc_device_frame.argtypes = [C.POINTER(A.ENCODER), C.c_void_p, C.POINTER(A.BUF)]
    hip.dsv2hip_enc_device_frame.restype = C.c_int

    ncpu = os.cpu_count() or 8
    S = max(1, args.streams)
    K, Wm = args.steps, args.warmup
    total = Wm + K
    pkg = load_pkg()

    # distinct picture content per stream; frames pre-generated on the host, then parked in HBM
    nuniq = min(total + 8, 24)  # frames repeat ping-pong fashion beyond this to bound generation time
    nvid = min(S, 4)        # distinct videos; further streams start at a different frame of one of them
    vids = []
    for k in range(nvid):
        v = pkg.synth.SynthVideo(W_, H_, "420", seed=1 + rank * 8 + k)
        host = [np.frombuffer(v.frame_bytes(t), dtype=np.uint8) for t in range(nuniq)]
        vids.append([torch.from_numpy(h.copy()).cuda() for h in host])
    torch.cuda.synchronize()
    dev_frames = []
    for s in range(S):
        base, shift = vids[s % nvid], 5 * (s // nvid)
        dev_frames.append(base[shift % nuniq:] + base[:shift % nuniq])

    def frame_index(t):
        period = 2 * (nuniq - 1) if nuniq > 1 else 1
        k = t % period
        return k if k < nuniq else period - k

    meta = A.mk_meta(W_, H_, A.SUBSAMP_420)
    encs = []
    for s in range(S):
        e = A.ENCODER()
        configure_encoder(hip, e, meta, qp=QP, gop=GOP, effort=10)
        encs.append(e)
    out_bytes = [[] for _ in range(S)]
    barrier = threading.Barrier(S + 1)

    def worker(s, t0, t1, prof_phase):
        bufs = (A.BUF * 4)()
        barrier.wait()
        for t in range(t0, t1):
            n = hip.dsv2hip_enc_device_frame(C.byref(encs[s]), C.c_void_p(dev_frames[s][frame_index(t)].data_ptr()), bufs)
            for i in range(n):
                out_bytes[s].append(C.string_at(bufs[i].data, bufs[i].len))
                hip.dsv_buf_free(C.byref(bufs[i]))
        barrier.wait()

    hip.dsv2hip_enc_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch.restype = C.c_int
    encp = (C.POINTER(A.ENCODER) * S)(*[C.pointer(e) for e in encs])
    bbufs = (A.BUF * (4 * S))()
    bn = (C.c_int * S)()

    G = max(1, min(args.groups, S))
    group_of = [list(range(g, S, G)) for g in range(G)]

    def group_worker(g, t0, t1, gbar):
        ids = group_of[g]
        m = len(ids)
        gp = (C.POINTER(A.ENCODER) * m)(*[C.pointer(encs[s]) for s in ids])
        gb = (A.BUF * (4 * m))()
        gn = (C.c_int * m)()
        gbar.wait()
        for t in range(t0, t1):
            ptrs = (C.c_void_p * m)(*[dev_frames[s][frame_index(t)].data_ptr() for s in ids])
            hip.dsv2hip_enc_batch(m, gp, ptrs, gb, gn)
            for k, s in enumerate(ids):
                for i in range(gn[k]):
                    b = gb[4 * k + i]
                    out_bytes[s].append(C.string_at(b.data, b.len))
                    hip.dsv_buf_free(C.byref(b))
        gbar.wait()

    def run_phase_batch(t0, t1):
        gbar = threading.Barrier(G + 1)
        ths = [threading.Thread(target=group_worker, args=(g, t0, t1, gbar)) for g in range(G)]
        for th in ths:
            th.start()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        gbar.wait()
        gbar.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t_end = time.perf_counter()
        for th in ths:
            th.join()
        return t_end - t_start

    def run_phase(t0, t1):
        if args.mode == "batch":
            return run_phase_batch(t0, t1)
        ths = [threading.Thread(target=worker, args=(s, t0, t1, False)) for s in range(S)]
        for th in ths:
            th.start()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        barrier.wait()   # release the workers
        barrier.wait()   # all workers done (each dsv_enc call returns with its stream drained)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t_end = time.perf_counter()
        for th in ths:
            th.join()
        return t_end - t_start

    run_phase(0, Wm)                      # warm-up: first I frame + a few P frames, allocations, clocks
    hip.dsv2hip_prof_enable(0)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    elapsed = run_phase(Wm, total)        # timed: exactly K steps
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)

    # stage profile: a few more steps of the SAME configuration with HIP-event stage timing on (not timed)
    stage_ms, stage_launches, stage_units, prof_steps = None, None, None, 0
    if args.mode == "batch" and not args.no_profile:
        hip.dsv2hip_prof_enable(1)
        run_phase(total, total + args.profile_steps)
        if rank == 0:
            ms = (C.c_double * 8)()
            ln = (C.c_longlong * 8)()
            un = (C.c_longlong * 8)()
            fr = C.c_longlong(0)
            hip.dsv2hip_prof_read(ms, ln, C.byref(fr))
            hip.dsv2hip_prof_read_units(un)
            stage_ms, stage_launches, stage_units, prof_steps = list(ms), list(ln), list(un), fr.value
        hip.dsv2hip_prof_enable(0)

    # final ordered gather of the segment bytes (the only collective of the path)
    seg_bytes = sum(len(b) for s in range(S) for b in out_bytes[s])
    xdev = "cuda" if args.backend == "nccl" else "cpu"
    t_max = torch.tensor([elapsed], device=xdev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        # segment id = global stream index: rank-major, so the gathered file is the ordered concatenation
        segs = {rank * S + s: b"".join(out_bytes[s]) for s in range(S)}
        whole = pkg.sharding.gather_segments(dist, rank, world, segs, device=xdev)
        total_bytes = len(whole) if rank == 0 else 0
    else:
        total_bytes = seg_bytes
    elapsed = float(t_max.item())

    for s in range(S):
        hip.dsv_enc_free(C.byref(encs[s]))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    frames_total = S * K * world
    fps = frames_total / elapsed
    result = {
        "metric": "encoded frames/s, 1080p 4:2:0 qp=60 gop=48 (bit-exact .dsv)",
        "value": round(fps, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": round(1000.0 * elapsed / K, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/int32",
        "data": "synthetic",
        "config": {"workload": "1920x1080 4:2:0 -qp=60 -gop=48 effort=10 CRF, %d closed-GOP streams per GPU" % S,
                   "streams_per_gpu": S, "frames_per_step_per_gpu": S, "mode": args.mode, "groups": G, "host_cpu_cores_busy": round(host_cpu_s / elapsed, 2), "mpix_per_s": round(fps * N_PIX / 1e6, 1),
                   "stream_bytes_total": total_bytes, "host_cpus": ncpu},
    }
    if stage_ms is not None and prof_steps:
        # per stage: span (HIP events on the group's stream) per stream-frame, and the algorithmic
        # bytes of SURVEY.md 8(d) moved in that span
        per_unit = {STAGES[i]: (stage_ms[i] / stage_units[i] if stage_units[i] else 0.0) for i in range(8)}
        total_ms = {STAGES[i]: stage_ms[i] for i in range(8)}
        dom = max(total_ms, key=lambda k: total_ms[k])
        i = STAGES.index(dom)
        nl = max(1, stage_launches[i])
        avg_launch_ms = stage_ms[i] / nl
        bytes_per_launch = STAGE_BYTES[dom] * stage_units[i] / nl
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        # HBM traffic per launch of the dominant kernel: from the committed PMC passes (tools/profile_round.sh ->
        # profiles/pmc_traffic.json), valid for the configuration they were taken on
        traffic = None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("stage") == dom and pt.get("streams_per_gpu") == S and pt.get("groups") == G:
                traffic = pt.get("bytes_per_launch")
        except (OSError, ValueError):
            pass
        result["roofline"] = {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "stage": dom,
                              "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                              "avg_launch_us": round(avg_launch_ms * 1e3, 2),
                              "launches_per_step": round(nl / prof_steps * G, 1),
                              "algorithmic_bytes_per_launch": round(bytes_per_launch),
                              "stage_us_per_frame": {k: round(1e3 * v, 2) for k, v in per_unit.items()},
                              "whole_frame_algorithmic_GBps": round(104.0e6 * fps / 1e9, 1)}
    if not args.no_cpu_baseline and world == 1 and os.path.exists(A.REF_SO):
        from codec_run import encode_stream
        ref = A.load_ref()
        nfr = 24
        v = pkg.synth.SynthVideo(W_, H_, "420", seed=1)
        frames = [v.frame_bytes(t) for t in range(nfr)]
        t0 = time.perf_counter()
        encode_stream(ref, frames, W_, H_, A.SUBSAMP_420, qp=QP, gop=GOP, effort=10)
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(nfr / dt, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                                  "sample": "first %d frames (1 I + %d P) of stream 0, reference C library -O3, 1 thread" % (nfr, nfr - 1)}
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
