#!/usr/bin/env python3
"""Benchmark of the DSV2 encode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json north star): synthetic 1920x1080 4:2:0, -qp=60 -gop=48, CRF, effort 10
(quarter-pel + EPRM + in-loop filters).  Every rank drives S independent closed-GOP streams on its GPU -- the
reference's own segment-parallel recipe (parallel_encode_yuv.sh:31-52) -- as G lockstep groups: each group is
one host thread calling dsv2hip_enc_batch_host() for its S/G encoder instances, so a kernel launch serves a
whole group and the groups overlap each other's host and device phases.  A "step" is one frame of every
stream: S frames per rank per step.

What the timed region contains: the pictures start in PINNED HOST memory and every frame's
host-to-device upload happens inside the region (the next step's pictures go up on a copy stream under the
current step's kernels), then the whole encode including host controller and entropy coding, up to the
finished packets in host memory.  The streams' GOP phases are staggered (stream s begins s mod 48 steps
before the others, in an untimed pre-roll), so every step -- and any window of steps -- carries the
steady-state share of intra pictures (1/48 of the streams per step) instead of one all-intra step per GOP.

Bit-exactness is checked inside the run: a number of streams (two per lockstep group) are re-encoded by the
real reference (oracle/_ref, CPU) and compared byte for byte, and every stream has a twin with identical
input in ANOTHER lockstep group whose packets must be identical over the whole run.  A mismatch is fatal.

One JSON line is printed by rank 0: frames/s aggregated over all ranks (weak scaling), the roofline object of
the dominant kernel (HIP-event stage spans measured in a short extra pass of the same configuration), the CPU
baselines (the real reference on one thread, and 8 processes on closed-GOP streams as in
parallel_encode_yuv.sh) and -- at N=1 -- the other BASELINE.json configurations and the decoder.

`--gpus N` without a launcher starts the N ranks itself (one process per GPU, torch.distributed over
RCCL); under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.
"""
import argparse
import ctypes as C
import json
import os
import socket
import struct
import subprocess
import sys
import tempfile
import threading
import time

# the lockstep groups each drive their own HIP stream; give them hardware queues of their own
# (ROCm maps streams onto 4 by default).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

STAGES = ["ingest_pyramid", "hme", "predict_subtract", "fwd_sbt", "quant_compact", "inv_sbt", "recon_filters", "extend", "hme_level0"]
NST = len(STAGES)
# dominant-kernel name per stage (rocprofv3 --kernel-trace name prefix)
STAGE_KERNEL = {"hme": "k_hme_rows_b_*", "hme_level0": "k_hme_rows_b_fast_l0_w2", "fwd_sbt": "k_fwd_haar/k_fwd_rows/k_fwd_cols", "inv_sbt": "k_inv_haar/k_inv_cols/k_inv_rows",
                "quant_compact": "k_quant_level", "recon_filters": "k_inter_filters", "predict_subtract": "k_predict_w",
                "ingest_pyramid": "k_extend/k_ds2x", "extend": "k_extend"}
HBM_PEAK_GBS = 8000.0
NREF_STREAMS = 8   # streams re-encoded by the reference for the parity check (and the 8-process CPU baseline)
NREF_FRAMES = 56   # frames of each of them (crosses the GOP boundary at 48)


def stage_bytes(w, h, fmt):
    """algorithmic bytes per frame and stage, SURVEY.md section 8(d) (P-frame column)"""
    n = w * h
    p = n * 3 // 2 if fmt == "420" else n * 3
    # the search reads three luma pyramids (4 N) + chroma (N); its level-0 launch reads the three full-size lumas + chroma = 4 N of those 5 N
    return {"ingest_pyramid": 2 * p + 2.67 * n / 2, "hme": 5 * n, "hme_level0": 4 * n, "predict_subtract": 4 * p, "fwd_sbt": 5 * p,
            "quant_compact": 8 * p, "inv_sbt": 5 * p, "recon_filters": 3 * p + 2 * n, "extend": p}, 27 * p + 9.67 * n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSV2_STREAMS", "768")),
                    help="independent closed-GOP streams (encoder instances) per GPU")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("DSV2_GROUPS", "4")),
                    help="lockstep groups per GPU, one host thread + HIP stream each")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference CPU runs (also skips the parity check against them)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE.json configurations and the decode leg")
    ap.add_argument("--no-profile", action="store_true", help="skip the extra stage-timing pass that feeds the roofline object")
    ap.add_argument("--no-stagger", action="store_true", help="all streams start their GOP together (one all-intra step per GOP)")
    ap.add_argument("--device-resident", action="store_true",
                    help="pictures parked in HBM before the clock starts (kernel-side figure; NOT the SURVEY 8d metric)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for exercising the N>1 path "
                                                      "on a box with fewer GPUs than ranks, together with DSV2_FORCE_DEVICE)")
    ap.add_argument("--profile-steps", type=int, default=6)
    ap.add_argument("--gen-procs", type=int, default=-1,
                    help="helper processes that generate the synthetic pictures (default: the usable cores, 1 under a profiler: "
                         "forked children of a process with rocprofv3's tool library loaded can hang at exit)")
    return ap.parse_args()


def usable_cpus():
    """cores this process may actually use: affinity mask and cgroup quota, not os.cpu_count()"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def spawn_ranks(args):
    """--gpus N without a launcher: N fresh processes, one per GPU.  This parent never touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


# ---- synthetic pictures (generated in forked helpers BEFORE torch / HIP come up) ---------------------------
def _gen_video(spec):
    w, h, fmt, seed, nf = spec
    from conftest import load_pkg
    v = load_pkg().synth.SynthVideo(w, h, fmt, seed=seed)
    return [v.frame_bytes(t) for t in range(nf)]


def gen_videos(specs, nproc):
    import multiprocessing as mp
    if nproc <= 1 or len(specs) == 1:
        return [_gen_video(s) for s in specs]
    with mp.get_context("fork").Pool(min(nproc, len(specs))) as pool:
        return pool.map(_gen_video, specs)


class EncodeRun:
    """S encoder instances of one geometry in G lockstep groups, pictures in pinned host memory."""

    def __init__(self, hip, A, torch, w, h, fmt, qp, gop, effort, S, G, videos, stagger, device_resident=False):
        from codec_run import configure_encoder
        self.hip, self.A, self.torch = hip, A, torch
        self.w, self.h, self.fmt, self.qp, self.gop, self.effort = w, h, fmt, qp, gop, effort
        self.S, self.G = S, max(1, min(G, S))
        self.P = len(videos[0][0])
        self.NV, self.NF = len(videos), len(videos[0])
        self.device_resident = device_resident
        # pictures: one pinned host block per video (or, for the kernel-side figure, one HBM tensor)
        self.vbase, self._keep = [], []
        for frames in videos:
            if device_resident:
                import numpy as np
                t = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).cuda()
                self._keep.append(t)
                self.vbase.append(t.data_ptr())
            else:
                p = hip.dsv2hip_host_alloc(self.P * self.NF)
                assert p, "pinned host allocation failed"
                for i, fb in enumerate(frames):
                    C.memmove(p + i * self.P, fb, self.P)
                self.vbase.append(p)
        torch.cuda.synchronize()
        # stream s: twin pairs (2u, 2u+1) share their input and their GOP phase and sit in different groups
        self.video = [(s // 2) % self.NV for s in range(S)]
        self.shift = [2 * (((s // 2) // self.NV) % max(1, self.NF // 2)) for s in range(S)]
        self.R = gop if (stagger and gop > 1) else 0
        self.r0 = [(s // 2) % self.R if self.R else 0 for s in range(S)]
        self.group_of = [list(range(g, S, self.G)) for g in range(self.G)]
        subsamp = A.SUBSAMP_420 if fmt == "420" else A.SUBSAMP_444
        meta = A.mk_meta(w, h, subsamp)
        self.encs = []
        for s in range(S):
            e = A.ENCODER()
            configure_encoder(hip, e, meta, qp=qp, gop=gop, effort=effort)
            self.encs.append(e)
        self.out = [[] for _ in range(S)]  # per stream, per frame: list of packets (bytes)
        self.step = 0

    def frame_index(self, s, t):
        k = self.shift[s] + t
        period = 2 * (self.NF - 1) if self.NF > 1 else 1
        k %= period
        return k if k < self.NF else period - k

    def ptr(self, s, t):
        return self.vbase[self.video[s]] + self.P * self.frame_index(s, t)

    def _group_worker(self, g, g0, g1, bar):
        hip, A = self.hip, self.A
        ids_all = self.group_of[g]
        bar.wait()
        for step in range(g0, g1):
            ids = [s for s in ids_all if self.r0[s] <= step]
            m = len(ids)
            if not m:
                continue
            gp = (C.POINTER(A.ENCODER) * m)(*[C.pointer(self.encs[s]) for s in ids])
            gb = (A.BUF * (4 * m))()
            gn = (C.c_int * m)()
            cur = (C.c_void_p * m)(*[self.ptr(s, step - self.r0[s]) for s in ids])
            if self.device_resident:
                rc = hip.dsv2hip_enc_batch(m, gp, cur, gb, gn)
            else:
                nxt = (C.c_void_p * m)(*[self.ptr(s, step + 1 - self.r0[s]) for s in ids])
                rc = hip.dsv2hip_enc_batch_host(m, gp, cur, nxt, gb, gn)
            assert rc == 0
            for k, s in enumerate(ids):
                pk = []
                for i in range(gn[k]):
                    b = gb[4 * k + i]
                    pk.append(C.string_at(b.data, b.len))
                    hip.dsv_buf_free(C.byref(b))
                self.out[s].append(pk)
        bar.wait()

    def run(self, nsteps, dist=None):
        """advance every (started) stream by nsteps frames; returns the wall time bracketed by barrier + synchronize"""
        torch = self.torch
        g0, g1 = self.step, self.step + nsteps
        bar = threading.Barrier(self.G + 1)
        ths = [threading.Thread(target=self._group_worker, args=(g, g0, g1, bar)) for g in range(self.G)]
        for th in ths:
            th.start()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        bar.wait()
        bar.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t_end = time.perf_counter()
        for th in ths:
            th.join()
        self.step = g1
        return t_end - t_start

    def frames_in(self, g0, g1):
        return sum(max(0, g1 - max(g0, self.r0[s])) for s in range(self.S))

    def twins_equal(self):
        """every stream's packets == its twin's (same input, other lockstep group), over the whole run"""
        pairs = bad = 0
        for u in range(self.S // 2):
            pairs += 1
            if self.out[2 * u] != self.out[2 * u + 1]:
                bad += 1
        return pairs, bad

    def stream_bytes(self, s):
        return b"".join(p for fr in self.out[s] for p in fr)

    def free(self):
        for e in self.encs:
            self.hip.dsv_enc_free(C.byref(e))
        if not self.device_resident:
            for p in self.vbase:
                self.hip.dsv2hip_host_free(p)
        self._keep = []


class RefWorkers:
    """reference encodes on the host CPU (tools/ref_encode_worker.py): parity oracle + CPU baselines"""

    def __init__(self, jobs):
        # jobs: list of (w, h, fmt, seed, qp, gop, effort, [frame indices])
        self.tmp = tempfile.mkdtemp(prefix="dsv2bench")
        self.procs, self.paths = [], []
        env = dict(os.environ)
        env.pop("RANK", None)
        for i, (w, h, fmt, seed, qp, gop, effort, idx) in enumerate(jobs):
            path = os.path.join(self.tmp, "ref%d.bin" % i)
            cmd = [sys.executable, os.path.join(ROOT, "tools", "ref_encode_worker.py"), str(w), str(h), fmt, str(seed), str(qp), str(gop), str(effort),
                   path, ",".join(str(k) for k in idx)]
            self.procs.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env))
            self.paths.append(path)
        for p in self.procs:
            line = p.stdout.readline().strip()
            assert line == "ready", "reference worker failed to start: %r" % line

    def go(self, which, n):
        for i in which:
            self.procs[i].stdin.write("go %d\n" % n)
            self.procs[i].stdin.flush()
        return [json.loads(self.procs[i].stdout.readline()) for i in which]

    def frames(self, i):
        data, out, off = open(self.paths[i], "rb").read(), [], 0
        while off < len(data):
            (n,) = struct.unpack_from("<I", data, off)
            out.append(data[off + 4:off + 4 + n])
            off += 4 + n
        return out

    def close(self):
        for p in self.procs:
            try:
                p.stdin.write("quit\n")
                p.stdin.flush()
            except OSError:
                pass
            p.wait()
        for path in self.paths:
            if os.path.exists(path):
                os.unlink(path)
        os.rmdir(self.tmp)


def decode_leg(hip, A, run, g0, nsteps, nstreams, groups):
    """lockstep batch decoder over the packets the encode run produced for global steps [g0, g0 + nsteps)"""
    hip.dsv2hip_dec_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.DECODER)), C.POINTER(A.BUF), C.POINTER(C.POINTER(A.FRAME)),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    hip.dsv2hip_dec_batch.restype = C.c_int
    D = min(nstreams, run.S)
    G = max(1, min(groups, D))
    decs = [A.DECODER() for _ in range(D)]
    group_of = [list(range(g, D, G)) for g in range(G)]
    decoded = [0] * G

    def feed(g, ids, packets):
        m = len(ids)
        bufs = (A.BUF * m)()
        for i, pk in enumerate(packets):
            hip.dsv_mk_buf(C.byref(bufs[i]), len(pk) + 64)
            C.memmove(bufs[i].data, pk, len(pk))
            bufs[i].len = len(pk)
        decp = (C.POINTER(A.DECODER) * m)(*[C.pointer(decs[s]) for s in ids])
        outs = (C.POINTER(A.FRAME) * m)()
        fns = (C.c_uint32 * m)()
        rets = (C.c_int * m)()
        hip.dsv2hip_dec_batch(m, decp, bufs, outs, fns, rets)
        for i in range(m):
            if rets[i] == A.DEC_OK and outs[i]:
                decoded[g] += 1
                hip.dsv_frame_ref_dec(outs[i])

    def worker(g, t0, t1, bar):
        ids = group_of[g]
        bar.wait()
        for t in range(t0, t1):
            # a stream decodes from its own first frame on: local frame index t
            live = [s for s in ids if t < len(run.out[s])]
            with_meta = [s for s in live if len(run.out[s][t]) > 1]
            if with_meta:
                feed(g, with_meta, [run.out[s][t][0] for s in with_meta])
            if live:
                feed(g, live, [run.out[s][t][-1] for s in live])
        bar.wait()

    def phase(t0, t1):
        bar = threading.Barrier(G + 1)
        ths = [threading.Thread(target=worker, args=(g, t0, t1, bar)) for g in range(G)]
        for th in ths:
            th.start()
        ts = time.perf_counter()
        bar.wait()
        bar.wait()
        te = time.perf_counter()
        for th in ths:
            th.join()
        return te - ts

    nfr = min(len(run.out[s]) for s in range(D))
    warm = min(4, max(1, nfr - nsteps))
    phase(0, warm)
    before = sum(decoded)
    elapsed = phase(warm, min(nfr, warm + nsteps))
    n = sum(decoded) - before
    for d in decs:
        hip.dsv_dec_free(C.byref(d))
    return {"value": round(n / elapsed, 2), "unit": "frames/s", "frames": n, "decoders": D, "groups": G,
            "mpix_per_s": round(n / elapsed * run.w * run.h / 1e6, 1),
            "note": "lockstep batch decoder (dsv2hip_dec_batch) over this run's own packets, pictures delivered to host memory as DSV_FRAMEs"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        sys.stderr.write("[bench] --gpus %d but WORLD_SIZE=%d: running %d ranks\n" % (args.gpus, world, world))
    ncpu = usable_cpus()
    # host phases run on a worker pool inside the library: share the usable cores between the ranks
    os.environ.setdefault("DSV2_HOST_THREADS", str(min(48, max(8, 3 * ncpu // max(1, world)))))
    local = int(os.environ.get("DSV2_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    extras = not args.no_extras and world == 1

    # ---- pictures first: forked generators must not inherit an initialised GPU runtime ----
    W_, H_, GOP, QP = 1920, 1080, 48, 60
    NV, NF = (8, 32) if args.streams >= 16 else (max(1, min(4, args.streams // 2)), 24)
    specs = [(W_, H_, "420", 1 + rank * NV + k, NF) for k in range(NV)]
    if extras:
        specs += [(1280, 720, "420", 101 + k, 24) for k in range(4)]
        specs += [(W_, H_, "444", 201, 12)]
    t_gen = time.perf_counter()
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ)
    gen_procs = args.gen_procs if args.gen_procs > 0 else (1 if under_profiler else max(1, min(8, ncpu // max(1, world))))
    vids = gen_videos(specs, gen_procs)
    t_gen = time.perf_counter() - t_gen

    import torch
    import dsvabi as A
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
    hip.dsv2hip_enc_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(A.BUF), C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch.restype = C.c_int
    hip.dsv2hip_enc_batch_host.argtypes = [C.c_int, C.POINTER(C.POINTER(A.ENCODER)), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(A.BUF),
                                           C.POINTER(C.c_int)]
    hip.dsv2hip_enc_batch_host.restype = C.c_int
    hip.dsv2hip_host_alloc.argtypes = [C.c_size_t]
    hip.dsv2hip_host_alloc.restype = C.c_void_p
    hip.dsv2hip_host_free.argtypes = [C.c_void_p]
    from conftest import load_pkg
    pkg = load_pkg()

    S, K, Wm = max(1, args.streams), args.steps, args.warmup
    effort = int(os.environ.get("DSV2_BENCH_EFFORT", "10"))  # (experiments only: the headline is effort 10)
    run = EncodeRun(hip, A, torch, W_, H_, "420", QP, GOP, effort, S, args.groups, vids[:NV], not args.no_stagger, args.device_resident)
    G = run.G
    hip.dsv2hip_prof_enable(0)
    run.run(run.R + Wm)                   # untimed: GOP-phase pre-roll + warm-up (allocations, clocks)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    g_timed = run.step
    elapsed = run.run(K, dist)            # timed: exactly K steps
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    frames_rank = run.frames_in(g_timed, g_timed + K)
    intra_rank = sum(1 for s in range(S) for t in range(g_timed - run.r0[s], g_timed + K - run.r0[s]) if t % GOP == 0)

    # stage profile: a few more steps of the SAME configuration with HIP-event stage timing on (not timed)
    stage_ms, stage_launches, stage_units, prof_steps = None, None, None, 0
    if not args.no_profile:
        hip.dsv2hip_prof_enable(1)
        run.run(args.profile_steps)
        if rank == 0:
            ms = (C.c_double * 16)()
            ln = (C.c_longlong * 16)()
            un = (C.c_longlong * 16)()
            fr = C.c_longlong(0)
            hip.dsv2hip_prof_read(ms, ln, C.byref(fr))
            hip.dsv2hip_prof_read_units(un)
            stage_ms, stage_launches, stage_units, prof_steps = list(ms), list(ln), list(un), fr.value
        hip.dsv2hip_prof_enable(0)

    # ---- parity, part 1 (every rank): twin streams in different lockstep groups produced identical packets ----
    pairs, bad = run.twins_equal()
    if bad:
        sys.stderr.write("[bench] rank %d: %d of %d twin stream pairs DIFFER -- output is not deterministic\n" % (rank, bad, pairs))
        sys.exit(3)

    # final ordered gather of the segment bytes (the only collective of the path)
    xdev = "cuda" if args.backend == "nccl" else "cpu"
    t_max = torch.tensor([elapsed], device=xdev, dtype=torch.float64)
    counts = torch.tensor([frames_rank, intra_rank, pairs], device=xdev, dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        # segment id = global stream index: rank-major, so the gathered file is the ordered concatenation
        segs = {rank * S + s: run.stream_bytes(s) for s in range(S)}
        whole = pkg.sharding.gather_segments(dist, rank, world, segs, device=xdev)
        total_bytes = len(whole) if rank == 0 else 0
        del whole, segs
    else:
        total_bytes = sum(len(p) for s in range(S) for fr in run.out[s] for p in fr)
    elapsed = float(t_max.item())
    frames_total, intra_total, pairs_total = (int(x) for x in counts.tolist())

    if rank != 0:
        run.free()
        if dist is not None:
            dist.destroy_process_group()
        return

    fps = frames_total / elapsed
    sb, frame_bytes = stage_bytes(W_, H_, "420")
    result = {
        "metric": "encoded frames/s, 1080p 4:2:0 qp=60 gop=48, incl. H2D upload of every frame (bit-exact .dsv)",
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
        "config": {"workload": "1920x1080 4:2:0 -qp=60 -gop=48 effort=%d CRF, %d closed-GOP streams per GPU in %d lockstep groups; "
                               "pictures in %s; GOP phases %s" %
                               (effort, S, G, "HBM before the clock starts (kernel-side figure)" if args.device_resident else
                                "pinned host memory, every frame uploaded inside the timed region (double-buffered copy stream)",
                                "staggered over %d untimed pre-roll steps: every step codes 1/%d of the streams as intra pictures" % (run.R, GOP)
                                if run.R else "aligned: one all-intra step per GOP"),
                   "streams_per_gpu": S, "frames_per_step_per_gpu": S, "groups": G, "frames_timed": frames_total, "intra_frames_timed": intra_total,
                   "input": "pinned_host" if not args.device_resident else "device_resident", "h2d_bytes_per_step_per_gpu": 0 if args.device_resident else S * run.P,
                   "distinct_videos_per_gpu": NV, "unique_frames_per_video": NF,
                   "host_cpu_cores_busy": round(host_cpu_s / elapsed, 2), "mpix_per_s": round(fps * W_ * H_ / 1e6, 1),
                   "stream_bytes_total": total_bytes, "host_cpus_usable": ncpu, "host_threads": int(os.environ["DSV2_HOST_THREADS"]),
                   "setup_s": {"generate_pictures": round(t_gen, 1)}},
        "parity_checked": {"twin_pairs_equal": pairs_total, "twin_pairs": pairs_total,
                           "note": "twins = same input, different lockstep group, compared over every packet of the run"},
    }
    if stage_ms is not None and prof_steps:
        # per stage: span (HIP events on the group's stream) per stream-frame, and the algorithmic
        # bytes of SURVEY.md 8(d) moved in that span
        per_unit = {STAGES[i]: (stage_ms[i] / stage_units[i] if stage_units[i] else 0.0) for i in range(NST)}
        total_ms = {STAGES[i]: stage_ms[i] for i in range(8)}
        dom = max(total_ms, key=lambda k: total_ms[k])
        if dom == "hme":
            dom = "hme_level0"  # the dominant KERNEL: the level-0 launch of the search (its span is measured on its own)
        i = STAGES.index(dom)
        nl = max(1, stage_launches[i])
        avg_launch_ms = stage_ms[i] / nl
        bytes_per_launch = sb[dom] * stage_units[i] / nl
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        # HBM traffic per launch of the dominant kernel cannot be read inside this process: it comes from the separate
        # rocprofv3 --pmc passes of tools/profile_round.sh (profiles/pmc_traffic.json) and is only quoted for the
        # configuration those passes were taken on
        traffic, traffic_source = None, None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("stage") == dom and pt.get("streams_per_gpu") == S and pt.get("groups") == G and pt.get("kernel") == STAGE_KERNEL[dom]:
                traffic = pt.get("bytes_per_launch")
                traffic_source = "committed PMC passes, not this run: " + pt.get("source", "profiles/pmc_traffic.json")
        except (OSError, ValueError):
            pass
        result["roofline"] = {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "stage": dom,
                              "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source,
                              "avg_launch_us": round(avg_launch_ms * 1e3, 2),
                              "launches_per_step": round(nl / prof_steps * G, 1),
                              "algorithmic_bytes_per_launch": round(bytes_per_launch),
                              "stage_us_per_frame": {k: round(1e3 * v, 2) for k, v in per_unit.items()},
                              "whole_frame_algorithmic_GBps": round(frame_bytes * fps / 1e9, 1)}

    # ---- parity, part 2 + CPU baselines: the real reference (oracle/_ref) on the host cores ----
    if not args.no_cpu_baseline and os.path.exists(A.REF_SO):
        # two streams per lockstep group, early GOP phases (they have run longest)
        sel = []
        for u in range(min(NREF_STREAMS, S // 2 if S > 1 else 1)):
            s = 2 * u + ((u >> 1) & 1)
            sel.append(s if s < S else 2 * u)
        nfr = [min(NREF_FRAMES, len(run.out[s])) for s in sel]
        jobs = [(W_, H_, "420", 1 + rank * NV + run.video[s], QP, GOP, effort, [run.frame_index(s, t) for t in range(nfr[i])]) for i, s in enumerate(sel)]
        try:
            rw = RefWorkers(jobs)
            one = rw.go([0], min(24, nfr[0]))[0]             # one reference thread, alone on the box
            allr = rw.go(list(range(len(sel))), max(nfr))    # all workers at once (each stops at its own frame count)
            mism = []
            for i, s in enumerate(sel):
                want = rw.frames(i)
                got = [b"".join(fr) for fr in run.out[s][:len(want)]]
                if want != got:
                    first = next((t for t, (a, b) in enumerate(zip(want, got)) if a != b), min(len(want), len(got)))
                    mism.append((s, first))
            rw.close()
        except (OSError, AssertionError, ValueError) as e:  # the reference side failed: say so beside the headline
            result["parity_checked"]["vs_reference_error"] = repr(e)
            print(json.dumps(result))
            sys.exit(5)
        result["parity_checked"].update({"vs_reference_streams": len(sel), "vs_reference_frames_each": nfr, "streams": sel,
                                         "groups_covered": sorted(set(s % G for s in sel)), "mismatches": len(mism)})
        result["cpu_baseline"] = {"value": round(one["frames"] / (one["t1"] - one["t0"]), 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                                  "sample": "first %d frames (1 I + %d P) of stream %d, reference C library -O3, 1 thread, alone on the box"
                                            % (one["frames"], one["frames"] - 1, sel[0])}
        span = max(r["t1"] for r in allr) - min(r["t0"] for r in allr)
        result["cpu_baseline_8proc"] = {"value": round(sum(r["frames"] for r in allr) / span, 3), "unit": "frames/s", "cores": len(sel), "kind": "reference",
                                        "sample": "%d reference processes at once, one closed-GOP stream each (%s frames), as parallel_encode_yuv.sh does"
                                                  % (len(sel), "/".join(str(r["frames"]) for r in allr))}
        if mism:
            print(json.dumps(result))
            sys.stderr.write("[bench] MISMATCH against the reference: (stream, first differing frame) = %s\n" % mism)
            sys.exit(4)

    # ---- the decoder on this run's packets, and the other BASELINE.json configurations (N = 1 only) ----
    # (the headline above is complete: whatever goes wrong below is reported beside it, never instead of it)
    if extras:
        try:
            result["decode"] = decode_leg(hip, A, run, 0, 32, 256, 4)
        except Exception as e:  # noqa: BLE001
            result["decode"] = {"error": repr(e)}
    run.free()
    if extras:
        try:
            result["configs"] = other_configs(hip, A, torch, args, vids, NV, S, W_, H_)
        except Exception as e:  # noqa: BLE001
            result["configs"] = {"error": repr(e)}
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def other_configs(hip, A, torch, args, vids, NV, S, W_, H_):
    """the other BASELINE.json configurations, each a short run of the same engine (N = 1 only)"""
    cfgs = {}
    # C2: 1280x720 4:2:0 -qp=60 -gop=48 effort 10
    r2 = EncodeRun(hip, A, torch, 1280, 720, "420", 60, 48, 10, S, args.groups, vids[NV:NV + 4], not args.no_stagger)
    r2.run(r2.R + 4)
    k2 = 24
    g2 = r2.step
    e2 = r2.run(k2)
    p2, b2 = r2.twins_equal()
    f2 = r2.frames_in(g2, g2 + k2)
    cfgs["c2_720p_420_qp60_gop48"] = {"value": round(f2 / e2, 2), "unit": "frames/s", "streams": S, "steps": k2, "ms_per_step": round(1e3 * e2 / k2, 3),
                                       "mpix_per_s": round(f2 / e2 * 1280 * 720 / 1e6, 1), "twin_pairs_equal": p2 - b2, "twin_pairs": p2,
                                       "input": "pinned_host, staggered GOP phases"}
    r2.free()
    # C3: 1920x1080 4:2:0 -qp=60 -gop=60 (the headline's geometry with the CLI's default GOP)
    r3 = EncodeRun(hip, A, torch, W_, H_, "420", 60, 60, 10, S, args.groups, vids[:NV], not args.no_stagger)
    r3.run(r3.R + 4)
    k3 = 16
    g3 = r3.step
    e3 = r3.run(k3)
    p3, b3 = r3.twins_equal()
    f3 = r3.frames_in(g3, g3 + k3)
    cfgs["c3_1080p_420_qp60_gop60"] = {"value": round(f3 / e3, 2), "unit": "frames/s", "streams": S, "steps": k3, "ms_per_step": round(1e3 * e3 / k3, 3),
                                        "mpix_per_s": round(f3 / e3 * W_ * H_ / 1e6, 1), "twin_pairs_equal": p3 - b3, "twin_pairs": p3,
                                        "input": "pinned_host, staggered GOP phases"}
    r3.free()
    # C4: 1920x1080 4:4:4 lossless (-qp=100), every stream from its first (intra) frame; round trip through the decoder
    s4 = min(32, S)
    r4 = EncodeRun(hip, A, torch, W_, H_, "444", 100, 60, 10, s4, 2, vids[NV + 4:NV + 5], False)
    r4.run(2)
    k4 = 8
    e4 = r4.run(k4)
    cfgs["c4_1080p_444_lossless"] = {"value": round(s4 * k4 / e4, 2), "unit": "frames/s", "streams": s4, "steps": k4, "ms_per_step": round(1e3 * e4 / k4, 3),
                                      "mpix_per_s": round(s4 * k4 / e4 * W_ * H_ / 1e6, 1), "frames": "P frames 2..9 of each stream (general ME routine)",
                                      "round_trip": lossless_round_trip(hip, A, r4, vids[NV + 4])}
    r4.free()
    return cfgs


def lossless_round_trip(hip, A, run, frames):
    """decode stream 0 of a lossless run with the GPU decoder and compare every picture with its source"""
    import numpy as np
    from codec_run import decode_stream
    packets = [p for fr in run.out[0] for p in fr]
    dec = decode_stream(hip, packets)
    ok = 0
    for t, (_, y, u, v) in enumerate(dec):
        src = frames[run.frame_index(0, t)]
        got = y.tobytes() + u.tobytes() + v.tobytes()
        ok += int(got == src)
    return {"frames": len(dec), "identical_to_source": ok}


if __name__ == "__main__":
    main()
