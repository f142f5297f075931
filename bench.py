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
STAGE_KERNEL = {"hme": "k_hme_rows_*", "hme_level0": "k_hme_rows_l0", "fwd_sbt": "k_fwd_haar/k_fwd_rows/k_fwd_cols", "inv_sbt": "k_inv_haar/k_inv_cols/k_inv_rows",
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
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSV2_STREAMS", "768")),
                    help="independent closed-GOP streams (encoder instances) per GPU")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("DSV2_GROUPS", "4")),
                    help="lockstep groups per GPU, one host thread + HIP stream each")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference CPU runs (also skips the parity check against them)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE.json configurations and the decode leg")
    ap.add_argument("--no-profile", action="store_true", help="skip the extra stage-timing pass that feeds the roofline object")
    ap.add_argument("--no-stagger", action="store_true", help="all streams start their GOP together (one all-intra step per GOP)")
    ap.add_argument("--no-phase-align", action="store_true",
                    help="spread every step's intra pictures over all lockstep groups (round 2's layout) instead of giving them to one group")
    ap.add_argument("--host-cores", type=int, default=0,
                    help="pin this rank to N of its usable cores (sched_setaffinity, after the pictures are generated and before anything "
                         "touches the GPU) and size the library's worker pool from them: does the host side fit N cores per GPU?")
    ap.add_argument("--no-mix", action="store_true", help="every stream pans (round 2's content): no scene-cut / static / fast-motion classes")
    ap.add_argument("--no-batch-curve", action="store_true", help="skip the small-batch operating points (1 / 8 / 48 / 192 streams)")
    ap.add_argument("--only-batch-curve", action="store_true", help="of the extras, run only the small-batch operating points (experiments)")
    ap.add_argument("--no-host-share", action="store_true", help="skip the 2-host-cores re-run of the headline")
    ap.add_argument("--no-api-legs", action="store_true", help="skip the legs through the reference's own entry points (threads of dsv_enc / dsv_dec, 8 drop-in CLI processes)")
    ap.add_argument("--only-api-legs", action="store_true", help="of the extras, run only those legs (experiments)")
    ap.add_argument("--device-resident", action="store_true",
                    help="pictures parked in HBM before the clock starts (kernel-side figure; NOT the SURVEY 8d metric)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for exercising the N>1 path "
                                                      "on a box with fewer GPUs than ranks, together with DSV2_FORCE_DEVICE)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the segment gather even with ONE rank "
                    "(exercises RCCL init + collectives on a single GPU)")
    ap.add_argument("--no-multi-rank", action="store_true", help="skip the 8-ranks-on-one-GPU leg")
    ap.add_argument("--decode-too", action="store_true", help="with --no-extras: still run the decode leg on the run's packets (the 2-host-core re-run uses it)")
    ap.add_argument("--profile-steps", type=int, default=6)
    ap.add_argument("--gen-procs", type=int, default=-1,
                    help="helper processes that generate the synthetic pictures (default: the usable cores, 1 under a profiler: "
                         "forked children of a process with rocprofv3's tool library loaded can hang at exit)")
    return ap.parse_args()


def under_profiler_():
    """rocprofv3's tool library is loaded: picture generation stays in this process (forked children can hang at exit), so the content is the small set"""
    return "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ)


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


# ---- where a rank's host side should run: the cores and memory node next to ITS GPU --------------------------------------
def _parse_cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def gpu_host_locality(ordinal, sysfs="/sys"):
    """(pci address, numa node, local cpus) of HIP device `ordinal`, read from sysfs WITHOUT touching the GPU runtime: the KFD
    topology lists the GPU nodes in the order the runtime enumerates them (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES, when they
    are plain index lists, select from that order).  None when the box does not expose it."""
    try:
        base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
        gpus = []
        for n in sorted(os.listdir(base), key=int):
            props = dict(l.split(None, 1) for l in open(os.path.join(base, n, "properties")).read().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
                gpus.append("%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v and all(x.strip().isdigit() for x in v.split(",")):
                gpus = [gpus[int(x)] for x in v.split(",") if int(x) < len(gpus)]
        addr = gpus[ordinal]
        dev = os.path.join(sysfs, "bus/pci/devices", addr)
        node = int(open(os.path.join(dev, "numa_node")).read().strip())
        cpus = _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
        return addr, node, cpus
    except (OSError, ValueError, IndexError, KeyError):
        return None


def pci_locality(addr, sysfs="/sys"):
    """(numa node, local cpus) of the PCI device `addr` ("dddd:bb:dd.f"), or None"""
    try:
        dev = os.path.join(sysfs, "bus/pci/devices", addr)
        return int(open(os.path.join(dev, "numa_node")).read().strip()), _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
    except (OSError, ValueError):
        return None


def bind_rank_late(locality, world, torch, ordinal):
    """Containers that hide the KFD topology (this pool's do: PermissionError on the GPU nodes' properties) leave the runtime as the
    only source of a GPU's PCI address.  Asked AFTER it is up -- still ahead of everything that matters: the pinned pictures, the
    library's worker pool and the lockstep groups' threads are all made later and inherit this thread's cores."""
    if locality["pci"] is not None:
        return locality
    try:
        pr = torch.cuda.get_device_properties(ordinal)
        addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except (AttributeError, RuntimeError):
        return locality
    loc = pci_locality(addr)
    locality = dict(locality, pci=addr, source="runtime (KFD topology not readable)")
    if loc is None:
        return locality
    node, cpus = loc
    locality["numa_node"] = node
    mine = sorted(cpus & os.sched_getaffinity(0))
    if world > 1 and mine and os.environ.get("DSV2_NUMA_BIND", "1") != "0":
        os.sched_setaffinity(0, mine)  # (ranks that share a node share all of its cores here: their positions are not known without the topology)
        locality.update(cpus=len(mine), bound=True)
    return locality


def bind_rank_to_gpu_node(ordinal, world):
    """sched_setaffinity to the usable cores next to GPU `ordinal` (parallel_encode_yuv.sh's processes run wherever the scheduler
    puts them; here a rank pins ~0.8 GB of pictures and moves ~25 GB/s over ITS GPU's PCIe link: both want the local node).
    Must run before anything allocates pinned memory or starts the library's worker pool.  Returns what was done, for the line."""
    info = {"pci": None, "numa_node": None, "cpus": len(os.sched_getaffinity(0)), "bound": False}
    loc = gpu_host_locality(ordinal)
    if loc is None:
        return info
    addr, node, cpus = loc
    info.update(pci=addr, numa_node=node)
    mine = sorted(cpus & os.sched_getaffinity(0))
    if world > 1 and mine and os.environ.get("DSV2_NUMA_BIND", "1") != "0":
        # ranks that share a node share its cores evenly (by position among the GPUs of that node)
        peers = [o for o in range(world) if (gpu_host_locality(o) or (None, None, None))[1] == node]
        if len(peers) > 1 and len(mine) >= 2 * len(peers):
            k, per = peers.index(ordinal), len(mine) // len(peers)
            mine = mine[k * per:(k + 1) * per]
        os.sched_setaffinity(0, mine)
        info.update(cpus=len(mine), bound=True)
    return info


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
    """S encoder instances of one geometry in G lockstep groups, pictures in pinned host memory.

    Stream layout: streams 2u and 2u+1 are TWINS -- same input, different lockstep group -- whose packets must be identical
    frame for frame.  GOP phases (stream s codes its first picture in step r0[s] of an untimed pre-roll, so that every step
    carries the steady-state 1/gop share of intra pictures) are PHASE-ALIGNED with the groups when the group count divides
    the GOP length: group g holds the phases g, g + G, g + 2G ..., so in any one step the intra pictures of the whole GPU
    all belong to ONE group -- its launches carry them all, the other groups launch no intra-only kernel at all
    (dsv_encoder.c:1247-1271 decides the picture type from the frame number alone)."""

    CLASSES = ("pan", "cut", "static", "fast")

    def __init__(self, hip, A, torch, w, h, fmt, qp, gop, effort, S, G, videos, stagger, device_resident=False, seeds=None, phase_align=True,
                 mix=None, timed_from=0, timed_steps=48):
        from codec_run import configure_encoder
        self.hip, self.A, self.torch = hip, A, torch
        self.w, self.h, self.fmt, self.qp, self.gop, self.effort = w, h, fmt, qp, gop, effort
        self.S, self.G = S, max(1, min(G, S))
        self.P = len(videos[0][0])
        self.NV, self.NF = len(videos), len(videos[0])
        self.seeds = list(seeds) if seeds is not None else [None] * self.NV
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
        G = self.G
        self.video = [(s // 2) % self.NV for s in range(S)]
        self.shift = [2 * (((s // 2) // self.NV) % max(1, self.NF // 2)) for s in range(S)]
        self.R = gop if (stagger and gop > 1) else 0
        self.group_of = [list(range(g, S, G)) for g in range(G)]
        self.phase_aligned = bool(phase_align and self.R and self.R % G == 0 and G > 1)
        if self.phase_aligned:
            slots, nj = self.R // G, (S + G - 1) // G  # a group's phases g + G * slot; few streams: slots spread over the GOP
            self.r0 = [(s % G + G * (((s // G) * slots) // nj if nj < slots else (s // G) % slots)) % self.R for s in range(S)]
        else:
            self.r0 = [(s // 2) % self.R if self.R else 0 for s in range(S)]
        # Content classes (mix = shares per ten twin pairs, e.g. {"cut": 1, "static": 1, "fast": 1}: the rest pans):
        #   pan     the generator's own motion (1.5 / 1 pixels a frame + moving squares), frame t of the video
        #   cut     the same until a scene cut INSIDE the timed window, then another video: the scene-change test flips that
        #           P picture to an intra picture in mid-batch (dsv_encoder.c:545)
        #   static  one picture repeated: every block of every P picture is skipped
        #   fast    every third frame of the video: 4.5 / 3 pixels a frame, squares up to 12
        # Twins share class, video and local cut time, so their inputs stay identical.
        self.klass = [0] * S
        self.cut_t = [1 << 30] * S
        if mix:
            order = [c for c in ("cut", "static", "fast") for _ in range(int(mix.get(c, 0)))]
            for u in range((S + 1) // 2):
                c = order[u % 10] if u % 10 < len(order) else "pan"
                for s2 in (2 * u, 2 * u + 1):
                    if s2 < S:
                        self.klass[s2] = self.CLASSES.index(c)
                if c == "cut":
                    t_local = timed_from - min(self.r0[2 * u], self.r0[min(S - 1, 2 * u + 1)]) + 6 + (5 * u) % max(1, timed_steps - 16)
                    for s2 in (2 * u, 2 * u + 1):
                        if s2 < S:
                            self.cut_t[s2] = max(1, t_local)
        subsamp = A.SUBSAMP_420 if fmt == "420" else A.SUBSAMP_444
        meta = A.mk_meta(w, h, subsamp)
        self.encs = []
        for s in range(S):
            e = A.ENCODER()
            configure_encoder(hip, e, meta, qp=qp, gop=gop, effort=effort)
            self.encs.append(e)
        self.out = [[] for _ in range(S)]  # per stream, per frame: list of packets (bytes)
        self.step = 0
        self.step_ms = None  # per group: wall-clock duration of every step of the current run() (filled when a list)
        self.in_call_s = [0.0] * self.G
        # one host thread per lockstep group for the life of the run (a group keeps its thread from step to step and from
        # run() to run(), as a long-lived encoding service would)
        import queue
        self._state = [self._group_setup(g) for g in range(G)]
        self._cmd = [queue.Queue() for _ in range(G)]
        self._threads = [threading.Thread(target=self._group_thread, args=(g,), daemon=True) for g in range(G)]
        for th in self._threads:
            th.start()

    def source(self, s, t):
        """(video, frame of it) that stream s codes as its local frame t"""
        c = self.klass[s]
        k = self.shift[s] + (0 if c == 2 else (3 * t if c == 3 else t))
        period = 2 * (self.NF - 1) if self.NF > 1 else 1
        k %= period
        v = self.video[s] if t < self.cut_t[s] else (self.video[s] + max(1, self.NV // 2)) % self.NV
        return v, (k if k < self.NF else period - k)

    def frame_index(self, s, t):
        return self.source(s, t)[1]

    def ptr(self, s, t):
        v, k = self.source(s, t)
        return self.vbase[v] + self.P * k

    def ref_job(self, s, nframes):
        """this stream's first nframes as a job of tools/ref_encode_worker.py: (seed of the video, frame of it) per frame"""
        src = [self.source(s, t) for t in range(nframes)]
        assert all(self.seeds[v] is not None for v, _ in src)
        return (self.w, self.h, self.fmt, 0, self.qp, self.gop, self.effort, ["%d:%d" % (self.seeds[v], k) for v, k in src])

    def pick_reference_streams(self, n):
        """n streams to re-encode with the reference: GOP phases spread over the whole 0 .. gop-1 range, every lockstep group
        covered, no two of them twins"""
        S, G = self.S, self.G
        n = max(1, min(n, max(1, S // 2)))
        if not self.R:
            sel = []
            for u in range(n):
                s = 2 * u + ((u >> 1) & 1)
                sel.append(s if s < S else 2 * u)
            return sel
        sel, used_pairs, per_group = [], set(), [0] * G
        phases = sorted(set(self.r0))
        for k in range(n):
            want = phases[(k * (len(phases) - 1)) // max(1, n - 1)] if n > 1 else phases[0]
            cands = [s for s in range(S) if (s // 2) not in used_pairs]
            if not cands:
                break
            # nearest phase first, then the group that has been picked least, then a video not picked yet
            vids_used = {self.video[x] for x in sel}
            s = min(cands, key=lambda x: (abs(self.r0[x] - want), per_group[x % G], self.video[x] in vids_used, x))
            sel.append(s)
            used_pairs.add(s // 2)
            per_group[s % G] += 1
        return sel

    def _group_thread(self, g):
        while True:
            cmd = self._cmd[g].get()
            if cmd is None:
                return
            try:
                self._group_worker(g, *cmd)
            except BaseException:  # noqa: BLE001  (a failed group must not leave the others waiting at the barrier)
                import traceback
                traceback.print_exc()
                os._exit(7)

    def _group_setup(self, g):
        """per group, once: its streams ordered by GOP phase (the started ones are then always a prefix), the encoder
        pointer table, and the numbers the per-step picture pointers are computed from -- the step loop itself does no
        per-stream Python work"""
        import numpy as np
        ids = sorted(self.group_of[g], key=lambda s: (self.r0[s], s))
        M = len(ids)
        st = {"ids": ids, "M": M,
              "gp": (C.POINTER(self.A.ENCODER) * M)(*[C.pointer(self.encs[s]) for s in ids]),
              "r0": np.array([self.r0[s] for s in ids], dtype=np.int64),
              "shift": np.array([self.shift[s] for s in ids], dtype=np.int64),
              "klass": np.array([self.klass[s] for s in ids], dtype=np.int64),
              "cut_t": np.array([self.cut_t[s] for s in ids], dtype=np.int64),
              "base": np.array([self.vbase[self.video[s]] for s in ids], dtype=np.uint64),
              "base2": np.array([self.vbase[(self.video[s] + max(1, self.NV // 2)) % self.NV] for s in ids], dtype=np.uint64)}
        return st

    def _ptrs(self, st, m, step):
        """host (or device) address of the picture each of the first m streams codes in global step `step`"""
        import numpy as np
        t = step - st["r0"][:m]
        c = st["klass"][:m]
        k = st["shift"][:m] + np.where(c == 2, 0, np.where(c == 3, 3 * t, t))  # (same rule as source())
        period = 2 * (self.NF - 1) if self.NF > 1 else 1
        k %= period
        k = np.where(k < self.NF, k, period - k)
        base = np.where(t < st["cut_t"][:m], st["base"][:m], st["base2"][:m])
        return np.ascontiguousarray(base + (k * self.P).astype(np.uint64))

    def _group_worker(self, g, g0, g1, bar, bar_done):
        hip, A = self.hip, self.A
        st = self._state[g]
        import numpy as np
        pend = []  # per step: (m, packets, counts) as the library returned them; turned into bytes after the clock stops
        bar.wait()
        t_prev = time.perf_counter()
        for step in range(g0, g1):
            m = int(np.searchsorted(st["r0"], step, side="right"))  # streams whose first step has come
            if not m:
                continue
            gp = (C.POINTER(A.ENCODER) * m).from_buffer(st["gp"])
            gb = (A.BUF * (4 * m))()
            gn = (C.c_int * m)()
            cur_a = self._ptrs(st, m, step)
            cur = (C.c_void_p * m).from_buffer(cur_a)
            t_call = time.perf_counter()
            if self.device_resident:
                rc = hip.dsv2hip_enc_batch(m, gp, cur, gb, gn)
            else:
                nxt_a = self._ptrs(st, m, step + 1)
                nxt = (C.c_void_p * m).from_buffer(nxt_a)
                t_call = time.perf_counter()
                rc = hip.dsv2hip_enc_batch_host(m, gp, cur, nxt, gb, gn)
            self.in_call_s[g] += time.perf_counter() - t_call
            assert rc == 0
            pend.append((m, gb, gn))
            if self.step_ms is not None:
                t_now = time.perf_counter()
                self.step_ms[g].append(1e3 * (t_now - t_prev))
                t_prev = t_now
        bar.wait()
        # the packets are finished and in host memory (DSV_BUFs); copying them into Python objects for the checks is the
        # harness's business, not the codec's: outside the timed region
        ids = st["ids"]
        for m, gb, gn in pend:
            for k in range(m):
                pk = []
                for i in range(gn[k]):
                    b = gb[4 * k + i]
                    pk.append(C.string_at(b.data, b.len))
                    hip.dsv_buf_free(C.byref(b))
                self.out[ids[k]].append(pk)
        bar_done.wait()

    def run(self, nsteps, dist=None, record=False):
        """advance every (started) stream by nsteps frames; returns the wall time bracketed by barrier + synchronize"""
        torch = self.torch
        g0, g1 = self.step, self.step + nsteps
        self.step_ms = [[] for _ in range(self.G)] if record else None
        self.in_call_s = [0.0] * self.G  # seconds each group spent inside the library during this run()
        bar = threading.Barrier(self.G + 1)
        bar_done = threading.Barrier(self.G + 1)
        for g in range(self.G):
            self._cmd[g].put((g0, g1, bar, bar_done))
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
        bar_done.wait()  # (the groups have turned their packets into Python objects)
        self.step = g1
        return t_end - t_start

    def frames_in(self, g0, g1):
        return sum(max(0, g1 - max(g0, self.r0[s])) for s in range(self.S))

    def intra_in(self, g0, g1):
        return sum(1 for s in range(self.S) for t in range(max(0, g0 - self.r0[s]), g1 - self.r0[s]) if t % self.gop == 0)

    def class_report(self, g0, g1):
        """per content class over the global steps [g0, g1): streams, pictures, bytes per picture, and the intra pictures
        that are NOT at a GOP start (P pictures the scene-change test flipped)"""
        rep = {}
        for ci, name in enumerate(self.CLASSES):
            ss = [s for s in range(self.S) if self.klass[s] == ci]
            if not ss:
                continue
            pics = nbytes = flips = 0
            for s in ss:
                for t in range(max(0, g0 - self.r0[s]), min(len(self.out[s]), g1 - self.r0[s])):
                    pk = self.out[s][t][-1]
                    pics += 1
                    nbytes += sum(len(x) for x in self.out[s][t])
                    if not (pk[5] & 1) and t % self.gop:
                        flips += 1
            rep[name] = {"streams": len(ss), "pictures": pics, "bytes_per_picture": round(nbytes / max(1, pics)), "intra_flips": flips}
        return rep

    def twins_equal(self):
        """every stream's packets == its twin's (same input, other lockstep group and -- phase-aligned -- another GOP phase),
        frame for frame over everything both have coded"""
        pairs = bad = 0
        for u in range(self.S // 2):
            a, b = self.out[2 * u], self.out[2 * u + 1]
            n = min(len(a), len(b))
            pairs += 1
            if n == 0 or a[:n] != b[:n]:
                bad += 1
        return pairs, bad

    def stream_bytes(self, s):
        return b"".join(p for fr in self.out[s] for p in fr)

    def free(self):
        for q in self._cmd:
            q.put(None)
        for th in self._threads:
            th.join()
        for e in self.encs:
            self.hip.dsv_enc_free(C.byref(e))
        if not self.device_resident:
            for p in self.vbase:
                self.hip.dsv2hip_host_free(p)
        self._keep = []


class RefWorkers:
    """reference encodes / decodes on the host CPU (tools/ref_encode_worker.py): parity oracle + CPU baselines"""

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

    def _cmd(self, which, word, counts):
        for i, n in zip(which, counts):
            self.procs[i].stdin.write("%s %d\n" % (word, n))
            self.procs[i].stdin.flush()
        return [json.loads(self.procs[i].stdout.readline()) for i in which]

    def go(self, which, counts):
        """encode: worker i codes its first counts[k] frames, all the named workers at once"""
        return self._cmd(which, "go", counts)

    def dec(self, which, counts):
        """decode the packets of the last encode with the reference decoder: timing + md5 of every picture"""
        return self._cmd(which, "dec", counts)

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


class RefCheck:
    """what one leg of the bench hands to the reference for comparison: the job (how to regenerate the stream's input and
    encode it) and the bytes this library produced, frame by frame"""

    def __init__(self, leg, run, s, nframes):
        self.leg, self.stream = leg, s
        self.n = min(nframes, len(run.out[s]))
        self.job = run.ref_job(s, self.n)
        self.got = [b"".join(fr) for fr in run.out[s][:self.n]]
        self.group, self.phase = s % run.G, run.r0[s]
        self.dec_md5 = None  # (legs that also decode: md5 of every picture this library decoded from self.got, frame by frame)


def picture_planes(fp):
    """a decoded DSV_FRAME's three planes, rows packed tight (one copy; what the reference worker hashes)"""
    import numpy as np
    out = []
    for c in range(3):
        p = fp.contents.planes[c]
        out.append(np.ctypeslib.as_array(p.data, shape=(p.h * p.stride,)).reshape(p.h, p.stride)[:, :p.w].copy())
    return out


def planes_md5(planes):
    import hashlib
    h = hashlib.md5()
    for a in planes:
        h.update(a.tobytes())
    return h.hexdigest()


def decode_leg(hip, A, run, nsteps, nstreams, groups, check):
    """lockstep batch decoder over the packets the encode run produced: every decoder starts at its stream's first packet.
    The streams in `check` (the ones the reference re-encodes AND decodes) are among the decoders; their pictures are
    copied out of the returned DSV_FRAME inside the clock (3 MB each, a few per step) and hashed after it stops."""
    hip.dsv2hip_dec_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(A.DECODER)), C.POINTER(A.BUF), C.POINTER(C.POINTER(A.FRAME)),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    hip.dsv2hip_dec_batch.restype = C.c_int
    D = min(nstreams, run.S)
    ids_all = list(check) + [s for s in range(run.S) if s not in set(check)][:max(0, D - len(check))]
    D = len(ids_all)
    G = max(1, min(groups, D))
    decs = {s: A.DECODER() for s in ids_all}
    group_of = [ids_all[g::G] for g in range(G)]
    decoded = [0] * G
    held = {s: [] for s in check}

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
                if ids[i] in held:
                    held[ids[i]].append(picture_planes(outs[i]))
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

    import resource
    nfr = min(len(run.out[s]) for s in ids_all)
    prof_steps = 4 if nfr >= nsteps + 12 else 0          # a few more steps of the same configuration with stage events on (not timed)
    warm = min(4, max(1, nfr - nsteps - prof_steps))
    phase(0, warm)
    before = sum(decoded)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    elapsed = phase(warm, min(nfr - prof_steps, warm + nsteps))
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    n = sum(decoded) - before
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    out = {"value": round(n / elapsed, 2), "unit": "frames/s", "frames": n, "decoders": D, "groups": G,
           "mpix_per_s": round(n / elapsed * run.w * run.h / 1e6, 1),
           "host_cpu_cores_busy": round(host_cpu_s / elapsed, 2),
           "note": "lockstep batch decoder (dsv2hip_dec_batch) over this run's own packets, pictures delivered to host memory as DSV_FRAMEs; "
                   "host_cpu_cores_busy includes the serial entropy parse of the plane sections (hzcc.c:451-585) where the host does it "
                   "(plane_sections_parsed_on; csrc/dec_parse_dev.hip is the device form)"}
    if prof_steps:
        t0 = min(nfr - prof_steps, warm + nsteps)
        hip.dsv2hip_prof_enable(1)
        phase(t0, t0 + prof_steps)
        ms, ln, un, fr = (C.c_double * 16)(), (C.c_longlong * 16)(), (C.c_longlong * 16)(), C.c_longlong(0)
        hip.dsv2hip_prof_read(ms, ln, C.byref(fr))
        hip.dsv2hip_prof_read_units(un)
        hip.dsv2hip_prof_enable(0)
        nn = run.w * run.h
        pp = nn * 3 // 2 if run.fmt == "420" else nn * 3
        # algorithmic bytes of a P picture's decode, 15 P + 2 N: coefficient planes zeroed (4 P) + symbols scattered and dequantised
        # in place; inverse transform (4 P read, P written); motion-compensated reconstruction (reference P, residual P, picture P)
        # + in-loop luma filters (2 N); borders and the picture's way into the caller's frame (P read, 2 P written)
        dbytes = {"quant_compact": 4 * pp, "inv_sbt": 5 * pp, "recon_filters": 3 * pp + 2 * nn, "extend": 3 * pp}
        dkern = {"quant_compact": "k_zero_linear / k_dequant_level", "inv_sbt": "k_inv_haar_u8x4 / k_inv_haar / k_inv_rows / k_inv_cols",
                 "recon_filters": "k_predict_w<MC_RECONSTRUCT> / k_inter_filters_b", "extend": "k_extend / k_copy_linear"}
        per = {}
        for name in dbytes:
            i = STAGES.index(name)
            if un[i]:
                per[name] = {"ms": ms[i], "launch_groups": fr.value, "units": un[i], "us_per_frame": round(1e3 * ms[i] / un[i], 2),
                             "GBps": round(dbytes[name] * un[i] / (ms[i] * 1e-3) / 1e9, 1) if ms[i] > 0 else None}
        if per:
            dom = max(per, key=lambda k: per[k]["ms"])
            steps_prof = max(1, fr.value)  # lockstep steps (all groups) the events cover
            ach = per[dom]["GBps"]
            out["roofline"] = {"bound": "hbm", "stage": dom, "kernel": dkern[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                               "avg_stage_span_us": round(1e3 * per[dom]["ms"] / steps_prof, 1),
                               "algorithmic_bytes_per_frame": dbytes[dom], "pictures_per_step_per_group": D // G,
                               "stage_us_per_frame": {k: v["us_per_frame"] for k, v in per.items()},
                               "whole_frame_algorithmic_GBps": round((15 * pp + 2 * nn) * out["value"] / 1e9, 1),
                               "note": "stage spans = HIP events on each group's stream around the stage's launches (%d groups share the GPU: spans of "
                                       "different groups overlap); kernel durations of the same run: profiles/r05_decode_kernel_stats.txt" % G}
    md5 = {s: [planes_md5(pl) for pl in frames] for s, frames in held.items()}
    for d in decs.values():
        hip.dsv_dec_free(C.byref(d))
    return out, md5


def thread_cpu():
    """(name, user + system CPU seconds) of every thread of this process, from /proc"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (name, (int(rest[11]) + int(rest[12])) / tick)
        except (OSError, ValueError):
            pass
    return out


def bind_abi(hip, A):
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


def agree_once(what, world, mine):
    """The ranks of one job (children of one launcher on one node: torch.distributed.run, or spawn_ranks) adopt the FIRST rank's decision: it is
    written to a file named after the launcher's pid and the rendezvous port (O_EXCL: one writer), everybody else reads it.  Needed before
    torch.distributed exists (the pictures are made before the GPU runtime starts).  world 1: the caller's own value."""
    if world <= 1:
        return mine
    path = "/tmp/dsv2_bench_%s_%d_%s" % (what, os.getppid(), os.environ.get("MASTER_PORT", "0"))
    try:
        if os.path.exists(path) and time.time() - os.path.getmtime(path) > 3600:
            os.unlink(path)  # (a leftover of a launcher whose pid has come round again)
    except OSError:
        pass
    try:
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        os.write(fd, json.dumps(list(mine)).encode())
        os.close(fd)
        import atexit
        atexit.register(lambda: os.path.exists(path) and os.unlink(path))
        return mine
    except FileExistsError:
        for _ in range(200):
            try:
                return tuple(json.load(open(path)))
            except (OSError, ValueError):
                time.sleep(0.01)  # (created, not yet written)
        return mine


def census_report(hip, elapsed):
    """csrc/prio.h: per kernel site, resident wavefront-time over the timed region -> mean resident wavefronts per SIMD (1 024 SIMDs)
    and the mean lifetime of a workgroup, with every lockstep group running (nothing is serialised)"""
    import re
    buf = C.create_string_buffer(1 << 16)
    hip.dsv2hip_census_read.restype = C.c_int
    n = hip.dsv2hip_census_read(buf, len(buf))
    if n <= 0:
        return {"error": "this build of the library carries no census (make -C digital-subband-video-2_amd/csrc census; DSV2HIP_LIB=...)"}
    src = {}
    rows = []
    for ln in buf.raw[:n].decode().splitlines():
        f, line, ticks, groups, waves = ln.split()
        line, ticks, groups, waves = int(line), int(ticks), int(groups), int(waves)
        if f not in src:
            src[f] = open(os.path.join(ROOT, "digital-subband-video-2_amd", "csrc", f)).read().splitlines()
        name = "%s:%d" % (f, line)
        for k in range(min(line, len(src[f])) - 1, max(-1, line - 40), -1):  # the kernel the scope sits in: the nearest __global__ / HME_ROWS_P above (or on) its line
            m = re.search(r"void\s+(k_\w+)\s*\(", src[f][k]) if "__global__" in src[f][k] else re.search(r"HME_ROWS_P\((k_\w+)", src[f][k])
            if m is None and "_body(" in src[f][k] and "__device__" in src[f][k]:
                m = re.search(r"void\s+(\w+)\s*\(", src[f][k])
            if m:
                name = m.group(1)
                break
        rows.append({"kernel": name, "waves_per_simd": round(ticks / 1e8 / elapsed / 1024.0, 3), "workgroups": groups,
                     "mean_group_life_us": round(ticks / max(1, waves) / 100.0, 2)})
    rows.sort(key=lambda r: -r["waves_per_simd"])
    return {"resident_waves_per_simd": round(sum(r["waves_per_simd"] for r in rows), 2), "of_slots": 8, "elapsed_s": round(elapsed, 3),
            "note": "measured inside the kernels (first thread of every workgroup, 100 MHz real-time counter) while all lockstep groups run; "
                    "a workgroup counts from its first instruction to its last, waiting included",
            "kernels": rows}


def issue_roofline(fps, world):
    """instruction-issue roofline of the whole encode: vector wavefront-instructions per frame (committed PMC passes over
    every kernel, profiles/instruction_volume.json) x measured frames/s against what the chip's SIMDs can issue"""
    try:
        iv = json.load(open(os.path.join(ROOT, "profiles", "instruction_volume.json")))
    except (OSError, ValueError):
        return None
    simds, clock = 256 * 4, 2.4e9
    # a wave64 vector instruction occupies the SIMD-32 for 2 clocks; one wavefront alone issues one every 4 (MI355X_MICROARCH.md,
    # 'vector-instruction ISSUE cost'): two peaks -- what the SIMDs can issue with two or more wavefronts each, and what they
    # can with one
    peak2, peak4 = simds * clock / 2 / 1e9, simds * clock / 4 / 1e9
    ach = iv["vector_per_frame"] * fps / max(1, world) / 1e9
    occ = {}
    try:  # tools/profile_round.sh part `occ`: resident wavefronts per SIMD and issue shares of the four-group mix (committed passes)
        oc = json.load(open(os.path.join(ROOT, "profiles", "occupancy.json")))
        occ = {"resident_waves_per_simd": oc.get("resident_waves_per_simd"), "resident_waves_note": "lower bound: per-kernel wave-clocks measured alone x the "
               "launches of the un-serialised trace's timed region; " + oc.get("source", "profiles/occupancy.json")}
    except (OSError, ValueError):
        pass
    return {**occ, "bound": "vector issue", "vector_inst_per_frame": iv["vector_per_frame"], "scalar_inst_per_frame": iv.get("scalar_per_frame"),
            "achieved": round(ach, 1), "peak": round(peak2, 1), "unit": "G wave-instructions/s per GPU", "frac": round(ach / peak2, 4),
            "peak_one_wave_per_simd": round(peak4, 1), "frac_of_one_wave_rate": round(ach / peak4, 4),
            "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 2 clocks per wave64 vector instruction (two or more wavefronts per SIMD); / 4 clocks "
                         "is what one wavefront per SIMD can issue -- the search runs at three per SIMD and is parked on memory counters 43 % of its wave-cycles",
            "source": "committed PMC passes, not this run: " + iv.get("source", "profiles/instruction_volume.json")}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        sys.stderr.write("[bench] --gpus %d but WORLD_SIZE=%d: running %d ranks\n" % (args.gpus, world, world))
    local = int(os.environ.get("DSV2_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    # this rank's cores: those next to its GPU (before the picture generators fork, the pictures are pinned, the runtime starts)
    locality = bind_rank_to_gpu_node(local, world if "DSV2_FORCE_DEVICE" not in os.environ else 1)
    ncpu_box = usable_cpus()
    extras = not args.no_extras and world == 1

    # ---- pictures first: forked generators must not inherit an initialised GPU runtime ----
    W_, H_, GOP, QP = 1920, 1080, 48, 60
    # Content: 96 distinct videos of 64 unique frames per GPU for the headline (no input shared beyond the twin of a stream and,
    # at 768 streams, three more streams that start 2 / 4 / 6 frames into the same video) -- 19 GB of pinned pictures per rank,
    # taken only when the host has room for every rank's share twice over; else 8 x 32 (round 4's content), said on the line.
    def mem_available():
        try:
            return int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) * 1024
        except (OSError, StopIteration, ValueError):
            return 0
    if args.streams >= 192 and not os.environ.get("DSV2_BENCH_SMALL_CONTENT") and not under_profiler_():
        NV, NF = 96, 64
        if mem_available() < 2.2 * world * NV * NF * (W_ * H_ * 3 // 2):
            NV, NF = 8, 32
        NV, NF = agree_once("content", world, (NV, NF))  # ONE decision per job: ranks that looked at /proc/meminfo at different moments must not differ
    elif args.streams >= 16:
        NV, NF = 8, 32
    else:
        NV, NF = max(1, min(4, args.streams // 2)), 24
    seeds = [1 + rank * NV + k for k in range(NV)]
    specs = [(W_, H_, "420", seeds[k], NF) for k in range(NV)]
    if extras:
        specs += [(1280, 720, "420", 101 + k, 24) for k in range(4)]
        specs += [(W_, H_, "444", 201, 12)]
        specs += [(3840, 2160, "420", 301 + k, 16) for k in range(2)]
    t_gen = time.perf_counter()
    under_profiler = under_profiler_()
    gen_procs = args.gen_procs if args.gen_procs > 0 else (1 if under_profiler else max(1, min(16, ncpu_box // (1 if locality["bound"] else max(1, world)))))
    vids = gen_videos(specs, gen_procs)
    t_gen = time.perf_counter() - t_gen

    # ---- the host budget of this rank: pinned BEFORE the GPU runtime and the library's worker pool come up ----
    if args.host_cores > 0:
        cores = sorted(os.sched_getaffinity(0))
        k0 = (rank * args.host_cores) % max(1, len(cores))
        mine = [cores[(k0 + i) % len(cores)] for i in range(min(args.host_cores, len(cores)))]
        os.sched_setaffinity(0, mine)
    ncpu = usable_cpus() if args.host_cores <= 0 else min(usable_cpus(), args.host_cores)
    # host phases run on a worker pool inside the library: share the usable cores between the ranks
    if args.host_cores > 0:
        os.environ.setdefault("DSV2_HOST_THREADS", str(max(2, 2 * ncpu)))
    else:
        os.environ.setdefault("DSV2_HOST_THREADS", str(min(48, max(8, 3 * ncpu // (1 if locality["bound"] else max(1, world))))))

    import torch
    import dsvabi as A
    dist = None
    if world > 1 or args.force_dist:
        if world == 1 and "MASTER_PORT" not in os.environ:
            so = socket.socket()
            so.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(so.getsockname()[1]), RANK="0", WORLD_SIZE="1")
            so.close()
        import torch.distributed as dist_
        dist = dist_
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    locality = bind_rank_late(locality, world if "DSV2_FORCE_DEVICE" not in os.environ else 1, torch, local)
    hip = A.load_hip()
    assert hip.dsv2hip_device_ok() == 0, "no HIP device: the product has no CPU path"
    hip.dsv2hip_set_device(local)
    bind_abi(hip, A)
    from conftest import load_pkg
    pkg = load_pkg()

    S, K, Wm = max(1, args.streams), args.steps, args.warmup
    effort = int(os.environ.get("DSV2_BENCH_EFFORT", "10"))  # (experiments only: the headline is effort 10)
    align = not args.no_phase_align
    mix = None if args.no_mix else MIX
    torch.cuda.synchronize()
    hbm_free0 = torch.cuda.mem_get_info()[0]
    run = EncodeRun(hip, A, torch, W_, H_, "420", QP, GOP, effort, S, args.groups, vids[:NV], not args.no_stagger, args.device_resident,
                    seeds=seeds, phase_align=align, mix=mix, timed_from=(GOP if not args.no_stagger else 0) + Wm, timed_steps=K)
    G = run.G
    hip.dsv2hip_prof_enable(0)
    run.run(run.R + Wm)                   # untimed: GOP-phase pre-roll + warm-up (allocations, clocks)
    torch.cuda.synchronize()
    hbm_per_instance = (hbm_free0 - torch.cuda.mem_get_info()[0]) / S  # every instance has coded intra and inter pictures by now
    import resource
    thr0 = thread_cpu() if os.environ.get("DSV2_BENCH_THREADS") else None
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    g_timed = run.step
    census_on = bool(os.environ.get("DSV2_CENSUS"))
    if census_on:  # a `make census` build (DSV2HIP_LIB): resident wavefront-time per kernel, measured inside the kernels under load
        torch.cuda.synchronize()
        hip.dsv2hip_census_reset()
    elapsed = run.run(K, dist, record=True)   # timed: exactly K steps
    census = census_report(hip, elapsed) if census_on else None
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    if thr0 is not None:  # where the host CPU of the timed region went, by thread (name, seconds)
        thr1 = thread_cpu()
        rows = sorted(((thr1[t][1] - thr0.get(t, (None, 0.0))[1], thr1[t][0], t) for t in thr1), reverse=True)
        sys.stderr.write("[bench] host CPU by thread over %.2f s: %s\n" % (elapsed, ", ".join("%s/%d %.2f" % (nm, t, d) for d, nm, t in rows[:24] if d > 0.005)))
    if os.environ.get("DSV2_BENCH_STEP_SERIES"):  # drift inside the timed region: mean step time of each group, per dozen steps
        for g, ser in enumerate(run.step_ms):
            sys.stderr.write("[bench] group %d step ms per dozen: %s\n" % (g, " ".join("%.1f" % (sum(ser[i:i + 12]) / max(1, len(ser[i:i + 12]))) for i in range(0, len(ser), 12))))
    step_ms = sorted(x for g in run.step_ms for x in g)
    in_call_share = [round(x / elapsed, 3) for x in run.in_call_s]
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    frames_rank = run.frames_in(g_timed, g_timed + K)
    intra_rank = run.intra_in(g_timed, g_timed + K)
    classes = run.class_report(g_timed, g_timed + K)

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

    # final ordered gather of the segment bytes (the only exchange of the path); a rank's streams are the closed-GOP
    # segments sharding.assign_segments deals to it round-robin: global id = local index * world + rank
    xdev = "cuda" if args.backend == "nccl" else "cpu"
    t_max = torch.tensor([elapsed], device=xdev, dtype=torch.float64)
    counts = torch.tensor([frames_rank, intra_rank, pairs], device=xdev, dtype=torch.int64)
    gather_s = None
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        assert pkg.sharding.assign_segments(S * world, world)[rank] == [pkg.sharding.segment_id(rank, world, s) for s in range(S)]
        # Every segment of the job travels to rank 0 and is checked there against its producer's md5 -- in pieces of at most
        # 256 MB that rank 0 folds into running digests and drops (sharding.gather_segments_streaming): eight ranks of the
        # headline produce ~58 GB, which no rank ever holds.  (DSV2_GATHER_OUT=<file>: rank 0 also writes every piece at its
        # final offset of that file, the `cat` of parallel_encode_yuv.sh:50.)
        out_path = os.environ.get("DSV2_GATHER_OUT")
        out_fd = os.open(out_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC) if (rank == 0 and out_path) else -1
        sink = (lambda sid, base, off, piece: os.pwrite(out_fd, piece, base + off)) if out_fd >= 0 else None
        chunk = int(float(os.environ.get("DSV2_GATHER_CHUNK_MB", "256")) * (1 << 20))
        t_g = time.perf_counter()
        res = pkg.sharding.gather_segments_streaming(dist, rank, world, [pkg.sharding.segment_id(rank, world, s) for s in range(S)],
                                                     lambda sid: run.stream_bytes((sid - rank) // world), device=xdev, chunk=chunk, sink=sink)
        gather_s = time.perf_counter() - t_g
        if out_fd >= 0:
            os.close(out_fd)
        total_bytes = res["bytes"] if rank == 0 else 0
        gather_ok = None
        if rank == 0:
            gather_ok = {"segments": res["segments"], "segments_verified": res["segments_verified"], "bytes": res["bytes"], "segments_of_job": S * world,
                         "segments_over_the_wire": res["segments_over_the_wire"], "digest": res["digest"],
                         "rank0_peak_bytes_held": res["peak_bytes_held"],
                         "form": "streaming: a digest per segment folded piece by piece on a checker thread while the next piece is on the wire, pieces dropped"}
            if res["segments_verified"] != res["segments"] or res["segments"] != S * world:
                sys.stderr.write("[bench] gathered segment bytes DIFFER from what the ranks produced (%d of %d ok, job has %d)\n"
                                 % (res["segments_verified"], res["segments"], S * world))
                sys.exit(8)
        # per rank: where its host side ran and what it moved
        mine_info = {"rank": rank, "gpu": local, "pci": locality["pci"], "numa_node": locality["numa_node"], "cpus": locality["cpus"], "bound": locality["bound"],
                     "frames_per_s": round(frames_rank / elapsed, 1),
                     "h2d_GBps": 0.0 if args.device_resident else round(frames_rank * run.P / elapsed / 1e9, 2)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_info)
    else:
        total_bytes = sum(len(p) for s in range(S) for fr in run.out[s] for p in fr)
        gather_ok = None
        per_rank = [{"rank": 0, "gpu": local, "pci": locality["pci"], "numa_node": locality["numa_node"], "cpus": locality["cpus"], "bound": locality["bound"],
                     "frames_per_s": round(frames_rank / elapsed, 1),
                     "h2d_GBps": 0.0 if args.device_resident else round(frames_rank * run.P / elapsed / 1e9, 2)}]
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
                               "pictures in %s; GOP phases %s; content: %s" %
                               (effort, S, G, "HBM before the clock starts (kernel-side figure)" if args.device_resident else
                                "pinned host memory, every frame uploaded inside the timed region (double-buffered copy stream)",
                                ("staggered over %d untimed pre-roll steps: every step codes 1/%d of the streams as intra pictures%s" %
                                 (run.R, GOP, ", all of them in ONE lockstep group (phase-aligned groups)" if run.phase_aligned else ""))
                                if run.R else "aligned: one all-intra step per GOP",
                                "of ten twin pairs seven pan, one has a scene cut inside the timed window, one is static, one moves three frames per step"
                                if mix else "every stream pans"),
                   "streams_per_gpu": S, "frames_per_step_per_gpu": S, "groups": G, "frames_timed": frames_total, "intra_frames_timed": intra_total,
                   "phase_aligned_groups": run.phase_aligned,
                   "content_classes": classes,
                   "input": "pinned_host" if not args.device_resident else "device_resident", "h2d_bytes_per_step_per_gpu": 0 if args.device_resident else S * run.P,
                   "distinct_videos_per_gpu": NV, "unique_frames_per_video": NF,
                   "hbm_bytes_per_instance": int(hbm_per_instance), "hbm_bytes_instances_total": int(hbm_per_instance * S),
                   "ms_per_frame_p50": round(step_ms[len(step_ms) // 2], 3) if step_ms else None,
                   "group_time_inside_library": in_call_share,
                   "host_cpu_cores_busy": round(host_cpu_s / elapsed, 2), "mpix_per_s": round(fps * W_ * H_ / 1e6, 1),
                   "stream_bytes_total": total_bytes, "host_cpus_usable": ncpu, "host_cores_pinned": args.host_cores or None,
                   "host_threads": int(os.environ["DSV2_HOST_THREADS"]),
                   "final_gather_s": round(gather_s, 3) if gather_s is not None else None, "final_gather_check": gather_ok,
                   "exchange_backend": (args.backend if dist is not None else None),
                   "per_rank": per_rank,
                   "setup_s": {"generate_pictures": round(t_gen, 1)}},
        "parity_checked": {"twin_pairs_equal": pairs_total, "twin_pairs": pairs_total,
                           "note": "twins = same input, different lockstep group (and GOP phase), compared frame by frame over the run"},
    }
    if census is not None:
        result["census"] = census
    if stage_ms is not None and prof_steps:
        # per stage: span (HIP events on the group's stream) per stream-frame, and the algorithmic
        # bytes of SURVEY.md 8(d) moved in that span
        per_unit = {STAGES[i]: (stage_ms[i] / stage_units[i] if stage_units[i] else 0.0) for i in range(NST)}
        total_ms = {STAGES[i]: stage_ms[i] for i in range(8)}
        # The dominant KERNEL is the level-0 launch of the search (its span is measured on its own): the largest single kernel of
        # the rocprofv3 summary of this command (profiles/r05_rocprof_kernel_stats.txt).  The largest STAGE span is reported beside
        # it -- a stage is several kernels (quantise + compact + entropy is ~25 launches per plane class), and under four
        # concurrent groups its span is mostly the other groups' kernels sharing the chip.
        largest_stage = max(total_ms, key=lambda k: total_ms[k])
        dom = "hme_level0"
        i = STAGES.index(dom)
        nl = max(1, stage_launches[i])
        avg_launch_ms = stage_ms[i] / nl
        bytes_per_launch = sb[dom] * stage_units[i] / nl
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        # HBM traffic per launch of the dominant kernel cannot be read inside this process: it comes from the separate
        # rocprofv3 --pmc passes of tools/profile_round.sh (profiles/pmc_traffic.json) and is only quoted for the
        # configuration those passes were taken on (streams, groups, GOP-phase layout)
        traffic, traffic_source = None, None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            same = (pt.get("stage") == dom and pt.get("streams_per_gpu") == S and pt.get("groups") == G and pt.get("kernel") == STAGE_KERNEL[dom]
                    and bool(pt.get("stagger", False)) == bool(run.R) and bool(pt.get("phase_aligned", False)) == run.phase_aligned)
            # ... and only for the KERNEL they were taken on: the committed figure carries a hash of the search's sources
            import hashlib
            hh = hashlib.sha256()
            for fn in ("hme.hip", "hme_fast.h", "hme.h", "blockstat.h", "dev.h"):
                hh.update(open(os.path.join(ROOT, "digital-subband-video-2_amd", "csrc", fn), "rb").read())
            fresh = pt.get("kernel_source_sha16") == hh.hexdigest()[:16]
            if same and fresh:
                traffic = pt.get("bytes_per_launch")
                traffic_source = "committed PMC passes, not this run: " + pt.get("source", "profiles/pmc_traffic.json")
            elif same:
                traffic_source = "profiles/pmc_traffic.json is STALE (taken on other search sources: kernel_source_sha16 differs): not quoted"
        except (OSError, ValueError):
            pass
        # (advisor, round 5) the stage with the largest span, priced the same way: its algorithmic bytes over its span per stream-frame
        ls_i = STAGES.index(largest_stage)
        ls_rate = (sb[largest_stage] / (per_unit[largest_stage] * 1e-3) / 1e9) if per_unit.get(largest_stage) else 0.0
        result["roofline"] = {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "stage": dom,
                              "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source,
                              "avg_launch_us": round(avg_launch_ms * 1e3, 2),
                              "launches_per_step": round(nl / prof_steps * G, 1),
                              "algorithmic_bytes_per_launch": round(bytes_per_launch),
                              "stage_us_per_frame": {k: round(1e3 * v, 2) for k, v in per_unit.items()},
                              "largest_stage_span": largest_stage,
                              "largest_stage_roofline": {"stage": largest_stage, "launches_per_step": round(stage_launches[ls_i] / prof_steps * G, 1),
                                                         "algorithmic_bytes_per_frame": round(sb[largest_stage]),
                                                         "span_us_per_frame": round(1e3 * per_unit[largest_stage], 2), "achieved": round(ls_rate, 2),
                                                         "frac": round(ls_rate / HBM_PEAK_GBS, 6),
                                                         "note": "a stage is several kernels; under four concurrent groups its span is mostly other groups' kernels sharing the chip"},
                              "whole_frame_algorithmic_GBps": round(frame_bytes * fps / 1e9, 1),
                              "note": "the search is an integer kernel bound by instruction issue and dependent latencies, not by bandwidth: "
                                      "see roofline.issue for the roofline that binds the pipeline"}
        iss = issue_roofline(fps, world)
        if iss:
            result["roofline"]["issue"] = iss

    # ---- legs whose packets the reference will re-encode (collected now, compared at the end, all at once) ----
    checks = []
    sel = run.pick_reference_streams(NREF_STREAMS)
    for s in sel:
        checks.append(RefCheck("headline", run, s, NREF_FRAMES))

    # ---- the decoder on this run's packets (N = 1 only) ----
    # (the headline above is complete: whatever goes wrong below is reported beside it, never instead of it)
    dec_md5 = {}
    if (extras or args.decode_too) and not args.only_batch_curve and not args.only_api_legs:
        try:
            # (DSV2_DEC_GROUPS / DSV2_DEC_STREAMS: experiments with the decode leg's shape)
            result["decode"], dec_md5 = decode_leg(hip, A, run, 32, int(os.environ.get("DSV2_DEC_STREAMS", "256")), int(os.environ.get("DSV2_DEC_GROUPS", "4")),
                                                   sel if (extras and not args.no_cpu_baseline) else [])
            # the same leg with the P pictures' plane sections parsed on the DEVICE (csrc/dec_parse_dev.hip): the operating point of a
            # host with few cores per GPU -- what the library picks by itself below 12 usable cores (the 2-core re-run below takes it)
            hip.dsv2hip_dec_parse_mode.restype = C.c_int
            mode0 = hip.dsv2hip_dec_parse_mode()
            result["decode"]["plane_sections_parsed_on"] = ["host", "device (P pictures)", "device"][mode0]
            if mode0 == 0 and extras:
                hip.dsv2hip_dec_set_parse_mode(1)
                try:
                    leg, md5b = decode_leg(hip, A, run, 32, int(os.environ.get("DSV2_DEC_STREAMS", "256")), int(os.environ.get("DSV2_DEC_GROUPS", "4")),
                                           sel if (extras and not args.no_cpu_baseline) else [])
                    leg["plane_sections_parsed_on"] = "device (P pictures)"
                    leg["pictures_equal_to_the_host_parsed_leg"] = bool(md5b == dec_md5)
                    leg.pop("roofline", None)
                    result["decode_device_parse"] = leg
                finally:
                    hip.dsv2hip_dec_set_parse_mode(-1)
        except Exception as e:  # noqa: BLE001
            result["decode"] = {"error": repr(e)}
    run.free()
    del run

    # ---- the other BASELINE.json configurations and the small-batch operating points (N = 1 only) ----
    if extras and not args.only_batch_curve and not args.only_api_legs:
        try:
            result["configs"] = other_configs(hip, A, torch, args, vids, NV, S, W_, H_, seeds, checks)
        except Exception as e:  # noqa: BLE001
            result["configs"] = {"error": repr(e)}
    if extras and not args.no_batch_curve and not args.only_api_legs:
        try:
            result["batch_curve"] = batch_curve(hip, A, torch, args, vids, NV, seeds, W_, H_, QP, GOP, effort, checks)
            one = next((p for p in result["batch_curve"] if p.get("streams") == 1), None)
            if one:  # BASELINE.json config 5 as literally written: ONE closed-GOP segment per GPU at a time
                result["config5_one_stream_per_gpu"] = {
                    "value": one["value"], "unit": "frames/s per GPU", "ms_per_frame_p50": one["ms_per_frame_p50"],
                    "note": "one stream alone on the GPU: every frame is the chain coarse search levels -> level 0 -> in-loop filter sweep, "
                            "the reference's own data dependencies; the headline is the same hardware with 768 such segments in flight "
                            "(parallel_encode_yuv.sh:31-52 runs as many segments as it has processes)"}
        except Exception as e:  # noqa: BLE001
            result["batch_curve"] = {"error": repr(e)}
    if extras and not args.no_mix and not args.only_batch_curve and not args.only_api_legs:
        try:
            result["content_class_legs"] = class_legs(hip, A, torch, args, vids, NV, seeds, W_, H_, QP, GOP, effort, checks)
        except Exception as e:  # noqa: BLE001
            result["content_class_legs"] = {"error": repr(e)}

    # ---- the reference's own entry points and its own parallel recipe (row h of the round-3 review) ----
    if extras and not args.only_batch_curve and not args.no_api_legs:
        try:
            result["api_legs"] = api_thread_legs(hip, A, vids, NV, seeds, W_, H_, QP, GOP, effort, checks)
        except Exception as e:  # noqa: BLE001
            result["api_legs"] = {"error": repr(e)}
        try:
            result["api_legs"]["processes"] = api_process_leg(vids, W_, H_, QP, GOP)
        except Exception as e:  # noqa: BLE001
            result["api_legs"]["processes"] = {"error": repr(e)}

    # ---- does the host side fit the cores an 8-GPU node leaves per rank?  The headline again, pinned to 2 cores ----
    if extras and not args.no_host_share and not args.only_batch_curve and not args.only_api_legs and args.host_cores <= 0:
        result["host_share"] = host_share(args, 2, fps)

    if extras and not args.no_multi_rank and not args.only_batch_curve and not args.only_api_legs and args.host_cores <= 0:
        result["multi_rank_one_gpu"] = multi_rank_one_gpu(args, fps)

    # ---- parity, part 2 + CPU baselines: the real reference (oracle/_ref) on the host cores ----
    rc = 0
    if not args.no_cpu_baseline and os.path.exists(A.REF_SO):
        try:
            rc = reference_phase(result, checks, dec_md5, sel)
        except (OSError, AssertionError, ValueError) as e:  # the reference side failed: say so beside the headline
            result["parity_checked"]["vs_reference_error"] = repr(e)
            rc = 5
    if isinstance(result.get("decode_device_parse"), dict) and result["decode_device_parse"].get("pictures_equal_to_the_host_parsed_leg") is False:
        sys.stderr.write("[bench] pictures decoded with the device parser DIFFER from the host-parsed ones\n")
        rc = rc or 9
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


def reference_phase(result, checks, dec_md5, sel):
    """Every stream the legs above set aside is re-encoded by the real reference (one process each, CPU) and compared byte
    for byte; the headline's streams are also DECODED by the reference decoder and every picture's md5 compared with what
    the lockstep decoder delivered in the decode leg.  The CPU baselines are timed here too: worker 0 alone on the box
    (encode, then decode), then the headline's 8 workers at once (parallel_encode_yuv.sh's recipe)."""
    rw = RefWorkers([c.job for c in checks])
    try:
        head = [i for i, c in enumerate(checks) if c.leg == "headline"]
        rest = [i for i, c in enumerate(checks) if c.leg != "headline"]
        one = rw.go([head[0]], [min(48, checks[head[0]].n)])[0]              # one reference thread, alone on the box: a whole GOP
        one_dec = rw.dec([head[0]], [min(48, checks[head[0]].n)])[0]
        allr = rw.go(head, [checks[i].n for i in head])                      # the headline's workers at once
        decr = rw.dec(head, [checks[i].n for i in head]) if dec_md5 else []
        dec_rest = {}
        if rest:
            rw.go(rest, [checks[i].n for i in rest])                         # every other leg's streams at once
            wd = [i for i in rest if checks[i].dec_md5 is not None]
            if wd:
                for i, r in zip(wd, rw.dec(wd, [checks[i].n for i in wd])):
                    dec_rest[i] = r["md5"]
        mism, per_leg = [], {}
        for i, c in enumerate(checks):
            want = rw.frames(i)
            ok = want == c.got
            per_leg.setdefault(c.leg, {"streams": 0, "frames": 0, "mismatches": 0})
            per_leg[c.leg]["streams"] += 1
            per_leg[c.leg]["frames"] += len(want)
            if not ok:
                first = next((t for t, (a, b) in enumerate(zip(want, c.got)) if a != b), min(len(want), len(c.got)))
                mism.append((c.leg, c.stream, first))
                per_leg[c.leg]["mismatches"] += 1
            if i in dec_rest:  # this leg's decoder output against the reference decoder's, picture by picture
                nd = len(c.dec_md5)
                per_leg[c.leg]["decoded_pictures_compared"] = per_leg[c.leg].get("decoded_pictures_compared", 0) + nd
                if nd == 0 or c.dec_md5 != dec_rest[i][:nd]:
                    mism.append((c.leg + " (decode)", c.stream, -1))
                    per_leg[c.leg]["mismatches"] += 1
        dec_bad, dec_pics = [], 0
        for k, i in enumerate(head if dec_md5 else []):
            got = dec_md5.get(checks[i].stream, [])
            want = decr[k]["md5"][:len(got)]
            dec_pics += len(got)
            if not got or got != want:
                dec_bad.append(checks[i].stream)
    finally:
        rw.close()
    hc = [checks[i] for i in head]
    result["parity_checked"].update({"vs_reference_streams": len(hc), "vs_reference_frames_each": [c.n for c in hc], "streams": [c.stream for c in hc],
                                     "gop_phases": [c.phase for c in hc], "groups_covered": sorted(set(c.group for c in hc)),
                                     "mismatches": per_leg.get("headline", {}).get("mismatches", 0),
                                     "legs": per_leg, "mismatches_all_legs": len(mism),
                                     "decode_vs_reference_decoder": {"streams": len(head) if dec_md5 else 0, "pictures_md5_compared": dec_pics,
                                                                     "streams_differing": len(dec_bad)}})
    result["cpu_baseline"] = {"value": round(one["frames"] / (one["t1"] - one["t0"]), 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                              "sample": "first %d frames (1 I + %d P) of stream %d, reference C library -O3, 1 thread, alone on the box"
                                        % (one["frames"], one["frames"] - 1, hc[0].stream)}
    span = max(r["t1"] for r in allr) - min(r["t0"] for r in allr)
    result["cpu_baseline_8proc"] = {"value": round(sum(r["frames"] for r in allr) / span, 3), "unit": "frames/s", "cores": len(head), "kind": "reference",
                                    "sample": "%d reference processes at once, one closed-GOP stream each (%s frames), as parallel_encode_yuv.sh does"
                                              % (len(head), "/".join(str(r["frames"]) for r in allr))}
    if isinstance(result.get("decode"), dict) and "error" not in result["decode"]:
        result["decode"]["cpu_baseline_decode"] = {"value": round(one_dec["frames"] / max(1e-9, one_dec["t1"] - one_dec["t0"]), 2), "unit": "frames/s", "cores": 1,
                                                   "kind": "reference", "sample": "the reference decoder (dsv_dec) over the first %d pictures of stream %d, "
                                                   "1 thread, alone on the box" % (one_dec["frames"], hc[0].stream)}
        result["decode"]["vs_reference_decoder"] = {"streams": len(head) if dec_md5 else 0, "pictures_md5_compared": dec_pics, "streams_differing": len(dec_bad)}
    # the legs' own lines carry their verdicts too
    for leg, v in per_leg.items():
        if leg.startswith("c") and isinstance(result.get("configs"), dict) and leg in result["configs"]:
            result["configs"][leg]["vs_reference"] = v
        if leg.startswith("class_") and isinstance(result.get("content_class_legs"), dict) and leg[6:] in result["content_class_legs"]:
            result["content_class_legs"][leg[6:]]["vs_reference"] = v
        if leg.startswith("api_") and isinstance(result.get("api_legs"), dict):
            for name, pt in result["api_legs"].items():
                if isinstance(pt, dict) and pt.get("check_leg") == leg:
                    pt["vs_reference"] = v
        if leg.startswith("batch") and isinstance(result.get("batch_curve"), list):
            for pt in result["batch_curve"]:
                if "batch_%d" % pt["streams"] == leg:
                    pt["vs_reference_mismatches"] = v["mismatches"]
                    pt["vs_reference_frames"] = v["frames"]
    if mism:
        sys.stderr.write("[bench] MISMATCH against the reference: (leg, stream, first differing frame) = %s\n" % mism)
        return 4
    if dec_bad:
        sys.stderr.write("[bench] decoded pictures DIFFER from the reference decoder's: streams %s\n" % dec_bad)
        return 6
    return 0


MIX = {"cut": 1, "static": 1, "fast": 1}  # of every ten twin pairs; the other seven pan


def timed_leg(run, warm, k):
    """pre-roll + warm-up, then k timed steps: (frames, seconds, sorted per-step wall times of the groups)"""
    run.run(run.R + warm)
    g = run.step
    e = run.run(k, record=True)
    ms = sorted(x for grp in run.step_ms for x in grp)
    return run.frames_in(g, g + k), e, ms


def other_configs(hip, A, torch, args, vids, NV, S, W_, H_, seeds, checks):
    """the other BASELINE.json configurations, each a short run of the same engine (N = 1 only); two streams of each are
    set aside for the reference re-encode (another video each, another lockstep group each)"""
    cfgs = {}
    align = not args.no_phase_align

    def leg(name, run, warm, k, nref_frames, extra):
        f, e, _ = timed_leg(run, warm, k)
        p, b = run.twins_equal()
        cfgs[name] = {"value": round(f / e, 2), "unit": "frames/s", "streams": run.S, "steps": k, "ms_per_step": round(1e3 * e / k, 3),
                      "mpix_per_s": round(f / e * run.w * run.h / 1e6, 1), "twin_pairs_equal": p - b, "twin_pairs": p}
        cfgs[name].update(extra)
        for s in run.pick_reference_streams(2):
            checks.append(RefCheck(name, run, s, nref_frames))
        if b:
            raise AssertionError("%s: %d of %d twin stream pairs differ" % (name, b, p))

    # C2: 1280x720 4:2:0 -qp=60 -gop=48 effort 10
    r2 = EncodeRun(hip, A, torch, 1280, 720, "420", 60, 48, 10, S, args.groups, vids[NV:NV + 4], not args.no_stagger, seeds=[101 + k for k in range(4)],
                   phase_align=align)
    leg("c2_720p_420_qp60_gop48", r2, 4, 24, 24, {"input": "pinned_host, staggered GOP phases"})
    r2.free()
    # C3: 1920x1080 4:2:0 -qp=60 -gop=60 (the headline's geometry with the CLI's default GOP)
    r3 = EncodeRun(hip, A, torch, W_, H_, "420", 60, 60, 10, S, args.groups, vids[:min(NV, 8)], not args.no_stagger, seeds=seeds, phase_align=align)
    leg("c3_1080p_420_qp60_gop60", r3, 4, 16, 20, {"input": "pinned_host, staggered GOP phases"})
    r3.free()
    # C4: 1920x1080 4:4:4 lossless (-qp=100), every stream from its first (intra) frame; round trip through the decoder
    s4 = min(128, S)
    r4 = EncodeRun(hip, A, torch, W_, H_, "444", 100, 60, 10, s4, min(4, args.groups), vids[NV + 4:NV + 5], False, seeds=[201])
    leg("c4_1080p_444_lossless", r4, 2, 8, 10, {"frames": "P frames 2..9 of each stream (4:4:4 instance of the fast level-0 search)"})
    cfgs["c4_1080p_444_lossless"]["round_trip"] = lossless_round_trip(hip, A, r4, vids[NV + 4])
    r4.free()
    # 3840x2160 4:2:0: 32 x 32 blocks (dsv_encoder.c:1203-1211) -- the search's 32 x 32 forms (csrc/hme_fast32.h: k_hme_rows_l0_32, k_hme_rows_lx32);
    # every stream from its first (intra) picture; 64 streams = 16 pictures per launch in four groups
    r5 = EncodeRun(hip, A, torch, 3840, 2160, "420", 60, 48, 10, min(64, S), min(4, args.groups), vids[NV + 5:NV + 7], False, seeds=[301, 302])
    leg("c_2160p_420_qp60_gop48", r5, 2, 12, 8, {"frames": "P frames 2..13 of each stream; 32 x 32 blocks: the search's 32 x 32 block routines"})
    r5.free()
    return cfgs


BATCH_POINTS = [(1, 1), (8, 4), (16, 4), (48, 4), (192, 4)]  # (streams, lockstep groups): what a node that has fewer streams than the headline gets (groups: 4 re-measured against 1 - 24 per point, tools/probe/few_streams_groups.sh -- every lockstep group adds its ~100 launches per step to ONE submission path: 8 streams in 8 groups deliver 0.6 x what they do in 4)


def batch_curve(hip, A, torch, args, vids, NV, seeds, W_, H_, QP, GOP, effort, checks):
    """The headline's workload at 1 / 8 / 48 / 192 concurrent streams (BASELINE config 5 runs ONE closed-GOP segment per
    GPU; parallel_encode_yuv.sh:31-52 runs 8): frames/s, the median time a stream waits for its next frame, and stream 0's
    first frames set aside for the reference re-encode.  Same timed region as the headline (upload inside)."""
    out = []
    for S, G in BATCH_POINTS:
        G = min(G, S)
        k = 48
        run = EncodeRun(hip, A, torch, W_, H_, "420", QP, GOP, effort, S, G, vids[:min(NV, 8)], not args.no_stagger, seeds=seeds, phase_align=not args.no_phase_align,
                        mix=None if (args.no_mix or S < 20) else MIX, timed_from=(GOP if not args.no_stagger else 0) + 4, timed_steps=k)
        f, e, ms = timed_leg(run, 4, k)
        p, b = run.twins_equal() if S > 1 else (0, 0)
        out.append({"streams": S, "groups": run.G, "value": round(f / e, 2), "unit": "frames/s", "steps": k, "ms_per_step": round(1e3 * e / k, 3),
                    "ms_per_frame_p50": round(ms[len(ms) // 2], 3), "ms_per_frame_p90": round(ms[(len(ms) * 9) // 10], 3),
                    "intra_frames_timed": run.intra_in(run.step - k, run.step), "twin_pairs_equal": p - b, "twin_pairs": p})
        checks.append(RefCheck("batch_%d" % S, run, 0, 32))
        run.free()
        if b:
            raise AssertionError("batch curve, %d streams: %d of %d twin stream pairs differ" % (S, b, p))
    return out


def class_legs(hip, A, torch, args, vids, NV, seeds, W_, H_, QP, GOP, effort, checks):
    """frames/s of each content class on its own: 192 streams of ONE class (4 lockstep groups), staggered GOP phases, 24 timed
    steps; one stream of each leg goes to the reference"""
    out = {}
    for name in EncodeRun.CLASSES:
        k = 24
        run = EncodeRun(hip, A, torch, W_, H_, "420", QP, GOP, effort, 192, 4, vids[:min(NV, 8)], not args.no_stagger, seeds=seeds, phase_align=not args.no_phase_align,
                        mix=None if name == "pan" else {name: 10}, timed_from=(GOP if not args.no_stagger else 0) + 4, timed_steps=k)
        f, e, _ = timed_leg(run, 4, k)
        p, b = run.twins_equal()
        rep = run.class_report(run.step - k, run.step)[name]
        out[name] = {"value": round(f / e, 2), "unit": "frames/s", "streams": run.S, "steps": k, "bytes_per_picture": rep["bytes_per_picture"],
                     "intra_flips": rep["intra_flips"], "twin_pairs_equal": p - b, "twin_pairs": p}
        sel = max(range(run.S), key=lambda s: (len(run.out[s]) >= 40, -run.r0[s]))  # a stream whose cut (if any) lies in its first frames? no: longest history
        checks.append(RefCheck("class_" + name, run, sel, 64))
        run.free()
        if b:
            raise AssertionError("content class %s: %d of %d twin stream pairs differ" % (name, b, p))
    return out


class ApiRun:
    """what RefCheck needs of a leg that is not an EncodeRun: per stream the packets of every frame, and how to regenerate its input"""

    def __init__(self, w, h, fmt, qp, gop, effort, seeds, src):
        self.w, self.h, self.fmt, self.qp, self.gop, self.effort, self.seeds = w, h, fmt, qp, gop, effort, seeds
        self.src = src  # per stream: [(video, frame of it)] per local frame
        self.G, self.r0 = 1, [0] * len(src)
        self.out = [[] for _ in src]

    def ref_job(self, s, nframes):
        return (self.w, self.h, self.fmt, 0, self.qp, self.gop, self.effort, ["%d:%d" % (self.seeds[v], k) for v, k in self.src[s][:nframes]])


API_THREADS = (1, 4, 16)


def api_thread_legs(hip, A, vids, NV, seeds, W_, H_, QP, GOP, effort, checks):
    """Throughput through the reference's OWN entry points, used the way the reference is used: T host threads, each looping
    plain dsv_enc (dsv_encoder.h:190-199) on an encoder of its own with ordinary pageable DSV_FRAMEs (dsv_load_planar_frame over
    the caller's memory), then T threads looping dsv_dec (dsv_decoder.h:54-61) over those packets.  Nothing library-specific is
    called; concurrent callers are merged into lockstep steps inside the library (csrc/batch.h: Coalescer).  Thread 0's packets of
    every leg go to the reference for re-encode AND decode."""
    import numpy as np
    from codec_run import configure_encoder
    for name in ("dsv2hip_enc_queue_stats", "dsv2hip_dec_queue_stats"):
        getattr(hip, name).argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        getattr(hip, name).restype = None
    NF = len(vids[0])
    period = 2 * (NF - 1) if NF > 1 else 1
    warm, K = 4, 32
    legs = {}
    for T in API_THREADS:
        src = [[(s % NV, (lambda k: k if k < NF else period - k)(t % period)) for t in range(warm + K)] for s in range(T)]
        run = ApiRun(W_, H_, "420", QP, GOP, effort, seeds, src)
        # pageable copies of the pictures, made before the clock starts (a caller's own frame buffers)
        pics = {}
        for s in range(T):
            for v, k in src[s]:
                if (s, v, k) not in pics:
                    pics[(s, v, k)] = np.frombuffer(vids[v][k], dtype=np.uint8).copy()
        meta = A.mk_meta(W_, H_, A.SUBSAMP_420)
        encs = [A.ENCODER() for _ in range(T)]
        for e in encs:
            configure_encoder(hip, e, meta, qp=QP, gop=GOP, effort=effort)
        held = [[] for _ in range(T)]  # per thread, per frame: (BUF array, count) -- turned into bytes after the clock stops
        bar = threading.Barrier(T + 1)

        def enc_worker(s):
            e = encs[s]
            for phase, (a, b) in enumerate(((0, warm), (warm, warm + K))):
                bar.wait()
                for t in range(a, b):
                    v, k = src[s][t]
                    fr = hip.dsv_load_planar_frame(A.SUBSAMP_420, pics[(s, v, k)].ctypes.data, W_, H_)
                    bufs = (A.BUF * 4)()
                    n = hip.dsv_enc(C.byref(e), fr, bufs)
                    held[s].append((bufs, n))
                bar.wait()

        ths = [threading.Thread(target=enc_worker, args=(s,)) for s in range(T)]
        for th in ths:
            th.start()
        bar.wait()
        bar.wait()  # warm-up done (allocations, the intra picture)
        hip.dsv2hip_enc_queue_stats(None, 1)
        t0 = time.perf_counter()
        bar.wait()
        bar.wait()
        t_enc = time.perf_counter() - t0
        for th in ths:
            th.join()
        st = (C.c_ulonglong * 4)()
        hip.dsv2hip_enc_queue_stats(st, 0)
        for s in range(T):
            for bufs, n in held[s]:
                pk = []
                for i in range(n):
                    pk.append(C.string_at(bufs[i].data, bufs[i].len))
                    hip.dsv_buf_free(C.byref(bufs[i]))
                run.out[s].append(pk)
        for e in encs:
            hip.dsv_enc_free(C.byref(e))
        same = sum(1 for s in range(NV, T) if run.out[s] == run.out[s % NV])  # threads beyond the distinct videos repeat one: same bytes
        chk = RefCheck("api_enc_%d" % T, run, 0, warm + K)
        legs["dsv_enc_threads_%d" % T] = {"value": round(T * K / t_enc, 2), "unit": "frames/s", "threads": T, "frames_per_thread": K, "ms_per_call": round(1e3 * t_enc / K, 3),
                                          "queue": {"calls": st[0], "lockstep_steps": st[1], "largest_step": st[2], "leader_wait_us_per_step": round(st[3] / max(1, st[1]), 1)},
                                          "repeat_threads_equal": "%d/%d" % (same, max(0, T - NV)), "check_leg": "api_enc_%d" % T}
        if same != max(0, T - NV):
            raise AssertionError("dsv_enc threads leg, T=%d: threads coding the same video produced different packets" % T)

        # ---- the decode twin: T threads, each looping dsv_dec over its stream's packets ----
        decs = [A.DECODER() for _ in range(T)]
        got = [[] for _ in range(T)]
        bar2 = threading.Barrier(T + 1)

        def dec_worker(s):
            d = decs[s]
            for a, b in ((0, warm), (warm, warm + K)):
                bar2.wait()
                for t in range(a, b):
                    for pk in run.out[s][t]:
                        buf = A.BUF()
                        hip.dsv_mk_buf(C.byref(buf), len(pk) + 64)
                        C.memmove(buf.data, pk, len(pk))
                        buf.len = len(pk)
                        fp = C.POINTER(A.FRAME)()
                        fn = C.c_uint32(0)
                        if hip.dsv_dec(C.byref(d), C.byref(buf), C.byref(fp), C.byref(fn)) == A.DEC_OK and fp:
                            got[s].append(picture_planes(fp) if s == 0 or s >= NV else None)
                            hip.dsv_frame_ref_dec(fp)
                bar2.wait()

        ths = [threading.Thread(target=dec_worker, args=(s,)) for s in range(T)]
        for th in ths:
            th.start()
        bar2.wait()
        bar2.wait()
        hip.dsv2hip_dec_queue_stats(None, 1)
        t0 = time.perf_counter()
        bar2.wait()
        bar2.wait()
        t_dec = time.perf_counter() - t0
        for th in ths:
            th.join()
        hip.dsv2hip_dec_queue_stats(st, 0)
        for d in decs:
            hip.dsv_dec_free(C.byref(d))
        ndec = sum(len(g) for g in got)
        chk.dec_md5 = [planes_md5(pl) for pl in got[0]]
        checks.append(chk)
        legs["dsv_dec_threads_%d" % T] = {"value": round(T * K / t_dec, 2), "unit": "frames/s", "threads": T, "pictures": ndec, "ms_per_call": round(1e3 * t_dec / K, 3),
                                          "queue": {"calls": st[0], "lockstep_steps": st[1], "largest_step": st[2], "leader_wait_us_per_step": round(st[3] / max(1, st[1]), 1)},
                                          "check_leg": "api_enc_%d" % T}
        if ndec != T * (warm + K):
            raise AssertionError("dsv_dec threads leg, T=%d: %d pictures for %d packets" % (T, ndec, T * (warm + K)))
    legs["note"] = ("T host threads, each looping the reference's plain dsv_enc / dsv_dec on an instance of its own with pageable DSV_FRAMEs; %d warm-up + %d timed "
                    "calls per thread; thread 0 of every leg re-encoded and decoded by the reference (parity_checked.legs api_enc_T)" % (warm, K))
    return legs


def api_process_leg(vids, W_, H_, QP, GOP):
    """The reference's own parallel recipe (parallel_encode_yuv.sh:31-52) with the reference's own CLI: P = 8 processes, each
    `e -sfr=.. -nfr=.. -noeos=1` on one raw .yuv file, once with the CLI linked against this library (oracle/_ref/dsv2_dropin: 8
    processes share the one GPU) and once with the pure reference build (oracle/_ref/dsv2_ref: 8 CPU processes), timed end to end
    (process start, file input, encode, file output).  The concatenated outputs must be identical."""
    import dsvabi as A
    dropin = os.path.join(ROOT, "oracle", "_ref", "dsv2_dropin")
    if not (os.path.exists(dropin) and os.path.exists(A.REF_CLI)):
        return {"error": "oracle/_ref CLIs not built"}
    P, chunk = 8, 48  # (chunk = one GOP, parallel_encode_yuv.sh's chunk_per_gop)
    tmp = tempfile.mkdtemp(prefix="dsv2api", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        yuv = os.path.join(tmp, "in.yuv")
        NF = len(vids[0])
        with open(yuv, "wb") as f:  # P segments of `chunk` frames each: segment p = frames of video p % len(vids)
            for p in range(P):
                for t in range(chunk):
                    f.write(vids[p % len(vids)][t % NF])
        base = ["-y", "-inp=" + yuv, "-w=%d" % W_, "-h=%d" % H_, "-fps_num=30", "-fps_den=1", "-gop=%d" % GOP, "-qp=%d" % QP, "-rc_mode=0"]

        # the children get the environment a user's shell would have: none of this harness's runtime settings
        clean = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "DSV2_HOST_THREADS") and not k.startswith("ROCP")}

        def recipe(exe, tag, nproc, extra_env=None):
            outs = [os.path.join(tmp, "%s%d.dsv" % (tag, p)) for p in range(nproc)]
            env = dict(clean, **(extra_env or {}))
            t0 = time.perf_counter()
            procs = [subprocess.Popen([exe, "e"] + base + ["-out=" + outs[p], "-sfr=%d" % (p * chunk), "-nfr=%d" % chunk, "-noeos=1"],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env) for p in range(nproc)]
            rcs = [pr.wait() for pr in procs]
            dt = time.perf_counter() - t0
            data = b"".join(open(o, "rb").read() for o in outs)
            return dt, data, rcs

        recipe(dropin, "w", 1)  # (first process of the box pages the runtime in)
        d1, one, rc1 = recipe(dropin, "a", 1)
        d8, all8, rc8 = recipe(dropin, "b", P)
        tuned = {"GPU_MAX_HW_QUEUES": "2", "DSV2_HOST_THREADS": "2"}  # (what to export when many processes share a GPU)
        d8q, all8q, rc8q = recipe(dropin, "c", P, tuned)
        r8, ref8, rcr = recipe(A.REF_CLI, "r", P)
        ok = all8 == ref8 and all8q == ref8 and one == ref8[:len(one)] and not any(rc1 + rc8 + rc8q + rcr)
        out = {"processes": P, "frames_per_process": chunk, "dropin_1_process_fps": round(chunk / d1, 2), "dropin_8_processes_fps": round(P * chunk / d8, 2),
               "dropin_8_processes_2_hw_queues_fps": round(P * chunk / d8q, 2),
               "ratio_8_to_1": round((P * chunk / min(d8, d8q)) / (chunk / d1), 2), "reference_8_processes_fps": round(P * chunk / r8, 2),
               "speedup_vs_reference_recipe": round(r8 / min(d8, d8q), 2), "bytes": len(all8), "identical_to_reference_output": bool(ok),
               "note": "end to end per process: exec, HIP runtime + device context start-up (~0.4 s), raw .yuv read, %d frames encoded, .dsv written, "
                       "runtime tear-down; 8 drop-in processes share ONE GPU (and the host with this bench process, which holds a context of "
                       "its own); second figure with GPU_MAX_HW_QUEUES=2 DSV2_HOST_THREADS=2 exported" % chunk}
        if not ok:
            raise AssertionError("drop-in CLI recipe: outputs differ from the reference's (rcs %s %s %s)" % (rc1, rc8, rcr))
        return out
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def multi_rank_one_gpu(args, fps_one_rank):
    """What one GPU can prove about the N > 1 path (the 1 -> 8 curve itself needs an 8-GPU node and is the driver's to measure):
    (a) EIGHT ranks of this bench sharing this one GPU (DSV2_FORCE_DEVICE=0), 96 streams and 2 host cores each -- the whole
    multi-rank code path (rank spawn, per-rank streams, barrier + max-over-ranks timing, the ordered segment gather over gloo,
    every gathered segment verified) with the aggregate beside the one-rank 768-stream figure; (b) ONE rank forced through the
    distributed path on RCCL (--backend nccl): process-group init, all_reduce, all_gather and the gather on the hardware."""
    out = {}
    base = [sys.executable, os.path.abspath(__file__), "--no-extras", "--no-profile", "--no-cpu-baseline"]
    env = dict(os.environ)
    env.pop("DSV2_HOST_THREADS", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def sub(cmd, env2, key):
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env2)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            c = j["config"]
            return {"value": j["value"], "unit": "frames/s", "ranks": j["n_gpus"], "streams_per_rank": c["streams_per_gpu"], "groups_per_rank": c["groups"],
                    "steps": j["steps"], "ms_per_step": j["ms_per_step"], "exchange_backend": c.get("exchange_backend"), "final_gather_s": c.get("final_gather_s"),
                    "final_gather_check": c.get("final_gather_check"), "twin_pairs_equal": j["parity_checked"]["twin_pairs_equal"],
                    "twin_pairs": j["parity_checked"]["twin_pairs"], "host_cores_per_rank": c.get("host_cores_pinned"), "rc": r.returncode}
        except Exception as e:  # noqa: BLE001
            return {"error": repr(e), "leg": key}

    e8 = dict(env, DSV2_FORCE_DEVICE="0", GPU_MAX_HW_QUEUES="2")
    a = sub(base + ["--gpus", "8", "--backend", "gloo", "--streams", "96", "--groups", "1", "--host-cores", "2", "--steps", "24", "--warmup", "4"], e8, "eight_ranks")
    if "value" in a:
        a["ratio_to_one_rank_768_streams"] = round(a["value"] / fps_one_rank, 3)
        a["note"] = "8 processes x 96 streams on ONE GPU (DSV2_FORCE_DEVICE=0), gloo exchange, 2 host cores per rank; aggregate over the ranks, max-over-ranks time"
    out["eight_ranks_one_gpu_gloo"] = a
    b = sub(base + ["--gpus", "1", "--force-dist", "--backend", "nccl", "--streams", "96", "--groups", "1", "--steps", "8", "--warmup", "2"], env, "one_rank_rccl")
    if "value" in b:
        b["note"] = "one rank through the distributed path on RCCL: init_process_group(nccl), all_reduce, all_gather and the segment gather executed on the GPU"
    out["one_rank_rccl_path"] = b
    out["scaling_1_to_8_gpus"] = "unmeasured here: needs an 8-GPU node (the driver's SCALE run)"
    return out


def host_share(args, cores, fps_unrestricted):
    """the headline once more in a fresh process pinned to `cores` host cores (an 8-GPU node's share per rank)"""
    cmd = [sys.executable, os.path.abspath(__file__), "--host-cores", str(cores), "--streams", str(args.streams), "--groups", str(args.groups),
           "--steps", str(min(args.steps, 24)), "--warmup", str(min(args.warmup, 4)), "--no-extras", "--no-cpu-baseline", "--no-profile", "--decode-too"]
    if args.no_stagger:
        cmd.append("--no-stagger")
    if args.no_phase_align:
        cmd.append("--no-phase-align")
    env = dict(os.environ)
    env.pop("DSV2_HOST_THREADS", None)  # (this process exported its own pool size: the child sizes its pool from its cores)
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        j = json.loads(line)
        return {"cores": cores, "value": j["value"], "unit": "frames/s", "ratio_to_unrestricted": round(j["value"] / fps_unrestricted, 3),
                "host_cpu_cores_busy": j["config"]["host_cpu_cores_busy"], "host_threads": j["config"]["host_threads"], "steps": j["steps"],
                "twin_pairs_equal": j["parity_checked"]["twin_pairs_equal"], "twin_pairs": j["parity_checked"]["twin_pairs"],
                "decode": {k: j.get("decode", {}).get(k) for k in ("value", "unit", "decoders", "host_cpu_cores_busy", "error") if k in j.get("decode", {})},
                "note": "separate process, sched_setaffinity to %d cores before the GPU runtime starts; same workload and timed region" % cores}
    except Exception as e:  # noqa: BLE001
        return {"cores": cores, "error": repr(e)}


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
