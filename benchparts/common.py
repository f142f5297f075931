"""bench.py, shared part: constants, the algorithmic byte counts of SURVEY 8(d), where a rank's host side runs (GPU locality), the
picture generators, and EncodeRun -- the lockstep groups that drive dsv2hip_enc_batch_host, whose run() is the TIMED REGION."""
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")  # what the legs that re-run the bench in a child process start
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


MIX = {"cut": 1, "static": 1, "fast": 1}  # of every ten twin pairs; the other seven pan


def timed_leg(run, warm, k):
    """pre-roll + warm-up, then k timed steps: (frames, seconds, sorted per-step wall times of the groups)"""
    run.run(run.R + warm)
    g = run.step
    e = run.run(k, record=True)
    ms = sorted(x for grp in run.step_ms for x in grp)
    return run.frames_in(g, g + k), e, ms
