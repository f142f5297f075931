"""bench.py, every leg beside the headline: the decoder, the other BASELINE.json configurations, the batch curve, content classes, the
reference's own entry points (threads / processes), eight ranks on one GPU, the 2-host-core re-run."""
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
from .common import *  # noqa: F401,F403
from .reference import RefCheck  # noqa: F401

def decode_traffic(stage, pictures_per_launch):
    """HBM-side counter bytes of one launch group of the decode leg's dominant stage: committed PMC passes (profiles/pmc_traffic_decode.json:
    FETCH_SIZE + WRITE_SIZE per decoded picture and stage) x the pictures a group's step carries; None without the file"""
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_decode.json")))
        return int(pt["bytes_per_picture_by_stage"][stage] * pictures_per_launch)
    except (OSError, ValueError, KeyError):
        return None


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
                               "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": decode_traffic(dom, D // G),
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
    # 1920x800 4:2:0 (a 2.40:1 crop of 1080p): 32 x 16 blocks (dsv_encoder.c:1203-1211: wider than 1280 and at least twice as wide as high)
    # -- the search's two-quadrant forms (k_hme_rows_l0_32w, k_hme_rows_lx32w; until late in round 6 the general block routine)
    r6 = EncodeRun(hip, A, torch, 1920, 800, "420", 60, 48, 10, min(192, S), min(4, args.groups), vids[NV + 7:NV + 9], False, seeds=[401, 402])
    leg("c_1920x800_420_qp60_gop48", r6, 2, 12, 8, {"frames": "P frames 2..13 of each stream; 32 x 16 blocks: the search's 32 x 16 block routines"})
    r6.free()
    return cfgs


BATCH_POINTS = [(1, 1), (8, 4), (16, 4), (48, 4), (192, 4), (1536, 4)]  # (streams, lockstep groups): what a node that has fewer streams than the headline gets -- and, last, twice as many (126 GB of encoder instances: deeper launches, +4 ... 5 % frames/s at twice the time a frame waits) (groups: 4 re-measured against 1 - 24 per point, tools/probe/few_streams_groups.sh -- every lockstep group adds its ~100 launches per step to ONE submission path: 8 streams in 8 groups deliver 0.6 x what they do in 4)


def batch_curve(hip, A, torch, args, vids, NV, seeds, W_, H_, QP, GOP, effort, checks):
    """The headline's workload at 1 / 8 / 48 / 192 concurrent streams (BASELINE config 5 runs ONE closed-GOP segment per
    GPU; parallel_encode_yuv.sh:31-52 runs 8): frames/s, the median time a stream waits for its next frame, and stream 0's
    first frames set aside for the reference re-encode.  Same timed region as the headline (upload inside)."""
    out = []
    for S, G in BATCH_POINTS:
        G = min(G, S)
        k = 48 if S < 1000 else 24
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
    base = [sys.executable, BENCH_PY, "--no-extras", "--no-profile", "--no-cpu-baseline"]
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
    cmd = [sys.executable, BENCH_PY, "--host-cores", str(cores), "--streams", str(args.streams), "--groups", str(args.groups),
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
