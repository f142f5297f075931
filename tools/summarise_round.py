#!/usr/bin/env python3
"""gpurun_out/prof<N> (tools/profile_round.sh, ROUND=<N>) -> profiles/r0<N>_*.txt, profiles/pmc_traffic.json, profiles/instruction_volume.json.
usage: python3 tools/summarise_round.py [round number, default 6]

The timed region of the traced headline is found from the trace itself: its level-0 search launches are the LAST
steps x groups launches of that kernel (nothing runs behind the headline with --no-extras); rounds 1-3 took "the last 60 % of the
search span", which at 48 pre-roll steps is mostly pre-roll and the stage-profile pass -- the source of round 3's "no kernel
running 19.7 % of the time"."""
import collections
import csv
import gzip
import hashlib
import io
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
ROUND = int(sys.argv[1]) if len(sys.argv) > 1 else 6
TAG = "r%02d" % ROUND
src = os.path.join(ROOT, "gpurun_out", "prof%d" % ROUND)
dst = os.path.join(ROOT, "profiles")


def persist_workers():
    """the level-0 launch's persistent row workers: the default in csrc/hme.hip (g_hme_persist), unless the run overrode it"""
    import re
    m = re.search(r"g_hme_persist\s*=[^;]*?:\s*(\d+)\s*;", open(os.path.join(ROOT, "digital-subband-video-2_amd", "csrc", "hme.hip")).read())
    return int(os.environ.get("DSV2_HME_PERSIST", m.group(1) if m else 0))
L0 = "k_hme_rows_l0"
N, P = 1920 * 1080, 1920 * 1080 * 3 // 2


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dsv2::", "")


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("hme.hip", "hme_fast.h", "hme.h", "blockstat.h", "dev.h"):
        h.update(open(os.path.join(ROOT, "digital-subband-video-2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def trace_part():
    traced = json.loads([l for l in open(os.path.join(src, "bench_traced.json")) if l.startswith("{")][-1])
    steps, groups = traced["steps"], traced["config"]["groups"]
    rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    tr = []
    for r in csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, "kernel_trace.csv.gz")))):
        tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
    tr.sort()
    l0 = [r for r in tr if L0 in r[2]]
    # the profiled pass (stage timing) follows the timed steps: leave its launches out
    prof_steps = 6 if traced.get("roofline") else 0
    last = len(l0) - prof_steps * groups
    timed = l0[last - steps * groups:last]
    lo, hi = timed[0][0], timed[-1][1]
    span = hi - lo
    with open(os.path.join(dst, TAG + "_rocprof_kernel_stats.txt"), "w") as f:
        f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras   (MI355X, %d streams / %d groups)\n"
                % (traced["config"]["streams_per_gpu"], groups))
        f.write("bench line under the profiler: %.1f frames/s, %.2f ms/step; sum of kernel durations over the whole run %.1f ms\n"
                % (traced["value"], traced["ms_per_step"], total / 1e6))
        f.write("(durations of concurrently running kernels overlap: the groups share the GPU)\n\n")
        f.write("%-78s %8s %12s %12s %7s\n" % ("kernel", "calls", "total ms", "avg us", "%"))
        for r in rows[:40]:
            f.write("%-78s %8s %12.2f %12.2f %7.2f\n" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        d = [(e - s) / 1e3 for s, e, _, _ in timed]
        f.write("\n%s (level-0 search, the dominant kernel; a persistent launch of %d row workers per lockstep group): the %d launches of the "
                "%d timed steps x %d groups: mean %.1f us, min %.1f, max %.1f -- bench.py HIP-event span of that launch in its profiled steps: %.1f us\n"
                % (L0, persist_workers(), len(d), steps, groups, sum(d) / len(d), min(d), max(d), traced.get("roofline", {}).get("avg_launch_us", float("nan"))))
        f.write("algorithmic bytes of a launch: 4 N x %d pictures = %.0f MB -> %.1f GB/s = %.4f of 8 TB/s\n"
                % (traced["config"]["streams_per_gpu"] // groups, 4 * N * (traced["config"]["streams_per_gpu"] // groups) / 1e6,
                   4 * N * (traced["config"]["streams_per_gpu"] // groups) / (sum(d) / len(d) * 1e-6) / 1e9,
                   4 * N * (traced["config"]["streams_per_gpu"] // groups) / (sum(d) / len(d) * 1e-6) / 8e12))
    # concurrency over the TRUE timed region
    ev = []
    for s, e, n, q in tr:
        s2, e2 = max(s, lo), min(e, hi)
        if e2 > s2:
            k = 1 if "k_hme_rows" in n else 0
            ev.append((s2, 1, k))
            ev.append((e2, -1, -k))
    ev.sort()
    conc, busy, srch, both, cur, cs, prev = {}, 0, 0, 0, 0, 0, lo
    for t, dd, k in ev:
        dt = t - prev
        if dt > 0:
            conc[min(cur, 8)] = conc.get(min(cur, 8), 0) + dt
            busy += dt if cur else 0
            srch += dt if cs else 0
            both += dt if cs and cur > cs else 0
        cur += dd
        cs += k
        prev = t
    by, cnt = collections.Counter(), collections.Counter()
    for s, e, n, q in tr:
        s2, e2 = max(s, lo), min(e, hi)
        if e2 > s2:
            by[n] += e2 - s2
            cnt[n] += 1
    with open(os.path.join(dst, TAG + "_kernel_concurrency.txt"), "w") as f:
        f.write("kernel concurrency over the %.1f ms of the headline's TIMED region (%d steps x %d groups, delimited by its own level-0 search launches;\n"
                "rounds 1-3 analysed 'the last 60 %% of the search span', which is mostly pre-roll + the stage-profile pass: round 3's '0 kernels 19.7 %%' came from there):\n"
                % (span / 1e6, steps, groups))
        f.write("  some kernel running %.1f %%, a search launch running %.1f %%, a search launch AND another kernel %.1f %%\n"
                % (100.0 * busy / span, 100.0 * srch / span, 100.0 * both / span))
        f.write("  kernels running at once: " + "  ".join("%d%s: %.1f %%" % (c, "+" if c == 8 else "", 100.0 * conc.get(c, 0) / span) for c in range(0, 9)) + "\n")
        f.write("  kernel time summed over the groups' streams: %.2f x the region\n" % (sum(by.values()) / span))
        f.write("  %-40s %9s %9s %9s\n" % ("kernel", "% region", "launches", "avg us"))
        for n, v in by.most_common(24):
            f.write("  %-40s %8.1f%% %9d %9.1f\n" % (n[:40], 100.0 * v / span, cnt[n], v / cnt[n] / 1e3))
    return traced


def pmc_part(traced):
    groups, steps = traced["config"]["groups"], 6
    agg = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rs = [r for r in csv.DictReader(open(os.path.join(src, "pmc_%s.csv" % c))) if r["Counter_Name"] == c and L0 in r["Kernel_Name"]]
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
        agg[c] = [float(r["Counter_Value"]) for r in rs][-steps * groups:]  # the timed steps' launches (every picture of a group started)
    nl = len(agg["FETCH_SIZE"])
    fetch, write = sum(agg["FETCH_SIZE"]) / nl, sum(agg["WRITE_SIZE"]) / max(1, len(agg["WRITE_SIZE"]))
    per_launch = (fetch + write) * 1024.0
    pics = traced["config"]["streams_per_gpu"] // groups
    with open(os.path.join(dst, TAG + "_pmc_hme.txt"), "w") as f:
        f.write("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-include-regex %s -- python3 bench.py --steps 6 --warmup 3 "
                "(the bench's own layout: %d streams, %d groups, staggered GOP phases, content mix)\n" % (L0, traced["config"]["streams_per_gpu"], groups))
        f.write("units: KiB per launch as reported; the search's loads are 2 - 4 bytes per lane, so the gfx950 half-count correction for 16-byte streaming "
                "reads does not apply.  The %d launches of the 6 timed steps (176 - 192 inter pictures each):\n" % nl)
        f.write("  FETCH_SIZE mean %.1f KiB, WRITE_SIZE mean %.1f KiB -> %.2f MB fetched + written per launch = %.2f x the algorithmic 4 N x %d pictures = %.2f MB\n"
                % (fetch, write, per_launch / 1e6, per_launch / (4.0 * N * pics), pics, 4.0 * N * pics / 1e6))
    json.dump({"stage": "hme_level0", "kernel": L0, "streams_per_gpu": traced["config"]["streams_per_gpu"], "groups": groups, "stagger": True,
               "phase_aligned": bool(traced["config"].get("phase_aligned_groups")), "bytes_per_launch": round(per_launch),
               "kernel_source_sha16": open(os.path.join(src, "kernel_source_sha16.txt")).read().strip(),
               "source": "profiles/%s_pmc_hme.txt (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)" % TAG},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)


def excl_part():
    r = json.loads([l for l in open(os.path.join(src, "excl.json")) if l.startswith("{")][-1])
    frames = r["config"]["streams_per_gpu"] * (r["steps"] + r["warmup"])
    rows = [x for x in csv.DictReader(open(os.path.join(src, "excl_kernel_stats.csv"))) if "rocclr_fillBuffer" not in x["Name"]]
    tot = sum(float(x["TotalDurationNs"]) for x in rows)
    with open(os.path.join(dst, TAG + "_exclusive_kernel_costs.txt"), "w") as f:
        f.write("one lockstep group of 96 streams alone on the GPU (rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 96 --groups 1 --no-stagger --no-mix "
                "--steps 12 --warmup 2): %.1f frames/s, %.2f ms/step; %d frames traced (1 intra + 13 inter per stream)\n" % (r["value"], r["ms_per_step"], frames))
        f.write("sum of kernel durations %.1f us per frame (set-up memsets left out)\n" % (tot / 1e3 / frames))
        for x in rows[:36]:
            f.write("  %-66s %6s calls %9.1f us avg %7.2f us/frame\n" % (x["Name"][:66], x["Calls"], float(x["AverageNs"]) / 1e3, float(x["TotalDurationNs"]) / 1e3 / frames))


def insts_part():
    frames = 32 * 6
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, "insts.csv.gz")))):
        agg[short(r["Kernel_Name"])[-44:]][r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for v in agg.values() for c in v})
    tot = collections.defaultdict(float)
    with open(os.path.join(dst, TAG + "_instruction_volume.txt"), "w") as f:
        f.write("rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD, every kernel of: bench.py --streams 32 --groups 1 --steps 4 --warmup 2 --no-stagger\n")
        f.write("wavefront-instructions per frame (1 intra + 5 inter pictures per stream), thousands: %s\n" % names)
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
            if sum(v.values()) / frames / 1e3 < 20:
                continue
            f.write("  %-46s %s\n" % (k, "  ".join("%9.1f" % (v.get(c, 0) / frames / 1e3) for c in names)))
            for c in names:
                tot[c] += v.get(c, 0) / frames / 1e3
        f.write("  %-46s %s\n" % ("TOTAL (kernels above)", "  ".join("%9.1f" % tot[c] for c in names)))
    json.dump({"vector_per_frame": round(tot["SQ_INSTS_VALU"] * 1e3), "scalar_per_frame": round(tot.get("SQ_INSTS_SALU", 0) * 1e3),
               "mix": "1 intra + 5 inter pictures per stream, 32 streams in one lockstep group",
               "source": "profiles/%s_instruction_volume.txt (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU, every kernel of the run summed)" % TAG},
              open(os.path.join(dst, "instruction_volume.json"), "w"))


def decode_part(tag="decode", mode="w"):
    """kernels of the decode leg: everything the traced process launched after its last encoder kernel (k_ent_out)"""
    line = json.loads([l for l in open(os.path.join(src, tag + ".json")) if l.startswith("{")][-1])
    tr = []
    for r in csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, tag + "_kernel_trace.csv.gz")))):
        tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    tr.sort()
    last_enc = max(e for s, e, n in tr if "k_ent_out" in n)
    dec = [(s, e, n) for s, e, n in tr if s > last_enc]
    span = dec[-1][1] - dec[0][0]
    by, cnt = collections.Counter(), collections.Counter()
    for s, e, n in dec:
        by[n] += e - s
        cnt[n] += 1
    d = line.get("decode", {})
    with open(os.path.join(dst, TAG + "_decode_kernel_stats.txt"), mode) as f:
        if mode == "a":
            f.write("\n\n==== the same leg with DSV2_DEC_DEVICE_PARSE=1: the P pictures' plane sections parsed on the device (k_dec_parse, csrc/dec_parse_dev.hip) ====\n")
        f.write("rocprofv3 --kernel-trace -- python3 bench.py --no-extras --decode-too --streams 256 --groups 4 --steps 24 (tools/profile_decode.sh): the kernels launched\n"
                "after the last encoder kernel = the decode leg (warm-up, %d timed pictures, the stage-event steps); 256 decoders in 4 lockstep groups\n" % d.get("frames", 0))
        f.write("decode leg under the profiler: %s frames/s, host_cpu_cores_busy %s; roofline object of the line: %s\n"
                % (d.get("value"), d.get("host_cpu_cores_busy"), json.dumps(d.get("roofline"))))
        f.write("span of the decode kernels %.1f ms; kernel time summed %.1f ms (groups overlap)\n\n" % (span / 1e6, sum(by.values()) / 1e6))
        f.write("%-44s %8s %12s %12s %8s\n" % ("kernel", "calls", "total ms", "avg us", "% span"))
        for n, v in by.most_common(30):
            f.write("%-44s %8d %12.2f %12.1f %7.1f%%\n" % (n[:44], cnt[n], v / 1e6, v / cnt[n] / 1e3, 100.0 * v / span))


def occ_part(traced):
    """profiles/<tag>_occupancy.txt: resident waves per SIMD and issue utilisation per kernel -- measured with each dispatch ALONE on the
    chip (a rocprofv3 --pmc run serialises the dispatches) in the headline's own launch shapes, then joined with the un-serialised
    trace of the same command (part `trace`): how long each kernel is in flight beside the other groups' kernels, and what the
    counters it brought along add up to over the timed region."""
    SIMDS = 1024.0
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    nl = collections.Counter()
    meta = {}
    for r in csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, "occ.csv.gz")))):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            nl[k] += 1
            meta[k] = (r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "0"), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"), r.get("Workgroup_Size", "?"))
    alone = collections.defaultdict(float)
    for r in csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, "occ_kernel_trace.csv.gz")))):
        alone[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    # the un-serialised trace: the timed region, as in trace_part
    steps, groups = traced["steps"], traced["config"]["groups"]
    tr = []
    for r in csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, "kernel_trace.csv.gz")))):
        tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    tr.sort()
    l0 = [r for r in tr if L0 in r[2]]
    prof_steps = 6 if traced.get("roofline") else 0
    last = len(l0) - prof_steps * groups
    timed = l0[last - steps * groups:last]
    lo, hi = timed[0][0], timed[-1][1]
    span_us = (hi - lo) / 1e3
    load_us, load_n = collections.defaultdict(float), collections.Counter()
    for s0, e0, n in tr:
        s2, e2 = max(s0, lo), min(e0, hi)
        if e2 > s2:
            load_us[n] += (e2 - s2) / 1e3
            load_n[n] += 1
    rows = []
    tot_valu = tot_salu = tot_wave = 0.0
    clocks = []
    for k, v in agg.items():
        n = nl[k]
        if not n or not v.get("GRBM_GUI_ACTIVE"):
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0 / n            # shader clocks per launch (the counter sums the 8 XCDs)
        us = alone[k] / n
        if us > 300:
            clocks.append((us, cyc / us / 1e3))
        wc = v.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / n      # wave-clocks per launch (the SQ counters tick in quad-cycles)
        res = wc / (cyc * SIMDS)
        valu = v.get("SQ_INSTS_VALU", 0.0) / n * 2.0 / (cyc * SIMDS)        # wave64 instruction = 2 clocks of a SIMD's issue (bench.py: roofline.issue)
        valu_busy = v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / n / (cyc * SIMDS)
        salu = v.get("SQ_INST_CYCLES_SALU", 0.0) * 4.0 / n / (cyc * SIMDS)
        parked = v.get("SQ_WAIT_ANY", 0.0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1.0))
        stall = v.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1.0))
        ln = load_n.get(k, 0)
        lus = load_us[k] / ln if ln else 0.0
        vg = meta[k][0]
        try:
            slots = min(8, 512 // (((int(vg) + int(meta[k][1] or 0)) + 7) // 8 * 8))
        except (ValueError, ZeroDivisionError):
            slots = 0
        rows.append(dict(k=k, n=n, waves=v.get("SQ_WAVES", 0.0) / n, us=us, res=res, valu=valu, valu_busy=valu_busy, salu=salu, parked=parked, stall=stall,
                         ln=ln, lus=lus, inflight=load_us[k] / span_us, vgpr=vg, slots=slots, lds=meta[k][3], wg=meta[k][4],
                         res_load=(res * us / lus if lus else 0.0)))
        tot_valu += ln * v.get("SQ_INSTS_VALU", 0.0) / n * 2.0
        tot_salu += ln * v.get("SQ_INST_CYCLES_SALU", 0.0) * 4.0 / n
        tot_wave += ln * wc
    clk = sum(u * c for u, c in clocks) / max(1e-9, sum(u for u, _ in clocks)) if clocks else 2.4   # GHz, weighted over the long dispatches
    region_cyc = span_us * 1e3 * clk
    rows.sort(key=lambda r: -r["inflight"])
    with open(os.path.join(dst, TAG + "_occupancy.txt"), "w") as f:
        f.write("What the machine is short of (tools/profile_round.sh part `occ` + part `trace`; summarised by tools/summarise_round.py).\n\n")
        f.write("ALONE: rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY\n"
                "GRBM_GUI_ACTIVE -- python3 bench.py --steps 4 --warmup 2 --no-extras (the headline's command: %d streams, %d groups).  A --pmc run executes the\n"
                "dispatches one at a time, so every kernel is measured in the headline's launch shape with the chip to itself: per launch,\n"
                "  clocks = GRBM_GUI_ACTIVE / 8;  resident waves per SIMD = SQ_WAVE_CYCLES x 4 / (clocks x 1 024 SIMDs);\n"
                "  VALU issue = SQ_INSTS_VALU x 2 clocks / (clocks x 1 024) (the convention of roofline.issue); 'VALU busy' = SQ_ACTIVE_INST_VALU x 4 / (clocks x 1 024);\n"
                "  SALU = SQ_INST_CYCLES_SALU x 4 / (clocks x 1 024); parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt, barriers); stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.\n"
                "  slots = wavefronts per SIMD the register allocation admits (512 / VGPRs, at most 8).  Effective clock of the long dispatches: %.2f GHz.\n"
                % (traced["config"]["streams_per_gpu"], groups, clk))
        f.write("UNDER LOAD: the un-serialised rocprofv3 --kernel-trace of the same command (%s_rocprof_kernel_stats.txt), timed region %.1f ms: mean launch,\n"
                "share of the region with a launch of the kernel in flight (summed over the groups), stretch = under load / alone, and\n"
                "'waves/SIMD under load' = resident waves alone x alone / under load: what the kernel holds on average while in flight if its\n"
                "wave-clocks are conserved -- a LOWER bound (waits lengthen under load); not a counter reading.\n\n" % (TAG, span_us / 1e3))
        f.write("%-30s %5s %4s %6s | %8s %9s %6s %6s %6s %6s %6s %6s | %9s %8s %7s %9s\n"
                % ("kernel", "VGPR", "slot", "wg", "launches", "alone us", "w/SIMD", "VALU", "busy", "SALU", "parked", "stall", "load us", "in fl.", "stretch", "w/SIMD ld"))
        for r in rows:
            if r["inflight"] < 0.002 and r["us"] < 50:
                continue
            f.write("%-30s %5s %4d %6s | %8d %9.1f %6.2f %5.1f%% %5.1f%% %5.1f%% %5.1f%% %5.1f%% | %9.1f %7.1f%% %7.2f %9.2f\n"
                    % (r["k"][:30], r["vgpr"], r["slots"], r["wg"], r["n"], r["us"], r["res"], 100 * r["valu"], 100 * r["valu_busy"], 100 * r["salu"],
                       100 * r["parked"], 100 * r["stall"], r["lus"], 100 * r["inflight"], (r["lus"] / r["us"] if r["us"] else 0), r["res_load"]))
        summary = {"resident_waves_per_simd": round(tot_wave / (region_cyc * SIMDS), 2), "valu_issue": round(tot_valu / (region_cyc * SIMDS), 4),
                   "salu_issue": round(tot_salu / (region_cyc * SIMDS), 4), "clock_GHz": round(clk, 3),
                   "source": "profiles/%s_occupancy.txt (per-kernel counters measured alone x launches in the timed region of the un-serialised trace)" % TAG}
        f.write("\nWHOLE CHIP over the timed region (counters each launch brings along, summed over the launches in the region, over region clocks x 1 024 SIMDs):\n"
                "  resident waves per SIMD >= %.2f (of 8 slots; wave-clocks as measured alone, a lower bound)\n  VALU issue %.1f %%   SALU %.1f %%\n"
                % (summary["resident_waves_per_simd"], 100 * summary["valu_issue"], 100 * summary["salu_issue"]))
    json.dump(summary, open(os.path.join(dst, "occupancy.json"), "w"), indent=1)


def census_part():
    """appends the in-kernel census (part `census`) to <tag>_occupancy.txt and makes it the figure bench.py quotes"""
    line = json.loads([l for l in open(os.path.join(src, "census.json")) if l.startswith("{")][-1])
    c = line.get("census") or {}
    if "kernels" not in c:
        return
    with open(os.path.join(dst, TAG + "_occupancy.txt"), "a") as f:
        f.write("\nMEASURED UNDER LOAD (csrc/prio.h, `make census` build; no profiler, nothing serialised): DSV2_CENSUS=1 python3 bench.py --no-extras --steps %d\n"
                "-> %.1f frames/s with the instrumented library (%d streams, %d groups).  The first thread of every workgroup reads the 100 MHz real-time counter\n"
                "at its first and last instruction; ticks x wavefronts of the group, summed per kernel, / 1e8 / %.3f s timed region / 1 024 SIMDs\n"
                "= mean resident wavefronts per SIMD of that kernel OVER THE WHOLE REGION (waiting included; 8 slots per SIMD):\n\n"
                % (line["steps"], line["value"], line["config"]["streams_per_gpu"], line["config"]["groups"], c["elapsed_s"]))
        f.write("  %-30s %12s %12s %16s\n" % ("kernel", "waves/SIMD", "workgroups", "mean life us"))
        for r in c["kernels"]:
            if r["waves_per_simd"] >= 0.002:
                f.write("  %-30s %12.3f %12d %16.2f\n" % (r["kernel"][:30], r["waves_per_simd"], r["workgroups"], r["mean_group_life_us"]))
        f.write("  %-30s %12.2f   of 8 slots\n" % ("ALL KERNELS", c["resident_waves_per_simd"]))
    p = os.path.join(dst, "occupancy.json")
    oc = json.load(open(p)) if os.path.exists(p) else {}
    oc["resident_waves_per_simd_lower_bound_from_counters"] = oc.get("resident_waves_per_simd")
    oc["resident_waves_per_simd"] = c["resident_waves_per_simd"]
    oc["source"] = "profiles/%s_occupancy.txt (measured inside the kernels under load: csrc/prio.h census build, all lockstep groups running)" % TAG
    json.dump(oc, open(p, "w"), indent=1)


def decode_pmc_part():
    """profiles/pmc_traffic_decode.json: counter traffic (FETCH_SIZE + WRITE_SIZE, separate passes) of the decode leg's kernels per decoded picture,
    by stage of bench.py's decode roofline.  The decode phase = the dispatches behind the last encoder kernel (k_ent_out); pictures = the rows of the
    k_copy_linear launches (one row per picture delivered)."""
    WIDE16 = ("k_inv_haar_u8x4", "k_copy_linear", "k_zero_linear", "k_inv_haar_i32x4", "k_inv_haar_tail")  # 16-byte loads: FETCH_SIZE counts half (gfx950)
    stage_of = [("k_zero_linear_if", "recon_filters"), ("k_zero_linear", "quant_compact"), ("k_dequant", "quant_compact"), ("k_dec_parse", "quant_compact"),
                ("k_inv_", "inv_sbt"), ("k_predict_w", "recon_filters"), ("k_inter_filters", "recon_filters"), ("k_intra_filter", "recon_filters"),
                ("k_extend", "extend"), ("k_copy_linear", "extend"), ("k_copy_plane", "extend"), ("k_to420", "extend")]
    per_stage = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: [0.0, 0.0])
    pictures = None
    for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        tr = list(csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, "decode_pmc_%s_trace.csv.gz" % c)))))
        tr.sort(key=lambda r: int(r["Dispatch_Id"]))
        last_enc = max(int(r["Dispatch_Id"]) for r in tr if "k_ent_out" in r["Kernel_Name"])
        if pictures is None:
            pictures = sum(int(r["Grid_Size_Y"]) for r in tr if int(r["Dispatch_Id"]) > last_enc and short(r["Kernel_Name"]).startswith("k_copy_linear"))
        for r in csv.DictReader(io.TextIOWrapper(gzip.open(os.path.join(src, "decode_pmc_%s.csv.gz" % c)))):
            if r["Counter_Name"] != c or int(r["Dispatch_Id"]) <= last_enc:
                continue
            k = short(r["Kernel_Name"])
            v = float(r["Counter_Value"]) * 1024.0  # KiB as reported
            if c == "FETCH_SIZE" and k.startswith(WIDE16):
                v *= 2.0
            per_kernel[k][ci] += v
            for pre, st in stage_of:
                if k.startswith(pre):
                    per_stage[st] += v
                    break
    pictures = max(1, pictures or 1)
    out = {"pictures": pictures, "bytes_per_picture_by_stage": {k: round(v / pictures) for k, v in per_stage.items()},
           "bytes_per_picture_by_kernel": {k: [round(a / pictures), round(b / pictures)] for k, (a, b) in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1]))[:16]},
           "note": "FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes) of the decode leg's dispatches, 256 decoders in 4 groups, plane sections parsed on "
                   "the host; FETCH_SIZE doubled for the kernels whose loads are 16 bytes per lane (gfx950 tallies their 128-byte requests at 64 B)",
           "source": "tools/profile_decode.sh -> tools/summarise_round.py"}
    json.dump(out, open(os.path.join(dst, "pmc_traffic_decode.json"), "w"), indent=1)
    with open(os.path.join(dst, TAG + "_decode_kernel_stats.txt"), "a") as f:
        f.write("\n\n==== counter traffic of the decode leg (profiles/pmc_traffic_decode.json), bytes per decoded picture ====\n")
        for k, v in sorted(out["bytes_per_picture_by_stage"].items(), key=lambda kv: -kv[1]):
            f.write("  stage %-16s %12d\n" % (k, v))
        for k, (a, b) in out["bytes_per_picture_by_kernel"].items():
            f.write("  %-40s fetched %12d written %12d\n" % (k[:40], a, b))


have = lambda f: os.path.exists(os.path.join(src, f))
traced = trace_part() if have("kernel_trace.csv.gz") else None
if traced and have("pmc_FETCH_SIZE.csv") and have("kernel_source_sha16.txt"):
    pmc_part(traced)
if traced and have("occ.csv.gz") and have("occ_kernel_trace.csv.gz"):
    occ_part(traced)
if have("census.json"):
    census_part()
if have("excl_kernel_stats.csv"):
    excl_part()
if have("insts.csv.gz"):
    insts_part()
if have("decode_kernel_trace.csv.gz") and have("decode.json"):
    decode_part()
    if have("decode_dev_kernel_trace.csv.gz") and have("decode_dev.json"):
        decode_part("decode_dev", "a")
    if have("decode_pmc_FETCH_SIZE.csv.gz") and have("decode_pmc_WRITE_SIZE.csv.gz"):
        decode_pmc_part()
print("summarised into", dst)
