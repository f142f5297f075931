"""bench.py, roofline objects that come from committed profile passes (instruction volume, occupancy) and the in-kernel census."""
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
    # a wave64 vector instruction occupies the SIMD-32 for 2 clocks (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'); one
    # wavefront alone was MEASURED to issue an independent vector instruction every 5.4 clocks and a dependent one every 9
    # (tools/probe/issue_rate.cpp; the guide says 4): two peaks -- what the SIMDs can issue with enough wavefronts each, and what
    # they can with one
    peak2, peak4 = simds * clock / 2 / 1e9, simds * clock / 5.4 / 1e9
    peak_meas = simds * clock * 0.39 / 1e9  # tools/probe/issue_rate_chip.cpp: 0.36 - 0.39 vector instructions per nominal clock per SIMD from 3 - 4 wavefronts up
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
            "peak_measured": round(peak_meas, 1), "frac_of_measured_peak": round(ach / peak_meas, 4),
            "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 2 clocks per wave64 vector instruction (enough wavefronts per SIMD); / 5.4 clocks "
                         "is what ONE wavefront per SIMD was measured to issue (9 when each instruction waits for the one before: tools/probe/issue_rate.cpp) "
                         "-- the search runs at three per SIMD and is parked on memory counters 43 % of its wave-cycles; peak_measured: every CU running streams of "
                         "v_add_u32 saturates at 0.36 - 0.39 instructions per nominal clock per SIMD from three to four wavefronts up (tools/probe/issue_rate_chip.cpp)",
            "source": "committed PMC passes, not this run: " + iv.get("source", "profiles/instruction_volume.json")}
