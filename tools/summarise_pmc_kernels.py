#!/usr/bin/env python3
"""Joins the three passes of tools/pmc_kernels.sh: per kernel the counter traffic (FETCH_SIZE + WRITE_SIZE, separate
rocprofv3 --pmc passes), its duration (kernel trace), the rate against the 8 TB/s HBM peak and the ratio to the
algorithmic bytes of SURVEY.md 8(d).

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-byte requests of a wide coalesced streaming
read (16 B per lane) at 64 B, i.e. reports exactly half of the bytes; kernels whose loads are 16 B per lane are listed in
WIDE16 and their FETCH_SIZE is doubled.  Other widths are uncalibrated in the guide and are quoted as counted.
"""
import collections
import csv
import sys

d, streams, frames_per_stream = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = streams * frames_per_stream
N = 1920 * 1080
P = N * 3 // 2

# kernels whose global loads are 16 bytes per lane (uint4): FETCH_SIZE x 2
WIDE16 = ("k_inv_haar_u8x4", "k_quant_level4", "k_scatter", "k_reconstruct_w", "k_ingest16", "k_copy_linear", "k_inv_haar_tail", "k_fwd_haar_tail")
# algorithmic bytes per FRAME by kernel family (P = picture bytes, N = luma pixels); None = no streaming model
ALGO = [
    ("k_hme_rows_l0", 4 * N, "3 full-size luma planes + chroma (4 N)"),
    ("k_hme_rows_lx", 1 * N, "the coarser pyramid levels of the three lumas (N)"),
    ("k_hme_src_stats", 2.5 * N, "levels 0 and 1 of the source and of the previous source picture (2.5 N)"),
    ("k_inter_filters_b", 2 * P, "picture read + written once (2 P)"),
    ("k_intra_filter_b", 2 * N / 14, "luma read + written, intra pictures only (1 in 14 here)"),
    ("k_predict_w", 4 * P, "source + reference read, prediction + residual written (4 P)"),
    ("k_reconstruct_w", 3 * P, "prediction + residual read, picture written (3 P)"),
    ("k_fwd_haar_u8x4", P + 4 * P, "u8 picture read, int32 bands written (5 P)"),
    ("k_inv_haar_u8x4", 4 * P + P, "int32 bands read, u8 picture written (5 P)"),
    ("k_quant_level4", 3 * 4 * P * 63 / 64, "int32 coefficients read + written in place + dense values written (12 P x 63/64)"),
    ("k_scatter", 4 * P + 0.5 * P, "dense values read, (pos, val) list written"),
    ("k_extend", 2 * P * 0.1, "border strips"),
    ("k_ingest16", 2 * P, "packed picture read, bordered planes written (2 P)"),
    ("k_copy_linear", 2 * 1.1 * P, "padded source copied to the working picture"),
    ("k_ds2x4", 1.25 * N * 1.33, "pyramid: every level read once, the next written"),
]


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("dsv2::", "").replace("void ", "")
    n = n.split("(")[0]
    for fam in ("k_quant_level4", "k_predict_w", "k_quant_level"):  # one row per family: the instantiations split the planes of ONE pass
        if n.startswith(fam + "<"):
            return fam + "<*>"
    return n[:46]


dur = collections.defaultdict(float)
calls = collections.Counter()
for r in csv.DictReader(open(d + "/kernel_trace.csv")):
    k = short(r["Kernel_Name"])
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    calls[k] += 1
cnt = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}
for c in cnt:
    for r in csv.DictReader(open(d + "/pmc_%s.csv" % c)):
        if r["Counter_Name"] == c:
            cnt[c][short(r["Kernel_Name"])] += float(r["Counter_Value"])
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
KB = 1024.0
print("counter traffic per kernel, %d streams in one lockstep group, %d frames (1 intra + %d inter per stream); us and MB are PER FRAME" % (streams, frames, frames_per_stream - 1))
print("%-46s %7s %9s %9s %9s %8s %7s %9s  %s" % ("kernel", "calls", "us", "fetch MB", "write MB", "GB/s", "% peak", "x algo", "algorithmic bytes"))
tot_us = tot_b = 0.0
rows = sorted(dur, key=lambda k: -dur[k])
for k in rows[:24]:
    f = cnt["FETCH_SIZE"].get(k, 0.0) * KB
    w = cnt["WRITE_SIZE"].get(k, 0.0) * KB
    wide = any(k.startswith(x) for x in WIDE16)
    if wide:
        f *= 2
    us = dur[k] / frames
    b = (f + w) / frames
    algo = next(((a, why) for (n, a, why) in ALGO if k.startswith(n)), None)
    print("%-46s %7d %9.2f %9.2f %9.2f %8.0f %7.2f %9s  %s" % (k + ("*" if wide else ""), calls[k], us, f / frames / 1e6, w / frames / 1e6, b / us / 1e3 if us else 0,
                                                       100 * b / us / 1e3 / 8000 if us else 0, ("%.2f" % (b / algo[0])) if algo else "-", algo[1] if algo else ""))
    tot_us += us
    tot_b += b
print("%-46s %7s %9.2f %19.2f MB %7.0f %7.2f" % ("TOTAL (listed)", "", tot_us, tot_b / 1e6, tot_b / tot_us / 1e3, 100 * tot_b / tot_us / 1e3 / 8000))
print("* = loads are 16 B per lane: FETCH_SIZE doubled (gfx950 tallies their 128-byte requests at 64 B)")
