"""bench.py, reference side: the real reference (oracle/_ref) re-encodes / re-decodes the streams the legs picked -- parity part 2 -- and
is timed as the CPU baseline, after the GPU legs' clocks have stopped."""
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
