"""Closed-GOP segment sharding across ranks and the final ordered gather.

The reference's only parallelism is process-level: `parallel_encode_yuv.sh:31-52` forks one encoder
per closed-GOP chunk (`-sfr=N -nfr=chunk -noeos=1`) and `cat`s the outputs in order (`:50,74`).  Here
each rank (one process per GPU) encodes the segments assigned to it with fresh encoder state and the
packet bytes are gathered to rank 0 in segment order.  There is no collective on the data path; the
gather is the only exchange:

  * one all_gather of a small int64 table per rank: (segment id, byte length) rows, built from a
    Python list in one go;
  * the payloads UNPADDED, point to point: every rank > 0 sends its segments as one flat byte tensor
    in pieces of at most `chunk` bytes, rank 0 receives each piece into one reusable buffer and drops
    it into a preallocated host array -- rank 0 never holds more than `chunk` bytes of another rank's
    data on the device, whatever the run's size (RCCL send/recv on GPUs, gloo in the CPU tests);
  * rank 0 writes every segment at its final offset of ONE output buffer (no per-segment joins).
"""
import numpy as np
import torch

CHUNK = 256 << 20  # bytes per point-to-point transfer


def assign_segments(nseg, world):
    """Round-robin: segment s belongs to rank s % world.  Returns a list (per rank) of segment ids."""
    return [[s for s in range(nseg) if s % world == r] for r in range(world)]


def segment_id(rank, world, local_index):
    """The global id of a rank's `local_index`-th segment under assign_segments' round-robin."""
    return local_index * world + rank


def frame_range(seg, gop, nframes):
    """Frames [start, end) of segment `seg` (the reference's -sfr / -nfr arguments)."""
    start = seg * gop
    return start, min(nframes, start + gop)


def _exchange_tables(dist, world, ids, lens, device):
    """every rank's (segment id, length) rows -> list (per rank) of int64 numpy arrays [n_r, 2]"""
    nmine = torch.tensor([len(ids)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(nmine) for _ in range(world)]
    dist.all_gather(counts, nmine)
    rows = max(1, max(int(c.item()) for c in counts))
    tab = np.full((rows, 2), -1, dtype=np.int64)
    if ids:
        tab[:len(ids), 0] = ids
        tab[:len(ids), 1] = lens
    table = torch.from_numpy(tab).to(device)
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    out = []
    for r in range(world):
        t = tables[r].cpu().numpy()
        out.append(t[:int(counts[r].item())])
    return out


def gather_segments(dist, rank, world, my_segments, device="cpu", chunk=CHUNK):
    """my_segments: dict seg_id -> bytes.  Returns the ordered concatenation on rank 0 (a bytes-like
    memoryview over one buffer), None elsewhere."""
    ids = sorted(my_segments)
    lens = [len(my_segments[s]) for s in ids]
    tables = _exchange_tables(dist, world, ids, lens, device)

    if rank != 0:
        total = sum(lens)
        if total:
            flat = np.empty(total, dtype=np.uint8)
            off = 0
            for s, n in zip(ids, lens):
                flat[off:off + n] = np.frombuffer(my_segments[s], dtype=np.uint8)
                off += n
            for a in range(0, total, chunk):
                piece = torch.from_numpy(flat[a:min(total, a + chunk)]).to(device)
                dist.send(piece, dst=0)
        return None

    # rank 0: where every segment lands in the output
    all_rows = [(int(s), int(n), r) for r in range(world) for s, n in tables[r]]
    all_rows.sort()
    seen = set()
    for s, _, _ in all_rows:
        if s in seen:
            raise ValueError("segment %d was produced by more than one rank" % s)
        seen.add(s)
    offset, at = {}, 0
    for s, n, _ in all_rows:
        offset[s] = at
        at += n
    out = np.empty(at, dtype=np.uint8)
    for s, n in zip(ids, lens):
        out[offset[s]:offset[s] + n] = np.frombuffer(my_segments[s], dtype=np.uint8)
    totals = [int(tables[r][:, 1].sum()) if len(tables[r]) else 0 for r in range(world)]
    rbuf = torch.empty(max(1, min(chunk, max(totals[1:], default=0))), dtype=torch.uint8, device=device)
    for r in range(1, world):
        total = totals[r]
        if not total:
            continue
        # the rank's payload is its segments back to back: (start in that flat order, length, segment id); every received
        # piece is dropped straight into the output at the offsets of the segments it overlaps (no second host copy)
        spans, off = [], 0
        for s, n in tables[r]:
            spans.append((off, int(n), int(s)))
            off += int(n)
        k = 0
        for a in range(0, total, chunk):
            n = min(total, a + chunk) - a
            view = rbuf[:n]
            dist.recv(view, src=r)
            piece = view.cpu().numpy()
            while k < len(spans) and spans[k][0] + spans[k][1] <= a:
                k += 1
            j = k
            while j < len(spans) and spans[j][0] < a + n:
                s0, sn, sid = spans[j]
                lo, hi = max(s0, a), min(s0 + sn, a + n)
                if hi > lo:
                    out[offset[sid] + (lo - s0):offset[sid] + (hi - s0)] = piece[lo - a:hi - a]
                j += 1
    return memoryview(out)


# ---- streaming form: any job size in O(chunk) memory on every rank ------------------------------------------------------
def _have_xxhash():
    try:
        import xxhash  # noqa: F401
        return True
    except Exception:
        return False


def _new_digest(kind):
    """kind 1: xxh3-128 (10+ GB/s on one core), 0: md5 (0.7 GB/s; the standard library's, when some rank lacks xxhash).
    Both give 16 bytes; the check is between a producer and rank 0 of one job, not a published checksum."""
    if kind == 1:
        import xxhash
        return xxhash.xxh3_128()
    import hashlib
    return hashlib.md5()


def _digest_words(h):
    d = h.digest()
    return int.from_bytes(d[:8], "little", signed=True), int.from_bytes(d[8:], "little", signed=True)


def gather_segments_streaming(dist, rank, world, seg_ids, seg_bytes, device="cpu", chunk=CHUNK, sink=None):
    """The same ordered gather for jobs whose output does not fit rank 0's memory (eight ranks of the headline produce ~58 GB).

    seg_ids: this rank's segment ids; seg_bytes(sid) -> bytes-like, called ONCE per segment and pass, in ascending id order (so
    a producer can build its segments lazily and drop them behind the call).
    sink(sid, final_offset, offset_in_segment, piece): called on rank 0 for every received piece (and for rank 0's own segments),
    e.g. an os.pwrite into the output file at final_offset + offset_in_segment; None = verify and drop.  It runs on rank 0's
    checker thread, one call at a time, in arrival order.

    Every rank first publishes (id, length, digest) of its segments in one all_gather of an int64 table (digest: xxh3-128 when
    every rank has it, else md5).  The payloads then travel unpadded, point to point, in pieces of at most `chunk` bytes:
      * a producer ships when rank 0 says so (a one-word go-ahead), out of TWO staging tensors it refills in turn, each only when
        the send that last used it has completed -- it never holds more than 2 x chunk beyond its own segments, whatever the
        backend does with a send call (RCCL's returns at once: the round-5 form queued a fresh tensor per piece);
      * rank 0 receives into two buffers in turn and hands each piece to a checker thread that folds it into the running digest
        of the segment(s) it overlaps, compares at each segment's end, feeds the sink and drops it -- the next piece is on the
        wire while this one is hashed (the hashes release the interpreter lock).
    Returns on rank 0 {"segments", "segments_verified", "segments_over_the_wire", "bytes", "offsets": {sid: final offset},
    "digest", "peak_bytes_held": the most buffer bytes this rank held at one moment, measured}, None elsewhere.  Rank 0's own
    segments never leave the rank: for them "verified" means both passes of seg_bytes agreed."""
    import queue
    import threading
    ids = sorted(seg_ids)
    nmine = torch.tensor([len(ids), int(_have_xxhash())], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(nmine) for _ in range(world)]
    dist.all_gather(counts, nmine)
    counts = [c.cpu().numpy() for c in counts]
    kind = int(min(int(c[1]) for c in counts))
    # (length and digest need the bytes: taken in a first pass; the second pass ships them.  A producer that cannot afford to
    # build a segment twice keeps it: seg_bytes may simply return a stored object.)
    rows = []
    for s in ids:
        b = seg_bytes(s)
        h = _new_digest(kind)
        h.update(b)
        lo, hi = _digest_words(h)
        rows.append((s, len(b), lo, hi))
    nrows = max(1, max(int(c[0]) for c in counts))
    tab = np.full((nrows, 4), -1, dtype=np.int64)
    if rows:
        tab[:len(rows)] = np.array(rows, dtype=np.int64)
    table = torch.from_numpy(tab).to(device)
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    tables = [tables[r].cpu().numpy()[:int(counts[r][0])] for r in range(world)]
    go = torch.zeros(1, dtype=torch.int64, device=device)

    if rank != 0:
        total = sum(r[1] for r in rows)
        if not total:
            return None
        dist.recv(go, src=0)  # rank 0 is ready for THIS rank's bytes: nothing is queued at a peer that is not listening
        size = min(chunk, total)
        host = [np.empty(size, dtype=np.uint8) for _ in range(2)]
        stage = [torch.empty(size, dtype=torch.uint8, device=device) for _ in range(2)] if device != "cpu" else [torch.from_numpy(x) for x in host]
        work = [None, None]
        cur, fill = 0, 0

        def ship(n):
            nonlocal cur, fill
            if device != "cpu":
                stage[cur][:n].copy_(torch.from_numpy(host[cur][:n]))
            work[cur] = dist.isend(stage[cur][:n], dst=0)
            cur ^= 1
            if work[cur] is not None:  # the buffer about to be refilled: its send has to be over
                work[cur].wait()
                if device != "cpu":
                    torch.cuda.current_stream().synchronize()
                work[cur] = None
            fill = 0

        for s in ids:
            b = np.frombuffer(seg_bytes(s), dtype=np.uint8)
            at = 0
            while at < len(b):
                n = min(len(b) - at, size - fill)
                host[cur][fill:fill + n] = b[at:at + n]
                fill += n
                at += n
                if fill == size:
                    ship(size)
        if fill:
            ship(fill)
        for w in work:
            if w is not None:
                w.wait()
        if device != "cpu":
            torch.cuda.current_stream().synchronize()
        return None

    all_rows = sorted((int(t[0]), int(t[1]), r, k) for r in range(world) for k, t in enumerate(tables[r]))
    seen = set()
    for s, _, _, _ in all_rows:
        if s in seen:
            raise ValueError("segment %d was produced by more than one rank" % s)
        seen.add(s)
    offsets, at = {}, 0
    for s, n, _, _ in all_rows:
        offsets[s] = at
        at += n
    totals = [int(tables[r][:, 1].sum()) if len(tables[r]) else 0 for r in range(world)]
    size = max(1, min(chunk, max(totals[1:], default=0)))
    rbuf = [torch.empty(size, dtype=torch.uint8, device=device) for _ in range(2)]
    free = [threading.Semaphore(1), threading.Semaphore(1)]
    jobs = queue.Queue()
    state = {"good": 0, "nseg": 0, "wire": 0, "held": 0, "peak": 2 * size, "err": None}

    def checker():
        cur_r, t, k, off_in, h = -1, None, 0, 0, None
        try:
            while True:
                job = jobs.get()
                if job is None:
                    return
                if job[0] == "own":  # one of rank 0's own segments, second pass
                    _, sid, b = job
                    row = next(x for x in tables[0] if int(x[0]) == sid)
                    hh = _new_digest(kind)
                    hh.update(b)
                    state["good"] += int(_digest_words(hh) == (int(row[2]), int(row[3])) and len(b) == int(row[1]))
                    state["nseg"] += 1
                    if sink is not None:
                        sink(sid, offsets[sid], 0, memoryview(b))
                    continue
                _, r, slot, n = job
                if r != cur_r:
                    cur_r, t, k, off_in, h = r, tables[r], 0, 0, _new_digest(kind)
                piece = rbuf[slot][:n].cpu().numpy() if device != "cpu" else rbuf[slot][:n].numpy()
                state["peak"] = max(state["peak"], 2 * size + (n if device != "cpu" else 0))
                p = 0
                while True:
                    while k < len(t) and off_in == int(t[k][1]):  # segments that are complete (empty ones included)
                        state["good"] += int(_digest_words(h) == (int(t[k][2]), int(t[k][3])))
                        state["nseg"] += 1
                        state["wire"] += 1
                        k, off_in, h = k + 1, 0, _new_digest(kind)
                    if p >= n:
                        break
                    m = min(n - p, int(t[k][1]) - off_in)
                    h.update(piece[p:p + m])
                    if sink is not None:
                        sink(int(t[k][0]), offsets[int(t[k][0])], off_in, memoryview(piece[p:p + m]))
                    off_in += m
                    p += m
                del piece
                free[slot].release()
        except BaseException as e:  # (handed to the caller: a checker that dies must not leave the receiver waiting for a buffer)
            state["err"] = e
            for f in free:
                f.release()

    th = threading.Thread(target=checker, name="dsv2-gather-check", daemon=True)
    th.start()
    for s in ids:
        jobs.put(("own", s, seg_bytes(s)))
    npiece = 0
    for r in range(1, world):
        total = totals[r]
        if not total:
            # (a rank whose segments are all empty ships nothing; its rows are closed here)
            for x in tables[r]:
                h0 = _new_digest(kind)
                state["good"] += int(_digest_words(h0) == (int(x[2]), int(x[3])))
                state["nseg"] += 1
                state["wire"] += 1
            continue
        dist.send(go, dst=r)
        for a in range(0, total, size):
            n = min(total, a + size) - a
            slot = npiece & 1
            npiece += 1
            free[slot].acquire()
            if state["err"] is not None:
                raise state["err"]
            dist.recv(rbuf[slot][:n], src=r)
            jobs.put(("piece", r, slot, n))
    jobs.put(None)
    th.join()
    if state["err"] is not None:
        raise state["err"]
    return {"segments": state["nseg"], "segments_verified": state["good"], "segments_over_the_wire": state["wire"], "bytes": at, "offsets": offsets,
            "digest": "xxh3_128" if kind == 1 else "md5", "peak_bytes_held": state["peak"]}
