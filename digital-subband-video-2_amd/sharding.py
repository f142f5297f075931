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
def _md5_words(b):
    import hashlib
    d = hashlib.md5(b).digest()
    return int.from_bytes(d[:8], "little", signed=True), int.from_bytes(d[8:], "little", signed=True)


def gather_segments_streaming(dist, rank, world, seg_ids, seg_bytes, device="cpu", chunk=CHUNK, sink=None):
    """The same ordered gather for jobs whose output does not fit rank 0's memory (eight ranks of the headline produce ~58 GB).

    seg_ids: this rank's segment ids; seg_bytes(sid) -> bytes-like, called ONCE per segment, in ascending id order, right before
    the segment is shipped (so a producer can build its segments lazily and drop them behind the call).
    sink(sid, final_offset, offset_in_segment, piece): called on rank 0 for every received piece (and for rank 0's own segments),
    e.g. an os.pwrite into the output file at final_offset + offset_in_segment; None = verify and drop.

    Every rank first publishes (id, length, md5) of its segments in one all_gather of an int64 table; the payloads then travel
    unpadded in pieces of at most `chunk` bytes, and rank 0 folds every piece into the running md5 of the segment(s) it
    overlaps and compares at each segment's end: no rank ever holds more than `chunk` bytes beyond its own segments.
    Returns on rank 0 {"segments", "segments_verified", "bytes", "offsets": {sid: final offset}}, None elsewhere."""
    import hashlib
    ids = sorted(seg_ids)
    # (length and digest need the bytes: taken in a first pass; the second pass ships them.  A producer that cannot afford to
    # build a segment twice keeps it: seg_bytes may simply return a stored object.)
    rows = []
    for s in ids:
        b = seg_bytes(s)
        lo, hi = _md5_words(b)
        rows.append((s, len(b), lo, hi))
    nmine = torch.tensor([len(ids)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(nmine) for _ in range(world)]
    dist.all_gather(counts, nmine)
    nrows = max(1, max(int(c.item()) for c in counts))
    tab = np.full((nrows, 4), -1, dtype=np.int64)
    if rows:
        tab[:len(rows)] = np.array(rows, dtype=np.int64)
    table = torch.from_numpy(tab).to(device)
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    tables = [tables[r].cpu().numpy()[:int(counts[r].item())] for r in range(world)]

    if rank != 0:
        buf = np.empty(chunk, dtype=np.uint8)
        fill = 0
        for s in ids:
            b = np.frombuffer(seg_bytes(s), dtype=np.uint8)
            at = 0
            while at < len(b):
                n = min(len(b) - at, chunk - fill)
                buf[fill:fill + n] = b[at:at + n]
                fill += n
                at += n
                if fill == chunk:
                    dist.send(torch.from_numpy(buf).to(device), dst=0)
                    fill = 0
        if fill:
            dist.send(torch.from_numpy(buf[:fill].copy()).to(device), dst=0)
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
    good, nseg = 0, 0
    # rank 0's own segments
    for s in ids:
        b = seg_bytes(s)
        row = next(t for t in tables[0] if int(t[0]) == s)
        good += int(_md5_words(b) == (int(row[2]), int(row[3])) and len(b) == int(row[1]))
        nseg += 1
        if sink is not None:
            sink(s, offsets[s], 0, memoryview(b))
    totals = [int(tables[r][:, 1].sum()) if len(tables[r]) else 0 for r in range(world)]
    rbuf = torch.empty(max(1, min(chunk, max(totals[1:], default=0))), dtype=torch.uint8, device=device)
    for r in range(1, world):
        total = totals[r]
        t = tables[r]
        k, off_in = 0, 0  # segment being received and how much of it has arrived
        h = hashlib.md5()

        def close_finished():
            nonlocal k, off_in, h, good, nseg
            while k < len(t) and off_in == int(t[k][1]):
                d = h.digest()
                good += int((int.from_bytes(d[:8], "little", signed=True), int.from_bytes(d[8:], "little", signed=True)) == (int(t[k][2]), int(t[k][3])))
                nseg += 1
                k += 1
                off_in = 0
                h = hashlib.md5()

        close_finished()  # (leading empty segments)
        for a in range(0, total, chunk):
            n = min(total, a + chunk) - a
            view = rbuf[:n]
            dist.recv(view, src=r)
            piece = view.cpu().numpy()
            p = 0
            while p < n:
                m = min(n - p, int(t[k][1]) - off_in)
                h.update(piece[p:p + m])
                if sink is not None:
                    sink(int(t[k][0]), offsets[int(t[k][0])], off_in, memoryview(piece[p:p + m]))
                off_in += m
                p += m
                close_finished()
    return {"segments": nseg, "segments_verified": good, "bytes": at, "offsets": offsets}
