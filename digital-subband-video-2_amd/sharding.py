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
