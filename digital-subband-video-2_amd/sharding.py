"""Closed-GOP segment sharding across ranks and the final ordered gather.

The reference's only parallelism is process-level: `parallel_encode_yuv.sh:31-52` forks one encoder
per closed-GOP chunk (`-sfr=N -nfr=chunk -noeos=1`) and `cat`s the outputs in order.  Here each rank
(one process per GPU) encodes the segments assigned to it with fresh encoder state and the packet
bytes are gathered to rank 0 in segment order.  There is no collective on the data path; the
gather is the only exchange: lengths by all_gather, payloads by gather of padded byte tensors
(RCCL on GPUs, gloo in the CPU tests).
"""
import torch


def assign_segments(nseg, world):
    """Round-robin: segment s belongs to rank s % world.  Returns a list (per rank) of segment ids."""
    return [[s for s in range(nseg) if s % world == r] for r in range(world)]


def frame_range(seg, gop, nframes):
    """Frames [start, end) of segment `seg` (the reference's -sfr / -nfr arguments)."""
    start = seg * gop
    return start, min(nframes, start + gop)


def gather_segments(dist, rank, world, my_segments, device="cpu"):
    """my_segments: dict seg_id -> bytes.  Returns the ordered concatenation on rank 0, None elsewhere."""
    ids = sorted(my_segments)
    blob = b"".join(my_segments[s] for s in ids)
    # per-rank table: (segment id, length) pairs, padded to a common row count
    nmax = torch.tensor([len(ids)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(nmax) for _ in range(world)]
    dist.all_gather(counts, nmax)
    rows = int(max(int(c.item()) for c in counts))
    table = torch.full((max(rows, 1), 2), -1, dtype=torch.int64, device=device)
    for k, s in enumerate(ids):
        table[k, 0], table[k, 1] = s, len(my_segments[s])
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    totals = [int(t[:, 1].clamp(min=0).sum().item()) for t in tables]
    cap = max(1, max(totals))
    payload = torch.zeros(cap, dtype=torch.uint8, device=device)
    if blob:
        payload[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    gathered = [torch.zeros_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, gathered, dst=0)
    if rank != 0:
        return None
    pieces = {}
    for r in range(world):
        data = gathered[r].cpu().numpy().tobytes()
        off = 0
        for k in range(tables[r].shape[0]):
            s, n = int(tables[r][k, 0].item()), int(tables[r][k, 1].item())
            if s >= 0:
                pieces[s] = data[off:off + n]
                off += n
    return b"".join(pieces[s] for s in sorted(pieces))
