#!/usr/bin/env python3
"""host-to-device rate of pinned memory on this box (what bounds the upload of 2.4 GB of pictures per step)"""
import time
import torch
n = 600 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for rep in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("H2D %d MB pinned: %.1f GB/s" % (n >> 20, n / dt / 1e9))
hs = [torch.empty(3110400, dtype=torch.uint8).pin_memory() for _ in range(192)]
ds = [torch.empty(3110400, dtype=torch.uint8, device="cuda") for _ in range(192)]
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for a, b in zip(ds, hs):
        a.copy_(b, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("H2D 192 x 3.1 MB pinned copies: %.1f GB/s" % (192 * 3110400 / dt / 1e9))
