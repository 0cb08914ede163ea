#!/usr/bin/env python3
"""Known-good HBM ceilings on the box (torch fill = pure write stream, copy = read+write stream),
at the two footprints bench.py uses (236 MB ~ Infinity-Cache resident, 1.9 GB ~ HBM streaming)."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (201, 1610, 6000):
    n = mb * 1000 * 1000 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty(n, dtype=torch.float64, device="cuda")
    tf = t(lambda: a.fill_(1.5)); tc = t(lambda: b.copy_(a))
    print("%5d MB  fill (write only) %.2f TB/s   copy (read+write) %.2f TB/s of moved bytes" % (mb, n * 8 / tf / 1e12, 2 * n * 8 / tc / 1e12))
    del a, b
