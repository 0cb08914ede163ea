"""PCIe-inclusive rate of the host-memory solve (C3 shape): pageable caller memory (staged through the arena's pinned
halves) and page-locked caller memory (direct DMA), solves/s.    python tools/host_path_bench.py   (GPU box)"""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth

B, S = 65536, 16
wp, tm = synth.make_batch(B, S, config_id=3)
out = np.empty((B, S, 3, 8))
out.fill(0.0)          # touch the pages once: first-touch faults are not the transfer
pwp, ptm = torch.from_numpy(wp).pin_memory(), torch.from_numpy(tm).pin_memory()
pout = torch.empty((B, S, 3, 8), dtype=torch.float64).pin_memory()
row = {"B": B, "S": S, "bytes_per_solve": 3608}
for name, args in (("pageable", (wp, tm, out)), ("pinned", (pwp.numpy(), ptm.numpy(), pout.numpy()))):
    csp.solve_batch(args[0], args[1], order=4, out=args[2])
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        csp.solve_batch(args[0], args[1], order=4, out=args[2])
    dt = (time.perf_counter() - t0) / n
    row[name + "_ms"] = round(dt * 1e3, 2)
    row[name + "_solves_per_s"] = "%.3g" % (B / dt)
    row[name + "_GBps"] = round(B * 3608 / dt / 1e9, 1)
print(json.dumps(row))
