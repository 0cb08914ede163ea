"""Row N4 (altitude optimiser, csp_alt_*): host-memory calls, batch of problems and one long problem.
    python tools/alt_bench.py        (GPU box; wall time includes staging)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import importlib
csp = importlib.import_module("cs-pathplan_amd")

rng = np.random.default_rng(3)


def problem(n):
    xy = np.cumsum(rng.uniform(20, 60, size=(n, 2)), axis=0)
    z = 100 + np.cumsum(rng.normal(0, 8, n))
    elev = 80 + 10 * np.sin(np.arange(n) / 5.0) + rng.normal(0, 2, n)
    return np.column_stack([xy, z]), elev


def timed(fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


for B, n in ((8192, 256), (1, 2000), (1, 20000)):
    ps = [problem(n) for _ in range(min(B, 64))]
    xyz = np.concatenate([ps[i % len(ps)][0] for i in range(B)])
    elev = np.concatenate([ps[i % len(ps)][1] for i in range(B)])
    off = np.arange(B + 1, dtype=np.int64) * n
    us1 = timed(lambda: csp.alt_optimize_heights_batch(xyz, elev, off, 1.0, 0.5, 50.0, 2.0))
    us2 = timed(lambda: csp.alt_global_smooth_batch(xyz[:, 2].copy(), xyz, off, 1.0, 2.0))
    print(json.dumps({"problems": B, "samples_each": n, "optimize_heights_us": round(us1, 1), "global_smooth_us": round(us2, 1),
                      "optimize_rows_per_s": round(B * n / us1 * 1e6)}), flush=True)
