"""Times uniform long trajectories and fp32 storage: span / chunked (multi-lane, workspace-free) vs generic kernel.
    python tools/long_bench.py          (GPU box; one JSON line per configuration)"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import importlib
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth

for order, S, B, dt in ((4, 64, 65536, np.float64), (4, 32, 65536, np.float64), (4, 64, 65536, np.float32),
                        (3, 64, 65536, np.float32), (5, 64, 65536, np.float32), (4, 16, 65536, np.float32), (4, 256, 8192, np.float64),
                        (4, 1024, 2048, np.float64)):
    wp, tm = synth.make_batch(B, S, config_id=7)
    d_wp, d_tm = torch.from_numpy(wp.astype(dt)).cuda(), torch.from_numpy(tm.astype(dt)).cuda()
    width = 4 if dt == np.float32 else 8
    row = {"order": order, "S": S, "B": B, "dtype": np.dtype(dt).name, "bytes_per_solve": synth.algorithmic_bytes(S, order, width)}
    outs = {}
    for name, force, span in (("span", False, True), ("chunked", False, False), ("generic", True, False)):
        ps = csp.PreparedSolve(d_wp, d_tm, order=order, force_generic=force, span=span)
        if not ps.kernel.startswith(name):
            continue
        for _ in range(3):
            ps.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ps.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        outs[name] = ps.out.cpu().numpy().astype(np.float64)
        row[name] = {"kernel": ps.kernel, "us": round(us, 1), "solves_per_s": round(B / us * 1e6, 0),
                     "algorithmic_GBps": round(B * row["bytes_per_solve"] / us * 1e-3, 1)}
    row["rel_err_vs_generic"] = float(max(synth.rel_err(outs[k], outs["generic"]) for k in outs if k != "generic"))
    print(json.dumps(row), flush=True)
