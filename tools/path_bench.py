"""Times the path-penalty solve: register-resident kernel vs generic kernel.
    python tools/path_bench.py [B]            (GPU box; prints one JSON line per configuration)"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import importlib
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for order, S in ((4, 16), (4, 8), (3, 16), (2, 16)):
    wp, tm = synth.make_batch(B, S, config_id=7)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    row = {"order": order, "S": S, "B": B, "bytes_per_solve": synth.algorithmic_bytes(S, order)}
    outs = {}
    for name, force in (("fixedpath", False), ("generic", True)):
        ps = csp.PreparedSolve(d_wp, d_tm, order=order, path_weight=0.1, vel_zero_weight=0.01, force_generic=force)
        for _ in range(3):
            ps.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ps.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        outs[name] = ps.out.cpu().numpy().copy()
        row[name] = {"kernel": ps.kernel, "us": round(us, 1), "solves_per_s": round(B / us * 1e6, 0),
                     "algorithmic_GBps": round(B * row["bytes_per_solve"] / us * 1e-3, 1)}
    row["rel_err_fixed_vs_generic"] = float(synth.rel_err(outs["fixedpath"], outs["generic"]))
    print(json.dumps(row), flush=True)
