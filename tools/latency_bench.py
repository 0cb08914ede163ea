"""Single-trajectory latency through the host-memory C-ABI (what one UavPathPlanner::getPlan call sees):
README uav31_0 waypoints, plan (time allocation + re-solve loop) and sampling.   python tools/latency_bench.py"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import importlib
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth

wp = np.asarray(synth.README_UAV31_ENU, dtype=np.float64)[None]
for order, pw, vw, v in ((3, 0.0, 0.0, 30.0), (4, 0.0, 0.0, 30.0), (2, 1e-7, 0.01, 200.0), (4, 0.3, 0.0, 30.0)):
    def once():
        plan = csp.plan_batch(wp, v, 1.0, order=order, path_weight=pw, vel_zero_weight=vw)
        return plan, csp.sample_batch(plan.times, plan.coeffs, 30.0, 4096)
    for _ in range(3):
        once()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        plan, (samples, counts, stats) = once()
    us = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        csp.solve_batch(wp, plan.times, order=order, path_weight=pw, vel_zero_weight=vw)
    us_solve = (time.perf_counter() - t0) / n * 1e6
    print(json.dumps({"order": order, "path_weight": pw, "vel_zero_weight": vw, "segments": wp.shape[1] - 1,
                      "plan_plus_sample_us": round(us, 1), "solve_only_us": round(us_solve, 1),
                      "resolve_iterations": int(plan.iterations[0]), "samples": int(counts[0])}), flush=True)
