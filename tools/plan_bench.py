"""Times the GenerateTrajectoryMatrix pieces on the device: plan (time
allocation + re-solve loop) and sampling.   python tools/plan_bench.py [B]   (GPU box)"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import importlib
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = 16


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, out


wp, _ = synth.make_batch(B, S, config_id=21)
d_wp = torch.from_numpy(wp * 4.0).cuda()
for order, pw, vw in ((3, 0.0, 0.0), (4, 0.0, 0.02), (3, 0.5, 0.0), (4, 0.3, 0.0), (4, 1e-3, 0.01)):
    us_plan, plan = timed(lambda: csp.plan_batch(d_wp, 5.0, 0.1, order=order, path_weight=pw, vel_zero_weight=vw))
    cap = 256
    bufs = csp.sample_batch(plan.times, plan.coeffs, 0.7, cap)
    us_samp, (samples, counts, stats) = timed(lambda: csp.sample_batch(plan.times, plan.coeffs, 0.7, cap, out=bufs))
    bufs1 = csp.sample_batch(plan.times, plan.coeffs, 0.7, cap, one_lane=True)
    us_one, _ = timed(lambda: csp.sample_batch(plan.times, plan.coeffs, 0.7, cap, out=bufs1, one_lane=True))
    assert all(torch.equal(x, y) for x, y in zip(bufs, bufs1))
    n_s = counts.double().mean().item()
    print(json.dumps({"order": order, "path_weight": pw, "B": B, "S": S,
                      "plan_us": round(us_plan, 1), "plans_per_s": round(B / us_plan * 1e6),
                      "mean_resolve_iterations": round(plan.iterations.double().mean().item(), 3),
                      "sample_us": round(us_samp, 1), "trajectories_sampled_per_s": round(B / us_samp * 1e6),
                      "sample_us_one_lane_kernel": round(us_one, 1),
                      "mean_samples_kept": round(n_s, 1), "max_samples_kept": int(counts.max().item())}), flush=True)
