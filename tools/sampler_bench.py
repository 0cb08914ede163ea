"""Sampling (row N1) of long legs: the three samplers on a batch of flights with kilometre legs (hundreds of
candidates per segment), microseconds per call.    python tools/sampler_bench.py        (GPU box)"""
import importlib
import json
import sys

import torch

sys.path.insert(0, ".")
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth

for B, S, scale in ((1, 6, 4000.0), (64, 6, 4000.0), (4096, 6, 1500.0)):
    wp, _ = synth.make_batch(B, S, config_id=26)
    plan = csp.plan_batch(torch.from_numpy(wp * scale).cuda(), 200.0, 1.0, order=2)
    cap = 1 << 13
    row = {"B": B, "S": S, "candidates_per_flight": int(float(plan.times[0].sum()) / 0.1)}
    for name, kw in (("wave_per_trajectory", dict(long_segments=True)), ("lane_per_segment", dict()), ("one_lane", dict(one_lane=True))):
        if name == "one_lane" and B > 64:
            continue
        for _ in range(3):
            r = csp.sample_batch(plan.times, plan.coeffs, 300.0, cap, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            r = csp.sample_batch(plan.times, plan.coeffs, 300.0, cap, **kw)
        e1.record()
        torch.cuda.synchronize()
        row[name + "_us"] = round(e0.elapsed_time(e1) / n * 1e3, 1)
    print(json.dumps(row), flush=True)
