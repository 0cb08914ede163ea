"""Path-penalty kernel (row A7 / the shipped yaml's configuration) at B = 65536: us per launch and fraction of the HBM
roofline by algorithmic bytes, per order and segment count; `--plan` adds the whole re-solve loop (csp_minsnap_plan_batch).
    python tools/path_bench.py [--plan]"""
import importlib
import json
import sys

import torch

sys.path.insert(0, ".")
import bench
from tests import synth

csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
for (o, S, pw, vw) in ((2, 16, 1e-7, 0.01), (3, 16, 1e-7, 0.01), (4, 16, 1e-7, 0.01), (4, 8, 0.1, 0.01), (2, 6, 1e-7, 0.01)):
    rec, prep, wp, tm = bench.bench_uniform(csp, dev, 65536, S, o, 20, 3, 3, pw=pw, vw=vw)
    print(json.dumps({"order": o, "S": S, "kernel": rec["kernel"], "us": round(rec["kernel_ms"] * 1e3, 1),
                      "solves_per_s": "%.3g" % rec["solves_per_s"], "frac_hbm": round(rec["frac_of_hbm_peak"], 3)}), flush=True)
    if "--plan" in sys.argv:
        d_wp = torch.from_numpy(wp * 4.0).to(dev)
        ms = bench.timed(lambda: csp.plan_batch(d_wp, 5.0, 0.1, order=o, path_weight=0.3), 5, 2, dev)
        print(json.dumps({"order": o, "S": S, "plan_batch_ms_all_passes": round(ms, 3)}), flush=True)
