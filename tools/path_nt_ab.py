"""A/B of non-temporal coefficient stores in the path-penalty kernel for batches beyond the Infinity Cache:
    CSP_NT_STORES=0 python tools/path_nt_ab.py ; CSP_NT_STORES=1 python tools/path_nt_ab.py ; python tools/path_nt_ab.py
(the environment variable is read once per process; unset = the launcher's own choice)."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import bench

csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
for o in (2, 3, 4):
    for B in (65536, 131072, 262144, 524288):
        rec, prep, wp, tm = bench.bench_uniform(csp, dev, B, 16, o, 20, 3, 3, pw=1e-7, vw=0.01)
        print(json.dumps({"nt_env": os.environ.get("CSP_NT_STORES"), "order": o, "B": B, "kernel": rec["kernel"],
                          "us": round(rec["kernel_ms"] * 1e3, 1), "frac_hbm": round(rec["frac_of_hbm_peak"], 3)}), flush=True)
        del prep
        torch.cuda.empty_cache()
