"""A/B of non-temporal coefficient stores in the unpenalised register-resident kernels, per order, beyond the Infinity Cache:
    CSP_NT_STORES=0 python tools/fixed_nt_ab.py ; CSP_NT_STORES=1 python tools/fixed_nt_ab.py"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import bench

csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
for o in (2, 3, 4, 5):
    for B, S in ((524288, 16), (262144, 16), (524288, 8)):
        rec, prep, wp, tm = bench.bench_uniform(csp, dev, B, S, o, 20, 3, 3)
        print(json.dumps({"nt_env": os.environ.get("CSP_NT_STORES"), "order": o, "B": B, "S": S, "kernel": rec["kernel"],
                          "us": round(rec["kernel_ms"] * 1e3, 1), "frac_hbm": round(rec["frac_of_hbm_peak"], 3)}), flush=True)
        del prep
        torch.cuda.empty_cache()
