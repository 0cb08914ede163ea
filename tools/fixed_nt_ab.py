"""A/B of non-temporal coefficient stores in the unpenalised register-resident kernels, per order:
    CSP_NT_STORES=0 python tools/fixed_nt_ab.py [orders] [B:S,B:S,...] ; CSP_NT_STORES=1 python tools/fixed_nt_ab.py ...
(no CSP_NT_STORES: the launcher's own rule).  Default orders 2,3,4,5 at B = 524288 / 262144 x S = 16 and B = 524288 x S = 8."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import bench

csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
orders = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 3, 4, 5]
sizes = [tuple(int(v) for v in x.split(":")) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [(524288, 16), (262144, 16), (524288, 8)]
for o in orders:
    for B, S in sizes:
        if o == 5 and S > 8:
            continue
        rec, prep, wp, tm = bench.bench_uniform(csp, dev, B, S, o, 20, 3, 3)
        print(json.dumps({"nt_env": os.environ.get("CSP_NT_STORES"), "order": o, "B": B, "S": S, "kernel": rec["kernel"],
                          "us": round(rec["kernel_ms"] * 1e3, 1), "frac_hbm": round(rec["frac_of_hbm_peak"], 3)}), flush=True)
        del prep
        torch.cuda.empty_cache()
