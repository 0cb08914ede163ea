"""Small-batch launches of the register-resident kernel: us per launch for several batch sizes, for each
trajectories-per-workgroup setting (CSP_SLICE_W, read once per process -> one subprocess each).
    python tools/c2_bench.py            (on the GPU box)"""
import json
import os
import subprocess
import sys

CHILD = r'''
import importlib, json, sys, torch
sys.path.insert(0, ".")
from tests import synth
import bench
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
out = {}
for (B, S, o) in ((4096, 8, 4), (1024, 8, 4), (16384, 8, 4), (4096, 16, 4), (30000, 16, 4), (4096, 8, 2)):
    rec, prep, wp, tm = bench.bench_uniform(csp, dev, B, S, o, 300, 30, 2)
    out["B%d_S%d_o%d" % (B, S, o)] = round(rec["kernel_ms"] * 1e3, 2)
print(json.dumps(out))
'''
for w in ("auto", "64", "32", "16", "8"):
    env = dict(os.environ)
    if w != "auto":
        env["CSP_SLICE_W"] = w
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    print("slice_w=%-5s" % w, line[-1] if line else r.stderr[-400:], flush=True)
