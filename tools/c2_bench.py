"""Small-batch launches of the register-resident kernel: us per launch for several batch sizes; A/B of the
three-lanes-per-trajectory mapping (CSP_AXIS_LANES) and of the slice width (CSP_SLICE_W) -- both are read once per
process, hence one subprocess per setting.     python tools/c2_bench.py            (on the GPU box)"""
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
for (B, S, o) in ((4096, 8, 4), (1024, 8, 4), (64, 8, 4), (8192, 8, 4), (4096, 16, 4), (4096, 8, 3)):
    rec, prep, wp, tm = bench.bench_uniform(csp, dev, B, S, o, 300, 30, 2)
    out["B%d_S%d_o%d" % (B, S, o)] = round(rec["kernel_ms"] * 1e3, 2)
print(json.dumps(out))
'''
for tag, env_add in (("axis lanes (default)", {}), ("axis lanes, slice 8", {"CSP_SLICE_W": "8"}), ("axis lanes, slice 16", {"CSP_SLICE_W": "16"}),
                     ("one lane per trajectory", {"CSP_AXIS_LANES": "0"}), ("one lane, slice 64", {"CSP_AXIS_LANES": "0", "CSP_SLICE_W": "64"})):
    env = dict(os.environ)
    env.update(env_add)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    print("%-26s" % tag, line[-1] if line else r.stderr[-400:], flush=True)
