#!/usr/bin/env python3
"""Diagnostic: where does a wave of the order-2 path-penalty kernel (the shipped yaml) spend its cycles?  Uses the stamps
build (python cs-pathplan_amd/build.py --stamps).  Shares only -- never quote this build's run time."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import synth
lib = ctypes.CDLL(os.path.join(ROOT, "cs-pathplan_amd", "libcsp_minsnap_stamps.so"))
csp = importlib.import_module("cs-pathplan_amd")
csp._lib = lib
lib.csp_minsnap_solve_batch.restype = ctypes.c_int
lib.csp_minsnap_solve_batch.argtypes = [ctypes.POINTER(csp.Desc)] + [ctypes.c_void_p] * 7 + [ctypes.c_size_t, ctypes.c_void_p]
lib.csp_minsnap_workspace_bytes.restype = ctypes.c_size_t
lib.csp_minsnap_workspace_bytes.argtypes = [ctypes.POINTER(csp.Desc)]
lib.csp_minsnap_kernel_name.restype = ctypes.c_char_p
lib.csp_minsnap_kernel_name.argtypes = [ctypes.POINTER(csp.Desc)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = 16
wp, tm = synth.make_batch(B, S)
d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
for _ in range(5):
    csp.solve_batch(d_wp, d_tm, order=2, path_weight=1e-7, vel_zero_weight=0.01)
torch.cuda.synchronize()
n = 4096 * 8
buf = (ctypes.c_ulonglong * n)()
lib.csp_debug_read_stamps_path_o2(buf, n)
raw = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)[:4096].astype(np.int64)
raw = raw[(raw[:, 0] > 0) & (raw[:, 4] > 0)]
st = raw[:, :5]
rt0, rt1 = raw[:, 5], raw[:, 6]
d = np.diff(st, axis=1)
names = ["start sleep + copy-in + barrier", "pass A (pre-solve + 17-sample search)", "pass B forward + exchange + middle solve", "pass B backward + stores"]
tot = st[:, 4] - st[:, 0]
print("waves sampled:", len(st), " median wave lifetime (shader cycles):", int(np.median(tot)))
for i, nme in enumerate(names):
    print("  %-46s median %8d  share %.1f%%" % (nme, np.median(d[:, i]), 100 * d[:, i].sum() / tot.sum()))
span = rt1.max() - rt0.min()
print("kernel span: %.1f us (s_memrealtime, 100 MHz)" % (span / 100.0))
life = (rt1 - rt0) / 100.0
print("wave lifetime us: median %.2f p10 %.2f p90 %.2f" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
start = (rt0 - rt0.min()) / 100.0
end = (rt1 - rt0.min()) / 100.0
print("wave start histogram (us):", [(round(float(e), 1), int(h)) for e, h in zip(*np.histogram(start, bins=10)[::-1])])
print("wave end histogram (us):", [(round(float(e), 1), int(h)) for e, h in zip(*np.histogram(end, bins=10)[::-1])])
