"""Randomised differential check on the GPU box: random (order, S, B, dtype, weights, boundary conditions,
ragged or uniform) configurations, every eligible kernel family against the generic kernel.
    python tools/fuzz_kernels.py [iterations] [seed]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import importlib
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
worst = {}
for it in range(iters):
    order = int(rng.integers(1, 6))
    ragged = bool(rng.integers(0, 2))
    f32 = bool(rng.integers(0, 3) == 0)
    B = int(rng.choice([1, 2, 63, 64, 65, 200, 1000, 4097]))
    smax = int(rng.choice([1, 2, 3, 8, 16, 17, 40, 64, 130, 300]))
    pw = float(rng.choice([0.0, 0.0, 0.3])) if not f32 else 0.0
    vw = float(rng.choice([0.0, 0.05]))
    per_bc = bool(rng.integers(0, 2))
    dt = np.float32 if f32 else np.float64
    if ragged:
        S_b = rng.integers(1, smax + 1, size=B)
        S_b[-1] = smax
    else:
        S_b = np.full(B, smax)
    wps, tms = [], []
    for S in S_b:
        p0 = rng.uniform(-10, 10, size=(1, 3))
        wps.append(np.concatenate([p0, p0 + np.cumsum(rng.normal(size=(int(S), 3)), axis=0)]))
        tms.append(rng.uniform(0.5, 2.0, size=int(S)))
    bc = rng.normal(size=(B if per_bc else 1, 4, 3))
    off = np.concatenate([[0], np.cumsum(S_b)]).astype(np.int64)
    if ragged:
        wp, tm = np.concatenate(wps).astype(dt), np.concatenate(tms).astype(dt)
        kw = dict(seg_offsets=torch.from_numpy(off).cuda(), max_segments=int(smax))
    else:
        wp, tm = np.stack(wps).astype(dt), np.stack(tms).astype(dt)
        kw = {}
    args = (torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda(), torch.from_numpy(bc.astype(dt)).cuda())
    common = dict(order=order, path_weight=pw, vel_zero_weight=vw, want_status=True, want_max_dev=True, **kw)
    ref = csp.solve_batch(*args, force_generic=True, **common)
    for extra in ({}, {"span": True}, {"no_persistent": True}):
        r = csp.solve_batch(*args, **extra, **common)
        if r.kernel == ref.kernel:
            continue
        torch.cuda.synchronize()
        a = r.coeffs.cpu().numpy().astype(np.float64).reshape(-1, 3 * 2 * order)
        g = ref.coeffs.cpu().numpy().astype(np.float64).reshape(-1, 3 * 2 * order)
        errs = []
        for b in range(B):
            lo, hi = (off[b], off[b + 1]) if ragged else (b * smax, (b + 1) * smax)
            den = max(np.max(np.abs(g[lo:hi])), 1e-300)
            errs.append(np.max(np.abs(a[lo:hi] - g[lo:hi])) / den)
        e = float(max(errs))
        fam = r.kernel.split("_")[0] + ("_f32" if f32 else "") + ("_o5" if order == 5 else "")
        worst[fam] = max(worst.get(fam, 0.0), e)
        tol = 1e-5 if f32 else (1e-5 if order == 5 else 1e-7)
        ok = e < tol and bool((r.status == ref.status).all()) and float((r.max_dev - ref.max_dev).abs().max()) < 1e-7 * max(1.0, float(ref.max_dev.abs().max()))
        if not ok:
            print("MISMATCH", json.dumps({"it": it, "kernel": r.kernel, "order": order, "B": B, "smax": smax, "ragged": ragged,
                                          "f32": f32, "pw": pw, "vw": vw, "per_bc": per_bc, "err": e}), flush=True)
            sys.exit(1)
print("ok", iters, "configurations; worst relative difference to the generic kernel per family:", json.dumps(worst))
