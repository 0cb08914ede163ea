"""Diagnostic (not collected by pytest; run on a GPU box as `python tests/probe_f32_storage.py`): error of fp32 STORAGE
(CSP_DTYPE_F32, fp64 arithmetic) and of fp64 against the 80-bit oracle on a ragged mix, per order.  Lives under tests/
because it uses the oracle as the checker."""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csp = importlib.import_module("cs-pathplan_amd")
import oracle
from tests import synth
trajs = synth.make_ragged(400)
for o in (3, 4, 5):
    sel = [t for t in trajs if t[0] == o]
    sel.sort(key=lambda t: len(t[2]))
    wp = np.concatenate([t[1] for t in sel]); tm = np.concatenate([t[2] for t in sel])
    off = np.concatenate([[0], np.cumsum([len(t[2]) for t in sel])]).astype(np.int64)
    r64 = csp.solve_batch(wp, tm, order=o, seg_offsets=off, want_status=True)
    r32 = csp.solve_batch(wp.astype(np.float32), tm.astype(np.float32), order=o, seg_offsets=off, want_status=True)
    errs32, errs64 = [], []
    for i, (_, w, t) in enumerate(sel):
        ref, _ = oracle.solve(o, w, np.zeros((2, 3)), np.zeros((2, 3)), t, long_double=True)
        a = r32.coeffs[off[i]:off[i + 1]].astype(np.float64).ravel(); b = r64.coeffs[off[i]:off[i + 1]].ravel(); rr = ref.ravel()
        errs32.append(np.max(np.abs(a - rr)) / np.max(np.abs(rr))); errs64.append(np.max(np.abs(b - rr)) / np.max(np.abs(rr)))
    errs32 = np.array(errs32); errs64 = np.array(errs64)
    print("order %d n=%d  f64 err max %.2e  | f32 err median %.2e p90 %.2e max %.2e  status32 any=%s" % (o, len(sel), errs64.max(), np.median(errs32), np.percentile(errs32, 90), errs32.max(), bool(r32.status.any())))
