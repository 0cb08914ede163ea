"""Host-side bucketing for mixed batches (BASELINE config C5: per-trajectory segment count and
order).  One C-ABI call serves one derivative order; inside a call trajectories are ragged.
Trajectories are grouped by order and sorted by segment count, so the 64 lanes of a wave run
loops of similar length."""
import importlib

import numpy as np


def solve_mixed(trajs, dtype=np.float32, f32_arith=False, device=None):
    """trajs: list of (order, waypoints [S+1,3], times [S]).  Returns a list of coefficient arrays
    [S,3,2*order] (numpy, `dtype`) in the input order, plus the list of kernel names used."""
    csp = importlib.import_module("cs-pathplan_amd")
    out = [None] * len(trajs)
    kernels = []
    for order in sorted({t[0] for t in trajs}):
        idx = [i for i, t in enumerate(trajs) if t[0] == order]
        idx.sort(key=lambda i: len(trajs[i][2]))
        wp = np.concatenate([np.asarray(trajs[i][1]) for i in idx]).astype(dtype)
        tm = np.concatenate([np.asarray(trajs[i][2]) for i in idx]).astype(dtype)
        off = np.concatenate([[0], np.cumsum([len(trajs[i][2]) for i in idx])]).astype(np.int64)
        if device is not None:
            import torch
            r = csp.solve_batch(torch.from_numpy(wp).to(device), torch.from_numpy(tm).to(device), order=order,
                                seg_offsets=torch.from_numpy(off).to(device), max_segments=int(np.max(np.diff(off))),
                                f32_arith=f32_arith)
            co = r.coeffs.cpu().numpy()
        else:
            r = csp.solve_batch(wp, tm, order=order, seg_offsets=off, f32_arith=f32_arith)
            co = r.coeffs
        kernels.append(r.kernel)
        for j, i in enumerate(idx):
            out[i] = co[off[j]:off[j + 1]]
    return out, kernels
