"""Mixed batches (BASELINE config C5: per-trajectory segment count AND derivative order) -- a thin caller of
csp_minsnap_solve_mixed (include/csp_minsnap.h).  The bucketing by (order, length class) and the launch schedule live in the
library now (cs-pathplan_amd/csrc/minsnap_mixed.hip: device-side histogram / scan / scatter, one persistent launch per order);
this module only concatenates a Python list of trajectories into the C-ABI's ragged layout."""
import importlib

import numpy as np


def pack(trajs, dtype=np.float32):
    """trajs: list of (order, waypoints [S+1,3], times [S]) -> (orders [B] i32, waypoints [sum(S)+B,3], times [sum S], seg_offsets [B+1])."""
    orders = np.array([t[0] for t in trajs], dtype=np.int32)
    lens = np.array([len(t[2]) for t in trajs], dtype=np.int64)
    wp = np.concatenate([np.asarray(t[1]) for t in trajs]).astype(dtype)
    tm = np.concatenate([np.asarray(t[2]) for t in trajs]).astype(dtype)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return orders, wp, tm, off


def solve_mixed(trajs, dtype=np.float32, device=None):
    """Returns the list of coefficient arrays [S,3,2*order] (numpy, `dtype`) in the input order and the per-trajectory status."""
    csp = importlib.import_module("cs-pathplan_amd")
    orders, wp, tm, off = pack(trajs, dtype)
    if device is not None:
        import torch
        r = csp.solve_mixed(torch.from_numpy(orders).to(device), torch.from_numpy(wp).to(device), torch.from_numpy(tm).to(device),
                            torch.from_numpy(off).to(device), want_status=True)
        torch.cuda.synchronize(device)
        co, cof, st = r.coeffs.cpu().numpy(), r.coeff_offsets.cpu().numpy(), r.status.cpu().numpy()
    else:
        r = csp.solve_mixed(orders, wp, tm, off, want_status=True)
        co, cof, st = r.coeffs, r.coeff_offsets, r.status
    out = [co[cof[i]:cof[i] + 6 * t[0] * len(t[2])].reshape(len(t[2]), 3, 2 * t[0]) for i, t in enumerate(trajs)]
    return out, st


class MixedBatch:
    """A mixed batch resident on `device`, solved many times (bench.py's C5 record, a planner's inner loop): `run()` is ONE
    C-ABI call -- device-side bucketing included, coefficients in the caller's order.  `coeffs(i)` returns trajectory i's
    [S,3,2*order] block (a view into the flat output)."""

    def __init__(self, csp, trajs, device, dtype=None):
        import torch
        dtype = dtype or torch.float32
        npdt = np.float32 if dtype == torch.float32 else np.float64
        width = 4 if dtype == torch.float32 else 8
        self.dev = device
        orders, wp, tm, off = pack(trajs, npdt)
        self.shapes = [(len(t[2]), int(t[0])) for t in trajs]
        self.prep = csp.PreparedMixed(torch.from_numpy(orders).to(device), torch.from_numpy(wp).to(device), torch.from_numpy(tm).to(device),
                                      torch.from_numpy(off).to(device), max_segments=int(np.max(np.diff(off))))
        pad = 4 if dtype == torch.float32 else 2
        self.offsets = np.concatenate([[0], np.cumsum([(6 * o * n + pad - 1) // pad * pad for n, o in self.shapes])]).astype(np.int64)
        self.algorithmic_bytes = int(sum(width * (3 * (n + 1) + n) + width * 3 * n * 2 * o for n, o in self.shapes))
        self.launches = 1
        self.kernels = ["mixed (device-side bucketing + ONE persistent launch of the lane-pair sweep for every order, S <= 64; "
                        "longer trajectories: one persistent chunked launch per order)"]

    def run(self):
        self.prep.run()

    def coeffs(self, i):
        n, o = self.shapes[i]
        return self.prep.out[self.offsets[i]:self.offsets[i] + 6 * o * n].reshape(n, 3, 2 * o)
