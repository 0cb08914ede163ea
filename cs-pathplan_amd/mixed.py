"""Host-side bucketing for mixed batches (BASELINE config C5: per-trajectory segment count and
order).  One C-ABI call serves one derivative order; inside a call trajectories are ragged.
Trajectories are grouped by order, sorted by segment count and cut into power-of-two length classes
(one ragged call each), so the lanes of a wave run loops of similar length."""
import importlib

import numpy as np

# The workspace-free ragged kernel gives every trajectory of a call the same number of lanes, chosen
# from the call's longest trajectory (4 segments per lane): one call per power-of-two length class
# keeps the lanes of short trajectories busy.  Classes below 16 segments are not split further: at
# B = 65536 the extra launches cost more than the idle lanes (measured with bench.py --workload c5).
LENGTH_CLASSES = (16, 32, 64, 128, 256)


def length_classes(sorted_lens):
    """sorted_lens: ascending segment counts.  Returns [(lo, hi)] index ranges, one per non-empty class."""
    sorted_lens = np.asarray(sorted_lens)
    out, lo = [], 0
    for cap in LENGTH_CLASSES + (None,):
        hi = len(sorted_lens) if cap is None else int(np.searchsorted(sorted_lens, cap, side="right"))
        if hi > lo:
            out.append((lo, hi))
        lo = hi
    return out


def solve_mixed(trajs, dtype=np.float32, f32_arith=False, device=None):
    """trajs: list of (order, waypoints [S+1,3], times [S]).  Returns a list of coefficient arrays
    [S,3,2*order] (numpy, `dtype`) in the input order, plus the list of kernel names used."""
    csp = importlib.import_module("cs-pathplan_amd")
    out = [None] * len(trajs)
    kernels = []
    for order in sorted({t[0] for t in trajs}):
        idx = [i for i, t in enumerate(trajs) if t[0] == order]
        idx.sort(key=lambda i: len(trajs[i][2]))
        lens = np.array([len(trajs[i][2]) for i in idx])
        for lo, hi in length_classes(lens):
            sub = idx[lo:hi]
            wp = np.concatenate([np.asarray(trajs[i][1]) for i in sub]).astype(dtype)
            tm = np.concatenate([np.asarray(trajs[i][2]) for i in sub]).astype(dtype)
            off = np.concatenate([[0], np.cumsum(lens[lo:hi])]).astype(np.int64)
            if device is not None:
                import torch
                r = csp.solve_batch(torch.from_numpy(wp).to(device), torch.from_numpy(tm).to(device), order=order,
                                    seg_offsets=torch.from_numpy(off).to(device), max_segments=int(lens[hi - 1]),
                                    f32_arith=f32_arith)
                co = r.coeffs.cpu().numpy()
            else:
                r = csp.solve_batch(wp, tm, order=order, seg_offsets=off, f32_arith=f32_arith)
                co = r.coeffs
            kernels.append(r.kernel)
            for j, i in enumerate(sub):
                out[i] = co[off[j]:off[j + 1]]
    return out, kernels


class MixedBatch:
    """A mixed batch prepared once and solved many times (bench.py's C5 record, a planner's inner loop):
    trajectories bucketed by order and length class, inputs resident on `device` in `dtype` storage, one
    PreparedSolve per bucket.  `run()` enqueues every bucket; a bucket alone cannot fill 1024 SIMDs, so the
    buckets are dealt over four HIP streams, longest first, and joined on torch's current stream.
    `coeffs(i)` returns trajectory i's [S,3,2*order] block (a view into its bucket's output)."""

    def __init__(self, csp, trajs, device, dtype=None, streams=4):
        import torch
        dtype = dtype or torch.float32
        npdt = np.float32 if dtype == torch.float32 else np.float64
        width = 4 if dtype == torch.float32 else 8
        self.dev = device
        self.buckets, self.where = [], [None] * len(trajs)
        self.algorithmic_bytes = 0
        for order in sorted({t[0] for t in trajs}):
            idx = [i for i, t in enumerate(trajs) if t[0] == order]
            idx.sort(key=lambda i: len(trajs[i][2]))
            lens = np.array([len(trajs[i][2]) for i in idx])
            for lo, hi in length_classes(lens):
                sub = idx[lo:hi]
                wp = torch.from_numpy(np.concatenate([np.asarray(trajs[i][1]) for i in sub]).astype(npdt)).to(device)
                tm = torch.from_numpy(np.concatenate([np.asarray(trajs[i][2]) for i in sub]).astype(npdt)).to(device)
                off_h = np.concatenate([[0], np.cumsum(lens[lo:hi])]).astype(np.int64)
                ps = csp.PreparedSolve(wp, tm, order=order, seg_offsets=torch.from_numpy(off_h).to(device), max_segments=int(lens[hi - 1]))
                for j, i in enumerate(sub):
                    self.where[i] = (len(self.buckets), int(off_h[j]), int(off_h[j + 1]))
                self.buckets.append(ps)
                self.algorithmic_bytes += int(sum(width * (3 * (int(n) + 1) + int(n)) + width * 3 * int(n) * 2 * order for n in lens[lo:hi]))
        self.order_of_launch = sorted(range(len(self.buckets)), key=lambda b: -self.buckets[b].tm.numel())
        self.streams = [torch.cuda.Stream(device=device) for _ in range(min(streams, len(self.buckets)))]
        self.launches = len(self.buckets)
        self.kernels = sorted({ps.kernel for ps in self.buckets})

    def run(self):
        import torch
        main = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            st.wait_stream(main)
        for k, b in enumerate(self.order_of_launch):
            self.buckets[b].run(self.streams[k % len(self.streams)].cuda_stream)
        for st in self.streams:
            main.wait_stream(st)

    def coeffs(self, i):
        b, s0, s1 = self.where[i]
        return self.buckets[b].out[s0:s1]
