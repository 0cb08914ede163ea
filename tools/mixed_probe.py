"""Probe of the mixed-order entry: one order at a time, against the round-2 way (one ragged solve_batch launch per (order,
power-of-two length class), trajectories sorted by length and physically re-packed, dealt over four streams).
    python tools/mixed_probe.py [B]"""
import importlib, json, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from tests import synth
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
CLASSES = (16, 32, 64, 128, 256)


def per_class(trajs):
    preps = []
    for order in sorted({t[0] for t in trajs}):
        idx = [i for i, t in enumerate(trajs) if t[0] == order]
        idx.sort(key=lambda i: len(trajs[i][2]))
        lens = np.array([len(trajs[i][2]) for i in idx])
        lo = 0
        for cap in CLASSES:
            hi = int(np.searchsorted(lens, cap, side="right"))
            if hi > lo:
                sub = idx[lo:hi]
                wp = torch.from_numpy(np.concatenate([trajs[i][1] for i in sub]).astype(np.float32)).to(dev)
                tm = torch.from_numpy(np.concatenate([trajs[i][2] for i in sub]).astype(np.float32)).to(dev)
                off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens[lo:hi])]).astype(np.int64)).to(dev)
                preps.append(csp.PreparedSolve(wp, tm, order=order, seg_offsets=off, max_segments=int(lens[hi - 1])))
            lo = hi
    preps.sort(key=lambda p: -p.tm.numel())
    streams = [torch.cuda.Stream(device=dev) for _ in range(min(4, len(preps)))]

    def run():
        main = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(main)
        for k, p in enumerate(preps):
            p.run(streams[k % len(streams)].cuda_stream)
        for st in streams:
            main.wait_stream(st)
    return run, [p.kernel for p in preps]


for orders in ((4,), (3,), (5,), (3, 4, 5)):
    trajs = synth.make_ragged(B, orders=orders)
    o = np.array([t[0] for t in trajs], dtype=np.int32)
    lens = np.array([len(t[2]) for t in trajs])
    wp = np.concatenate([t[1] for t in trajs]).astype(np.float32)
    tm = np.concatenate([t[2] for t in trajs]).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    d = [torch.from_numpy(x).to(dev) for x in (o, wp, tm, off)]
    p = csp.PreparedMixed(d[0], d[1], d[2], d[3])
    ms = bench.timed(p.run, 10, 3, dev)
    run, kernels = per_class(trajs)
    ms_pc = bench.timed(run, 10, 3, dev)
    print(json.dumps({"orders": orders, "mixed_entry_us": round(ms * 1e3, 1), "per_class_launches_us": round(ms_pc * 1e3, 1), "kernels": kernels}), flush=True)
    del p
