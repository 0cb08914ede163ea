"""Probe of the mixed-order entry: one order at a time, against the round-2 per-class launches.  python tools/mixed_probe.py"""
import importlib, json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from tests import synth
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for orders in ((4,), (3,), (5,), (3, 4, 5)):
    for sort in (False, True):
        trajs = synth.make_ragged(B, orders=orders)
        if sort:
            trajs.sort(key=lambda t: (t[0], len(t[2])))
        o = np.array([t[0] for t in trajs], dtype=np.int32)
        lens = np.array([len(t[2]) for t in trajs])
        wp = np.concatenate([t[1] for t in trajs]).astype(np.float32)
        tm = np.concatenate([t[2] for t in trajs]).astype(np.float32)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        d = [torch.from_numpy(x).to(dev) for x in (o, wp, tm, off)]
        p = csp.PreparedMixed(d[0], d[1], d[2], d[3])
        ms = bench.timed(p.run, 10, 3, dev)
        print(json.dumps({"orders": orders, "sorted_input": sort, "mixed_us": round(ms * 1e3, 1)}), flush=True)
        del p
