"""How long ONE work unit of the lane-pair sweep takes: B = 64 x (number of persistent workgroups) trajectories of one order and
one segment count -> every workgroup gets exactly one unit, the call's time (minus ~22 us of bucketing) is the unit's.
    python tools/twist_unit_time.py"""
import importlib, json, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
g = np.random.default_rng(5)
WG = 512
for order in (2, 3, 4, 5):
    row = {}
    for S in (8, 16, 32, 64):
        B = 64 * WG
        lens = np.full(B, S)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        wp = np.cumsum(g.normal(size=(int(off[-1]) + B, 3)), axis=0).astype(np.float32)
        tm = g.uniform(0.5, 2.0, size=int(off[-1])).astype(np.float32)
        o = np.full(B, order, dtype=np.int32)
        d = [torch.from_numpy(x).to(dev) for x in (o, wp, tm, off)]
        p = csp.PreparedMixed(d[0], d[1], d[2], d[3])
        ms = bench.timed(p.run, 10, 3, dev)
        row[S] = round(ms * 1e3, 1)
        del p
    print(json.dumps({"order": order, "call_us_by_S": row, "us_per_segment_pair": round((row[64] - row[32]) / 16, 2)}), flush=True)
