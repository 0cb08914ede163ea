"""rocprofv3 --pmc ... -- python3 tools/mixed_pmc.py : order-4-only C5 through the mixed entry and through per-class launches (2 runs each)."""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
from tests import synth
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
B = 65536
trajs = synth.make_ragged(B, orders=(4,))
o = np.array([t[0] for t in trajs], dtype=np.int32)
lens = np.array([len(t[2]) for t in trajs])
wp = np.concatenate([t[1] for t in trajs]).astype(np.float32)
tm = np.concatenate([t[2] for t in trajs]).astype(np.float32)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
d = [torch.from_numpy(x).to(dev) for x in (o, wp, tm, off)]
p = csp.PreparedMixed(d[0], d[1], d[2], d[3])
for _ in range(2):
    p.run()
torch.cuda.synchronize()
idx = sorted(range(B), key=lambda i: lens[i])
sl = lens[idx]
lo = 0
for cap in (16, 32, 64):
    hi = int(np.searchsorted(sl, cap, side="right"))
    sub = idx[lo:hi]
    w = torch.from_numpy(np.concatenate([trajs[i][1] for i in sub]).astype(np.float32)).to(dev)
    t = torch.from_numpy(np.concatenate([trajs[i][2] for i in sub]).astype(np.float32)).to(dev)
    of = torch.from_numpy(np.concatenate([[0], np.cumsum(sl[lo:hi])]).astype(np.int64)).to(dev)
    ps = csp.PreparedSolve(w, t, order=4, seg_offsets=of, max_segments=int(sl[hi - 1]))
    for _ in range(2):
        ps.run()
    torch.cuda.synchronize()
    lo = hi
