import importlib, sys, json, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from tests import synth
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
B = 524288
g = np.random.Generator(np.random.Philox(key=[77, 0]))
S = g.integers(4, 65, size=B)
o = g.integers(3, 6, size=B).astype(np.int32)
off = np.concatenate([[0], np.cumsum(S)]).astype(np.int64)
tot = int(off[-1])
wp = np.cumsum(g.normal(size=(tot + B, 3)), axis=0).astype(np.float32) % 50.0   # any finite waypoints
tm = g.uniform(0.5, 2.0, size=tot).astype(np.float32)
d = [torch.from_numpy(x).to(dev) for x in (o, wp, tm, off)]
p = csp.PreparedMixed(d[0], d[1], d[2], d[3], want_status=True)
p.out.fill_(float("nan"))
ms = bench.timed(p.run, 5, 2, dev)
st = p.status.cpu().numpy()
blk = csp.mixed_block_elements(o, off, True)
host_off = np.concatenate([[0], np.cumsum(blk)])
ok_off = np.array_equal(p.coeff_offsets.cpu().numpy(), host_off)
out = p.out
nan_count = int(torch.isnan(out).sum().item())
pad = int((blk - np.diff(off) * 6 * o).sum())
width = 4
nbytes = int(np.sum(width * (3 * (S + 1) + S) + width * 3 * S * 2 * o))
print(json.dumps({"B": B, "ms": ms, "solves_per_s": B / ms * 1e3, "frac_hbm": nbytes / (ms * 1e-3) / 8e12, "status_nonzero": int((st != 0).sum()), "offsets_ok": ok_off,
                  "nan_elements": nan_count, "padding_elements": pad}))
# a few trajectories against single calls
for i in (0, 1234, B // 2, B - 1):
    n, oo = int(S[i]), int(o[i])
    a = out[host_off[i]:host_off[i] + 6 * oo * n].reshape(n, 3, 2 * oo).cpu().numpy()
    one = csp.solve_batch(wp[off[i] + i: off[i + 1] + i + 1], tm[off[i]:off[i + 1]], order=oo, seg_offsets=np.array([0, n]), max_segments=n)
    print(i, n, oo, float(np.max(np.abs(a - one.coeffs)) / np.max(np.abs(one.coeffs))))
