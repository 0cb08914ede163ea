"""The lane-pair sweep (minsnap_twist_impl.h) against the chunked kernels inside csp_minsnap_solve_mixed: same batches, one
process per setting of CSP_MIXED_TWIST (the library reads it once), results compared per power, times side by side.
    python tools/twist_probe.py [B] [f32|f64]"""
import importlib, json, os, subprocess, sys
import numpy as np

B = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 65536
F32 = not (len(sys.argv) > 2 and sys.argv[2] == "f64")
STATUS = os.environ.get("TWIST_PROBE_STATUS", "0") == "1"   # bench.py's c5 record runs without the per-trajectory status
CASES = ((4,), (3,), (5,), (2,), (3, 4, 5))


def child(mask, out):
    import torch
    sys.path.insert(0, ".")
    import bench
    from tests import synth
    csp = importlib.import_module("cs-pathplan_amd")
    dev = torch.device("cuda", 0)
    res = {}
    for orders in CASES:
        trajs = synth.make_ragged(B, orders=orders)
        o = np.array([t[0] for t in trajs], dtype=np.int32)
        lens = np.array([len(t[2]) for t in trajs])
        ft = np.float32 if F32 else np.float64
        wp = np.concatenate([t[1] for t in trajs]).astype(ft)
        tm = np.concatenate([t[2] for t in trajs]).astype(ft)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        d = [torch.from_numpy(x).to(dev) for x in (o, wp, tm, off)]
        p = csp.PreparedMixed(d[0], d[1], d[2], d[3], want_status=STATUS)
        p.out.zero_()
        p.run()
        torch.cuda.synchronize()
        co = p.out.cpu().numpy().copy()
        st = p.status.cpu().numpy().copy() if STATUS else np.zeros(1, np.int32)
        ms = bench.timed(p.run, 10, 3, dev)
        key = "o" + "".join(map(str, orders))
        np.save(f"{out}_{key}.npy", co)
        res[key] = {"us": round(ms * 1e3, 1), "status_nonzero": int((st != 0).sum()), "nonfinite": int((~np.isfinite(co)).sum())}
        del p
    print(json.dumps(res), flush=True)


if len(sys.argv) > 3 and sys.argv[3] == "--child":
    child(int(sys.argv[4]), sys.argv[5])
    sys.exit(0)

os.makedirs("/tmp/twp", exist_ok=True)
runs = {}
for mask in (0, 15):
    env = dict(os.environ, CSP_MIXED_TWIST=str(mask))
    r = subprocess.run([sys.executable, __file__, str(B), "f32" if F32 else "f64", "--child", str(mask), f"/tmp/twp/m{mask}"], env=env,
                       capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-4000:])
        sys.exit(1)
    runs[mask] = json.loads(r.stdout.strip().splitlines()[-1])
sys.path.insert(0, ".")
from tests import synth
for orders in CASES:
    key = "o" + "".join(map(str, orders))
    a, b = np.load(f"/tmp/twp/m0_{key}.npy"), np.load(f"/tmp/twp/m15_{key}.npy")
    denom = np.maximum(np.abs(a), 1e-300)
    rel = np.abs(a - b) / np.maximum(denom, np.abs(a).max() * 1e-12)
    print(json.dumps({"orders": orders, "chunked": runs[0][key], "twist": runs[15][key], "max_rel_diff": float(rel.max()),
                      "n_diff_gt_1e-4": int((rel > 1e-4).sum()), "n": int(a.size)}), flush=True)
