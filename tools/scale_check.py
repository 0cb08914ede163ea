import sys, json, numpy as np, torch
sys.path.insert(0, ".")
import importlib
csp = importlib.import_module("cs-pathplan_amd")
from tests import synth
# large batches through every kernel family: finite outputs, clean status, spot parity vs the generic kernel
def run(order, S, B, dt, **kw):
    wp, tm = synth.make_batch(B, S, config_id=9)
    d_wp, d_tm = torch.from_numpy(wp.astype(dt)).cuda(), torch.from_numpy(tm.astype(dt)).cuda()
    r = csp.solve_batch(d_wp, d_tm, order=order, want_status=True, **kw)
    torch.cuda.synchronize()
    bad = int((r.status != 0).sum())
    idx = torch.tensor([0, 1, B // 3, B // 2, B - 2, B - 1]).cuda()
    g = csp.solve_batch(d_wp[idx], d_tm[idx], order=order, force_generic=True, **kw)
    err = float(synth.rel_err(r.coeffs[idx].cpu().numpy().astype(np.float64), g.coeffs.cpu().numpy().astype(np.float64)))
    fin = bool(torch.isfinite(r.coeffs).all())
    print(json.dumps({"kernel": r.kernel, "B": B, "S": S, "bad_status": bad, "finite": fin, "spot_rel_err_vs_generic": err}), flush=True)
    assert bad == 0 and fin and err < 1e-6
run(4, 16, 524288, np.float64)
run(4, 16, 524288, np.float64, path_weight=0.2, vel_zero_weight=0.01)
run(4, 64, 262144, np.float64)
run(3, 33, 524288, np.float32)
run(5, 8, 524288, np.float64)
run(4, 200, 32768, np.float32)
