"""Round-3 GPU tests (through the C-ABI): whole-line coefficient stores of orders 2 / 3 / 5 (and of the path-penalty
kernels), the mixed-order C-ABI entry, the one-pass batched sampler, ...

PARITY UNPINNED w.r.t. the real Eigen build (oracle/dense_oracle.c header); tolerances are stated per test.
"""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

TOL_WELL = 1e-8


def _old_path_rows(csp, torch, d_wp, d_tm, d_bc, order, starts, **kw):
    """The same trajectories in batches of 63: ragged slices take the record-at-a-time store path (FULL = false), whose
    arithmetic per trajectory is the whole-slice kernels' -- a bit-exact reference for the store rewrite."""
    out = {}
    for s in starts:
        # .clone(): the C-ABI wants 16-byte aligned device pointers, which a row offset into the batch need not be
        r = csp.solve_batch(d_wp[s:s + 63].clone(), d_tm[s:s + 63].clone(), None if d_bc is None else d_bc[s:s + 63].clone(), order=order, **kw)
        torch.cuda.synchronize()
        out[s] = r.coeffs.cpu().numpy()
    return out


# ring-eligible buckets: trajectories that start on 128-byte lines (S * 48 * order % 128 == 0); S = 4, 12 (order 2) and
# S = 8 (orders 3, 5) have a line CUT by the role boundary (each role stores its part)
@pytest.mark.parametrize("order,S", [(2, 4), (2, 8), (2, 12), (2, 16), (3, 8), (3, 16), (5, 8)])
def test_whole_line_stores_of_the_other_orders(csp, oracle_mod, order, S):
    """fixedk::LineRing: the records of orders 2 / 3 / 5 leave as whole 128-byte lines through a ring in the staging tile.
    Bit-equal with the record-at-a-time path on sampled 63-trajectory windows, against the generic kernel on the whole
    batch (a misplaced line is an O(1) error), against the oracle on samples; persistent (shared boundary conditions,
    several slices per workgroup) and one-workgroup-per-slice (per-trajectory boundary conditions) forms."""
    import torch
    rng = np.random.default_rng(300 + 10 * order + S)
    B = 512 * 64 * 2 + 64 * 5                      # > 2 * CUs slices: persistent workgroups walk several
    wp, tm = synth.make_batch(B, S, config_id=70 + order)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    starts = [0, 64 * 100 + 1, B - 63]
    # shared boundary conditions -> persistent kernel
    r = csp.solve_batch(d_wp, d_tm, order=order, want_status=True)
    assert r.kernel == "fixed_o%d_s%d_f64" % (order, S)
    g = csp.solve_batch(d_wp, d_tm, order=order, force_generic=True)
    torch.cuda.synchronize()
    a, gg = r.coeffs.cpu().numpy(), g.coeffs.cpu().numpy()
    assert not r.status.cpu().numpy().any()
    assert synth.rel_err(a, gg) < (1e-7 if order == 5 else 1e-9)
    for s, ref in _old_path_rows(csp, torch, d_wp, d_tm, None, order, starts).items():
        assert np.array_equal(a[s:s + 63], ref), (order, S, s)
    idx = np.array([0, 63, 64, 4097, B - 1])
    ref, _ = oracle_mod.solve_batch(order, wp[idx], tm[idx])
    assert synth.rel_err_per_power(a[idx], ref) < (1e-6 if order == 5 else 1e-7)
    # per-trajectory boundary conditions and weights -> one workgroup per slice (same fixed_body, FULL = true)
    Bs = 64 * 300
    bc = rng.normal(size=(Bs, 4, 3))
    vw = rng.uniform(0.0, 0.3, size=Bs)
    d_bc, d_vw = torch.from_numpy(bc).cuda(), torch.from_numpy(vw).cuda()
    r = csp.solve_batch(d_wp[:Bs], d_tm[:Bs], d_bc, order=order, vel_zero_weight_per_traj=d_vw)
    g = csp.solve_batch(d_wp[:Bs], d_tm[:Bs], d_bc, order=order, vel_zero_weight_per_traj=d_vw, force_generic=True)
    torch.cuda.synchronize()
    a, gg = r.coeffs.cpu().numpy(), g.coeffs.cpu().numpy()
    assert synth.rel_err(a, gg) < (1e-7 if order == 5 else 1e-9)
    for s in (0, 64 * 17 + 3):
        ro = csp.solve_batch(d_wp[s:s + 63].clone(), d_tm[s:s + 63].clone(), d_bc[s:s + 63].clone(), order=order,
                             vel_zero_weight_per_traj=d_vw[s:s + 63].clone())
        torch.cuda.synchronize()
        assert np.array_equal(a[s:s + 63], ro.coeffs.cpu().numpy()), (order, S, s)


@pytest.mark.parametrize("order,S,B", [(2, 16, 192 * 1024), (3, 16, 128 * 1024), (5, 8, 160 * 1024)])
def test_whole_line_stores_beyond_the_infinity_cache(csp, order, S, B):
    """More than 256 MB of coefficients: the launcher turns on non-temporal stores, which is only sound (and only fast)
    because every line leaves whole.  Bit-equal with the two halves solved separately (below the threshold: ordinary
    stores)."""
    import torch
    assert B * S * 48 * order > 256 * 1024 * 1024
    wp, tm = synth.make_batch(B, S, config_id=80 + order)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    whole = csp.solve_batch(d_wp, d_tm, order=order).coeffs
    h = B // 2
    lo = csp.solve_batch(d_wp[:h], d_tm[:h], order=order).coeffs
    hi = csp.solve_batch(d_wp[h:], d_tm[h:], order=order).coeffs
    torch.cuda.synchronize()
    assert torch.equal(whole[:h], lo) and torch.equal(whole[h:], hi)
    assert bool(torch.isfinite(whole).all())
