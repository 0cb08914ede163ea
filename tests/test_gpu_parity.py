"""Parity of the HIP path (through the C-ABI) against the oracle and the golden fixtures.

Bar (BASELINE.json north_star): polynomial coefficients within 1e-6 relative of the fp64
reference on identical waypoint/time inputs.  The tolerance used below for well-scaled inputs is
1e-8 (two orders tighter), stated per test.  PARITY UNPINNED w.r.t. the real Eigen build: the
oracle is a restatement (oracle/dense_oracle.c header).
"""
import numpy as np
import pytest

from tests import synth
from tests.conftest import load_cases

pytestmark = pytest.mark.gpu

# Every gate below is the PER-POWER relative error (tests/synth.py::parity_gate); tolerances placed from a survey run on
# the MI355X (CSP_PARITY_SURVEY, numbers in DESIGN.md section 3)
# the MI355X (CSP_PARITY_SURVEY=file; round 3, 7828 gate evaluations; the measured maximum is quoted beside each):
TOL_WELL = 5e-8         # HIP vs the fp64 dense oracle, well-scaled inputs, orders <= 4: measured <= 6.2e-9 (the oracle's own rounding)
TOL_O5 = 1e-6           # order-5 fixtures (cond ~1e9 in the dense formulation): 6.7e-8
TOL_PEN = 5e-8          # with the path penalty: 3.7e-9
TOL_KERNELS = 1e-8      # two HIP kernels against each other, orders <= 4: 3.1e-9 (chunked/span), 2e-10 (fixed)
TOL_KERNELS_O5 = 1e-7   # order 5, S <= 8: 5.5e-9
TOL_KERNELS_PATH = 1e-9 # path kernel vs generic: 5.8e-11
# Order 5 beyond a few segments: the fp64 dense oracle itself is 3e-6 ... 9e-6 per power from the 80-bit answer at S = 64
# (DESIGN.md section 3), so order-5 results are gated against the 80-bit oracle at the north-star tolerance; two HIP kernel
# families agree to 8.2e-6 per power at S = 69 (cond(R_PP) ~ 1e8 -- the spline problem's own conditioning): gate 5e-5.
TOL_KERNELS_O5_LONG = 5e-5
NORTH_STAR_TOL = 1e-6


def _solve_case(csp, c, **kw):
    r = csp.solve_batch(c["path"][None], c["time"][None], c["bc"][None], order=c["order"],
                        path_weight=c["path_weight"], vel_zero_weight=c["vel_zero_weight"],
                        want_max_dev=True, want_status=True, **kw)
    return r


def test_device_present(csp):
    assert csp.device_count() >= 1, "no gfx950 device: the product path has no CPU fallback"


@pytest.mark.parametrize("fname", ["F1_kat.json", "F3_wellscaled.json", "F5_ragged.json"])
def test_golden_unpenalised(csp, fname):
    for c in load_cases(fname):
        for force in (True, False):
            r = _solve_case(csp, c, force_generic=force)
            # order 5 systems are the worst conditioned of the fixture set (cond ~1e9 in the
            # dense formulation): both sides carry ~1e-9 of their own rounding
            tol = TOL_O5 if c["order"] == 5 else TOL_WELL
            synth.parity_gate(r.coeffs[0], c["coeff"], tol, (fname, c["name"], r.kernel))
            assert int(r.status[0]) == 0, (c["name"], r.status)
            assert abs(r.max_dev[0] - c["max_dev"]) < 1e-9


def test_golden_penalties(csp):
    for c in load_cases("F6_penalties.json"):
        r = _solve_case(csp, c)
        synth.parity_gate(r.coeffs[0], c["coeff"], TOL_PEN, ("F6", c["name"], r.kernel))
        assert abs(r.max_dev[0] - c["max_dev"]) < 1e-7 * max(1.0, abs(c["max_dev"])), (c["name"], r.max_dev[0], c["max_dev"])


@pytest.mark.parametrize("fname", ["F2_readme_uav31.json", "F2b_readme_uav31_merged.json"])
def test_golden_readme_uav31(csp, oracle_mod, fname):
    """README waypoints (config C1): the seven of readme.md:14-20 (F2, six segments) and the six getPlan really feeds after its
    200 m merge (F2b, five segments; uavPathPlanning.cpp:2643-2664).  cond(M) reaches 1e16..1e20 in the reference's raw-time dense
    formulation, so the fp64 dense answer is itself only accurate to a few digits; the 80-bit
    long-double oracle is the yardstick.  The gate is the PER-POWER relative error (every coefficient
    power judged against its own magnitude: the t^7 coefficients are ~1e-14 of the constant term here,
    and the constant term is just the waypoint copied through, so the norm-wise figure of SURVEY.md 8d
    -- printed beside it -- is vacuous on this fixture)."""
    for c in load_cases(fname):
        ld, _ = oracle_mod.solve(c["order"], c["path"], c["vel"], c["acc"], c["time"], c["path_weight"],
                                 c["vel_zero_weight"], long_double=True)
        S, m = c["segments"], 2 * c["order"]
        ld = ld.reshape(S, 3, m)
        r = _solve_case(csp, c)
        e_gpu = synth.rel_err_per_power(r.coeffs.reshape(S, 3, m), ld)
        e_dense = synth.rel_err_per_power(c["coeff"], ld)
        print("%s cond(M)=%.1e  per-power: gpu-vs-ld %.2e  dense-fp64-vs-ld %.2e   (norm-wise: %.2e / %.2e)" % (
            c["name"], c["cond_M"], e_gpu, e_dense, synth.rel_err(r.coeffs.reshape(1, -1), ld.reshape(1, -1)),
            synth.rel_err(c["coeff"].reshape(1, -1), ld.reshape(1, -1))))
        if c["cond_M"] < 1e12:
            assert e_gpu < NORTH_STAR_TOL, (c["name"], e_gpu)
        else:
            # the structured solve must be at least as close to the long-double answer as the
            # dense fp64 restatement is (it avoids the ill-conditioned M inverse altogether)
            assert e_gpu <= max(10 * e_dense, NORTH_STAR_TOL), (c["name"], e_gpu, e_dense)


@pytest.mark.parametrize("S,B", [(8, 256), (16, 256)])
def test_seeded_batch_vs_oracle(csp, oracle_mod, S, B):
    wp, tm = synth.make_batch(B, S, config_id=2 if S == 8 else 3)
    ref, _ = oracle_mod.solve_batch(4, wp, tm, nthreads=oracle_mod.max_threads())
    for force in (True, False):
        r = csp.solve_batch(wp, tm, order=4, want_status=True, force_generic=force)
        synth.parity_gate(r.coeffs, ref, TOL_WELL, ("seeded", S, r.kernel))
        assert not r.status.any()


@pytest.mark.parametrize("S", list(range(2, 17)))
def test_fixed_kernel_every_bucket_and_ragged_tails(csp, oracle_mod, S):
    """Every fixed-size bucket (2 <= S <= 16, either parity: the top role owns ceil(S/2) segments), batch sizes around the 64-trajectory slice boundary,
    per-trajectory boundary conditions and per-trajectory zero-velocity weights."""
    rng = np.random.default_rng(100 + S)
    for B in (1, 63, 64, 65, 130):
        wp, tm = synth.make_batch(B, S, config_id=30 + S)
        bc = rng.normal(size=(B, 4, 3))
        vw = rng.uniform(0.0, 0.5, size=B)
        r = csp.solve_batch(wp, tm, bc, order=4, vel_zero_weight_per_traj=vw, want_status=True, want_max_dev=True)
        assert r.kernel == "fixed_o4_s%d_f64" % S
        assert not r.status.any() and not r.max_dev.any()
        g = csp.solve_batch(wp, tm, bc, order=4, vel_zero_weight_per_traj=vw, force_generic=True)
        synth.parity_gate(r.coeffs, g.coeffs, TOL_KERNELS, ("every_bucket vs generic", S, B))
        for b in (0, B - 1):
            ref, _ = oracle_mod.solve(4, wp[b], bc[b, [0, 1]], bc[b, [2, 3]], tm[b], 0.0, float(vw[b]))
            synth.parity_gate(r.coeffs[b], ref, TOL_WELL, ("every_bucket vs oracle", S, B, b))


@pytest.mark.parametrize("S", [2, 3, 4, 6, 7, 8, 13, 16])
def test_persistent_workgroups_walk_several_slices(csp, oracle_mod, S):
    """B > 2*CUs*64 makes every persistent workgroup solve more than one 64-trajectory slice, so the
    LDS-DMA prefetch, its counted wait and the tile/exchange reuse across slices are exercised for
    every store-burst shape (paired, single, straddling middle pair at S=6)."""
    import torch
    B = 512 * 64 * 2 + 64 * 37 + 5
    wp, tm = synth.make_batch(B, S, config_id=50 + S)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    for seg_major in (False, True):
        a = csp.solve_batch(d_wp, d_tm, order=4, segment_major=seg_major).coeffs
        g = csp.solve_batch(d_wp, d_tm, order=4, force_generic=True, segment_major=seg_major).coeffs
        torch.cuda.synchronize()
        a, g = a.cpu().numpy(), g.cpu().numpy()
        if seg_major:
            a, g = np.transpose(a, (1, 0, 2, 3)), np.transpose(g, (1, 0, 2, 3))
        synth.parity_gate(a, g, TOL_KERNELS, ("persistent vs generic", S, seg_major))
    idx = np.array([0, 63, 64, 32768, 40000, B - 6, B - 1])
    ref, _ = oracle_mod.solve_batch(4, wp[idx], tm[idx])
    synth.parity_gate(a[idx], ref, TOL_WELL, ("persistent vs oracle", S))


@pytest.mark.parametrize("order,S_list", [(2, [2, 5, 16]), (3, [3, 4, 11, 16]), (5, [2, 5, 8])])
def test_fixed_kernels_of_the_other_orders(csp, oracle_mod, order, S_list):
    """Orders 2 (the shipped yaml), 3 (the reference's default) and 5 have register-resident buckets
    too: compare with the generic kernel on a multi-slice batch and with the oracle on samples."""
    import torch
    rng = np.random.default_rng(order)
    for S in S_list:
        for B in (1, 65, 512 * 64 + 64 * 3 + 7):
            wp, tm = synth.make_batch(B, S, config_id=60 + order)
            bc = rng.normal(size=(B, 4, 3))
            vw = rng.uniform(0.0, 0.3, size=B)
            d = [torch.from_numpy(x).cuda() for x in (wp, tm, bc, vw)]
            r = csp.solve_batch(d[0], d[1], d[2], order=order, vel_zero_weight_per_traj=d[3], want_status=True)
            assert r.kernel == "fixed_o%d_s%d_f64" % (order, S), r.kernel
            g = csp.solve_batch(d[0], d[1], d[2], order=order, vel_zero_weight_per_traj=d[3], force_generic=True)
            torch.cuda.synchronize()
            assert not r.status.cpu().numpy().any()
            a, gg = r.coeffs.cpu().numpy(), g.coeffs.cpu().numpy()
            synth.parity_gate(a, gg, TOL_KERNELS_O5 if order == 5 else TOL_KERNELS, ("other orders vs generic", order, S, B))
            for b in sorted({0, B // 2, B - 1}):
                # order 5: against the 80-bit oracle (the fp64 dense restatement is 1.2e-6 per power off at S = 8 already)
                ref, _ = oracle_mod.solve(order, wp[b], bc[b, [0, 1]], bc[b, [2, 3]], tm[b], 0.0, float(vw[b]), long_double=order == 5)
                synth.parity_gate(a[b], ref, NORTH_STAR_TOL if order == 5 else TOL_WELL, ("other orders vs oracle", order, S, B, b))


@pytest.mark.parametrize("order,S_list", [(2, [2, 3, 6, 16]), (3, [2, 5, 12, 16]), (4, [2, 3, 7, 8, 13, 16])])
def test_path_penalty_register_kernel(csp, oracle_mod, order, S_list):
    """The path-deviation penalty (pre-solve, 17-sample t* pick, penalised solve, deviation metric;
    minimum_snap.cpp:347-469, :594-624) in the register-resident kernel: same coefficients, max_dev and
    status as the generic kernel on whole batches, and as the oracle on sampled trajectories."""
    import torch
    rng = np.random.default_rng(100 + order)
    for S in S_list:
        for B, pw in ((1, 0.5), (200, 1e-2), (64 * 9 + 13, 2.0)):
            wp, tm = synth.make_batch(B, S, config_id=80 + order)
            bc = rng.normal(size=(B, 4, 3))
            vw = rng.uniform(0.0, 0.3, size=B)
            d = [torch.from_numpy(x).cuda() for x in (wp, tm, bc, vw)]
            kw = dict(order=order, path_weight=pw, vel_zero_weight_per_traj=d[3], want_status=True, want_max_dev=True)
            r = csp.solve_batch(d[0], d[1], d[2], **kw)
            assert r.kernel == "fixedpath_o%d_s%d_f64" % (order, S), r.kernel
            g = csp.solve_batch(d[0], d[1], d[2], force_generic=True, **kw)
            torch.cuda.synchronize()
            assert not r.status.cpu().numpy().any()
            a, gg = r.coeffs.cpu().numpy(), g.coeffs.cpu().numpy()
            synth.parity_gate(a, gg, TOL_KERNELS_PATH, ("path kernel vs generic", order, S, B))
            md, mdg = r.max_dev.cpu().numpy(), g.max_dev.cpu().numpy()
            assert np.max(np.abs(md - mdg)) < 1e-8 * max(1.0, np.max(mdg)), (order, S, B)
            for b in sorted({0, B // 2, B - 1}):
                ref, ref_md = oracle_mod.solve(order, wp[b], bc[b, [0, 1]], bc[b, [2, 3]], tm[b], pw, float(vw[b]))
                synth.parity_gate(a[b], ref, TOL_PEN, ("path kernel vs oracle", order, S, B, b))
                assert abs(md[b] - ref_md) < 1e-7 * max(1.0, ref_md), (order, S, B, b)
        # batch-wide boundary conditions and weight, host-memory entry
        wp, tm = synth.make_batch(70, S, config_id=90 + order)
        bc1 = rng.normal(size=(1, 4, 3))
        r = csp.solve_batch(wp, tm, bc1, order=order, path_weight=0.3, vel_zero_weight=0.05, want_max_dev=True)
        assert r.kernel.startswith("fixedpath_")
        ref, ref_md = oracle_mod.solve_batch(order, wp, tm, bc1, path_weight=0.3, vel_zero_weight=0.05)
        synth.parity_gate(r.coeffs, ref, TOL_PEN, ("path kernel host entry vs oracle", order, S))
        assert np.max(np.abs(r.max_dev - ref_md)) < 1e-7 * max(1.0, np.max(ref_md))


def test_status_flags_bad_trajectories_only(csp):
    wp, tm = synth.make_batch(130, 16, config_id=3)
    tm[5, 3] = 0.0        # zero-length segment time -> 1/T = inf -> non-finite coefficients
    tm[77, 9] = float("nan")
    for force in (False, True):
        r = csp.solve_batch(wp, tm, order=4, want_status=True, force_generic=force)
        bad = np.flatnonzero(r.status)
        assert bad.tolist() == [5, 77], (r.kernel, bad)
        good = np.setdiff1d(np.arange(130), bad)
        assert np.isfinite(r.coeffs[good]).all()


def test_large_time_and_length_scales(csp, oracle_mod):
    """Kilometre-scale waypoints and segment times of minutes (README regime): compare both the HIP
    result and the dense fp64 oracle with the 80-bit long-double oracle."""
    wp, tm = synth.make_batch(16, 16, config_id=41)
    wp = wp * 2000.0
    tm = tm * 120.0
    r = csp.solve_batch(wp, tm, order=4)
    z = np.zeros((2, 3))
    e_gpu = e_dense = 0.0
    for b in range(16):
        ld, _ = oracle_mod.solve(4, wp[b], z, z, tm[b], long_double=True)
        dn, _ = oracle_mod.solve(4, wp[b], z, z, tm[b])
        # per-segment, per-power scale: coefficients of t^7 are ~1e-14 of those of t^0 here
        den = np.max(np.abs(ld.reshape(16, 3, 8)), axis=(0, 1))
        e_gpu = max(e_gpu, float(np.max(np.abs(r.coeffs[b] - ld.reshape(16, 3, 8)) / den)))
        e_dense = max(e_dense, float(np.max(np.abs(dn.reshape(16, 3, 8) - ld.reshape(16, 3, 8)) / den)))
    print("large scales: hip-vs-ld %.2e   dense-fp64-vs-ld %.2e (per-power relative)" % (e_gpu, e_dense))
    assert e_gpu < NORTH_STAR_TOL
    assert e_gpu <= 10 * e_dense + 1e-12


def test_large_scales_long_trajectories(csp, oracle_mod):
    """The same kilometre / minute regime on the multi-lane kernel (S = 40: chunk Schur complements,
    interface solve): it must stay as close to the long-double answer as the dense fp64 oracle does."""
    S = 40
    wp, tm = synth.make_batch(6, S, config_id=42)
    wp = wp * 2000.0
    tm = tm * 120.0
    r = csp.solve_batch(wp, tm, order=4)
    assert r.kernel.startswith("chunked_o4_f64"), r.kernel
    rc = csp.solve_batch(wp, tm, order=4, span=True)
    assert rc.kernel.startswith("span_o4_f64"), rc.kernel
    g = csp.solve_batch(wp, tm, order=4, force_generic=True)
    z = np.zeros((2, 3))
    e_gpu = e_gen = e_dense = 0.0
    e_chk = max(float(np.max(np.abs(rc.coeffs[b] - r.coeffs[b]) / np.max(np.abs(r.coeffs[b]), axis=(0, 1)))) for b in range(6))
    assert e_chk < 1e-9
    for b in range(6):
        ld, _ = oracle_mod.solve(4, wp[b], z, z, tm[b], long_double=True)
        dn, _ = oracle_mod.solve(4, wp[b], z, z, tm[b])
        ld = ld.reshape(S, 3, 8)
        den = np.max(np.abs(ld), axis=(0, 1))
        e_gpu = max(e_gpu, float(np.max(np.abs(r.coeffs[b] - ld) / den)))
        e_gen = max(e_gen, float(np.max(np.abs(g.coeffs[b] - ld) / den)))
        e_dense = max(e_dense, float(np.max(np.abs(dn.reshape(S, 3, 8) - ld) / den)))
    print("large scales, S=40: chunked-vs-ld %.2e  generic-vs-ld %.2e  dense-fp64-vs-ld %.2e" % (e_gpu, e_gen, e_dense))
    assert e_gpu < NORTH_STAR_TOL
    assert e_gpu <= 10 * max(e_dense, e_gen) + 1e-12


def test_device_memory_path_matches_host_path(csp):
    import torch
    wp, tm = synth.make_batch(1000, 16, config_id=3)
    host = csp.solve_batch(wp, tm, order=4)
    dev = csp.solve_batch(torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda(), order=4)
    torch.cuda.synchronize()
    assert np.array_equal(host.coeffs, dev.coeffs.cpu().numpy())


@pytest.mark.parametrize("B", [64, 200, 1])
def test_segment_major_layout(csp, B):
    """CSP_FLAG_SEGMENT_MAJOR is a pure permutation of the default layout (both kernels, ragged tail)."""
    wp, tm = synth.make_batch(B, 16, config_id=3)
    for force in (False, True):
        a = csp.solve_batch(wp, tm, order=4, force_generic=force).coeffs
        b = csp.solve_batch(wp, tm, order=4, force_generic=force, segment_major=True).coeffs
        assert b.shape == (16, B, 3, 8)
        assert np.array_equal(np.transpose(b, (1, 0, 2, 3)), a)


def _per_class_reference(csp, trajs, dtype, f32_arith=False):
    """The mixed batch solved the round-2 way -- one ragged csp_minsnap_solve_batch call per (order, length class), bucketed in
    Python; the class of a trajectory is the number of lanes the workspace-free kernel gives it (4 segments per lane, rounded up
    to a power of two), which is exactly the key csp_minsnap_solve_mixed buckets by on the device."""
    out = [None] * len(trajs)
    cls = lambda n: int(np.ceil(np.log2(max(n, 4) / 4.0)))
    for order in sorted({t[0] for t in trajs}):
        for c in sorted({cls(len(t[2])) for t in trajs if t[0] == order}):
            sub = [i for i, t in enumerate(trajs) if t[0] == order and cls(len(t[2])) == c]
            wp = np.concatenate([np.asarray(trajs[i][1]) for i in sub]).astype(dtype)
            tm = np.concatenate([np.asarray(trajs[i][2]) for i in sub]).astype(dtype)
            lens = np.array([len(trajs[i][2]) for i in sub])
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            r = csp.solve_batch(wp, tm, order=order, seg_offsets=off, max_segments=int(lens.max()), f32_arith=f32_arith)
            assert r.kernel.startswith(("chunked_o", "generic_o")) and r.kernel.endswith("_ragged"), r.kernel
            for j, i in enumerate(sub):
                out[i] = r.coeffs[off[j]:off[j + 1]]
    return out


def test_f32_storage_mixed_batch_c5(csp, oracle_mod):
    """BASELINE config C5 shape: S ~ U{4..64}, order ~ U{3,4,5}, fp32 storage, through csp_minsnap_solve_mixed (host-memory
    form here): device-side bucketing, coefficients in the caller's order.  Agrees with one ragged solve_batch call per
    (order, length class) to the final fp32 rounding.  The reference defines no fp32 behaviour (parity unpinned, builder-defined gates): with the
    default fp64 arithmetic the only loss is the final rounding to fp32 (gate 1e-6 relative to the 80-bit oracle run on the
    same fp32-rounded inputs); pure fp32 arithmetic (CSP_FLAG_F32_ARITH, solve_batch only -- the mixed entry does not offer
    it) is gated at 1e-3 / 5e-3 / 1e-1 for order 3 / 4 / 5 and its measured error is printed."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("csp_mixed", os.path.join(os.path.dirname(csp.__file__), "mixed.py"))
    mixed = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mixed)
    trajs = synth.make_ragged(240)
    got, status = mixed.solve_mixed(trajs, dtype=np.float32)
    assert not status.any()
    # against one ragged solve_batch call per (order, power-of-two length class): the same kernel body, but the mixed entry gives
    # a trajectory exactly ceil(S / 4) lanes where the one-order call rounds up to a power of two -- a different chunking of
    # the same elimination, so the fp32 results agree to the final rounding, not bit for bit
    ref_calls = _per_class_reference(csp, trajs, np.float32)
    for i, (c, r) in enumerate(zip(got, ref_calls)):
        assert c.dtype == np.float32
        synth.parity_gate(c, r, 1e-6, ("mixed entry vs per-class calls, fp32 storage", i))
    got32 = _per_class_reference(csp, trajs, np.float32, f32_arith=True)
    worst = {3: 0.0, 4: 0.0, 5: 0.0}
    worst32 = {3: 0.0, 4: 0.0, 5: 0.0}
    z = np.zeros((2, 3))
    for (o, w, t), c, c32 in zip(trajs, got, got32):
        w32, t32 = w.astype(np.float32).astype(np.float64), t.astype(np.float32).astype(np.float64)
        ref, _ = oracle_mod.solve(o, w32, z, z, t32, long_double=True)
        den = np.max(np.abs(ref))
        worst[o] = max(worst[o], float(np.max(np.abs(c.astype(np.float64).reshape(ref.shape) - ref)) / den))
        worst32[o] = max(worst32[o], float(np.max(np.abs(c32.astype(np.float64).reshape(ref.shape) - ref)) / den))
    print("C5 f32 storage / f64 arithmetic worst rel err per order:", worst)
    print("C5 f32 storage / f32 arithmetic worst rel err per order:", worst32)
    assert all(v < 1e-6 for v in worst.values()), worst
    assert worst32[3] < 1e-3 and worst32[4] < 5e-3 and worst32[5] < 1e-1, worst32


def test_ragged_batch(csp, oracle_mod):
    trajs = [t for t in synth.make_ragged(96, smin=1, smax=40) if t[0] == 4]
    wp = np.concatenate([t[1] for t in trajs])
    tm = np.concatenate([t[2] for t in trajs])
    off = np.concatenate([[0], np.cumsum([len(t[2]) for t in trajs])]).astype(np.int64)
    r = csp.solve_batch(wp, tm, order=4, seg_offsets=off, want_status=True)
    assert not r.status.any()
    for i, (o, w, t) in enumerate(trajs):
        ref, _ = oracle_mod.solve(4, w, np.zeros((2, 3)), np.zeros((2, 3)), t)
        got = r.coeffs[off[i]:off[i + 1]]
        synth.parity_gate(got, ref, TOL_WELL, ("ragged vs oracle", i))


@pytest.mark.parametrize("span", [False, True])
@pytest.mark.parametrize("order", [2, 3, 4, 5])
def test_chunked_kernel_ragged_and_long(csp, oracle_mod, order, span):
    """The workspace-free multi-lane kernels (ragged batches, S > 16, fp32 storage): whole batches against
    the generic kernel, sampled trajectories against the oracle.  Covers every lanes-per-trajectory
    bucket, chunk sizes 1..4 (chunked kernel) and span sizes 1..16 (span kernel: beyond 256 segments, or from
    17 segments with CSP_FLAG_SPAN), per-trajectory boundary conditions and weights."""
    import torch
    rng = np.random.default_rng(200 + order)
    tol_g = TOL_KERNELS_O5_LONG if order == 5 else TOL_KERNELS
    classes = ((1, 4, 300), (1, 9, 257), (3, 16, 200), (5, 33, 150), (17, 64, 130), (60, 130, 40), (200, 256, 9))
    if span:
        classes = classes[3:] + ((1, 40, 333),)
    classes = classes + ((300, 700, 6),)
    for smin, smax, B in classes:
        S_b = rng.integers(smin, smax + 1, size=B)
        S_b[0], S_b[-1] = smin, smax
        off = np.concatenate([[0], np.cumsum(S_b)]).astype(np.int64)
        wps, tms = [], []
        for i, S in enumerate(S_b):
            p0 = rng.uniform(-10.0, 10.0, size=(1, 3))
            wps.append(np.concatenate([p0, p0 + np.cumsum(rng.normal(size=(int(S), 3)), axis=0)]))
            tms.append(rng.uniform(0.5, 2.0, size=int(S)))
        wp, tm = np.concatenate(wps), np.concatenate(tms)
        bc = rng.normal(size=(B, 4, 3))
        vw = rng.uniform(0.0, 0.3, size=B)
        d = [torch.from_numpy(x).cuda() for x in (wp, tm, bc, vw)]
        d_off = torch.from_numpy(off).cuda()
        kw = dict(order=order, seg_offsets=d_off, max_segments=int(smax), vel_zero_weight_per_traj=d[3], want_status=True, want_max_dev=True)
        r = csp.solve_batch(d[0], d[1], d[2], span=span, **kw)
        fam = "span" if (smax > 256 or (smax > 16 and span)) else "chunked"
        assert r.kernel.startswith("%s_o%d_f64_l" % (fam, order)) and r.kernel.endswith("_ragged"), r.kernel
        g = csp.solve_batch(d[0], d[1], d[2], force_generic=True, **kw)
        torch.cuda.synchronize()
        assert not r.status.cpu().numpy().any(), (order, smax)
        assert not r.max_dev.cpu().numpy().any()
        a, gg = r.coeffs.cpu().numpy(), g.coeffs.cpu().numpy()
        for i in range(B):
            synth.parity_gate(a[off[i]:off[i + 1]], gg[off[i]:off[i + 1]], tol_g, ("chunked/span vs generic", order, span, smax, i, int(S_b[i])))
        for i in sorted({0, B // 2, B - 1}):
            if (S_b[i] > 64 and order == 5) or S_b[i] > 260:
                continue   # the dense oracle itself is ill-conditioned / too slow there
            ref, _ = oracle_mod.solve(order, wps[i], bc[i, [0, 1]], bc[i, [2, 3]], tms[i], 0.0, float(vw[i]), long_double=order == 5)
            # order 5 beyond 16 segments: cond(R_PP) ~ 1e8 and the t^9 coefficients are tiny: 1.6e-6 per power measured at S = 17
            # (1.1e-7 norm-wise) against the 80-bit oracle; gate 5e-6 there, the north-star 1e-6 otherwise
            tol_o = (5e-6 if S_b[i] > 16 else NORTH_STAR_TOL) if order == 5 else TOL_WELL
            synth.parity_gate(a[off[i]:off[i + 1]], ref, tol_o, ("chunked/span vs oracle", order, span, smax, i))
    # uniform long trajectories, batch-wide boundary conditions, host-memory entry, fp64 and fp32 storage
    for S in (17, 32, 100):
        wp, tm = synth.make_batch(77, S, config_id=400 + order)
        bc1 = rng.normal(size=(1, 4, 3))
        r = csp.solve_batch(wp, tm, bc1, order=order, vel_zero_weight=0.05, want_status=True, span=span)
        assert r.kernel == ("chunked_o%d_f64_l%d" % (order, 8 if S <= 32 else 32) if not span else
                            "span_o%d_f64_l%d" % (order, 2 if S <= 32 else 8)), r.kernel
        g = csp.solve_batch(wp, tm, bc1, order=order, vel_zero_weight=0.05, force_generic=True)
        assert not r.status.any()
        synth.parity_gate(r.coeffs, g.coeffs, tol_g, ("chunked/span uniform vs generic", order, span, S))
        r32 = csp.solve_batch(wp.astype(np.float32), tm.astype(np.float32), bc1.astype(np.float32), order=order, vel_zero_weight=0.05,
                              span=span)
        g32 = csp.solve_batch(wp.astype(np.float32), tm.astype(np.float32), bc1.astype(np.float32), order=order, vel_zero_weight=0.05, force_generic=True)
        assert r32.kernel.startswith("%s_o%d_f32io_f64_l" % ("span" if span else "chunked", order)), r32.kernel
        assert r32.coeffs.dtype == np.float32
        synth.parity_gate(r32.coeffs, g32.coeffs, 1e-6, ("chunked/span f32 storage vs generic", order, span, S))


def test_chunked_kernel_status(csp):
    wp, tm = synth.make_batch(70, 40, config_id=3)
    tm[5, 3] = 0.0
    tm[33, 39] = float("nan")
    for span in (False, True):
        r = csp.solve_batch(wp, tm, order=4, want_status=True, span=span)
        assert r.kernel.startswith("span_" if span else "chunked_"), r.kernel
        assert np.flatnonzero(r.status).tolist() == [5, 33]
        good = np.setdiff1d(np.arange(70), [5, 33])
        assert np.isfinite(r.coeffs[good]).all()


def test_full_size_properties(csp):
    """BASELINE C3 size (B=65536, S=16): size-independent properties instead of the oracle."""
    import torch
    B, S, o = 65536, 16, 4
    wp, tm = synth.make_batch(B, S, config_id=3)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    r = csp.solve_batch(d_wp, d_tm, order=o, want_status=True)
    c = r.coeffs.cpu().numpy()  # [B,S,3,8]
    assert not r.status.cpu().numpy().any()
    T = tm[:, :, None]
    pw = np.arange(7, -1, -1)

    def deriv_at(coef, t, j):
        # j-th derivative of sum_i coef_i t^pw_i
        fac = np.array([np.prod(np.arange(p, p - j, -1)) if p >= j else 0.0 for p in pw])
        e = np.clip(pw - j, 0, None)
        return np.sum(coef * fac * t[..., None] ** e, axis=-1)

    # (1) interpolation: p_k(0) = waypoint k, p_k(T_k) = waypoint k+1
    p0 = c[..., 7]
    assert np.max(np.abs(p0 - wp[:, :-1, :])) == 0.0
    pT = deriv_at(c, np.broadcast_to(T, c.shape[:3]), 0)
    scale = np.max(np.abs(wp))
    assert np.max(np.abs(pT - wp[:, 1:, :])) < 1e-9 * scale
    # (2) K2 continuity: derivatives 1..6 continuous at interior waypoints (natural-spline property)
    for j in range(1, 7):
        end = deriv_at(c[:, :-1], np.broadcast_to(T[:, :-1], c[:, :-1].shape[:3]), j)
        start = deriv_at(c[:, 1:], np.zeros(c[:, 1:].shape[:3]), j)
        mag = np.maximum(np.max(np.abs(start)), 1.0)
        assert np.max(np.abs(end - start)) < (1e-9 if j <= 3 else 1e-5) * mag, j
    # (3) zero boundary velocity/acceleration/jerk
    for j in (1, 2, 3):
        assert np.max(np.abs(deriv_at(c[:, 0], np.zeros((B, 3)), j))) < 1e-12
        assert np.max(np.abs(deriv_at(c[:, -1], np.broadcast_to(T[:, -1], (B, 3)), j))) < 1e-7
    # (4) K6 axis permutation / K5 time reversal on the whole batch
    perm = torch.from_numpy(np.ascontiguousarray(wp[:, :, [2, 0, 1]])).cuda()
    rp = csp.solve_batch(perm, d_tm, order=o).coeffs.cpu().numpy()
    assert np.array_equal(rp, c[:, :, [2, 0, 1], :])
    rev = csp.solve_batch(torch.from_numpy(np.ascontiguousarray(wp[:, ::-1])).cuda(),
                          torch.from_numpy(np.ascontiguousarray(tm[:, ::-1])).cuda(), order=o).coeffs.cpu().numpy()
    # q(t) = p(T - t): compare values at mid-segment
    mid = 0.5 * np.broadcast_to(T, c.shape[:3])
    v_fwd = deriv_at(c, mid, 0)
    v_rev = deriv_at(rev[:, ::-1], mid, 0)
    assert np.max(np.abs(v_fwd - v_rev)) < 1e-8 * scale
    # (5) linearity in the waypoints: solve(a*P + shift) = a*solve(P) + shift on the constant term
    r2 = csp.solve_batch(torch.from_numpy(2.0 * wp + 3.0).cuda(), d_tm, order=o).coeffs.cpu().numpy()
    exp = 2.0 * c
    exp[..., 7] += 3.0
    assert np.max(np.abs(r2 - exp)) < 1e-9 * np.max(np.abs(exp))


@pytest.mark.parametrize("span", [False, True])
@pytest.mark.parametrize("order,S,B", [(4, 64, 16384), (3, 37, 20000), (4, 200, 2048)])
def test_long_trajectory_properties_at_scale(csp, order, S, B, span):
    """The chunked kernel on batches of thousands of waves: interpolation, continuity of the 2(o-1)
    derivatives the optimum leaves continuous at interior waypoints (chunk interfaces included),
    boundary conditions, axis permutation -- no oracle involved."""
    import torch
    m = 2 * order
    wp, tm = synth.make_batch(B, S, config_id=11)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    r = csp.solve_batch(d_wp, d_tm, order=order, want_status=True, span=span)
    assert r.kernel.startswith("%s_o%d_f64" % ("span" if span else "chunked", order)), r.kernel
    c = r.coeffs.cpu().numpy()
    assert not r.status.cpu().numpy().any()
    T = tm[:, :, None]
    pw = np.arange(m - 1, -1, -1)

    def deriv_at(coef, t, j):
        fac = np.array([np.prod(np.arange(p, p - j, -1)) if p >= j else 0.0 for p in pw])
        e = np.clip(pw - j, 0, None)
        return np.sum(coef * fac * t[..., None] ** e, axis=-1)

    assert np.max(np.abs(c[..., m - 1] - wp[:, :-1, :])) == 0.0
    scale = np.max(np.abs(wp))
    assert np.max(np.abs(deriv_at(c, np.broadcast_to(T, c.shape[:3]), 0) - wp[:, 1:, :])) < 1e-9 * scale
    for j in range(1, 2 * order - 1):
        end = deriv_at(c[:, :-1], np.broadcast_to(T[:, :-1], c[:, :-1].shape[:3]), j)
        start = deriv_at(c[:, 1:], np.zeros(c[:, 1:].shape[:3]), j)
        mag = np.maximum(np.max(np.abs(start)), 1.0)
        assert np.max(np.abs(end - start)) < (1e-9 if j < order else 1e-5) * mag, j
    for j in range(1, order):
        assert np.max(np.abs(deriv_at(c[:, 0], np.zeros((B, 3)), j))) < 1e-12
        assert np.max(np.abs(deriv_at(c[:, -1], np.broadcast_to(T[:, -1], (B, 3)), j))) < 1e-7
    perm = torch.from_numpy(np.ascontiguousarray(wp[:, :, [2, 0, 1]])).cuda()
    rp = csp.solve_batch(perm, d_tm, order=order, span=span).coeffs.cpu().numpy()
    assert np.array_equal(rp, c[:, :, [2, 0, 1], :])


def test_single_process_sharded_entry(csp):
    """csp_minsnap_solve_batch_sharded (one process, contiguous chunks over ngpu devices): with the
    devices this box has it must reproduce the single-device call bit for bit -- uniform and ragged
    batches, per-trajectory boundary conditions and weights -- and reject more devices than exist."""
    rng = np.random.default_rng(7)
    n = csp.device_count()
    wp, tm = synth.make_batch(1000, 16, config_id=3)
    bc = rng.normal(size=(1000, 4, 3))
    vw = rng.uniform(0.0, 0.2, size=1000)
    kw = dict(order=4, vel_zero_weight_per_traj=vw, want_status=True, want_max_dev=True)
    a = csp.solve_batch(wp, tm, bc, **kw)
    for g in sorted({1, n, 0}):
        b = csp.solve_batch(wp, tm, bc, ngpu=g, **kw)
        assert np.array_equal(a.coeffs, b.coeffs) and np.array_equal(a.status, b.status) and np.array_equal(a.max_dev, b.max_dev)
    trajs = [t for t in synth.make_ragged(120, smin=1, smax=40) if t[0] == 3]
    rwp = np.concatenate([t[1] for t in trajs])
    rtm = np.concatenate([t[2] for t in trajs])
    off = np.concatenate([[0], np.cumsum([len(t[2]) for t in trajs])]).astype(np.int64)
    a = csp.solve_batch(rwp, rtm, order=3, seg_offsets=off)
    b = csp.solve_batch(rwp, rtm, order=3, seg_offsets=off, ngpu=n)
    assert np.array_equal(a.coeffs, b.coeffs)
    with pytest.raises(csp.CspError) as e:
        csp.solve_batch(wp, tm, bc, ngpu=n + 1, **kw)
    assert e.value.code == -1


def test_time_alloc(csp, oracle_mod):
    wp, _ = synth.make_batch(300, 16, config_id=3)
    for (v, mt) in [(5.0, 0.1), (200.0, 1.0), (0.0, 0.7)]:
        got = csp.time_alloc_batch(wp, v, mt)
        ref = np.stack([oracle_mod.time_alloc(w, v, mt) for w in wp])
        assert np.max(np.abs(got - ref)) <= 4e-16 * np.max(ref)


def test_error_codes(csp):
    wp, tm = synth.make_batch(4, 4)
    with pytest.raises(csp.CspError) as e:
        csp.solve_batch(wp, tm, order=6)
    assert e.value.code == -2
    with pytest.raises(csp.CspError) as e:
        csp.solve_batch(wp, tm, order=0)
    assert e.value.code == -1
