"""Round-2 GPU parity tests, all through the C-ABI:
  * BASELINE config C4 at its single-GPU shard size AND at its full size on one GPU (B = 65536 x 8 = 524288,
    S = 16, order 4): size-independent properties on the device, an oracle subsample spread over the whole
    batch, the one-process sharded entry bit-equal with the plain call;
  * the committed round-2 fixtures: F4 (Minisnap_EN / _3D marshalling), F5b (S = 64), F7 (near-ties of the t*
    arg-max and of the thinning test);
  * degenerate segment times (0, denormal, tiny negative) in the samplers: the call returns."""
import numpy as np
import pytest

from tests import synth
from tests.conftest import load_cases
from tests.test_golden_r2 import argmax_cases, load_f4, load_f7

pytestmark = pytest.mark.gpu
NORTH_STAR_TOL = 1e-6


def _deriv_at(torch, c, t, j, order):
    """j-th derivative of sum_i c[..., i] t^(m-1-i) at t (broadcast over the leading dims), on the device."""
    m = 2 * order
    pw = torch.arange(m - 1, -1, -1, device=c.device, dtype=torch.float64)
    fac = torch.ones(m, device=c.device, dtype=torch.float64)
    for q in range(j):
        fac = fac * torch.clamp(pw - q, min=0.0)
    e = torch.clamp(pw - j, min=0.0)
    return torch.sum(c * fac * t[..., None] ** e, dim=-1)


def test_c4_full_size_on_one_gpu(csp, oracle_mod):
    """BASELINE C4: B = 524288 trajectories x 16 segments, order 4, fp64 -- the whole 8-GPU batch on ONE device
    (1.9 GB of inputs + outputs, 7x the Infinity Cache)."""
    import torch
    B, S, o = 524288, 16, 4
    m = 2 * o
    wp, tm = synth.make_batch(B, S, config_id=4)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    r = csp.solve_batch(d_wp, d_tm, order=o, want_status=True)
    torch.cuda.synchronize()
    assert r.kernel == "fixed_o4_s16_f64"
    assert int(r.status.abs().max()) == 0
    c = r.coeffs                                                 # [B,S,3,8] on the device
    T = d_tm[:, :, None].expand(B, S, 3)
    scale = float(d_wp.abs().max())
    # (1) interpolation: p_k(0) == waypoint k bit for bit, p_k(T_k) = waypoint k+1
    assert torch.equal(c[..., m - 1], d_wp[:, :-1, :])
    assert float((_deriv_at(torch, c, T, 0, o) - d_wp[:, 1:, :]).abs().max()) < 1e-9 * scale
    # (2) K2: derivatives 1..6 continuous at interior waypoints
    for j in range(1, 7):
        end = _deriv_at(torch, c[:, :-1], T[:, :-1], j, o)
        start = _deriv_at(torch, c[:, 1:], torch.zeros_like(T[:, 1:]), j, o)
        mag = max(float(start.abs().max()), 1.0)
        assert float((end - start).abs().max()) < (1e-9 if j <= 3 else 1e-5) * mag, j
        del end, start
    # (3) zero boundary velocity / acceleration / jerk
    for j in (1, 2, 3):
        assert float(_deriv_at(torch, c[:, 0], torch.zeros(B, 3, device="cuda", dtype=torch.float64), j, o).abs().max()) < 1e-12
        assert float(_deriv_at(torch, c[:, -1], T[:, -1], j, o).abs().max()) < 1e-7
    # (4) K6 axis permutation (bit for bit) and linearity in the waypoints
    rp = csp.solve_batch(d_wp[:, :, [2, 0, 1]].contiguous(), d_tm, order=o).coeffs
    assert torch.equal(rp, c[:, :, [2, 0, 1], :])
    del rp
    r2 = csp.solve_batch(2.0 * d_wp + 3.0, d_tm, order=o).coeffs
    exp = 2.0 * c
    exp[..., m - 1] += 3.0
    assert float((r2 - exp).abs().max()) < 1e-9 * float(exp.abs().max())
    del r2, exp
    # (5) 256 trajectories spread over the whole batch against the CPU oracle (fp64 dense and 80-bit)
    idx = np.linspace(0, B - 1, 256).astype(np.int64)
    ref, _ = oracle_mod.solve_batch(o, wp[idx], tm[idx], nthreads=oracle_mod.max_threads())
    ld, _ = oracle_mod.solve_batch(o, wp[idx], tm[idx], nthreads=oracle_mod.max_threads(), long_double=True)
    got = c[torch.from_numpy(idx).cuda()].cpu().numpy()
    e_ref, e_ld = synth.rel_err_per_power(got, ref), synth.rel_err_per_power(got, ld)
    print("C4 subsample: per-power rel err vs fp64 dense oracle %.2e, vs long double %.2e (norm-wise %.2e)" %
          (e_ref, e_ld, synth.rel_err(got, ref)))
    assert e_ref < NORTH_STAR_TOL and e_ld < 1e-8
    # (6) the one-process sharded entry (host memory, every device of this box) reproduces the plain call bit for bit
    full = c.cpu().numpy()
    n = csp.device_count()
    sh = csp.solve_batch(wp, tm, order=o, ngpu=n)
    assert np.array_equal(sh.coeffs, full)
    # (7) a checksum of checksums over the shards C4 would use (65536 per GPU): each shard solved on its own equals its
    # slice of the full batch -- what the 8-rank run computes, rank by rank
    for g in (0, 3, 7):
        lo = g * 65536
        part = csp.solve_batch(d_wp[lo:lo + 65536].contiguous(), d_tm[lo:lo + 65536].contiguous(), order=o).coeffs
        assert torch.equal(part, c[lo:lo + 65536])


def test_path_kernel_beyond_the_infinity_cache(csp, oracle_mod):
    """Order 4 with both penalties at B = 131072 x 16 segments (403 MB of coefficients): the launch takes non-temporal
    coefficient stores (launch_path_s); its two halves solved on their own (201 MB each: ordinary stores) must give the
    same bits -- coefficients, max_dev and status -- and a subsample spread over the batch agrees with the oracle.
    Order 2 (the shipped yaml; ordinary stores at every size) the same way."""
    import torch
    for o, tol in ((4, NORTH_STAR_TOL), (2, NORTH_STAR_TOL)):
        B, S, H = 131072, 16, 65536
        wp, tm = synth.make_batch(B, S, config_id=40 + o)
        d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
        kw = dict(order=o, path_weight=1e-7, vel_zero_weight=0.01, want_status=True, want_max_dev=True)
        r = csp.solve_batch(d_wp, d_tm, **kw)
        torch.cuda.synchronize()
        assert r.kernel == "fixedpath_o%d_s16_f64" % o, r.kernel
        assert int(r.status.abs().max()) == 0
        for lo in (0, H):
            part = csp.solve_batch(d_wp[lo:lo + H].contiguous(), d_tm[lo:lo + H].contiguous(), **kw)
            assert torch.equal(part.coeffs, r.coeffs[lo:lo + H]), (o, lo)
            assert torch.equal(part.max_dev, r.max_dev[lo:lo + H]), (o, lo)
            del part
        idx = np.linspace(0, B - 1, 96).astype(np.int64)
        ref, dev = oracle_mod.solve_batch(o, wp[idx], tm[idx], path_weight=1e-7, vel_zero_weight=0.01, nthreads=oracle_mod.max_threads())
        got = r.coeffs[torch.from_numpy(idx).cuda()].cpu().numpy()
        e = synth.rel_err_per_power(got, ref)
        print("path kernel order %d, B = %d: per-power rel err vs the oracle %.2e" % (o, B, e))
        assert e < tol, (o, e)
        del r, d_wp, d_tm
        torch.cuda.empty_cache()


def test_f5b_s64_fixtures(csp, oracle_mod):
    for c in load_cases("F5b_ragged_s64.json"):
        o, S = c["order"], c["segments"]
        ld, _ = oracle_mod.solve(o, c["path"], c["vel"], c["acc"], c["time"], long_double=True)
        ld = ld.reshape(S, 3, 2 * o)
        for force in (False, True):
            r = csp.solve_batch(c["path"][None], c["time"][None], c["bc"][None], order=o, force_generic=force, want_status=True)
            e_fix, e_ld = synth.rel_err_per_power(r.coeffs[0], c["coeff"]), synth.rel_err_per_power(r.coeffs[0], ld)
            print("%s %s: per-power vs fixture %.2e, vs long double %.2e (fixture vs long double %.2e)" %
                  (c["name"], r.kernel, e_fix, e_ld, synth.rel_err_per_power(c["coeff"], ld)))
            assert int(r.status[0]) == 0
            assert e_ld < NORTH_STAR_TOL, (c["name"], r.kernel, e_ld)
            assert e_fix < max(NORTH_STAR_TOL, 1e-13 * c["cond_M"]), (c["name"], r.kernel, e_fix)


@pytest.mark.parametrize("force_generic", [False, True])
def test_f7_argmax_near_ties(csp, force_generic):
    """The strict `>` of minimum_snap.cpp:435 on mirror samples 4 / 12 whose squared distances differ by a relative
    1e-8 .. 1e-10: the HIP pre-solve (Hermite-basis evaluation, FMA) must fall the way the oracle falls.  A flipped
    decision moves the coefficients by 2e-4 per power (tests/test_golden_r2.py), the gate is 1e-7."""
    kernels = set()
    for c in argmax_cases():
        r = csp.solve_batch(c["path"][None], c["time"][None], c["bc"][None], order=c["order"], path_weight=c["path_weight"],
                            vel_zero_weight=c["vel_zero_weight"], want_max_dev=True, force_generic=force_generic)
        kernels.add(r.kernel)
        e = synth.rel_err_per_power(r.coeffs[0], c["coeff"])
        assert e < 1e-7, (c["name"], r.kernel, e)
        assert abs(r.max_dev[0] - c["max_dev"]) < 1e-8 * max(1.0, c["max_dev"]), (c["name"], r.max_dev[0], c["max_dev"])
    assert kernels == ({"generic_o4_f64"} if force_generic else {"fixedpath_o4_s3_f64"}), kernels


@pytest.mark.parametrize("sampler", ["segment", "one_lane", "wave"])
def test_f7_thinning_near_ties(csp, sampler):
    """`dist >= sample_distance` (minimum_snap.cpp:145) with sample_distance exactly ON a candidate's distance
    (exactly representable data: both sides compute the same double) and a relative 1e-10 below / above it."""
    import torch
    for c in load_f7():
        cfg = c["config"]
        P = c["waypoints"]
        plan = csp.plan_batch(P[None], cfg["V_avg"], cfg["min_time_s"], order=cfg["order"])
        assert not plan.status.any()
        t, co = torch.from_numpy(plan.times).cuda(), torch.from_numpy(plan.coeffs).cuda()
        s, n, _ = csp.sample_batch(t, co, cfg["sample_distance"], 4096, one_lane=sampler == "one_lane", long_segments=sampler == "wave")
        torch.cuda.synchronize()
        n = int(n[0])
        assert n == c["n_samples"], (c["name"], sampler, n, c["n_samples"])
        got = s[0, :n].cpu().numpy()
        if c["kind"] == "exact":
            assert np.array_equal(got, c["samples"]), (c["name"], sampler)
        else:
            assert np.max(np.abs(got - c["samples"])) < 1e-9 * np.max(np.abs(c["samples"])), (c["name"], sampler)


def test_f4_marshalling_fixture_through_the_c_abi(csp):
    """F4 through csp_minsnap_plan_batch + csp_minsnap_sample_batch, marshalled the way Minisnap_EN / Minisnap_3D do
    (uavPathPlanning.cpp:4401-4474); tests/test_gpu_host_shim.py runs the same fixture through the C++ class shim."""
    for c in load_f4():
        P, cfg = c["waypoints_enu"], c["effective"]
        route = P.copy()
        if c["mode"] == "en":
            route[:, 2] = 0.0
        plan = csp.plan_batch(route[None], cfg["V_avg"], cfg["min_time_s"], order=cfg["order"],
                              path_weight=cfg.get("path_weight", 0.0), vel_zero_weight=cfg.get("vel_zero_weight", 0.0))
        assert plan.iterations[0] == c["iterations"], c["name"]
        assert abs(plan.vel_zero_weight[0] - float.fromhex(c["vel_zero_weight_final"])) <= 1e-15
        cap = c["n_result"] + 16
        s, n, stats = csp.sample_batch(plan.times, plan.coeffs, cfg["sample_distance"], cap)
        n = int(n[0])
        got = s[0, :n].copy()
        if c["mode"] == "en":
            got[:, 2] = P[0, 2]
        assert n == c["n_result"], (c["name"], n, c["n_result"])
        assert np.max(np.abs(got - c["result_enu"])) < 1e-6 * np.max(np.abs(c["result_enu"])), c["name"]
        mc = float.fromhex(c["max_climb_rate"])
        assert abs(stats[0, 0] - mc) < 1e-6 * max(1.0, mc), c["name"]


@pytest.mark.parametrize("sampler", ["segment", "one_lane", "wave"])
def test_degenerate_segment_times_do_not_hang_the_samplers(csp, sampler):
    """T = 0 (two coincident waypoints with min_time_s = 0), a denormal T and T in (-1e-12, 0) make the reference's
    candidate loop `for (t = dt; t <= T + 1e-12; t += dt)` spin forever (dt = T/10); on the device that would be a
    hang.  Such segments get no candidates (minsnap_plan.hip t_end); the solve flags the trajectory."""
    import torch
    B, S, o = 8, 4, 3
    wp, tm = synth.make_batch(B, S, config_id=31)
    tm = tm.copy()
    tm[1, 2] = 0.0
    tm[2, 0] = 1e-300
    tm[3, 3] = -1e-13
    tm[4, 1] = 5e-324
    tm[5, 2] = float("nan")
    tm[6, 0] = 1e-13        # legal: 110 candidates
    r = csp.solve_batch(wp, tm, order=o, want_status=True)
    assert r.status[0] == 0 and all(r.status[b] != 0 for b in (1, 2, 4, 5)), r.status
    co = np.nan_to_num(r.coeffs, nan=0.0, posinf=0.0, neginf=0.0)   # sample whatever the solve produced, finite
    s, n, _ = csp.sample_batch(torch.from_numpy(tm).cuda(), torch.from_numpy(co).cuda(), 0.5, 512,
                               one_lane=sampler == "one_lane", long_segments=sampler == "wave")
    torch.cuda.synchronize()
    n = n.cpu().numpy()
    assert (n >= 1).all() and (n <= 512).all(), n
    # the healthy trajectories are untouched by their neighbours' bad times
    s_ok, n_ok, _ = csp.sample_batch(torch.from_numpy(tm[:1]).cuda(), torch.from_numpy(r.coeffs[:1]).cuda(), 0.5, 512)
    assert int(n_ok[0]) == n[0] and torch.equal(s_ok[0, :n[0]], s[0, :n[0]])
    # plan_batch with min_time_s = 0 and coincident waypoints produces T = 0 itself
    wp2 = wp.copy()
    wp2[0, 2] = wp2[0, 1]
    plan = csp.plan_batch(wp2, 5.0, 0.0, order=o)
    assert plan.times[0, 1] == 0.0 and plan.status[0] != 0
    co2 = np.nan_to_num(plan.coeffs, nan=0.0, posinf=0.0, neginf=0.0)
    s2, n2, _ = csp.sample_batch(plan.times, co2, 0.5, 512)
    assert (n2 >= 1).all()


def test_host_memory_calls_reuse_a_cached_arena(csp, oracle_mod):
    """CSP_MEM_HOST calls stage through a per-device arena that survives the call (minsnap_hoststage.h): results must
    not depend on what the arena held before -- growing, shrinking, interleaved entry points, small (one pinned copy)
    and large (streamed through the pinned halves, > 16 MB) transfers."""
    import torch
    wp_big, tm_big = synth.make_batch(30000, 16, config_id=3)          # 108 MB of coefficients: streamed path
    dev = csp.solve_batch(torch.from_numpy(wp_big).cuda(), torch.from_numpy(tm_big).cuda(), order=4).coeffs.cpu().numpy()
    one = csp.solve_batch(wp_big[:1], tm_big[:1], order=4, want_status=True, want_max_dev=True)   # small first: arena grows later
    assert np.array_equal(one.coeffs, dev[:1])
    big = csp.solve_batch(wp_big, tm_big, order=4, want_status=True)
    assert np.array_equal(big.coeffs, dev) and not big.status.any()
    for n in (1, 7, 4096, 2, 30000, 1):                               # shrink and grow again
        r = csp.solve_batch(wp_big[:n], tm_big[:n], order=4)
        assert np.array_equal(r.coeffs, dev[:n]), n
        t = csp.time_alloc_batch(wp_big[:n], 5.0, 0.1)
        k = min(n, 16)
        assert np.allclose(t[:k], np.stack([oracle_mod.time_alloc(w, 5.0, 0.1) for w in wp_big[:k]]), rtol=0, atol=1e-15)
    csp.release_cached_memory()
    r = csp.solve_batch(wp_big[:3], tm_big[:3], order=4)
    assert np.array_equal(r.coeffs, dev[:3])
    # the plan's host form honours per-trajectory starting weights like the device form
    wp, _ = synth.make_batch(40, 6, config_id=21)
    wp4 = np.ascontiguousarray(wp * 4.0)
    vw0 = np.linspace(0.0, 0.3, 40)
    desc_kw = dict(order=3, path_weight=0.4)
    import ctypes
    def plan(host):
        B, S, m = 40, 6, 6
        if host:
            times, co = np.empty((B, S)), np.empty((B, S, 3, m))
            md, vw, it = np.empty(B), np.empty(B), np.empty(B, dtype=np.int32)
            d = csp.make_desc(3, B, S, csp.DTYPE_F64, 0.4, 0.0, csp.MEM_HOST, False, vw_per_ptr=vw0.ctypes.data)
            bc = np.zeros((1, 4, 3))
            rc = csp.raw_lib().csp_minsnap_plan_batch(ctypes.byref(d), wp4.ctypes.data, 5.0, 0.1, bc.ctypes.data, times.ctypes.data,
                                                     co.ctypes.data, md.ctypes.data, vw.ctypes.data, it.ctypes.data, None, None, 0, None)
            assert rc == 0
            return co, vw, it
        t_wp, t_vw = torch.from_numpy(wp4).cuda(), torch.from_numpy(vw0).cuda()
        times, co = torch.empty((B, S), dtype=torch.float64, device="cuda"), torch.empty((B, S, 3, m), dtype=torch.float64, device="cuda")
        md, vw = torch.empty(B, dtype=torch.float64, device="cuda"), torch.empty(B, dtype=torch.float64, device="cuda")
        it = torch.empty(B, dtype=torch.int32, device="cuda")
        bc = torch.zeros((1, 4, 3), dtype=torch.float64, device="cuda")
        d = csp.make_desc(3, B, S, csp.DTYPE_F64, 0.4, 0.0, csp.MEM_DEVICE, False, vw_per_ptr=t_vw.data_ptr(), device_id=0)
        need = int(csp.raw_lib().csp_minsnap_plan_workspace_bytes(ctypes.byref(d)))
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        rc = csp.raw_lib().csp_minsnap_plan_batch(ctypes.byref(d), t_wp.data_ptr(), 5.0, 0.1, bc.data_ptr(), times.data_ptr(), co.data_ptr(),
                                                 md.data_ptr(), vw.data_ptr(), it.data_ptr(), None, ws.data_ptr(), need,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        return co.cpu().numpy(), vw.cpu().numpy(), it.cpu().numpy()
    (c_h, v_h, i_h), (c_d, v_d, i_d) = plan(True), plan(False)
    assert np.array_equal(c_h, c_d) and np.array_equal(v_h, v_d) and np.array_equal(i_h, i_d)
    assert i_h.max() > 0 and (v_h >= vw0).all() and len(set(np.round(v_h, 6))) > 3     # the per-trajectory starts were used


def test_host_calls_from_several_threads(csp):
    """Concurrent CSP_MEM_HOST calls borrow distinct arenas from the device's pool."""
    import threading
    wp, tm = synth.make_batch(2048, 16, config_id=3)
    want = csp.solve_batch(wp, tm, order=4).coeffs
    errs = []

    def work(k):
        try:
            for i in range(6):
                lo = (k * 97 + i * 131) % 1500
                n = 1 + (k * 37 + i * 211) % 500
                r = csp.solve_batch(wp[lo:lo + n], tm[lo:lo + n], order=4)
                if not np.array_equal(r.coeffs, want[lo:lo + n]):
                    errs.append((k, i))
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_sharded_entry_pins_the_kernel_choice(csp):
    """An order-5 batch of 17+ segments picks the span kernel only when the WHOLE batch has >= 65536 span lanes
    (minsnap_capi.hip use_span); the sharded entry must make that choice once for all chunks.  On this one-GPU box
    the chunk IS the batch, so the check is that the sharded call reports / reproduces the plain call bit for bit
    at both sides of the threshold."""
    n = csp.device_count()
    for B in (300, 33000):                                            # 33000 x 2 span lanes (S = 32) >= 65536: span kernel
        wp, tm = synth.make_batch(B, 32, config_id=9)
        a = csp.solve_batch(wp, tm, order=5)
        b = csp.solve_batch(wp, tm, order=5, ngpu=n)
        assert a.kernel.startswith("span_o5" if B == 33000 else "chunked_o5"), a.kernel
        assert np.array_equal(a.coeffs, b.coeffs), B


@pytest.mark.parametrize("S", [2, 3, 8, 15, 16])
def test_axis_per_lane_mapping_of_small_batches(csp, oracle_mod, S):
    """Order-4 batches of up to 32 x CUs trajectories run in slices of 16 with THREE lanes per trajectory (one axis each,
    minsnap_fixed_impl.h NAX = 1).  Same arithmetic per axis as the one-lane-per-trajectory kernels: the small batch must
    equal, bit for bit, the same trajectories solved inside a large batch (persistent kernel), for ragged slice tails,
    per-trajectory boundary conditions and weights; plus the oracle and the status flags."""
    import torch
    rng = np.random.default_rng(S)
    big_B = 40000
    wp, tm = synth.make_batch(big_B, S, config_id=2)
    bc = rng.normal(size=(big_B, 4, 3))
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    big = csp.solve_batch(d_wp, d_tm, order=4).coeffs
    for B in (1, 5, 16, 17, 100, 4096):
        r = csp.solve_batch(d_wp[:B].contiguous(), d_tm[:B].contiguous(), order=4, want_status=True)
        assert r.kernel == "fixed_o4_s%d_f64" % S
        assert torch.equal(r.coeffs, big[:B]), (S, B)
        assert int(r.status.abs().max()) == 0
    B = 333
    vw = rng.uniform(0.0, 0.3, size=B)
    a = csp.solve_batch(wp[:B], tm[:B], bc[:B], order=4, vel_zero_weight_per_traj=vw, want_status=True)
    g = csp.solve_batch(wp[:B], tm[:B], bc[:B], order=4, vel_zero_weight_per_traj=vw, force_generic=True)
    assert synth.rel_err_per_power(a.coeffs, g.coeffs) < 1e-9 and not a.status.any()
    ref, _ = oracle_mod.solve_batch(4, wp[:64], tm[:64], bc[:64], vel_zero_weight=0.0, nthreads=oracle_mod.max_threads())
    r0 = csp.solve_batch(wp[:64], tm[:64], bc[:64], order=4)
    assert synth.rel_err_per_power(r0.coeffs, ref) < NORTH_STAR_TOL
    bad = tm[:40].copy()
    bad[7, :] = -bad[7, :]          # every block negative definite: not SPD
    bad[21, 0] = float("nan")
    rb = csp.solve_batch(wp[:40], bad, order=4, want_status=True)
    assert np.flatnonzero(rb.status).tolist() == [7, 21], rb.status
    ok = np.setdiff1d(np.arange(40), [7, 21])
    assert np.array_equal(rb.coeffs[ok], big[:40].cpu().numpy()[ok])


def test_c5_full_size_mixed_batch(csp, oracle_mod):
    """BASELINE C5 at the size bench.py times: B = 65536 mixed trajectories (S ~ U{4..64}, order ~ U{3,4,5}), fp32 storage /
    fp64 arithmetic, through csp_minsnap_solve_mixed (round 3: ONE C-ABI call, device-side bucketing by (order, length class),
    one persistent launch per order, coefficients in the CALLER'S order; cs-pathplan_amd/mixed.py::MixedBatch is a thin
    caller).  Size-independent properties on every trajectory (interpolation of the waypoints at both segment ends,
    continuity of velocity and acceleration at interior waypoints, zero boundary velocity / acceleration -- all to fp32
    resolution), the device-computed coefficient offsets against the host's, status clean, and 192 trajectories spread over
    the batch against the 80-bit oracle on the fp32-rounded inputs."""
    import importlib.util, os
    import torch
    spec = importlib.util.spec_from_file_location("csp_mixed", os.path.join(os.path.dirname(csp.__file__), "mixed.py"))
    mixed = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mixed)
    B = 65536
    trajs = synth.make_ragged(B)
    dev = torch.device("cuda", 0)
    orders, wp_h, tm_h, off_h = mixed.pack(trajs, np.float32)
    d_or, d_wp, d_tm, d_off = (torch.from_numpy(x).to(dev) for x in (orders, wp_h, tm_h, off_h))
    prep = csp.PreparedMixed(d_or, d_wp, d_tm, d_off, want_status=True)
    prep.out.fill_(float("nan"))      # every element must be written
    prep.run()
    prep.run()                        # a second call on the same workspace (counters and cursors are reset by the call)
    torch.cuda.synchronize()
    assert int(prep.status.abs().max()) == 0
    blk = csp.mixed_block_elements(orders, off_h, True)        # 6 * order * S rounded up to whole 16-byte pieces (fp32: multiples of 4)
    tight = np.diff(off_h) * 6 * orders.astype(np.int64)
    assert ((blk - tight) >= 0).all() and ((blk - tight) <= 2).all() and (blk % 4 == 0).all()
    host_off = np.concatenate([[0], np.cumsum(blk)])
    assert np.array_equal(prep.coeff_offsets.cpu().numpy(), host_off)
    out = prep.out
    pad_mask = torch.ones_like(out, dtype=torch.bool)          # every element but the padding must have been written
    pad_idx = np.concatenate([np.arange(host_off[i] + tight[i], host_off[i + 1]) for i in np.nonzero(blk != tight)[0]])
    pad_mask[torch.from_numpy(pad_idx).to(dev)] = False
    assert bool(torch.isfinite(out[pad_mask]).all()) and bool(torch.isnan(out[~pad_mask]).all())
    lens = d_off[1:] - d_off[:-1]
    seg_traj = torch.repeat_interleave(torch.arange(B, device=dev), lens)                 # trajectory of each segment
    seg_idx = torch.arange(int(d_off[-1]), device=dev)
    seg_local = seg_idx - d_off[seg_traj]
    seg_order = d_or.long()[seg_traj]
    seg_coef = prep.coeff_offsets[seg_traj] + seg_local * 6 * seg_order                   # element offset of each record
    first_all = torch.zeros_like(seg_idx, dtype=torch.bool)
    first_all[d_off[:-1]] = True
    last_all = torch.zeros_like(first_all)
    last_all[d_off[1:] - 1] = True
    for o in (3, 4, 5):
        m = 2 * o
        sel = seg_order == o
        g = seg_idx[sel]
        co = out[(seg_coef[sel][:, None] + torch.arange(3 * m, device=dev)[None, :])].reshape(-1, 3, m).double()
        wp = d_wp.double()
        p_start, p_end = wp[g + seg_traj[sel]], wp[g + seg_traj[sel] + 1]
        T = d_tm.double()[g][:, None].expand(-1, 3)
        first, last = first_all[sel], last_all[sel]
        # fp32 coefficients: a value of the polynomial is only as good as 6e-8 x the sum of the magnitudes of its terms
        # (they cancel), so every check is scaled by that sum
        def mag_at(t, j):
            return _deriv_at(torch, co.abs(), t, j, o)
        eps = 4e-7
        assert float((co[..., m - 1] - p_start).abs().max()) == 0.0                      # constant term = waypoint, bit for bit (fp32)
        assert bool(((_deriv_at(torch, co, T, 0, o) - p_end).abs() <= eps * mag_at(T, 0)).all())
        zero = torch.zeros_like(T)
        for j in (1, 2):
            end, start = _deriv_at(torch, co, T, j, o), _deriv_at(torch, co, zero, j, o)
            m_end, m_start = mag_at(T, j), mag_at(zero, j)
            inner = ~last[:-1]                                       # segment i and i+1 belong to the same trajectory
            gap = (end[:-1] - start[1:]).abs()
            assert bool((gap[inner] <= eps * (m_end[:-1] + m_start[1:])[inner] + 1e-30).all()), (o, j)
            assert float(start[first].abs().max()) == 0.0, (o, j)    # zero boundary velocity / acceleration at the start: exact
            assert bool((end[last].abs() <= eps * m_end[last]).all()), (o, j)
    idx = np.linspace(0, B - 1, 192).astype(np.int64)
    z = np.zeros((2, 3))
    worst = {3: 0.0, 4: 0.0, 5: 0.0}
    out_h = out.cpu().numpy()
    for i in idx:
        o, w, t = trajs[i]
        w32, t32 = w.astype(np.float32).astype(np.float64), t.astype(np.float32).astype(np.float64)
        ref, _ = oracle_mod.solve(o, w32, z, z, t32, long_double=True)
        got = out_h[host_off[i]:host_off[i] + tight[i]].astype(np.float64).reshape(ref.shape)
        worst[o] = max(worst[o], float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
    print("C5 full size, 192 of 65536 trajectories vs the 80-bit oracle (norm-wise, fp32 storage):", worst)
    assert all(v < 1e-6 for v in worst.values()), worst
    # MixedBatch (bench.py's object) is the same call
    mb = mixed.MixedBatch(csp, trajs[:5000], dev, dtype=torch.float32)
    mb.prep.out.zero_()               # the 2-float paddings after odd-order, odd-length blocks are never written
    mb.run()
    torch.cuda.synchronize()
    small = csp.PreparedMixed(d_or[:5000], d_wp[:int(off_h[5000]) + 5000], d_tm[:int(off_h[5000])], d_off[:5001])
    small.out.zero_()
    small.run()
    torch.cuda.synchronize()
    assert torch.equal(mb.prep.out, small.out)
    assert torch.equal(mb.coeffs(4999), small.out[int(host_off[4999]):int(host_off[4999] + tight[4999])].reshape(len(trajs[4999][2]), 3, 2 * trajs[4999][0]))


@pytest.mark.parametrize("S", [8, 11, 13, 14, 15, 16])
def test_order2_path_kernel_dense_residency(csp, S):
    """Order 2 with the path penalty, S >= 8: the variant that shares a CU between four workgroups (staging tiles on
    top of the LDS input image, pass B from a register copy of its inputs, staggered start of the first round).
    Same coefficients, max_dev, status and t* picks as the generic kernel -- with and without the status outputs
    (two kernel instantiations), for a partial last slice and for a grid of more than one resident round."""
    import torch
    rng = np.random.default_rng(500 + S)
    for B, pw, outputs in ((64 * 3 + 5, 0.3, True), (64 * 3 + 5, 0.3, False), (70000 + 13, 1e-7, False), (70000 + 13, 2.0, True)):
        wp, tm = synth.make_batch(B, S, config_id=300 + S)
        bc = rng.normal(size=(B, 4, 3))
        vw = rng.uniform(0.0, 0.3, size=B)
        d = [torch.from_numpy(x).cuda() for x in (wp, tm, bc, vw)]
        kw = dict(order=2, path_weight=pw, vel_zero_weight_per_traj=d[3], want_status=outputs, want_max_dev=outputs)
        r = csp.solve_batch(d[0], d[1], d[2], **kw)
        assert r.kernel == "fixedpath_o2_s%d_f64" % S, r.kernel
        g = csp.solve_batch(d[0], d[1], d[2], force_generic=True, **kw)
        torch.cuda.synchronize()
        num = (r.coeffs - g.coeffs).abs().reshape(B, -1).amax(dim=1)
        den = g.coeffs.abs().reshape(B, -1).amax(dim=1)
        assert float((num / den).max()) < 1e-8, (S, B, pw)
        if outputs:
            assert not bool(r.status.any()) and not bool(g.status.any())
            assert float((r.max_dev - g.max_dev).abs().max()) < 1e-8 * max(1.0, float(g.max_dev.max())), (S, B)


@pytest.mark.parametrize("order,S", [(2, 16), (2, 6), (3, 12), (4, 16), (4, 7), (5, 8)])
def test_status_flags_with_the_two_coefficient_test(csp, order, S):
    """The register-resident kernels test the highest-power and the constant coefficient of every record for NaN/Inf
    (every endpoint quantity of the segment enters the former): a zero or NaN segment time, a NaN or infinite
    waypoint and a NaN boundary condition are flagged on exactly the trajectories that hold them -- in the first,
    a middle and the last segment, with and without the path penalty -- like the generic kernel does."""
    B = 200
    wp, tm = synth.make_batch(B, S, config_id=40 + order)
    bc = np.zeros((B, 4, 3))
    tm[5, 0] = 0.0
    tm[17, S // 2] = float("nan")
    tm[64, S - 1] = 0.0
    wp[70, 0, 1] = float("nan")
    wp[99, S // 2, 2] = float("inf")
    wp[130, S, 0] = float("nan")
    bc[150, 0, 0] = float("nan")
    bc[199, 1, 2] = float("nan")
    bad = [5, 17, 64, 70, 99, 130, 150, 199]
    for pw in ((0.0, 0.3) if order <= 4 else (0.0,)):
        for force in (False, True):
            r = csp.solve_batch(wp, tm, bc, order=order, path_weight=pw, want_status=True, force_generic=force)
            if not force:
                assert r.kernel.startswith("fixedpath_" if pw else "fixed_"), r.kernel
            flagged = np.flatnonzero(r.status & 1).tolist()
            assert flagged == bad, (order, S, pw, r.kernel, flagged)
            good = np.setdiff1d(np.arange(B), bad)
            assert np.isfinite(r.coeffs[good]).all()


def test_large_host_batches_pageable_and_pinned(csp):
    """Host-memory batches beyond the 16 MB staging block: pageable caller memory streams through the pinned halves with
    the staging copies spread over helper threads; page-locked caller memory (here: torch pinned tensors viewed as
    numpy) is read and written by the DMA engine directly.  Both bit-equal with the device-memory call."""
    import torch
    B, S = 40000, 16
    wp, tm = synth.make_batch(B, S, config_id=77)
    ref = csp.solve_batch(torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda(), order=4).coeffs.cpu().numpy()
    got = csp.solve_batch(wp, tm, order=4).coeffs                       # pageable in, pageable out (123 MB)
    assert np.array_equal(got, ref)
    pwp, ptm = torch.from_numpy(wp).pin_memory(), torch.from_numpy(tm).pin_memory()
    pout = torch.empty((B, S, 3, 8), dtype=torch.float64).pin_memory()
    pout.fill_(-1.0)
    r = csp.solve_batch(pwp.numpy(), ptm.numpy(), order=4, out=pout.numpy())
    assert np.array_equal(r.coeffs, ref) and np.array_equal(pout.numpy(), ref)
    # mixed: pinned inputs, pageable output
    assert np.array_equal(csp.solve_batch(pwp.numpy(), ptm.numpy(), order=4).coeffs, ref)
