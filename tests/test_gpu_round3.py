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
    synth.parity_gate(a, gg, 1e-7 if order == 5 else 1e-9, ("ring stores vs generic", order, S))
    for s, ref in _old_path_rows(csp, torch, d_wp, d_tm, None, order, starts).items():
        assert np.array_equal(a[s:s + 63], ref), (order, S, s)
    idx = np.array([0, 63, 64, 4097, B - 1])
    ref, _ = oracle_mod.solve_batch(order, wp[idx], tm[idx], long_double=order == 5)
    synth.parity_gate(a[idx], ref, 1e-6 if order == 5 else 5e-8, ("ring stores vs oracle", order, S))
    # per-trajectory boundary conditions and weights -> one workgroup per slice (same fixed_body, FULL = true)
    Bs = 64 * 300
    bc = rng.normal(size=(Bs, 4, 3))
    vw = rng.uniform(0.0, 0.3, size=Bs)
    d_bc, d_vw = torch.from_numpy(bc).cuda(), torch.from_numpy(vw).cuda()
    r = csp.solve_batch(d_wp[:Bs], d_tm[:Bs], d_bc, order=order, vel_zero_weight_per_traj=d_vw)
    g = csp.solve_batch(d_wp[:Bs], d_tm[:Bs], d_bc, order=order, vel_zero_weight_per_traj=d_vw, force_generic=True)
    torch.cuda.synchronize()
    a, gg = r.coeffs.cpu().numpy(), g.coeffs.cpu().numpy()
    synth.parity_gate(a, gg, 1e-7 if order == 5 else 1e-9, ("ring stores, per-trajectory bc, vs generic", order, S))
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


@pytest.mark.parametrize("order,S", [(2, 4), (3, 8), (3, 16), (4, 8), (4, 16)])
def test_path_kernel_whole_line_stores_and_the_skip_mask(csp, oracle_mod, order, S):
    """The path-penalty kernels store through the same ring (LineRing::flush<PRED = true>): dead rows -- the ragged last
    slice, and trajectories the re-solve loop has finished (skip mask) -- must stay untouched while their neighbours'
    lines leave whole.  (a) one solve against the generic kernel on a ragged multi-slice batch; (b) the re-solve loop on a
    batch where slice 0 converges at once and slice 1 converges every other trajectory (as
    test_resolve_loop_with_converged_slices does for order 4, S = 6), against the oracle's loop."""
    import torch
    rng = np.random.default_rng(40 + order * 17 + S)
    B = 64 * 11 + 29
    wp, tm = synth.make_batch(B, S, config_id=85 + order)
    bc = rng.normal(size=(B, 4, 3))
    d = [torch.from_numpy(x).cuda() for x in (wp, tm, bc)]
    kw = dict(order=order, path_weight=0.7, vel_zero_weight=0.03, want_status=True, want_max_dev=True)
    r = csp.solve_batch(d[0], d[1], d[2], **kw)
    g = csp.solve_batch(d[0], d[1], d[2], force_generic=True, **kw)
    torch.cuda.synchronize()
    assert r.kernel == "fixedpath_o%d_s%d_f64" % (order, S)
    synth.parity_gate(r.coeffs.cpu().numpy(), g.coeffs.cpu().numpy(), 1e-9, ("path ring stores vs generic", order, S))
    assert float((r.max_dev - g.max_dev).abs().max()) < 1e-8 * max(1.0, float(g.max_dev.max()))
    # (b) the loop
    B = 64 * 3 + 17
    wig, _ = synth.make_batch(B, S, config_id=26)
    wig = wig * 4.0
    t = np.linspace(0.0, 1.0, S + 1)[None, :, None]
    span = rng.uniform(20, 60, size=(B, 1, 3))
    straight = rng.uniform(-50, 50, size=(B, 1, 3)) + t * span + rng.normal(scale=0.02, size=(B, S + 1, 3))
    vdir = 5.0 * span[:, 0, :] / np.linalg.norm(span[:, 0, :], axis=1, keepdims=True)
    wp, bc = wig.copy(), np.zeros((B, 4, 3))
    sel = np.zeros(B, dtype=bool)
    sel[:64] = True
    sel[64:128:2] = True
    wp[sel] = straight[sel]
    bc[sel, 0] = vdir[sel]
    bc[sel, 1] = vdir[sel]
    plan = csp.plan_batch(torch.from_numpy(wp).cuda(), 5.0, 0.1, bc=torch.from_numpy(bc).cuda(), order=order,
                          path_weight=0.3, vel_zero_weight=0.0)
    torch.cuda.synchronize()
    it = plan.iterations.cpu().numpy()
    co, md, vwo = plan.coeffs.cpu().numpy(), plan.max_dev.cpu().numpy(), plan.vel_zero_weight.cpu().numpy()
    n_loop = 0
    for b in (0, 5, 63, 64, 65, 66, 67, 127, 128, 150, B - 1):
        ref, info = oracle_mod.generate_trajectory(wp[b], order=order, path_weight=0.3, vel_zero_weight=0.0, v_avg=5.0,
                                                   min_time_s=0.1, sample_distance=1.0, bc=bc[b])
        assert it[b] == info["iters"], (b, it[b], info["iters"])
        n_loop += int(it[b] > 0)
        assert abs(vwo[b] - info["vel_zero_weight"]) <= 1e-15
        assert abs(md[b] - info["max_dev"]) < 1e-7 * max(1.0, info["max_dev"])
        synth.parity_gate(co[b], info["coeff"], 1e-6, ("path ring stores, loop, vs oracle", order, S, b))
    assert n_loop > 0 and (it[:64] == 0).all()


def test_mixed_entry_edge_cases(csp, oracle_mod):
    """csp_minsnap_solve_mixed: orders 2..5 together, fp64 storage, every length class (1 .. 256 segments), per-trajectory
    boundary conditions and weights (host-memory form), unsupported trajectories flagged CSP_TRAJ_SKIPPED and left alone, an
    empty batch, and the validation errors."""
    import torch
    rng = np.random.default_rng(77)
    lens = np.array([1, 2, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65, 128, 129, 256, 3, 7, 40, 100, 200] * 3)
    B = len(lens)
    orders = rng.integers(2, 6, size=B).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    wps, tms = [], []
    for n in lens:
        p0 = rng.uniform(-10, 10, size=(1, 3))
        wps.append(np.concatenate([p0, p0 + np.cumsum(rng.normal(size=(int(n), 3)), axis=0)]))
        tms.append(rng.uniform(0.5, 2.0, size=int(n)))
    wp, tm = np.concatenate(wps), np.concatenate(tms)
    bc = rng.normal(size=(B, 4, 3))
    vw = rng.uniform(0, 0.2, size=B)
    r = csp.solve_mixed(orders, wp, tm, off, bc=bc, vel_zero_weight_per_traj=vw, want_status=True)
    assert not r.status.any()
    for i in range(B):
        o, n = int(orders[i]), int(lens[i])
        got = r.coeffs[r.coeff_offsets[i]:r.coeff_offsets[i] + 6 * o * n].reshape(n, 3, 2 * o)
        one = csp.solve_batch(wps[i], tms[i], bc[i][None], order=o, seg_offsets=np.array([0, n]), vel_zero_weight=float(vw[i]),
                              max_segments=n, force_generic=n <= 0)
        if n <= 64 and not (o == 5 and n > 32):
            ref, _ = oracle_mod.solve(o, wps[i], bc[i, [0, 1]], bc[i, [2, 3]], tms[i], 0.0, float(vw[i]), long_double=o == 5)
            synth.parity_gate(got, ref, 1e-6 if o == 5 else 5e-8, ("mixed entry vs oracle", i, o, n))
        synth.parity_gate(got, one.coeffs, 5e-5 if o == 5 else 1e-8, ("mixed entry vs single ragged call", i, o, n))
    # device form, unsupported trajectories among good ones
    bad_orders = orders.copy()
    bad_orders[3], bad_orders[10] = 1, 6
    d = [torch.from_numpy(x).cuda() for x in (bad_orders, wp, tm, off)]
    p = csp.PreparedMixed(d[0], d[1], d[2], d[3], bc=torch.from_numpy(bc).cuda(), want_status=True)
    p.out.fill_(-7.0)
    p.run()
    torch.cuda.synchronize()
    st = p.status.cpu().numpy()
    assert st[3] == csp.TRAJ_SKIPPED and st[10] == csp.TRAJ_SKIPPED and not np.delete(st, [3, 10]).any()
    cof = p.coeff_offsets.cpu().numpy()
    o_h = p.out.cpu().numpy()
    assert (o_h[cof[3]:cof[4]] == -7.0).all() and (o_h[cof[10]:cof[11]] == -7.0).all()
    assert not (o_h[cof[4]:cof[10]] == -7.0).any()
    # empty batch, bad arguments
    e = csp.solve_mixed(np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros(0), np.zeros(1, np.int64))
    assert e.coeff_offsets.shape == (1,)
    with pytest.raises(csp.CspError):
        csp.solve_mixed(orders, wp, tm, off, max_segments=300)


def test_sharded_entry_with_a_device_resident_batch(csp):
    """csp_minsnap_solve_batch_sharded with CSP_MEM_DEVICE: the batch lives on the root GPU and is scattered / solved / gathered
    over RCCL from one process.  On a one-GPU box this is the ngpu = 1 degenerate path (no communicator; the root's shard cut
    into chunks and solved in place through the same schedule): bit-equal with the plain call -- fixed kernel (4 chunks),
    per-trajectory boundary conditions + status + max_dev with the path penalty, the generic kernel (workspace from the
    arena), the chunked kernel; ragged batches are refused."""
    import torch
    rng = np.random.default_rng(9)
    B, S = 20000, 16
    wp, tm = synth.make_batch(B, S, config_id=95)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    bc = torch.from_numpy(rng.normal(size=(B, 4, 3))).cuda()
    for kw in (dict(order=4), dict(order=2, path_weight=1e-3, vel_zero_weight=0.01, bc=bc, want_status=True, want_max_dev=True),
               dict(order=4, force_generic=True, want_status=True), dict(order=3)):
        kw = dict(kw)
        b = kw.pop("bc", None)
        one = csp.solve_batch(d_wp, d_tm, b, **kw)
        sh = csp.solve_batch(d_wp, d_tm, b, ngpu=1, **kw)
        torch.cuda.synchronize()
        assert torch.equal(one.coeffs, sh.coeffs), kw
        if kw.get("want_status"):
            assert torch.equal(one.status, sh.status)
        if kw.get("want_max_dev"):
            assert torch.equal(one.max_dev, sh.max_dev)
    wl, tl = synth.make_batch(9000, 40, config_id=96)     # chunked kernel, two chunks
    a = csp.solve_batch(torch.from_numpy(wl).cuda(), torch.from_numpy(tl).cuda(), order=4)
    b = csp.solve_batch(torch.from_numpy(wl).cuda(), torch.from_numpy(tl).cuda(), order=4, ngpu=1)
    torch.cuda.synchronize()
    assert a.kernel.startswith("chunked_") and torch.equal(a.coeffs, b.coeffs)
    off = torch.arange(0, 41, 8, dtype=torch.int64).cuda()
    with pytest.raises(csp.CspError) as ei:
        csp.solve_batch(d_wp[:45], d_tm.reshape(-1)[:40], order=4, seg_offsets=off, ngpu=1)
    assert ei.value.code == -2
    with pytest.raises(csp.CspError):
        csp.solve_batch(d_wp, d_tm, order=4, ngpu=csp.device_count() + 1)


def test_sharding_on_two_devices(csp):
    """Runs only where two gfx950 devices are visible (never on the development boxes -- the multi-GPU paths are UNVERIFIED on
    hardware until this has run once): ngpu = 2 through both forms of csp_minsnap_solve_batch_sharded against the one-device
    result, bit for bit, and bench.py --end-to-end (RootPipeline over RCCL, two ranks)."""
    import json, os, subprocess, sys
    import torch
    if csp.device_count() < 2:
        pytest.skip("needs two gfx950 devices")
    B, S = 40000, 16
    wp, tm = synth.make_batch(B, S, config_id=97)
    one = csp.solve_batch(wp, tm, order=4, want_status=True)
    host2 = csp.solve_batch(wp, tm, order=4, want_status=True, ngpu=2)
    assert np.array_equal(one.coeffs, host2.coeffs) and np.array_equal(one.status, host2.status)
    dev2 = csp.solve_batch(torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda(), order=4, want_status=True, ngpu=2)
    torch.cuda.synchronize()
    assert np.array_equal(one.coeffs, dev2.coeffs.cpu().numpy()) and np.array_equal(one.status, dev2.status.cpu().numpy())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8192", "--end-to-end",
                          "--no-side-records", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["end_to_end"]["bit_equal_to_one_device"] is True


@pytest.mark.parametrize("order,S", [(4, 8), (2, 16), (3, 5), (5, 8)])
def test_many_small_batches_in_one_launch(csp, order, S):
    """csp_minsnap_solve_multi: independent uniform batches of one shape, each with its own buffers, boundary conditions and
    status array, in one kernel launch (fixed-size kernels; table passed as a kernel argument).  Bit-equal with one
    csp_minsnap_solve_batch call per batch; batch sizes around the 64-trajectory slice, empty batches, more than 32 batches
    (two launches), a bad trajectory flagged in the right batch."""
    import torch
    rng = np.random.default_rng(order * 31 + S)
    sizes = [4096, 1, 63, 64, 65, 0, 200, 1000] + [rng.integers(1, 300) for _ in range(30)]
    wps, tms, bcs = [], [], []
    for k, B in enumerate(sizes):
        wp, tm = synth.make_batch(max(int(B), 1), S, config_id=110 + order, offset=1000 * k)
        wps.append(torch.from_numpy(wp[:B]).cuda())
        tms.append(torch.from_numpy(tm[:B]).cuda())
        bcs.append(torch.from_numpy(rng.normal(size=(1, 4, 3))).cuda())
    tms[6][17, 2] = float("nan")
    pm = csp.PreparedMulti(wps, tms, bcs, order=order, vel_zero_weight=0.02, want_status=True)
    for o in pm.out:
        o.fill_(7.0)
    pm.run()
    torch.cuda.synchronize()
    for k, B in enumerate(sizes):
        if B == 0:
            continue
        one = csp.solve_batch(wps[k], tms[k], bcs[k], order=order, vel_zero_weight=0.02, want_status=True)
        torch.cuda.synchronize()
        if k == 6:
            assert int(pm.status[k][17]) & csp.TRAJ_NONFINITE and int(pm.status[k].abs().sum()) == int(pm.status[k][17])
            ok = torch.ones(B, dtype=torch.bool, device="cuda")
            ok[17] = False
            assert torch.equal(pm.out[k][ok], one.coeffs[ok]), k
        else:
            assert torch.equal(pm.out[k], one.coeffs), (k, B)
            assert int(pm.status[k].abs().max()) == 0
        assert torch.equal(pm.status[k], one.status), k


def test_keep_decision_ulp_band_is_recorded(csp, oracle_mod):
    """The thinning test `dist >= sample_distance` (minimum_snap.cpp:142-150) is decided on positions the reference evaluates
    term by term with std::pow (:104-117) and the HIP samplers with a power ladder: a few ulp OF THE POSITION apart, so there
    is a band of sample_distance values around every candidate's distance in which the two keep different candidates -- its
    width is a few ulp of |position|, i.e. tens of ulp of the (much smaller) distance.  F7 covers exact ties and +-1e-10;
    this test measures the band itself: the SAME coefficients go to the oracle's sampling loop and to the one-lane HIP
    sampler while sample_distance is moved 0, +-1, +-2, +-4 .. +-4096 ulp off the first kept candidate's distance.
    Nothing inside the band is asserted (it is recorded, printed and written to gpurun_out/ulp_band.json for DESIGN.md
    section 3); asserted is only that the two agree again at +-4096 ulp (4.5e-13 relative)."""
    import json, os
    import torch
    rng = np.random.default_rng(123)
    S, order, ntraj = 6, 4, 40
    steps = [0] + [sg * (1 << e) for e in range(13) for sg in (-1, 1)]
    edges, flips = [], 0
    for it in range(ntraj):
        p0 = rng.uniform(-10, 10, size=(1, 3))
        wp = np.concatenate([p0, p0 + np.cumsum(rng.normal(size=(S, 3)) * 3.0, axis=0)])
        tm = csp.time_alloc_batch(wp[None], 5.0, 0.1)
        r = csp.solve_batch(wp[None], tm, order=order)
        co, T = r.coeffs[0], np.asarray(tm[0], dtype=np.float64)
        base = oracle_mod.sample(co, T, 0.7)
        d_star = float(np.sqrt(np.sum((base[1] - base[0]) ** 2)))    # the oracle's distance of the first kept candidate
        ulp = float(np.spacing(d_star))
        d_co, d_tm = torch.from_numpy(co[None].copy()).cuda(), torch.from_numpy(T[None].copy()).cuda()
        disagree = []
        for k in steps:
            sd = d_star + k * ulp
            ref = oracle_mod.sample(co, T, sd)
            smp, cnt, _ = csp.sample_batch(d_tm, d_co, float(sd), 512, order=order, one_lane=True)
            torch.cuda.synchronize()
            got = smp[0, :int(cnt[0])].cpu().numpy()
            same = got.shape == ref.shape and np.max(np.abs(got - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))
            if not same:
                disagree.append(k)
        if disagree:
            flips += 1
            edges.append(max(abs(k) for k in disagree))
            assert max(abs(k) for k in disagree) < 4096, (it, disagree)   # agreement is back at +-4096 ulp
    rec = {"trajectories": ntraj, "with_a_disagreement": flips, "outermost_disagreeing_offset_ulp_max": max(edges) if edges else 0,
           "outermost_disagreeing_offset_ulp_median": float(np.median(edges)) if edges else 0.0,
           "as_relative_offset_max": (max(edges) if edges else 0) * 2.0 ** -52, "offsets_probed_ulp": "0, +-1, +-2, +-4 .. +-4096"}
    print("keep-decision ulp band (HIP power ladder vs the oracle's pow):", rec)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rec, open(os.path.join("gpurun_out", "ulp_band.json"), "w"))


def test_device_memory_calls_can_be_captured_in_a_hip_graph(csp):
    """CSP_MEM_DEVICE calls only enqueue work on the caller's stream, so a planner can capture its per-tick solves in a HIP
    graph: 8 small solves and a path-penalty solve captured once and replayed, bit-equal with the eager calls.  (A graph
    removes host launch cost, not the kernels' own ~6 us latency: 16 x C2 replay at 6.4 us per batch against 1.8 us through
    csp_minsnap_solve_multi's single launch -- tools/graph_probe.py.)"""
    import torch
    dev = torch.device("cuda", 0)
    preps = []
    for k in range(8):
        wp, tm = synth.make_batch(1000 + 37 * k, 8, config_id=2, offset=5000 * k)
        preps.append(csp.PreparedSolve(torch.from_numpy(wp).to(dev), torch.from_numpy(tm).to(dev), order=4))
    wp, tm = synth.make_batch(3000, 16, config_id=3)
    preps.append(csp.PreparedSolve(torch.from_numpy(wp).to(dev), torch.from_numpy(tm).to(dev), order=2, path_weight=1e-3, vel_zero_weight=0.01))
    for p in preps:
        p.run()
    torch.cuda.synchronize()
    ref = [p.out.clone() for p in preps]
    for p in preps:
        p.out.zero_()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for p in preps:
                p.run(stream=s.cuda_stream)
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.out) for a, p in zip(ref, preps))


def test_hip_against_the_60_digit_kkt_solution(csp):
    """F8: the constrained QP solved directly in 60-digit arithmetic (oracle/gen_golden_kkt.py) -- an independent yardstick that
    shares nothing with the closed form's algebra or with either restatement's rounding.  Every HIP kernel family that serves
    the case (register-resident, generic) against it, per power; the errors are printed."""
    from tests.conftest import load_cases
    worst = {}
    for c in load_cases("F8_kkt_mpmath.json"):
        for force in (False, True):
            r = csp.solve_batch(c["path"][None], c["time"][None], c["bc"][None], order=c["order"], vel_zero_weight=c["vel_zero_weight"],
                                path_weight=c["path_weight"], want_status=True, want_max_dev=True, force_generic=force)
            if c["path_weight"]:
                assert abs(float(r.max_dev[0]) - c["max_dev"]) < 1e-9 * max(1.0, c["max_dev"]), c["name"]
            assert int(r.status[0]) == 0
            # measured (round 3): <= 3e-15 / 9e-14 / 9e-13 / 1.9e-11 per power at orders 2 / 3 / 4 / 5 -- three orders of magnitude
            # closer to the exact QP solution than the fp64 dense restatement of the reference's algorithm is (3.8e-8)
            tol = 1e-10 if c["path_weight"] else (5e-10 if c["order"] == 5 else 1e-11)   # with the path penalty: <= 1.3e-11 measured
            pp, _ = synth.parity_gate(r.coeffs[0], c["coeff"], tol, ("HIP vs 60-digit KKT", c["name"], r.kernel))
            key = (c["order"], ("generic" if force else r.kernel.split("_")[0]) + ("+path" if c["path_weight"] else ""))
            worst[key] = max(worst.get(key, 0.0), pp)
    print("HIP vs the 60-digit KKT solution, worst per-power error by (order, kernel family):", worst)


@pytest.mark.gpu
def test_mixed_entry_lane_pair_sweep_against_the_chunked_kernels(tmp_path):
    """csp_minsnap_solve_mixed serves trajectories of up to 64 segments with the lane-pair sweep (minsnap_twist_impl.h: two
    forward sweeps and one backward sweep in blocks) and CSP_MIXED_TWIST=0 puts them on the chunked kernels
    (substructuring): two different factorisation orders of the same system.  The library reads the switch once, so each
    setting runs in its own process; fp64 storage, orders 2..5, every segment count 1..64, per-trajectory boundary
    conditions; agreement per power at the level two HIP kernel families reach elsewhere (order 5: its conditioning)."""
    import subprocess, sys, os
    script = r'''
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
csp = importlib.import_module("cs-pathplan_amd")
g = np.random.default_rng(77)
lens = np.concatenate([np.arange(1, 65), g.integers(1, 65, size=700)])
orders = np.concatenate([np.repeat([2, 3, 4, 5], 16), g.integers(2, 6, size=700)]).astype(np.int32)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
wp = np.concatenate([np.cumsum(g.normal(size=(n + 1, 3)), axis=0) + g.uniform(-10, 10, 3) for n in lens])
tm = g.uniform(0.5, 2.0, size=int(off[-1]))
bc = g.normal(size=(len(lens), 4, 3))
r = csp.solve_mixed(orders, wp, tm, off, bc=bc, vel_zero_weight=0.25, want_status=True)
np.savez(sys.argv[1], co=np.asarray(r.coeffs), cf=np.asarray(r.coeff_offsets), st=np.asarray(r.status), orders=orders, lens=lens)
'''
    out = {}
    for mask in ("0", "15"):
        path = str(tmp_path / ("m%s.npz" % mask))
        env = dict(os.environ, CSP_MIXED_TWIST=mask)
        p = subprocess.run([sys.executable, "-c", script, path], env=env, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert p.returncode == 0, p.stderr[-3000:]
        out[mask] = np.load(path)
    a, b = out["0"], out["15"]
    assert np.array_equal(a["cf"], b["cf"]) and np.array_equal(a["st"], b["st"]) and not a["st"].any()
    worst = 0.0
    for i, (o, n) in enumerate(zip(a["orders"], a["lens"])):
        lo = int(a["cf"][i])
        x = a["co"][lo:lo + n * 6 * o].reshape(n, 3, 2 * o)
        y = b["co"][lo:lo + n * 6 * o].reshape(n, 3, 2 * o)
        pp, _ = synth.parity_gate(y, x, 5e-5 if o == 5 else 1e-8, ("lane-pair sweep vs chunked", i, int(o), int(n)))
        worst = max(worst, pp)
    print("lane-pair sweep vs chunked kernels, per power: %.2e" % worst)


@pytest.mark.gpu
def test_mixed_entry_status_bits_and_declared_maximum(csp):
    """The lane-pair sweep behind csp_minsnap_solve_mixed reports per trajectory: a NaN waypoint -> CSP_TRAJ_NONFINITE on that
    trajectory only (its neighbours in the same 64-trajectory work unit are untouched); a trajectory longer than the declared
    max_segments -> CSP_TRAJ_SKIPPED, its block left alone; fp32 and fp64 storage."""
    import torch
    rng = np.random.default_rng(5)
    for dt in (np.float32, np.float64):
        lens = np.concatenate([np.full(130, 12), np.full(70, 40), [9]])
        orders = np.resize(np.array([3, 4, 5, 2], dtype=np.int32), len(lens))
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        wp = np.cumsum(rng.normal(size=(int(off[-1]) + len(lens), 3)), axis=0).astype(dt)
        tm = rng.uniform(0.5, 2.0, size=int(off[-1])).astype(dt)
        bad = 37
        wp[off[bad] + bad + 5, 1] = np.nan
        d = [torch.from_numpy(x).cuda() for x in (orders, wp, tm, off)]
        p = csp.PreparedMixed(d[0], d[1], d[2], d[3], want_status=True, max_segments=40)
        clean_wp = wp.copy()
        clean_wp[off[bad] + bad + 5, 1] = 0.0
        q = csp.PreparedMixed(d[0], torch.from_numpy(clean_wp).cuda(), d[2], d[3], want_status=True, max_segments=40)
        p.out.zero_(); q.out.zero_()
        p.run(); q.run()
        torch.cuda.synchronize()
        st = p.status.cpu().numpy()
        assert st[bad] & csp.TRAJ_NONFINITE and not np.delete(st, bad).any(), (dt, st[bad], np.flatnonzero(st))
        assert not q.status.cpu().numpy().any()
        cof = p.coeff_offsets.cpu().numpy()
        a, b = p.out.cpu().numpy(), q.out.cpu().numpy()
        keep = np.ones(a.size, bool)
        keep[cof[bad]:cof[bad + 1]] = False
        assert np.array_equal(a[keep], b[keep])
        # declared maximum below the longest trajectory
        r = csp.PreparedMixed(d[0], torch.from_numpy(clean_wp).cuda(), d[2], d[3], want_status=True, max_segments=12)
        r.out.fill_(-3.0)
        r.run()
        torch.cuda.synchronize()
        st = r.status.cpu().numpy()
        long_ones = lens > 12
        assert (st[long_ones] == csp.TRAJ_SKIPPED).all() and not st[~long_ones].any()
        o = r.out.cpu().numpy()
        i = int(np.flatnonzero(long_ones)[0])
        assert (o[cof[i]:cof[i + 1]] == -3.0).all()
        assert np.array_equal(o[cof[0]:cof[1]], b[cof[0]:cof[1]])
