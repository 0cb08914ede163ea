"""Batched GenerateTrajectoryMatrix pieces on the GPU against the oracle's
restatement of minimum_snap.cpp:22-206: time allocation + re-solve loop (csp_minsnap_plan_batch) and
sampling / distance thinning / statistics (csp_minsnap_sample_batch)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _ref(oracle_mod, P, **kw):
    s, info = oracle_mod.generate_trajectory(P, **kw)
    return s, info


@pytest.mark.parametrize("order,pw,vw", [(3, 0.0, 0.0), (4, 0.0, 0.02), (3, 0.5, 0.0), (2, 1e-2, 0.01), (4, 0.3, 0.0)])
def test_plan_then_sample_matches_oracle(csp, oracle_mod, order, pw, vw):
    B, S = 24, 6
    wp, _ = synth.make_batch(B, S, config_id=21)
    wp = wp * 4.0
    v_avg, min_t, sd = 5.0, 0.1, 0.7
    plan = csp.plan_batch(wp, v_avg, min_t, order=order, path_weight=pw, vel_zero_weight=vw)
    assert not plan.status.any()
    cap = 4096
    samples, counts, stats = csp.sample_batch(plan.times, plan.coeffs, sd, cap)
    n_loop = 0
    for b in range(B):
        ref, info = _ref(oracle_mod, wp[b], order=order, path_weight=pw, vel_zero_weight=vw, v_avg=v_avg,
                         min_time_s=min_t, sample_distance=sd)
        n_loop += info["iters"] > 0
        assert np.allclose(plan.times[b], info["time"], rtol=0, atol=1e-15 * np.max(info["time"]))
        assert plan.iterations[b] == info["iters"], (b, plan.iterations[b], info["iters"])
        assert abs(plan.vel_zero_weight[b] - info["vel_zero_weight"]) <= 1e-15
        assert abs(plan.max_dev[b] - info["max_dev"]) < 1e-7 * max(1.0, info["max_dev"])
        scale = np.max(np.abs(info["coeff"]))
        assert np.max(np.abs(plan.coeffs[b] - info["coeff"])) < 1e-7 * scale
        assert counts[b] == len(ref), (b, counts[b], len(ref))
        assert np.max(np.abs(samples[b, :counts[b]] - ref)) < 1e-7 * np.max(np.abs(ref))
        assert abs(stats[b, 0] - info["max_climb_rate"]) < 1e-6 * max(1.0, info["max_climb_rate"])
        assert abs(stats[b, 1] - info["min_turn_radius"]) < 1e-5 * max(1.0, info["min_turn_radius"])
    if pw >= 0.3:
        assert n_loop > 0, "fixture does not exercise the re-solve loop"


def test_sample_capacity_overflow_is_reported(csp):
    wp, _ = synth.make_batch(4, 4, config_id=22)
    plan = csp.plan_batch(wp, 5.0, 0.1, order=3)
    samples, counts, _ = csp.sample_batch(plan.times, plan.coeffs, 1e-9, 8)
    assert (counts > 8).all()          # true counts come back; only 8 rows were written


def test_device_memory_plan_and_sample(csp, oracle_mod):
    import torch
    B, S = 512, 8
    wp, _ = synth.make_batch(B, S, config_id=23)
    d_wp = torch.from_numpy(wp * 3.0).cuda()
    plan = csp.plan_batch(d_wp, 5.0, 0.1, order=4)          # fixed kernel path (no path penalty)
    samples, counts, stats = csp.sample_batch(plan.times, plan.coeffs, 0.5, 2048)
    torch.cuda.synchronize()
    host = csp.plan_batch(wp * 3.0, 5.0, 0.1, order=4)
    assert np.array_equal(plan.coeffs.cpu().numpy(), host.coeffs)
    for b in (0, 17, 511):
        ref, info = oracle_mod.generate_trajectory(wp[b] * 3.0, order=4, v_avg=5.0, min_time_s=0.1, sample_distance=0.5)
        n = int(counts[b])
        assert n == len(ref)
        assert np.max(np.abs(samples[b, :n].cpu().numpy() - ref)) < 1e-7 * np.max(np.abs(ref))


@pytest.mark.parametrize("order,S,B,sd", [(4, 16, 1000, 0.7), (3, 1, 200, 0.3), (2, 7, 333, 5.0), (5, 8, 100, 0.05),
                                           (4, 64, 70, 1.5), (3, 37, 129, 100.0), (4, 2, 65, 1e-9)])
def test_segment_parallel_sampler_is_bitwise_the_one_lane_sampler(csp, order, S, B, sd):
    """Trajectories of up to 64 segments are sampled with one lane per (trajectory, segment) (two
    passes, prefix sums over the segments); the result -- samples, counts, statistics -- must be
    IDENTICAL to the sequential one-lane-per-trajectory kernel, which the oracle tests pin.  Covers
    segments that keep nothing (large distance), capacity overflow and the end-point rule."""
    import torch
    wp, _ = synth.make_batch(B, S, config_id=24)
    plan = csp.plan_batch(torch.from_numpy(wp * 3.0).cuda(), 5.0, 0.1, order=order)
    for cap in (4096, 5):
        a = csp.sample_batch(plan.times, plan.coeffs, sd, cap)
        o = csp.sample_batch(plan.times, plan.coeffs, sd, cap, one_lane=True)
        torch.cuda.synchronize()
        assert torch.equal(a[1], o[1]), (order, S, cap)
        assert torch.equal(a[0], o[0]), (order, S, cap)
        assert torch.equal(a[2], o[2]), (order, S, cap)
    assert int(a[1].min()) >= 2


def test_resolve_loop_with_converged_slices(csp, oracle_mod):
    """Re-solve loop on a batch where whole 64-trajectory slices converge at the first solve (nearly
    straight paths: deviation <= 0.2) while others keep doubling the weight: the later passes must
    leave the converged trajectories exactly as the first pass wrote them (the path kernel drops
    fully converged slices after one load, partially converged ones are computed but not stored)."""
    import torch
    rng = np.random.default_rng(5)
    B, S = 64 * 3 + 17, 6
    wig, _ = synth.make_batch(B, S, config_id=25)
    wig = wig * 4.0
    # straight, evenly spaced paths flown at V_avg from end to end (boundary velocity = V_avg along the
    # line): the time-parametrised chord is followed closely, so the deviation metric stays below 0.2
    t = np.linspace(0.0, 1.0, S + 1)[None, :, None]
    span = rng.uniform(20, 60, size=(B, 1, 3))
    straight = rng.uniform(-50, 50, size=(B, 1, 3)) + t * span + rng.normal(scale=0.02, size=(B, S + 1, 3))
    vdir = 5.0 * span[:, 0, :] / np.linalg.norm(span[:, 0, :], axis=1, keepdims=True)
    wp = wig.copy()
    bc = np.zeros((B, 4, 3))
    sel = np.zeros(B, dtype=bool)
    sel[:64] = True            # slice 0: all converge at once
    sel[64:128:2] = True       # slice 1: every other trajectory converges
    wp[sel] = straight[sel]
    bc[sel, 0] = vdir[sel]
    bc[sel, 1] = vdir[sel]
    plan = csp.plan_batch(torch.from_numpy(wp).cuda(), 5.0, 0.1, bc=torch.from_numpy(bc).cuda(), order=4,
                          path_weight=0.3, vel_zero_weight=0.0)
    torch.cuda.synchronize()
    it = plan.iterations.cpu().numpy()
    assert (it[:64] == 0).all() and (it[64:128:2] == 0).all() and it[128:].max() > 0
    co, md, vwo = plan.coeffs.cpu().numpy(), plan.max_dev.cpu().numpy(), plan.vel_zero_weight.cpu().numpy()
    for b in (0, 5, 63, 64, 65, 66, 127, 128, B - 1):
        ref, info = oracle_mod.generate_trajectory(wp[b], order=4, path_weight=0.3, vel_zero_weight=0.0, v_avg=5.0,
                                                   min_time_s=0.1, sample_distance=1.0, bc=bc[b])
        assert it[b] == info["iters"], (b, it[b], info["iters"])
        assert abs(vwo[b] - info["vel_zero_weight"]) <= 1e-15
        assert abs(md[b] - info["max_dev"]) < 1e-7 * max(1.0, info["max_dev"])
        assert np.max(np.abs(co[b] - info["coeff"])) < 1e-7 * np.max(np.abs(info["coeff"])), b


@pytest.mark.parametrize("order,scale,v_avg,sd", [(4, 3000.0, 30.0, 30.0), (3, 800.0, 12.0, 5.0), (2, 5000.0, 200.0, 300.0)])
def test_wave_cooperative_sampler_for_long_legs(csp, oracle_mod, order, scale, v_avg, sd):
    """Kilometre legs at 0.1-s candidates (hundreds to thousands per segment, the reference's own use):
    one wave per trajectory searches 64 candidates per round.  Must be bitwise the sequential sampler
    (same accumulated candidate times), on the device path (flag) and on the host path (chosen from
    the times), and agree with the oracle."""
    import torch
    B, S = 7, 5
    wp, _ = synth.make_batch(B, S, config_id=26)
    wp = wp * scale
    plan = csp.plan_batch(torch.from_numpy(wp).cuda(), v_avg, 1.0, order=order)
    cap = 1 << 15
    a = csp.sample_batch(plan.times, plan.coeffs, sd, cap, long_segments=True)
    o = csp.sample_batch(plan.times, plan.coeffs, sd, cap, one_lane=True)
    torch.cuda.synchronize()
    assert torch.equal(a[1], o[1]) and torch.equal(a[0], o[0]) and torch.equal(a[2], o[2])
    assert float(plan.times.max()) / 0.1 > 300          # really long segments
    h = csp.sample_batch(plan.times.cpu().numpy(), plan.coeffs.cpu().numpy(), sd, cap)   # host path picks the wave kernel
    assert np.array_equal(h[1], a[1].cpu().numpy())
    for b in range(B):   # rows beyond counts[b] are unspecified (the host path stages through a reused arena)
        assert np.array_equal(h[0][b, :h[1][b]], a[0][b, :h[1][b]].cpu().numpy()), b
    for b in (0, B - 1):
        ref, info = oracle_mod.generate_trajectory(wp[b], order=order, v_avg=v_avg, min_time_s=1.0, sample_distance=sd)
        n = int(a[1][b])
        assert n == len(ref), (b, n, len(ref))
        assert np.max(np.abs(a[0][b, :n].cpu().numpy() - ref)) < 1e-7 * np.max(np.abs(ref))
    # a non-finite segment time must not hang the device: that trajectory simply gets no candidates
    tm = plan.times.clone()
    tm[3, 2] = float("inf")
    for kw in (dict(long_segments=True), dict(), dict(one_lane=True)):
        r = csp.sample_batch(tm, plan.coeffs, sd, cap, **kw)
        torch.cuda.synchronize()
        assert int(r[1][3]) >= 1


def test_segment_wave_sampler_time_table_edges(csp):
    """The per-segment wave sampler of the host path reads accumulated candidate times from a per-device table when
    dt = 0.1 (minsnap_plan.hip: tacc).  Edges: a segment longer than the table (9000 candidates > 8192), a segment with
    dt = T/10 != 0.1 (no table), ordinary ones -- bitwise the sequential sampler."""
    import torch
    wp = np.array([[[0.0, 0.0, 0.0], [4500.0, 10.0, 0.0], [4502.5, 10.0, 1.0], [4652.5, -20.0, 0.0], [4800.0, 0.0, 5.0]],
                   [[10.0, 0.0, 0.0], [400.0, 50.0, 0.0], [401.0, 50.0, 0.0], [1000.0, 0.0, 9.0], [5200.0, 100.0, 0.0]]])
    for order in (2, 4):
        plan = csp.plan_batch(torch.from_numpy(wp).cuda(), 5.0, 0.1, order=order)
        tm = plan.times.cpu().numpy()
        assert tm.max() / 0.1 > 8192 + 64 and tm.min() < 1.0
        cap = 1 << 15
        o = csp.sample_batch(plan.times, plan.coeffs, 3.0, cap, one_lane=True)
        w = csp.sample_batch(plan.times, plan.coeffs, 3.0, cap, long_segments=True)   # one wave per trajectory, same table
        torch.cuda.synchronize()
        assert torch.equal(w[1], o[1]) and torch.equal(w[0], o[0]) and torch.equal(w[2], o[2])
        h = csp.sample_batch(tm, plan.coeffs.cpu().numpy(), 3.0, cap)     # host path: one wave per segment
        assert np.array_equal(h[1], o[1].cpu().numpy()), (order, h[1], o[1])
        for b in range(2):
            assert np.array_equal(h[0][b, :h[1][b]], o[0][b, :h[1][b]].cpu().numpy()), (order, b)
        assert np.array_equal(h[2], o[2].cpu().numpy())


def test_ragged_sampling_equals_per_trajectory_sampling(csp):
    """Ragged batches (seg_offsets) through all three samplers: trajectory b of the ragged call must be
    bitwise the uniform single-trajectory call."""
    rng = np.random.default_rng(9)
    order, sd, cap = 3, 0.6, 2048
    lens = [1, 5, 64, 2, 17, 33, 8]
    tms, cos = [], []
    for S in lens:
        wp, _ = synth.make_batch(1, S, config_id=27)
        plan = csp.plan_batch(wp * 3.0, 5.0, 0.1, order=order)
        tms.append(plan.times[0]); cos.append(plan.coeffs[0])
    tm, co = np.concatenate(tms), np.concatenate(cos)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for kw in (dict(), dict(one_lane=True), dict(long_segments=True)):
        smp, cnt, stats = csp.sample_batch(tm, co, sd, cap, order=order, seg_offsets=off, **kw)
        for b, S in enumerate(lens):
            s1, c1, st1 = csp.sample_batch(tms[b][None], cos[b][None], sd, cap, one_lane=True)
            assert cnt[b] == c1[0], (kw, b)
            assert np.array_equal(smp[b, :cnt[b]], s1[0, :c1[0]]), (kw, b)
            assert np.array_equal(stats[b], st1[0]), (kw, b)


def test_resolve_loop_on_the_generic_kernel(csp, oracle_mod):
    """More than 16 segments with the path penalty: the re-solve loop runs on the generic kernel, whose
    passes after the first also reuse the stored t* indices.  Iteration counts, weights, deviation and
    coefficients against the oracle."""
    B, S, order = 10, 20, 3
    wp, _ = synth.make_batch(B, S, config_id=28)
    wp = wp * 4.0
    plan = csp.plan_batch(wp, 5.0, 0.1, order=order, path_weight=0.5, vel_zero_weight=0.0)
    assert not plan.status.any()
    assert plan.iterations.max() > 0
    for b in range(B):
        _, info = oracle_mod.generate_trajectory(wp[b], order=order, path_weight=0.5, vel_zero_weight=0.0, v_avg=5.0,
                                                 min_time_s=0.1, sample_distance=1.0)
        assert plan.iterations[b] == info["iters"], (b, plan.iterations[b], info["iters"])
        assert abs(plan.vel_zero_weight[b] - info["vel_zero_weight"]) <= 1e-15
        assert abs(plan.max_dev[b] - info["max_dev"]) < 1e-7 * max(1.0, info["max_dev"])
        assert np.max(np.abs(plan.coeffs[b] - info["coeff"])) < 1e-7 * np.max(np.abs(info["coeff"])), b


def test_samplers_agree_with_fp32_storage(csp):
    """fp32 storage: the segment-parallel sampler takes its statistics inline (stored samples are rounded), the
    wave sampler always does; all three must still agree bit for bit on samples, counts and statistics."""
    import torch
    wp, _ = synth.make_batch(300, 9, config_id=29)
    plan = csp.plan_batch(torch.from_numpy((wp * 3.0).astype(np.float32)).cuda(), 5.0, 0.1, order=3)
    assert plan.coeffs.dtype == torch.float32
    ref = csp.sample_batch(plan.times, plan.coeffs, 0.6, 512, one_lane=True)
    for kw in (dict(), dict(long_segments=True)):
        got = csp.sample_batch(plan.times, plan.coeffs, 0.6, 512, **kw)
        torch.cuda.synchronize()
        assert got[0].dtype == torch.float32
        assert torch.equal(got[1], ref[1]) and torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]), kw


@pytest.mark.parametrize("order,pw,scale,v_avg,sd", [(2, 1e-7, 1500.0, 200.0, 300.0), (4, 0.0, 3.0, 5.0, 0.7), (3, 0.3, 4.0, 5.0, 0.5),
                                                     (2, 0.3, 40000.0, 60.0, 250.0)])
def test_generate_batch_is_plan_then_sample(csp, order, pw, scale, v_avg, sd):
    """csp_minsnap_generate_batch (the whole GenerateTrajectoryMatrix in one call) against csp_minsnap_plan_batch
    followed by csp_minsnap_sample_batch: bit for bit, from host memory (one upload / download / synchronisation; the
    fourth case is ONE long flight whose sample block exceeds 1 MB, fetched in two steps; the third raises weights, so
    the second round of the loop runs) and from device memory."""
    import torch
    for B in (1, 9):
        wp, _ = synth.make_batch(B, 6, config_id=33 + order)
        wp = wp * scale
        bc = np.zeros((1, 4, 3))
        bc[0, 0] = [0.3, -0.2, 0.1]
        cap = csp.sample_capacity(wp, v_avg, 1.0, order=order)
        p = csp.plan_batch(wp, v_avg, 1.0, bc=bc, order=order, path_weight=pw, vel_zero_weight=0.01)
        s = csp.sample_batch(p.times, p.coeffs, sd, cap)
        g = csp.generate_batch(wp, v_avg, 1.0, sd, bc=bc, order=order, path_weight=pw, vel_zero_weight=0.01)
        assert g.samples.shape[1] == cap and int(s[1].max()) <= cap
        if order == 3:
            assert p.iterations.max() > 0          # the loop really ran
        if scale == 40000.0 and B == 1:
            assert cap * 24 > (1 << 20)
        for name in ("times", "coeffs", "max_dev", "vel_zero_weight", "iterations", "status"):
            assert np.array_equal(getattr(g, name), getattr(p, name)), (name, B)
        assert np.array_equal(g.counts, s[1]) and np.array_equal(g.stats, s[2]), B
        for b in range(B):
            assert np.array_equal(g.samples[b, :g.counts[b]], s[0][b, :s[1][b]]), (B, b)
        # device memory: asynchronous, same results
        gd = csp.generate_batch(torch.from_numpy(wp).cuda(), v_avg, 1.0, sd, capacity=cap, bc=torch.from_numpy(bc).cuda(), order=order,
                                path_weight=pw, vel_zero_weight=0.01)
        pd = csp.plan_batch(torch.from_numpy(wp).cuda(), v_avg, 1.0, bc=torch.from_numpy(bc).cuda(), order=order, path_weight=pw,
                            vel_zero_weight=0.01)
        sd_ = csp.sample_batch(pd.times, pd.coeffs, sd, cap)
        torch.cuda.synchronize()
        assert torch.equal(gd.coeffs, pd.coeffs) and torch.equal(gd.counts, sd_[1]) and torch.equal(gd.samples, sd_[0])
        assert torch.equal(gd.stats, sd_[2]) and torch.equal(gd.iterations, pd.iterations)
        assert np.array_equal(gd.counts.cpu().numpy(), g.counts)


def test_generate_batch_ragged_host_call(csp):
    """csp_minsnap_generate_batch through the raw C-ABI with a RAGGED host batch (seg_offsets): every trajectory equals
    the uniform single-trajectory call, bit for bit -- with the path penalty on (the loop's state is per trajectory) and
    long legs (the per-segment wave sampler with runs sized from the host's estimate of the times)."""
    import ctypes
    lens = [3, 6, 2, 5]
    order, v_avg, sd = 3, 40.0, 25.0
    rng = np.random.default_rng(12)
    wps = [np.cumsum(rng.normal(scale=900.0, size=(n + 1, 3)), axis=0) for n in lens]
    wp = np.ascontiguousarray(np.concatenate(wps))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    B, total = len(lens), int(off[-1])
    bc = np.zeros((1, 4, 3))
    d = csp.make_desc(order, B, 0, csp.DTYPE_F64, 0.3, 0.01, csp.MEM_HOST, seg_offsets_ptr=off.ctypes.data, max_segments=max(lens))
    cap = int(csp.raw_lib().csp_minsnap_sample_capacity(ctypes.byref(d), wp.ctypes.data, v_avg, 1.0))
    assert cap > 0
    smp = np.zeros((B, cap, 3))
    cnt = np.zeros(B, dtype=np.int32)
    st = np.zeros((B, 2))
    tm = np.zeros(total)
    co = np.zeros((total, 3, 2 * order))
    md, vw = np.zeros(B), np.zeros(B)
    it, sta = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
    rc = csp.raw_lib().csp_minsnap_generate_batch(ctypes.byref(d), wp.ctypes.data, v_avg, 1.0, bc.ctypes.data, sd, cap, smp.ctypes.data,
                                                 cnt.ctypes.data, st.ctypes.data, tm.ctypes.data, co.ctypes.data, md.ctypes.data,
                                                 vw.ctypes.data, it.ctypes.data, sta.ctypes.data, None, 0, None)
    assert rc == 0, rc
    assert tm.max() / 0.1 > 300      # long legs
    assert it.max() > 0, it    # weights were raised: the second round of the call ran, after the folded first-pass update
    for b, n in enumerate(lens):
        # the ragged batch runs the generic kernel, the uniform single call the register-resident one: same results up to
        # rounding (the sampling decisions are far from ties here)
        one = csp.generate_batch(wps[b][None], v_avg, 1.0, sd, capacity=cap, order=order, path_weight=0.3, vel_zero_weight=0.01)
        assert np.array_equal(one.times[0], tm[off[b]:off[b + 1]]), b
        synth.parity_gate(co[off[b]:off[b + 1]], one.coeffs[0], 1e-9, ("ragged generate vs per-trajectory", b))
        assert one.counts[0] == cnt[b] and one.iterations[0] == it[b] and one.vel_zero_weight[0] == vw[b], b
        scale = np.max(np.abs(one.samples[0, :cnt[b]]))
        assert np.max(np.abs(one.samples[0, :cnt[b]] - smp[b, :cnt[b]])) < 1e-8 * scale, b
        assert np.allclose(one.stats[0], st[b], rtol=1e-6), b
    # and the ragged call against the two ragged calls it stands for: bit for bit
    tm2, co2 = np.zeros(total), np.zeros((total, 3, 2 * order))
    md2, vw2, it2, sta2 = np.zeros(B), np.zeros(B), np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
    rc = csp.raw_lib().csp_minsnap_plan_batch(ctypes.byref(d), wp.ctypes.data, v_avg, 1.0, bc.ctypes.data, tm2.ctypes.data, co2.ctypes.data,
                                             md2.ctypes.data, vw2.ctypes.data, it2.ctypes.data, sta2.ctypes.data, None, 0, None)
    assert rc == 0
    assert np.array_equal(tm2, tm) and np.array_equal(co2, co) and np.array_equal(it2, it) and np.array_equal(vw2, vw)
    s2 = csp.sample_batch(tm2, co2, sd, cap, order=order, seg_offsets=off)
    assert np.array_equal(s2[1], cnt) and np.array_equal(s2[2], st)
    for b in range(B):
        assert np.array_equal(s2[0][b, :cnt[b]], smp[b, :cnt[b]]), b
