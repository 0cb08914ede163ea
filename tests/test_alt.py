"""Row N4: the altitude optimiser's two pentadiagonal SPD solves (the reference's only
Eigen::SimplicialLDLT sites, uavPathPlanning.cpp:1575-1827).  PARITY UNPINNED (no Eigen, no goldens):
the C oracle (dense Cholesky) is pinned against an independent numpy restatement here; the HIP
kernels (banded LDL^T) are compared with the oracle on the GPU box."""
import numpy as np
import pytest

import oracle


def _problem(rng, n, with_gaps=True):
    xy = np.cumsum(rng.uniform(20, 60, size=(n, 2)), axis=0)
    z = 100 + np.cumsum(rng.normal(0, 8, n))
    elev = 80 + 10 * np.sin(np.arange(n) / 5.0) + rng.normal(0, 2, n)
    if with_gaps:
        elev[rng.integers(0, n, max(1, n // 7))] = np.nan
    return np.column_stack([xy, z]), elev


def _np_hessian(n, xyz, ls, mcr):
    H = np.zeros((n, n))
    if n >= 3 and ls > 0:
        for i in range(1, n - 1):
            c = np.array([1.0, -2.0, 1.0])
            H[i - 1:i + 2, i - 1:i + 2] += ls * np.outer(c, c)
    if mcr > 0:
        for i in range(n - 1):
            d = np.hypot(*(xyz[i + 1, :2] - xyz[i, :2]))
            if d <= 1e-9 or d * mcr <= 1e-12:
                continue
            w = 1.0 / (d * mcr) ** 2
            H[i, i] += w; H[i + 1, i + 1] += w; H[i, i + 1] -= w; H[i + 1, i] -= w
    return H


def _np_optimize(xyz, elev, ls, lf, sd, mcr):
    n = len(elev)
    H = _np_hessian(n, xyz, ls, mcr)
    b = np.zeros(n)
    for i in range(n):
        if not np.isnan(elev[i]):
            H[i, i] += lf
            b[i] += lf * max(xyz[i, 2], elev[i] + sd)
        H[i, i] += 1e-8
    z = np.linalg.solve(H, b)
    for i in range(n):
        if not np.isnan(elev[i]):
            z[i] = max(z[i], elev[i] + sd)
    return z


def _np_global(zin, xyz, ls, mcr):
    n = len(zin)
    active = np.zeros(n, bool)
    z = zin.copy()
    solves = 0
    for _ in range(10):
        H = _np_hessian(n, xyz, ls, mcr)
        b = np.zeros(n)
        H[0, 0] += 1e10; b[0] += 1e10 * zin[0]
        H[-1, -1] += 1e10; b[-1] += 1e10 * zin[-1]
        for i in range(1, n - 1):
            if active[i]:
                H[i, i] += 1e8; b[i] += 1e8 * zin[i]
        H[np.arange(n), np.arange(n)] += 1e-8
        z = np.linalg.solve(H, b)
        solves += 1
        new = (z < zin - 1e-3) & ~active
        active |= new
        if not new.any():
            break
    return np.maximum(z, zin), solves


@pytest.mark.parametrize("n", [1, 2, 3, 7, 120])
def test_oracle_matches_numpy(n):
    rng = np.random.default_rng(n)
    xyz, elev = _problem(rng, n, with_gaps=n > 3)
    for (ls, lf, sd, mcr) in [(1.0, 0.5, 50.0, 2.0), (3.0, 2.0, 30.0, 0.5), (1.0, 0.0, 50.0, 2.0)]:
        if lf == 0.0 and n < 3:
            continue
        z = oracle.alt_optimize(xyz, elev, ls, lf, sd, mcr)
        assert np.allclose(z, _np_optimize(xyz, elev, ls, lf, sd, mcr), rtol=1e-7, atol=1e-6)
    zin = xyz[:, 2] + 5.0
    g, k = oracle.alt_global_smooth(zin, xyz, 1.0, 2.0)
    g2, k2 = _np_global(zin, xyz, 1.0, 2.0)
    assert k == k2
    assert np.allclose(g, g2, rtol=1e-6, atol=1e-4)


@pytest.mark.gpu
def test_hip_banded_ldlt_matches_oracle(csp):
    rng = np.random.default_rng(11)
    sizes = [1, 2, 3, 5, 40, 333, 64, 7, 900]
    probs = [_problem(rng, n, with_gaps=n > 3) for n in sizes]
    xyz = np.concatenate([p[0] for p in probs])
    elev = np.concatenate([p[1] for p in probs])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    for (ls, lf, sd, mcr) in [(1.0, 0.5, 50.0, 2.0), (3.0, 2.0, 30.0, 0.5)]:
        z = csp.alt_optimize_heights_batch(xyz, elev, off, ls, lf, sd, mcr)
        for i, n in enumerate(sizes):
            ref = oracle.alt_optimize(probs[i][0], probs[i][1], ls, lf, sd, mcr)
            assert np.allclose(z[off[i]:off[i + 1]], ref, rtol=1e-8, atol=1e-6), (n, ls)
    zin = xyz[:, 2] + rng.uniform(0, 10, len(xyz))
    g, solves = csp.alt_global_smooth_batch(zin, xyz, off, 1.0, 2.0)
    for i, n in enumerate(sizes):
        ref, k = oracle.alt_global_smooth(zin[off[i]:off[i + 1]], probs[i][0], 1.0, 2.0)
        assert solves[i] == k, (n, solves[i], k)
        assert np.allclose(g[off[i]:off[i + 1]], ref, rtol=1e-6, atol=1e-4), n
        assert (g[off[i]:off[i + 1]] >= zin[off[i]:off[i + 1]]).all()


@pytest.mark.gpu
def test_hip_lane_and_wave_kernels_agree(csp):
    """Fewer than 2048 problems run one wave per problem, more run one lane per problem: a large batch
    of small problems (lane kernels) must reproduce, problem by problem, what the same problems give
    when solved in small batches (wave kernels) -- identical arithmetic per row, so bit for bit."""
    rng = np.random.default_rng(12)
    B = 2300
    sizes = rng.integers(1, 70, size=B)
    sizes[:4] = [1, 2, 3, 200]
    probs = [_problem(rng, int(n), with_gaps=n > 3) for n in sizes]
    xyz = np.concatenate([p[0] for p in probs])
    elev = np.concatenate([p[1] for p in probs])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    zin = xyz[:, 2] + rng.uniform(0, 10, len(xyz))
    z_lane = csp.alt_optimize_heights_batch(xyz, elev, off, 1.0, 0.5, 50.0, 2.0)
    g_lane, s_lane = csp.alt_global_smooth_batch(zin, xyz, off, 1.0, 2.0)
    pick = [0, 1, 2, 3, 100, 1000, B - 1]
    sub_off = np.concatenate([[0], np.cumsum(sizes[pick])]).astype(np.int64)
    sel = np.concatenate([np.arange(off[i], off[i + 1]) for i in pick])
    z_wave = csp.alt_optimize_heights_batch(xyz[sel], elev[sel], sub_off, 1.0, 0.5, 50.0, 2.0)
    g_wave, s_wave = csp.alt_global_smooth_batch(zin[sel], xyz[sel], sub_off, 1.0, 2.0)
    assert np.array_equal(z_wave, z_lane[sel])
    assert np.array_equal(g_wave, g_lane[sel])
    assert np.array_equal(s_wave, s_lane[pick])


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 300])
def test_banded_oracle_matches_dense_oracle(n):
    """The O(n) banded Cholesky of oracle/alt_oracle.c (the one-core CPU time of bench.py's single_altitude record) fills its
    three bands by the same triplet rules as the dense matrix: same solutions to rounding, same number of active-set solves."""
    rng = np.random.default_rng(40 + n)
    xyz, elev = _problem(rng, n)
    for lf in (0.0, 0.5):
        a = oracle.alt_optimize(xyz, elev, 1.0, lf, 50.0, 2.0)
        b = oracle.alt_optimize(xyz, elev, 1.0, lf, 50.0, 2.0, banded=True)
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(a)))
    (za, na), (zb, nb) = oracle.alt_global_smooth(xyz[:, 2], xyz, 1.0, 2.0), oracle.alt_global_smooth(xyz[:, 2], xyz, 1.0, 2.0, banded=True)
    assert na == nb and np.max(np.abs(za - zb)) <= 1e-9 * max(1.0, np.max(np.abs(za)))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 777, 2000, 2900, 3001, 20000])
def test_hip_cyclic_reduction_for_one_long_problem(csp, n):
    """A single long problem (the reference's own call pattern, uavPathPlanning.cpp:1670-1676, :1796-1799) runs block cyclic
    reduction, one 1024-thread workgroup, block rows in LDS up to ~2900 samples and in the workspace beyond: against the CPU
    oracle (dense Cholesky up to 777 samples, the banded one beyond) at 1e-8 relative, host- and device-memory forms, and a few
    problems of different lengths in one call."""
    import torch
    rng = np.random.default_rng(50 + n)
    xyz, elev = _problem(rng, n)
    off = np.array([0, n], dtype=np.int64)
    banded = n > 777
    for lf in (0.0, 0.5):
        ref = oracle.alt_optimize(xyz, elev, 1.0, lf, 50.0, 2.0, banded=banded)
        got = csp.alt_optimize_heights_batch(xyz, elev, off, 1.0, lf, 50.0, 2.0)
        assert np.max(np.abs(got - ref)) <= 1e-8 * np.max(np.abs(ref)), (n, lf)
    zref, nref = oracle.alt_global_smooth(xyz[:, 2], xyz, 1.0, 2.0, banded=banded)
    z, solves = csp.alt_global_smooth_batch(xyz[:, 2].copy(), xyz, off, 1.0, 2.0)
    assert int(solves[0]) == nref
    assert np.max(np.abs(z - zref)) <= 1e-8 * np.max(np.abs(zref))
    # three problems of different lengths, device memory
    lens = [n, 600, n // 2 + 300]
    ps = [_problem(rng, m) for m in lens]
    xyz3, elev3 = np.concatenate([p[0] for p in ps]), np.concatenate([p[1] for p in ps])
    off3 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    d = [torch.from_numpy(x).cuda() for x in (xyz3, elev3, off3)]
    got3 = csp.alt_optimize_heights_batch(d[0], d[1], d[2], 1.0, 0.5, 50.0, 2.0)
    torch.cuda.synchronize()
    got3 = got3.cpu().numpy()
    for k, (pxyz, pelev) in enumerate(ps):
        ref = oracle.alt_optimize(pxyz, pelev, 1.0, 0.5, 50.0, 2.0, banded=True)
        assert np.max(np.abs(got3[off3[k]:off3[k + 1]] - ref)) <= 1e-8 * np.max(np.abs(ref)), (n, k)
