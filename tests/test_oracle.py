"""The CPU oracle against the golden fixtures, the independent numpy restatement, the 80-bit
long-double build of itself and the known-answer tests of SURVEY.md §8c (K1..K8).

PARITY UNPINNED with respect to the real Eigen build (no Eigen in the image; the reference ships
no coefficient goldens) -- see oracle/dense_oracle.c.
"""
import numpy as np
import pytest

import oracle
from oracle import numpy_ref as nr
from tests import synth
from tests.conftest import load_cases

ALL = ["F1_kat.json", "F2_readme_uav31.json", "F2b_readme_uav31_merged.json", "F8_kkt_mpmath.json", "F3_wellscaled.json", "F5_ragged.json", "F6_penalties.json"]


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("fname", ALL)
def test_c_oracle_matches_golden(fname):
    for c in load_cases(fname):
        got, md = oracle.solve(c["order"], c["path"], c["vel"], c["acc"], c["time"], c["path_weight"], c["vel_zero_weight"])
        got = got.reshape(c["coeff"].shape)
        if c.get("cond_M", 1.0) > 1e12:
            # both restatements are rounding-dominated here; compare each to the long-double build
            ld, _ = oracle.solve(c["order"], c["path"], c["vel"], c["acc"], c["time"], c["path_weight"],
                                 c["vel_zero_weight"], long_double=True)
            assert _rel(got, ld.reshape(got.shape)) < 1e-6, c["name"]
            assert _rel(c["coeff"], ld.reshape(got.shape)) < 1e-6, c["name"]
        else:
            tol = 1e-7 if c["order"] == 5 else 1e-9
            assert _rel(got, c["coeff"]) < tol, (c["name"], _rel(got, c["coeff"]))
        assert abs(md - c["max_dev"]) < 1e-7 * max(1.0, abs(c["max_dev"]))


def test_k1_known_answers():
    kat = {2: [-2, 3, 0, 0], 3: [6, -15, 10, 0, 0, 0], 4: [-20, 70, -84, 35, 0, 0, 0, 0]}
    for o, c in kat.items():
        got, _ = oracle.solve(o, [[0, 0, 0], [1, 1, 1]], np.zeros((2, 3)), np.zeros((2, 3)), [1.0])
        assert np.max(np.abs(got.reshape(3, -1) - np.array(c, dtype=float))) < 1e-12


def test_selection_matrix_matches_literal_cascade():
    """selection_column() (waypoint picture) vs. the reference's literal branch cascade, exercised
    through full solves with distinct boundary values on every slot."""
    rng = np.random.default_rng(5)
    for o in (1, 2, 3, 4, 5):
        for S in (1, 2, 3, 5):
            p = rng.normal(size=(S + 1, 3))
            t = rng.uniform(0.5, 2, S)
            v, a = rng.normal(size=(2, 3)), rng.normal(size=(2, 3))
            c1, _ = oracle.solve(o, p, v, a, t)
            c2, _ = nr.solve_qp_closed_form(o, p, v, a, t)
            # order 5 is rounding-dominated in the dense formulation (cond ~1e9); this test is
            # about index algebra, a wrong column would show as an O(1) difference
            assert _rel(c1, c2) < (1e-5 if o == 5 else 1e-8), (o, S)


def _poly_deriv(c, t, j):
    m = len(c)
    out = 0.0
    for i, ci in enumerate(c):
        p = m - 1 - i
        if p >= j:
            f = 1.0
            for q in range(p, p - j, -1):
                f *= q
            out += ci * f * t ** (p - j)
    return out


def test_k2_continuity_and_k3_boundary():
    wp, tm = synth.make_batch(4, 8, config_id=77)
    for b in range(4):
        c, _ = oracle.solve(4, wp[b], np.zeros((2, 3)), np.zeros((2, 3)), tm[b])
        c = c.reshape(8, 3, 8)
        for k in range(7):
            for ax in range(3):
                for j in range(0, 7):   # derivatives 0..2o-2 continuous, 2o-1 jumps
                    a = _poly_deriv(c[k, ax], tm[b, k], j)
                    bb = _poly_deriv(c[k + 1, ax], 0.0, j)
                    assert abs(a - bb) < 1e-6 * max(1.0, abs(a)), (k, ax, j)
        for ax in range(3):
            for j in (1, 2, 3):
                assert abs(_poly_deriv(c[0, ax], 0.0, j)) < 1e-9
                assert abs(_poly_deriv(c[7, ax], tm[b, 7], j)) < 1e-6


def test_k5_time_reversal_and_k6_axis_permutation():
    wp, tm = synth.make_batch(1, 6, config_id=78)
    z = np.zeros((2, 3))
    c, _ = oracle.solve(4, wp[0], z, z, tm[0])
    cr, _ = oracle.solve(4, wp[0][::-1].copy(), z, z, tm[0][::-1].copy())
    c, cr = c.reshape(6, 3, 8), cr.reshape(6, 3, 8)
    for k in range(6):
        for ax in range(3):
            for s in (0.1, 0.5, 0.9):
                T = tm[0, k]
                assert abs(_poly_deriv(c[k, ax], s * T, 0) - _poly_deriv(cr[5 - k, ax], (1 - s) * T, 0)) < 1e-8
    cp, _ = oracle.solve(4, wp[0][:, [1, 2, 0]].copy(), z, z, tm[0])
    assert np.array_equal(cp.reshape(6, 3, 8), c[:, [1, 2, 0], :])


def test_k7_vel_zero_penalty_is_a_diagonal_shift():
    """A8: in derivative space the penalty adds 2w to every interior velocity diagonal of R_PP and
    nothing else -- checked by rebuilding R both ways with the numpy restatement."""
    o, S, w = 4, 5, 0.37
    T = np.array([0.7, 1.3, 0.9, 1.8, 1.1])
    M, CT, Q = nr.build_M(o, T), nr.build_CT(o, S), nr.build_Q(o, T)
    Mi = np.linalg.inv(M)
    R0 = CT.T @ Mi.T @ Q @ Mi @ CT
    m = 2 * o
    V = np.zeros_like(Q)
    for k in range(S):
        for t in (0.0, T[k]):
            pd = np.array([(m - 1 - i) * t ** (m - 2 - i) if m - 2 - i > 0 else (float(m - 1 - i) if m - 2 - i == 0 else 0.0) for i in range(m)])
            V[k * m:(k + 1) * m, k * m:(k + 1) * m] += np.outer(pd, pd)
    R1 = CT.T @ Mi.T @ (Q + w * V) @ Mi @ CT
    F = 2 * o + S - 1
    D = (R1 - R0)[F:, F:]
    exp = np.zeros_like(D)
    for k in range(S - 1):
        exp[k * (o - 1), k * (o - 1)] = 2 * w
    assert np.max(np.abs(D - exp)) < 1e-6
    assert np.max(np.abs((R1 - R0)[:F, F:])) < 1e-6   # R_FP untouched


def test_k8_path_weight_quirks():
    """First-max tie rule and the un-halved linear term (A7): a straight-line trajectory has zero
    deviation at every sample, so t* stays at sample 0 and the penalised solve still interpolates."""
    p = np.array([[0, 0, 0], [1, 2, 3], [2, 4, 6], [3, 6, 9.0]])
    t = np.array([1.0, 1.0, 1.0])
    z = np.zeros((2, 3))
    c, md = oracle.solve(3, p, z, z, t, path_weight=0.5)
    c2, md2 = nr.solve_qp_closed_form(3, p, z, z, t, 0.5)
    assert _rel(c, c2) < 1e-9
    assert abs(md - md2) < 1e-9


def test_time_alloc_and_generate_trajectory_match_numpy():
    P = synth.README_UAV31_ENU
    for v, mt_ in ((200.0, 1.0), (30.0, 1.0), (0.0, 0.3)):
        assert np.allclose(oracle.time_alloc(P, v, mt_), nr.time_allocation(P, v, mt_), rtol=0, atol=0)
    cfg = {"order": 2, "path_weight": 1e-7, "vel_zero_weight": 0.01, "V_avg": 200.0, "min_time_s": 1.0,
           "sample_distance": 300.0}
    s_np, info_np = nr.generate_trajectory(P, cfg)
    s_c, info_c = oracle.generate_trajectory(P, order=2, path_weight=1e-7, vel_zero_weight=0.01, v_avg=200.0,
                                             min_time_s=1.0, sample_distance=300.0)
    assert s_np.shape == s_c.shape
    assert np.max(np.abs(s_np - s_c)) < 1e-6
    assert info_np["iters"] == info_c["iters"]
    # bad shape -> empty (minimum_snap.cpp:54-57)
    e, _ = oracle.generate_trajectory(P[:1], order=2)
    assert e.size == 0


@pytest.mark.parametrize("order", [1, 2, 3, 4, 5])
def test_structured_cpu_solver_matches_the_dense_oracle(order):
    """oracle/structured_oracle.cpp (block-tridiagonal LDL^T on the CPU, the 'honest CPU' timing line of
    bench.py) against the dense restatement: boundary conditions, zero-velocity weight, S = 1..16."""
    import oracle
    rng = np.random.default_rng(40 + order)
    for S in (1, 2, 5, 16):
        wp, tm = synth.make_batch(7, S, config_id=3)
        bc = rng.normal(size=(7, 4, 3))
        for vw in (0.0, 0.11):
            a = oracle.struct_solve_batch(order, wp, tm, bc, vel_zero_weight=vw, nthreads=2)
            b, _ = oracle.solve_batch(order, wp, tm, bc, vel_zero_weight=vw)
            tol = 1e-4 if order == 5 else 1e-8      # the dense order-5 solve carries 1e-6 of its own rounding at S = 16
            assert synth.rel_err(a, b) < tol, (order, S, vw)
    one = oracle.struct_solve_batch(order, wp[:1], tm[:1], None)
    ld, _ = oracle.solve_batch(order, wp[:1], tm[:1], long_double=True)
    assert synth.rel_err(one, ld) < 1e-9


def test_closed_form_is_the_minimiser_of_the_constrained_qp():
    """F8 (oracle/gen_golden_kkt.py): the constrained QP behind the reference's closed form -- snap integral [+ zero-velocity
    penalty, + the path penalty in its two-stage form with the un-halved linear term, minimum_snap.cpp:347-469, :577-579] subject to interpolation, continuity of derivatives 1..o-1 and pinned end derivatives -- solved DIRECTLY through
    its KKT system in 60-digit arithmetic, no M inverse, no selection matrix.  Both restatements of the closed form (numpy,
    C) and the 80-bit build agree with it per power to their own rounding: the algebra of SURVEY.md rows A4-A10 is the QP's.
    (Says nothing about Eigen's rounding: parity stays unpinned.)"""
    worst = {"numpy": 0.0, "c_f64": 0.0, "c_ld": 0.0}
    for c in load_cases("F8_kkt_mpmath.json"):
        S, m = c["segments"], 2 * c["order"]
        kkt = c["coeff"].reshape(S, 3, m)
        pw = c["path_weight"]     # > 0: the two-stage definition with the un-halved linear term, t* from the 60-digit pre-solve
        npy, _ = nr.solve_qp_closed_form(c["order"], c["path"], c["vel"], c["acc"], c["time"], pw, c["vel_zero_weight"])
        f64, _ = oracle.solve(c["order"], c["path"], c["vel"], c["acc"], c["time"], pw, c["vel_zero_weight"])
        ld, _ = oracle.solve(c["order"], c["path"], c["vel"], c["acc"], c["time"], pw, c["vel_zero_weight"], long_double=True)
        e = {"numpy": synth.rel_err_per_power(np.asarray(npy).reshape(kkt.shape), kkt), "c_f64": synth.rel_err_per_power(f64.reshape(kkt.shape), kkt),
             "c_ld": synth.rel_err_per_power(ld.reshape(kkt.shape), kkt)}
        tol = 1e-6 if c["order"] == 5 else (1e-9 if c["cond_M"] < 1e12 else 1e-6)
        assert e["numpy"] < tol and e["c_f64"] < tol, (c["name"], e)
        assert e["c_ld"] < (1e-9 if c["order"] == 5 else 1e-11) or c["cond_M"] >= 1e12, (c["name"], e)
        for k in worst:
            worst[k] = max(worst[k], e[k])
    print("closed form vs the 60-digit KKT solution, worst per-power error:", worst)
