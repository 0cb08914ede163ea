"""Generates tests/golden/F8_kkt_mpmath.json -- TEST INFRASTRUCTURE.   Run in the dev container:  python oracle/gen_golden_kkt.py

An INDEPENDENT yardstick for the unpenalised / zero-velocity-penalised solve: not a third restatement of the reference's
closed form (dense M, Q, C_T, R = C^T M^-T Q M^-1 C, minimum_snap.cpp:247-592) but the constrained quadratic programme that
closed form solves, written down directly and solved through its KKT system in 60-digit arithmetic (mpmath):

    minimise   sum_k  int_0^{T_k} (p_k^(o)(t))^2 dt  [ + w (p_k'(0)^2 + p_k'(T_k)^2) ]       (minimum_snap.cpp:312-330, :473-509)
    subject to p_k(0) = w_k, p_k(T_k) = w_{k+1};  p_k^(j)(T_k) = p_{k+1}^(j)(0), j = 1..o-1 at interior waypoints
               (what the selection matrix C_T encodes, :268-310);  the start / end derivatives j = 1..o-1 pinned to
               (v, a, 0, ...) (:527-555).

No M inverse, no selection matrix, no block elimination: if the oracle's and the HIP kernels' coefficients agree with THIS,
the algebra of rows A4-A10 is right.  It says nothing about Eigen's rounding: PARITY STAYS UNPINNED with
respect to the real Eigen build (no Eigen in the image, no coefficient goldens in the reference).
Coefficients are stored rounded to double (hex), highest power first, local time -- the PolyCoeff convention (:220-223).
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import numpy_ref as nr  # noqa: E402
from oracle.gen_golden_r3 import merge_close_waypoints  # noqa: E402
from tests import synth  # noqa: E402

mp.mp.dps = 60
OUT = os.path.join(HERE, "..", "tests", "golden")


def ff(a, j):
    """falling factorial a (a-1) .. (a-j+1) as an exact integer"""
    r = 1
    for q in range(j):
        r *= a - q
    return r


def kkt_solve(order, path, vel, acc, time, vel_zero_weight=0.0, path_weight=0.0):
    """path_weight > 0: the reference's two-stage definition (minimum_snap.cpp:347-469) in the same direct form -- stage 1: the
    QP WITHOUT either penalty (Q_original, :349), its polynomial sampled at t = T s / 16, s = 0..16, first sample of maximal
    squared distance to the chord wins (strict >, :408-439); stage 2: the QP with w phi(t*) phi(t*)^T added per segment and the
    linear term f = -2 w L(t*) phi(t*) taken UN-halved, i.e. minimise c^T Q' c + 2 f^T c (the stationarity the reference
    solves, :577-579).  Returns (coefficients, t* sample indices)."""
    if path_weight:
        _, pre = kkt_solve(order, path, vel, acc, time, 0.0, 0.0)     # the pre-solve's coefficients at full precision
        return _kkt(order, path, vel, acc, time, vel_zero_weight, path_weight, pre)
    return _kkt(order, path, vel, acc, time, vel_zero_weight, 0.0, None)


def _kkt(order, path, vel, acc, time, vel_zero_weight, path_weight, pre):
    o, m, S = order, 2 * order, len(time)
    T = [mp.mpf(float(t)) for t in time]
    n = m * S

    def deriv_row(k, j, t):
        """row vector: p_k^(j)(t) as a linear function of the n coefficients (highest power first)"""
        row = [mp.mpf(0)] * n
        for i in range(m):
            a = m - 1 - i
            if a >= j:
                row[k * m + i] = ff(a, j) * (t ** (a - j) if a - j > 0 else mp.mpf(1))
        return row

    Q = mp.zeros(n, n)
    for k in range(S):
        for i in range(m):
            for l in range(m):
                a, b = m - 1 - i, m - 1 - l
                if a >= o and b >= o:
                    e = a + b - 2 * o + 1
                    Q[k * m + i, k * m + l] += mp.mpf(ff(a, o) * ff(b, o)) / e * T[k] ** e
        if vel_zero_weight:
            for t in (mp.mpf(0), T[k]):
                r = deriv_row(k, 1, t)
                for i in range(m):
                    for l in range(m):
                        Q[k * m + i, k * m + l] += mp.mpf(vel_zero_weight) * r[k * m + i] * r[k * m + l]
    lin = [[mp.mpf(0)] * n for _ in range(3)]       # linear term f per axis
    tstar = [0] * S
    if path_weight:
        w = mp.mpf(path_weight)
        for k in range(S):
            best, bs = mp.mpf(-1), 0
            for sidx in range(17):
                tt = T[k] * sidx / 16
                d2 = mp.mpf(0)
                for ax in range(3):
                    pt = sum(pre[k][ax][i] * (tt ** (m - 1 - i) if m - 1 - i > 0 else 1) for i in range(m))
                    L = mp.mpf(float(path[k, ax])) + (tt / T[k]) * (mp.mpf(float(path[k + 1, ax])) - mp.mpf(float(path[k, ax])))
                    d2 += (pt - L) ** 2
                if d2 > best:
                    best, bs = d2, sidx
            tstar[k] = bs
            tt = T[k] * bs / 16
            phi = deriv_row(k, 0, tt)
            for i in range(m):
                for l in range(m):
                    Q[k * m + i, k * m + l] += w * phi[k * m + i] * phi[k * m + l]
            for ax in range(3):
                L = mp.mpf(float(path[k, ax])) + (tt / T[k]) * (mp.mpf(float(path[k + 1, ax])) - mp.mpf(float(path[k, ax])))
                for i in range(m):
                    lin[ax][k * m + i] = -2 * w * L * phi[k * m + i]
    out = np.zeros((S, 3, m))
    exact = [[[None] * m for _ in range(3)] for _ in range(S)]
    for ax in range(3):
        rows, rhs = [], []
        for k in range(S):
            rows.append(deriv_row(k, 0, mp.mpf(0))); rhs.append(mp.mpf(float(path[k, ax])))
            rows.append(deriv_row(k, 0, T[k])); rhs.append(mp.mpf(float(path[k + 1, ax])))
        for j in range(1, o):
            bstart = float(vel[0, ax]) if j == 1 else float(acc[0, ax]) if j == 2 else 0.0
            bend = float(vel[1, ax]) if j == 1 else float(acc[1, ax]) if j == 2 else 0.0
            rows.append(deriv_row(0, j, mp.mpf(0))); rhs.append(mp.mpf(bstart))
            rows.append(deriv_row(S - 1, j, T[S - 1])); rhs.append(mp.mpf(bend))
            for k in range(S - 1):
                ra, rb = deriv_row(k, j, T[k]), deriv_row(k + 1, j, mp.mpf(0))
                rows.append([x - y for x, y in zip(ra, rb)]); rhs.append(mp.mpf(0))
        nc = len(rows)
        K = mp.zeros(n + nc, n + nc)
        for i in range(n):
            for l in range(n):
                K[i, l] = 2 * Q[i, l]
        for c in range(nc):
            for i in range(n):
                K[i, n + c] = rows[c][i]
                K[n + c, i] = rows[c][i]
        b = mp.zeros(n + nc, 1)
        for i in range(n):
            b[i] = -2 * lin[ax][i]
        for c in range(nc):
            b[n + c] = rhs[c]
        x = mp.lu_solve(K, b)
        for k in range(S):
            for i in range(m):
                out[k, ax, i] = float(x[k * m + i])
                exact[k][ax][i] = x[k * m + i]
    return out, (exact if pre is None else tstar)


def hx(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def main():
    rng = np.random.default_rng(20260503)
    cases = []

    def add(name, o, path, time, vel=None, acc=None, vw=0.0, pw=0.0, note=""):
        vel = np.zeros((2, 3)) if vel is None else vel
        acc = np.zeros((2, 3)) if acc is None else acc
        co, extra = kkt_solve(o, path, vel, acc, time, vw, pw)
        ref, md = nr.solve_qp_closed_form(o, path, vel, acc, time, pw, vw)
        ref = np.asarray(ref).reshape(co.shape)
        err = synth.rel_err_per_power(ref, co)
        c = {"name": name, "note": note, "order": o, "segments": len(time), "path_weight": pw, "vel_zero_weight": vw,
             "path": hx(path), "time": hx(time), "vel": hx(vel), "acc": hx(acc), "coeff": hx(co), "max_dev": float(md).hex(),
             "cond_M": float(np.linalg.cond(nr.build_M(o, time))), "numpy_ref_per_power_err": err}
        if pw:
            c["tstar_samples"] = [int(v) for v in extra]
        cases.append(c)
        print("%-28s o=%d S=%d  numpy closed form vs 60-digit KKT, per power: %.2e" % (name, o, len(time), err))

    for o in (2, 3, 4, 5):
        for S in (1, 3, 8):
            p0 = rng.uniform(-10, 10, size=(1, 3))
            path = np.concatenate([p0, p0 + np.cumsum(rng.normal(size=(S, 3)), axis=0)])
            time = rng.uniform(0.5, 2.0, size=S)
            bcs = rng.normal(size=(4, 3)) if S > 1 else np.zeros((4, 3))
            add("kkt_o%d_s%d" % (o, S), o, path, time, vel=bcs[:2], acc=bcs[2:], vw=0.05 if S == 3 else 0.0, note="well-scaled random walk")
    for o, S, pw in ((2, 3, 0.5), (2, 8, 1e-2), (3, 3, 0.5), (3, 8, 1e-2), (4, 3, 0.5), (4, 8, 1e-2), (4, 6, 2.0)):
        p0 = rng.uniform(-10, 10, size=(1, 3))
        path = np.concatenate([p0, p0 + np.cumsum(rng.normal(size=(S, 3)), axis=0)])
        time = rng.uniform(0.5, 2.0, size=S)
        bcs = rng.normal(size=(4, 3))
        add("kkt_pen_o%d_s%d" % (o, S), o, path, time, vel=bcs[:2], acc=bcs[2:], vw=0.02, pw=pw,
            note="path penalty: two-stage definition, un-halved linear term; t* from the 60-digit pre-solve")
    wp, tm = synth.make_batch(1, 16, config_id=3)
    add("kkt_c3_row0", 4, wp[0], tm[0], note="first trajectory of the headline workload C3")
    P = merge_close_waypoints(synth.README_UAV31_ENU)
    for o in (2, 3):
        add("kkt_uav31_merged_o%d_V200" % o, o, P, nr.time_allocation(P, 200.0, 1.0), vw=0.01 if o == 2 else 0.0,
            note="README flight after getPlan's merge; km-scale legs (ill-conditioned in the closed form)")
    json.dump({"fixture": "F8", "cases": cases}, open(os.path.join(OUT, "F8_kkt_mpmath.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
