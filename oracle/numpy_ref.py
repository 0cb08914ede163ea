"""Dense numpy restatement of the reference's closed-form minimum-snap solve.

TEST INFRASTRUCTURE ONLY -- imported by tests/, oracle/gen_golden.py and nothing else.
PARITY UNPINNED: the reference ships no golden coefficient vectors and cannot be compiled
here (Eigen is absent), so this file and oracle/dense_oracle.c are two independently
written restatements (numpy/LAPACK vs. hand-written C LU) that pin each other; neither is
pinned by an output of the real Eigen build.

Follows /root/reference/math_util/minimum_snap.cpp:
  * solve_qp_closed_form  <- SolveQPClosedForm            (:227-649)
  * time_allocation       <- GenerateTrajectoryMatrix      (:59-72)
  * generate_trajectory   <- GenerateTrajectoryMatrix      (:22-206)
All dense inverses use numpy.linalg.inv (LAPACK getrf/getri = partial-pivot LU, the same
algorithm class Eigen's dynamic-size MatrixXd::inverse() uses).
"""
import math
import numpy as np


def _fact(x):
    # :15-20 (C int; x <= 12 in every supported order)
    f = 1
    for i in range(x, 0, -1):
        f *= i
    return f


def build_M(o, T):
    # :247-266
    m = 2 * o
    S = len(T)
    M = np.zeros((S * m, S * m))
    for s in range(S):
        for j in range(o):
            for k in range(j, m):
                c = _fact(k) // _fact(k - j)
                M[s * m + j, s * m + m - 1 - k] = c * (0.0 ** (k - j) if k > j else 1.0)
                M[s * m + j + o, s * m + m - 1 - k] = c * T[s] ** (k - j)
    return M


def build_CT(o, S):
    # :268-310, literal branch order
    m = 2 * o
    N = m * S
    V = (S + 1) * o
    F = 2 * o + (S - 1)
    C = np.zeros((N, V))
    for i in range(N):
        if i < o:
            C[i, i] = 1
        elif i >= N - o:
            C[i, F - o + (i - (N - o))] = 1
        elif i % o == 0 and (i // o) % 2 == 1:
            C[i, i // (2 * o) + o] = 1
        elif i % o == 0 and (i // o) % 2 == 0:
            C[i, i // (2 * o) + o - 1] = 1
        elif i % o != 0 and (i // o) % 2 == 1:
            t0 = i // (2 * o) * (2 * o) + o
            t1 = i // (2 * o) * (o - 1) + i - t0 - 1
            C[i, F + t1] = 1
        else:
            t0 = (i - o) // (2 * o) * (2 * o) + o
            t1 = (i - o) // (2 * o) * (o - 1) + (i - o) - t0 - 1
            C[i, F + t1] = 1
    return C


def build_Q(o, T):
    # :312-330 (int prefactor incl. the int division, then * pow)
    m = 2 * o
    p = m - 1
    S = len(T)
    Q = np.zeros((S * m, S * m))
    for s in range(S):
        for i in range(m):
            for l in range(m):
                if m - i <= o or m - l <= o:
                    continue
                pre = (_fact(p - i) // _fact(p - o - i)) * (_fact(p - l) // _fact(p - o - l))
                e = p - i + p - l - (2 * o - 1)
                pre = pre // e  # C int division (exact for o <= 5)
                Q[s * m + i, s * m + l] = pre * T[s] ** e
    return Q


def _fill_fixed(o, S, path, vel, acc, axis, V, F):
    # :526-562
    d = np.zeros(V)
    d[0] = path[0, axis]
    if o >= 2:
        d[1] = vel[0, axis]
        d[F - o + 1] = vel[1, axis]
    if o >= 3:
        d[2] = acc[0, axis]
        d[F - o + 2] = acc[1, axis]
    d[F - o] = path[S, axis]
    for k in range(1, S):
        d[o + k - 1] = path[k, axis]
    return d


def _solve_axes(o, S, R, CT, Minv, path, vel, acc, f_valid):
    V = (S + 1) * o
    F = 2 * o + (S - 1)
    RPP = R[F:, F:]
    RFP = R[:F, F:]
    out = []
    for axis in range(3):
        d = _fill_fixed(o, S, path, vel, acc, axis, V, F)
        if V > F:
            rhs = RFP.T @ d[:F]
            if f_valid is not None:
                rhs = rhs + f_valid[axis][F:]
            d[F:] = -np.linalg.inv(RPP) @ rhs
        out.append(Minv @ (CT @ d))
    return out


def solve_qp_closed_form(order, path, vel, acc, time, path_weight=0.0, vel_zero_weight=0.0):
    """Returns (PolyCoeff [S, 3*2o], max_deviation).  path [S+1,3], vel/acc [2,3], time [S]."""
    o = int(order)
    path = np.asarray(path, dtype=np.float64)
    vel = np.asarray(vel, dtype=np.float64)
    acc = np.asarray(acc, dtype=np.float64)
    T = np.asarray(time, dtype=np.float64)
    S = len(T)
    m = 2 * o
    p_order = m - 1
    M = build_M(o, T)
    CT = build_CT(o, S)
    Q = build_Q(o, T)
    Minv = np.linalg.inv(M)
    MTinv = np.linalg.inv(M.T)

    best_t = np.zeros(S)
    f_coeff = None
    if path_weight > 0.0:
        # :347-469 pre-solve with the unpenalised Q, 17-sample argmax, rank-1 penalty
        R0 = CT.T @ MTinv @ Q @ Minv @ CT
        P0 = _solve_axes(o, S, R0, CT, Minv, path, vel, acc, None)
        A = np.zeros_like(Q)
        f_coeff = [np.zeros(S * m) for _ in range(3)]
        for k in range(S):
            Tk = T[k]
            bt, bd = 0.0, -1.0
            for s in range(17):
                tt = Tk * float(s) / 16.0
                phi = np.array([tt ** (p_order - i) if (p_order - i) > 0 else 1.0 for i in range(m)])
                pt = np.array([phi @ P0[a][k * m:(k + 1) * m] for a in range(3)])
                L = path[k] + (tt / Tk) * (path[k + 1] - path[k])
                d2 = float(np.sum((pt - L) ** 2))
                if d2 > bd:
                    bd, bt = d2, tt
            phi = np.array([bt ** (p_order - i) if (p_order - i) > 0 else 1.0 for i in range(m)])
            A[k * m:(k + 1) * m, k * m:(k + 1) * m] = np.outer(phi, phi)
            Lb = path[k] + (bt / Tk) * (path[k + 1] - path[k])
            for a in range(3):
                f_coeff[a][k * m:(k + 1) * m] = -2.0 * phi * Lb[a] * path_weight
            best_t[k] = bt
        Q = Q + path_weight * A

    if vel_zero_weight > 0.0:
        # :473-509
        Vm = np.zeros_like(Q)
        for k in range(S):
            for t in (0.0, T[k]):
                pd = np.zeros(m)
                for i in range(m):
                    power = p_order - i - 1
                    if power < 0:
                        pd[i] = 0.0
                    elif power == 0:
                        pd[i] = float(p_order - i)
                    else:
                        pd[i] = float(p_order - i) * t ** power
                Vm[k * m:(k + 1) * m, k * m:(k + 1) * m] += np.outer(pd, pd)
        Q = Q + vel_zero_weight * Vm

    R = CT.T @ MTinv @ Q @ Minv @ CT  # :511
    f_valid = None
    if path_weight > 0.0:
        f_valid = [CT.T @ MTinv @ f_coeff[a] for a in range(3)]  # :517-522
    P = _solve_axes(o, S, R, CT, Minv, path, vel, acc, f_valid)  # :524-592

    # :594-624 deviation metric at the recorded t*
    max_dev = 0.0
    for k in range(S):
        bt = best_t[k]
        phi = np.array([bt ** (p_order - i) if (p_order - i) > 0 else 1.0 for i in range(m)])
        pt = np.array([phi @ P[a][k * m:(k + 1) * m] for a in range(3)])
        Lb = path[k] + (bt / T[k]) * (path[k + 1] - path[k])
        dist = math.sqrt(float(np.sum((pt - Lb) ** 2)))
        seg_len = float(np.linalg.norm(path[k + 1] - path[k]))
        ratio = dist / seg_len if seg_len > 1e-6 else 0.0
        max_dev = max(max_dev, ratio)

    coeff = np.zeros((S, 3 * m))  # :626-648
    for k in range(S):
        for a in range(3):
            coeff[k, a * m:(a + 1) * m] = P[a][k * m:(k + 1) * m]
    return coeff, max_dev


def time_allocation(path, v_avg, min_time_s):
    # :63-72
    path = np.asarray(path, dtype=np.float64)
    S = path.shape[0] - 1
    T = np.zeros(S)
    for i in range(S):
        ln = math.sqrt(float(np.sum((path[i + 1] - path[i]) ** 2)))
        t = ln / v_avg if v_avg > 1e-6 else min_time_s
        T[i] = max(t, min_time_s)
    return T


def generate_trajectory(path, cfg, sample_distance_override=-1.0, v_avg_override=-1.0):
    """GenerateTrajectoryMatrix (:22-206).  cfg: dict with the MinimumSnapConfig fields
    (minimum_snap.hpp:9-33).  Returns (samples [N,3], info dict)."""
    path = np.asarray(path, dtype=np.float64)
    o = int(cfg.get("order", 3))
    v_avg = float(cfg.get("V_avg", 5.0))
    min_t = float(cfg.get("min_time_s", 0.1))
    sd = float(cfg.get("sample_distance", 1.0))
    vel = np.array([cfg.get("start_vel", [0, 0, 0]), cfg.get("end_vel", [0, 0, 0])], dtype=np.float64)
    acc = np.array([cfg.get("start_acc", [0, 0, 0]), cfg.get("end_acc", [0, 0, 0])], dtype=np.float64)
    pw = float(cfg.get("path_weight", 0.0))
    vw = float(cfg.get("vel_zero_weight", 0.0))
    if sample_distance_override > 0.0:
        sd = sample_distance_override
    if v_avg_override > 0.0:
        v_avg = v_avg_override
    if path.ndim != 2 or path.shape[0] < 2 or path.shape[1] < 3:
        return np.zeros((0, 0)), {}
    S = path.shape[0] - 1
    T = time_allocation(path, v_avg, min_t)
    it = 0
    while True:  # :80-90
        coeff, max_dev = solve_qp_closed_form(o, path, vel, acc, T, pw, vw)
        if max_dev > 0.2 and it < 10:
            vw = 0.01 if vw < 1e-6 else vw * 2.0
            it += 1
        else:
            break
    m = 2 * o

    def ev(seg, t):
        pt = np.zeros(3)
        for dim in range(3):
            c = coeff[seg, dim * m:(dim + 1) * m]
            val = 0.0
            for k in range(m):
                val += c[k] * math.pow(t, m - 1 - k)
            pt[dim] = val
        return pt

    samples = []
    for seg in range(S):  # :123-161
        Ts = T[seg]
        dt = 0.1
        if dt > Ts / 10.0:
            dt = Ts / 10.0
        t0 = ev(seg, 0.0)
        if not samples:
            samples.append(t0)
        prev = t0
        t = dt
        while t <= Ts + 1e-12:
            tt = min(t, Ts)
            cur = ev(seg, tt)
            if float(np.linalg.norm(cur - prev)) >= sd:
                prev = cur
                samples.append(cur)
            t += dt
        if seg == S - 1:
            endpt = ev(seg, Ts)
            if not samples or float(np.linalg.norm(samples[-1] - endpt)) > 1e-6:
                samples.append(endpt)
    out = np.array(samples).reshape(-1, 3)
    # :163-195 stats
    max_climb, min_r = 0.0, 1.0e12
    for i in range(len(out) - 1):
        dx, dy = out[i + 1, 0] - out[i, 0], out[i + 1, 1] - out[i, 1]
        dz = abs(out[i + 1, 2] - out[i, 2])
        hd = math.sqrt(dx * dx + dy * dy)
        if hd > 1e-6:
            max_climb = max(max_climb, dz / hd)
        if i > 0:
            p0, p1, p2 = out[i - 1], out[i], out[i + 1]
            a = np.linalg.norm(p1 - p0)
            b = np.linalg.norm(p2 - p1)
            c = np.linalg.norm(p2 - p0)
            area = 0.5 * np.linalg.norm(np.cross(p1 - p0, p2 - p0))
            if area > 1e-8:
                min_r = min(min_r, a * b * c / (4.0 * area))
    return out, {"time": T, "coeff": coeff, "vel_zero_weight": vw, "iters": it,
                 "max_dev": max_dev, "max_climb_rate": max_climb, "min_turn_radius": min_r}
