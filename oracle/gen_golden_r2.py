"""Generates the round-2 fixtures (tests/golden/F4, F5b, F7) -- TEST INFRASTRUCTURE.

Run in the dev container:  python oracle/gen_golden_r2.py
Separate from gen_golden.py (whose single random stream would otherwise shift under the round-1
fixtures).  Expected values come from oracle/numpy_ref.py, the independent numpy/LAPACK restatement
of /root/reference/math_util/minimum_snap.cpp:22-649; the reference ships no goldens for this path
and cannot be built here (no Eigen): PARITY UNPINNED with respect to the real Eigen build.

  F4  the Minisnap_EN / Minisnap_3D marshalling cases of SURVEY.md section 8c
      (uavPathPlanning.cpp:4401-4474: EN zeroes z on the way in and writes waypoint 0's `up` on the way
      out): ENU waypoints + MinimumSnapConfig in, the sampled ENU points out.
  F5b the S = 64 ragged cases round 1 left out "for size" (orders 2..5).
  F7  constructed NEAR-TIES for the two discrete decisions on the path:
        * the t* arg-max over 17 samples with a strict `>` (first maximum wins, minimum_snap.cpp:435):
          a point-symmetric 3-segment trajectory makes the middle segment's deviation profile
          symmetric with two mirror maxima (d2(s) == d2(16-s) mathematically); a small asymmetry eps then decides between
          the two mirror samples with a relative gap of 1e-8 .. 1e-10 in either direction;
        * the thinning test `dist >= sample_distance` (:145): sample_distance is placed a relative
          1e-10 below / above a candidate's distance, and -- exactly representable arithmetic, order 1,
          power-of-two data -- exactly ON it.
      Each case records which way the decision falls in the 80-bit long-double oracle, and the cases are
      only written if fp64 numpy, the C oracle and the long-double oracle agree on it.
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle  # noqa: E402
from oracle import numpy_ref as nr  # noqa: E402
from tests import synth  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def hx(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def solve_case(name, order, path, time, vel=None, acc=None, pw=0.0, vw=0.0, note="", extra=None):
    path, time = np.asarray(path, dtype=np.float64), np.asarray(time, dtype=np.float64)
    vel = np.zeros((2, 3)) if vel is None else np.asarray(vel, dtype=np.float64)
    acc = np.zeros((2, 3)) if acc is None else np.asarray(acc, dtype=np.float64)
    coeff, md = nr.solve_qp_closed_form(order, path, vel, acc, time, pw, vw)
    c = {"name": name, "note": note, "order": order, "segments": len(time), "path_weight": pw, "vel_zero_weight": vw,
         "path": hx(path), "time": hx(time), "vel": hx(vel), "acc": hx(acc), "coeff": hx(coeff),
         "max_dev": float(md).hex(), "cond_M": float(np.linalg.cond(nr.build_M(order, time)))}
    if extra:
        c.update(extra)
    return c


# ---------------------------------------------------------------------------------------------- F4
def f4_cases():
    out = []
    yaml_cfg = dict(order=2, path_weight=1e-7, vel_zero_weight=0.01, V_avg=200.0, min_time_s=1.0, sample_distance=300.0)
    wp, _ = synth.make_batch(2, 6, config_id=71)
    sets = [("readme_yaml", synth.README_UAV31_ENU, yaml_cfg, -1.0, -1.0),
            ("readme_o3_leader30", synth.README_UAV31_ENU, dict(yaml_cfg, order=3, path_weight=0.0, vel_zero_weight=0.0), 500.0, 30.0),
            ("synth_o3", wp[0] * 5.0, dict(order=3, V_avg=5.0, min_time_s=0.1, sample_distance=1.0), -1.0, -1.0),
            ("synth_o4_override", wp[1] * 8.0 + np.array([0.0, 0.0, 120.0]), dict(order=4, V_avg=5.0, min_time_s=0.1, sample_distance=1.0), 2.5, 12.0)]
    for name, P, cfg, sd_over, v_over in sets:
        for mode in ("en", "3d"):
            route = np.array(P, dtype=np.float64)
            if mode == "en":
                route = route.copy()
                route[:, 2] = 0.0                                   # uavPathPlanning.cpp:4412
            samples, info = nr.generate_trajectory(route, cfg, sd_over, v_over)
            res = samples.copy()
            if mode == "en":
                res[:, 2] = P[0][2]                                 # :4426-4431
            out.append({"name": "%s_%s" % (name, mode), "mode": mode, "config": cfg,
                        "sample_distance_override": sd_over, "v_avg_override": v_over,
                        "waypoints_enu": hx(P), "n_waypoints": len(P), "result_enu": hx(res), "n_result": len(res),
                        "iterations": info["iters"], "vel_zero_weight_final": float(info["vel_zero_weight"]).hex(),
                        "max_climb_rate": float(info["max_climb_rate"]).hex(), "min_turn_radius": float(info["min_turn_radius"]).hex()})
    return out


# --------------------------------------------------------------------------------------------- F5b
def f5b_cases():
    rng = np.random.default_rng(20260502)
    out = []
    for o in (2, 3, 4, 5):
        S = 64
        p = np.cumsum(rng.normal(size=(S + 1, 3)), axis=0) + rng.uniform(-10, 10, size=3)
        t = rng.uniform(0.5, 2.0, size=S)
        out.append(solve_case("ragged_S64_o%d" % o, o, p, t, rng.normal(size=(2, 3)), rng.normal(size=(2, 3))))
    return out


# ---------------------------------------------------------------------------------------------- F7
def deviation_profile(order, path, time, seg, long_double=True):
    """d2(s), s = 0..16, of the unpenalised pre-solve on segment `seg` (minimum_snap.cpp:408-439), from the
    long-double oracle's coefficients, evaluated in long double."""
    z = np.zeros((2, 3))
    co, _ = oracle.solve(order, path, z, z, time, 0.0, 0.0, long_double=long_double)
    m = 2 * order
    co = co.reshape(len(time), 3, m).astype(np.longdouble)
    T = np.longdouble(time[seg])
    d2 = []
    for s in range(17):
        tt = T * np.longdouble(s) / np.longdouble(16)
        P = np.array([sum(co[seg, a, i] * tt ** (m - 1 - i) for i in range(m)) for a in range(3)], dtype=np.longdouble)
        L = path[seg].astype(np.longdouble) + (tt / T) * (path[seg + 1] - path[seg]).astype(np.longdouble)
        d2.append(float(np.sum((P - L) ** 2)))
    return np.array(d2)


def tstar_from_solution(order, path, time, pw, vw):
    """Which sample the dense fp64 restatements picked: recovered from max_dev's own definition is not
    possible, so re-derive it the reference's way in fp64 (first strict maximum)."""
    d2 = deviation_profile(order, path, time, 1, long_double=False)
    best, bi = -1.0, 0
    for s in range(17):
        if d2[s] > best:
            best, bi = d2[s], s
    return bi


def f7_argmax_cases():
    out = []
    order = 4
    # point-symmetric about (5, 0, 0): the middle segment's deviation is ODD about its midpoint, so d2(s) == d2(16-s)
    # with a zero at s = 8 -- two mirror maxima
    base = np.array([[0.0, 0.0, 0.0], [3.0, 2.0, 0.5], [7.0, -2.0, -0.5], [10.0, 0.0, 0.0]])
    time = np.array([1.25, 1.5, 1.25])
    for target, direction in [(1e-8, +1), (1e-8, -1), (1e-10, +1), (1e-10, -1)]:
        # asymmetry: move waypoint 2 along y by eps; the gap between the two mirror maxima is ~linear in eps
        def gap(eps):
            p = base.copy()
            p[2, 1] += eps
            d2 = deviation_profile(order, p, time, 1)
            s_hi = int(np.argmax(d2))
            s_lo = 16 - s_hi
            first, second = min(s_hi, s_lo), max(s_hi, s_lo)
            return (d2[second] - d2[first]) / d2[first], first, second, p
        g1, first, second, _ = gap(1e-6)
        assert first != second, "the profile peaks at the middle sample: no mirror pair"
        eps = direction * target / (g1 / 1e-6)
        g, first, second, p = gap(eps)
        for _ in range(6):                                       # secant refinement towards the target gap
            eps *= (direction * target) / g
            g, first, second, p = gap(eps)
        assert abs(abs(g) / target - 1.0) < 0.2 and (g > 0) == (direction > 0), (g, target, direction)
        expect = second if g > 0 else first                       # strict `>`: the later sample wins only if strictly larger
        for pw in (0.5,):
            # all three CPU restatements must agree before the case is written
            t64 = tstar_from_solution(order, p, time, pw, 0.0)
            assert t64 == expect, (t64, expect, g)
            c = solve_case("argmax_gap%g_%s" % (target, "later" if g > 0 else "earlier"), order, p, time, pw=pw, vw=0.0,
                           note="middle segment: mirror samples %d and %d differ by a relative %.3e in d2 (long double); "
                                "strict > picks sample %d" % (first, second, g, expect),
                           extra={"tstar_middle_segment": expect, "mirror_samples": [first, second], "relative_gap": g})
            z = np.zeros((2, 3))
            cc, md = oracle.solve(order, p, z, z, time, pw, 0.0)
            assert np.max(np.abs(cc.ravel() - np.array([float.fromhex(v) for v in c["coeff"]]))) < 1e-8, "C oracle and numpy disagree"
            out.append(c)
            # what a flipped decision would cost: the same solve with the asymmetry mirrored picks the other sample
    return out


def f7_thinning_cases():
    out = []
    # (1) exact tie, exactly representable: order 1 (p(t) = 8 t), one segment, T = 1: candidates at the accumulated
    # t_k = 0.1 + 0.1 + ... ; distance from the start = 8 t_k exactly, sqrt(fl(x*x)) == x in IEEE arithmetic.
    t = 0.1
    acc = [t]
    for _ in range(9):
        t += 0.1
        acc.append(t)
    P = np.array([[0.0, 0.0, 0.0], [8.0, 0.0, 0.0]])
    for k in (2, 6):
        sd = 8.0 * acc[k]
        cfg = dict(order=1, V_avg=8.0, min_time_s=0.1, sample_distance=sd)
        samples, info = nr.generate_trajectory(P, cfg)
        assert abs(info["time"][0] - 1.0) == 0.0
        assert any(abs(s[0] - sd) == 0.0 for s in samples), "the tie candidate must be kept by >="
        out.append({"name": "thin_exact_tie_k%d" % k, "kind": "exact", "config": cfg, "waypoints": hx(P), "n_waypoints": 2,
                    "samples": hx(samples), "n_samples": len(samples),
                    "note": "sample_distance == the candidate's distance bit for bit (8*t_%d); >= keeps it" % k})
    # (2) near-ties on a curved order-4 trajectory.  The first candidate a run keeps lies at distance D from the
    # trajectory's start point and every earlier candidate is closer than the run's sample_distance <= D, so moving
    # sample_distance to D*(1 -/+ 1e-10) leaves the earlier decisions alone and puts THIS one on the boundary.
    wp, _ = synth.make_batch(1, 5, config_id=72)
    P = wp[0] * 3.0
    for sd0 in (0.9, 1.7):
        base_cfg = dict(order=4, V_avg=2.0, min_time_s=0.1, sample_distance=sd0)
        samples, info = nr.generate_trajectory(P, base_cfg)
        D = float(np.linalg.norm(samples[1] - samples[0]))
        assert D > sd0 * (1.0 + 1e-6)
        pair = []
        for rel in (-1e-10, +1e-10):
            sd = D * (1.0 + rel)
            cfg = dict(base_cfg, sample_distance=sd)
            s2, i2 = nr.generate_trajectory(P, cfg)
            so, io = oracle.generate_trajectory(P, order=4, v_avg=2.0, min_time_s=0.1, sample_distance=sd)
            assert s2.shape == so.shape and np.max(np.abs(s2 - so)) < 1e-9, "numpy and C oracle disagree on a near-tie"
            pair.append(s2)
            out.append({"name": "thin_near_tie_sd%g_%s" % (sd0, "below" if rel < 0 else "above"), "kind": "near", "config": cfg,
                        "waypoints": hx(P), "n_waypoints": len(P), "samples": hx(s2), "n_samples": len(s2),
                        "note": "sample_distance = D*(1%+.0e), D = distance of the first kept candidate from the start" % rel})
        # the pair must straddle the decision: `below` keeps the candidate at distance D, `above` drops it
        assert np.max(np.abs(pair[0][1] - samples[1])) == 0.0 and np.max(np.abs(pair[1][1] - samples[1])) > 1e-3
    return out


def main():
    oracle.build()
    os.makedirs(OUT, exist_ok=True)
    json.dump({"fixture": "F4", "cases": f4_cases()}, open(os.path.join(OUT, "F4_minisnap_en.json"), "w"), indent=1)
    json.dump({"fixture": "F5b", "cases": f5b_cases()}, open(os.path.join(OUT, "F5b_ragged_s64.json"), "w"), indent=1)
    json.dump({"fixture": "F7", "argmax": f7_argmax_cases(), "thinning": f7_thinning_cases()},
              open(os.path.join(OUT, "F7_near_ties.json"), "w"), indent=1)
    print("wrote F4, F5b, F7 to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
