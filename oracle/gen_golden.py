"""Generates tests/golden/*.json -- inputs and expected outputs for the minimum-snap path.

Run in the dev container:  python oracle/gen_golden.py
Expected coefficients come from oracle/numpy_ref.py (dense numpy/LAPACK restatement of
/root/reference/math_util/minimum_snap.cpp:227-649); analytic known answers (K1) are written
by hand.  The reference itself ships no coefficient goldens and cannot be built (no Eigen),
so these fixtures pin the C oracle and the HIP path against an independent restatement --
"parity unpinned" with respect to the real Eigen build.
Floats are stored as C99 hex strings (bit-exact round trip).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import numpy_ref as nr  # noqa: E402
from tests import synth  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def hx(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def case(name, order, path, time, vel=None, acc=None, pw=0.0, vw=0.0, note=""):
    path = np.asarray(path, dtype=np.float64)
    time = np.asarray(time, dtype=np.float64)
    vel = np.zeros((2, 3)) if vel is None else np.asarray(vel, dtype=np.float64)
    acc = np.zeros((2, 3)) if acc is None else np.asarray(acc, dtype=np.float64)
    coeff, md = nr.solve_qp_closed_form(order, path, vel, acc, time, pw, vw)
    S = len(time)
    M = nr.build_M(order, time)
    return {
        "name": name, "note": note, "order": order, "segments": S,
        "path_weight": pw, "vel_zero_weight": vw,
        "path": hx(path), "time": hx(time), "vel": hx(vel), "acc": hx(acc),
        "coeff": hx(coeff), "max_dev": float(md).hex(),
        "cond_M": float(np.linalg.cond(M)),
    }


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260501)

    # F1: analytic known answers (K1): single segment, rest-to-rest 0 -> 1, T = 1
    kat = {2: [-2, 3, 0, 0], 3: [6, -15, 10, 0, 0, 0], 4: [-20, 70, -84, 35, 0, 0, 0, 0]}
    f1 = []
    for o, c in kat.items():
        f1.append({"name": "K1_o%d" % o, "order": o, "segments": 1, "path_weight": 0.0,
                   "vel_zero_weight": 0.0, "path": hx([[0, 0, 0], [1, 1, 1]]), "time": hx([1.0]),
                   "vel": hx(np.zeros((2, 3))), "acc": hx(np.zeros((2, 3))),
                   "coeff": hx(np.tile(np.array(c, dtype=np.float64), 3)), "max_dev": (0.0).hex(),
                   "note": "analytic rest-to-rest Hermite polynomial, highest power first"})
    json.dump({"fixture": "F1", "cases": f1}, open(os.path.join(OUT, "F1_kat.json"), "w"), indent=1)

    # F2: README uav31_0 ENU waypoints (reference readme.md:14-20), o in {2,3,4}, V in {200, 30}
    f2 = []
    P = synth.README_UAV31_ENU
    for o in (2, 3, 4):
        for V in (200.0, 30.0):
            T = nr.time_allocation(P, V, 1.0)
            f2.append(case("uav31_o%d_V%d" % (o, int(V)), o, P, T,
                           note="README waypoints; ill-conditioned in the reference's raw-time formulation"))
    # the shipped yaml (minimum_snap_config.yaml:5-27): order 2, vel_zero 0.01, path 1e-7, V 200
    T = nr.time_allocation(P, 200.0, 1.0)
    f2.append(case("uav31_yaml", 2, P, T, pw=1e-7, vw=0.01, note="shipped yaml parameters"))
    json.dump({"fixture": "F2", "cases": f2}, open(os.path.join(OUT, "F2_readme_uav31.json"), "w"), indent=1)

    # F3: well-scaled synthetic, 3 seeds x S in {8,16}, o = 4, non-zero boundary conditions too
    f3 = []
    for S in (8, 16):
        wp, tm = synth.make_batch(3, S, config_id=90)
        for i in range(3):
            vel = rng.normal(size=(2, 3)) if i == 2 else None
            acc = rng.normal(size=(2, 3)) if i == 2 else None
            f3.append(case("synth_S%d_%d" % (S, i), 4, wp[i], tm[i], vel, acc))
    json.dump({"fixture": "F3", "cases": f3}, open(os.path.join(OUT, "F3_wellscaled.json"), "w"), indent=1)

    # F5: ragged shapes, S in {1,2,4,7,33}, o in {1..5}; S=64 kept out of the fixture for size
    f5 = []
    for S in (1, 2, 4, 7, 33):
        for o in (1, 2, 3, 4, 5):
            if S == 33 and o not in (3, 4):
                continue
            p = np.cumsum(rng.normal(size=(S + 1, 3)), axis=0) + rng.uniform(-10, 10, size=3)
            t = rng.uniform(0.5, 2.0, size=S)
            f5.append(case("ragged_S%d_o%d" % (S, o), o, p, t, rng.normal(size=(2, 3)), rng.normal(size=(2, 3))))
    json.dump({"fixture": "F5", "cases": f5}, open(os.path.join(OUT, "F5_ragged.json"), "w"), indent=1)

    # F6: penalties (K7 vel-zero, K8 path-weight incl. the un-halved f_P quirk)
    f6 = []
    for (o, S, pw, vw) in [(4, 8, 0.0, 0.01), (4, 8, 0.0, 5.0), (3, 6, 0.0, 0.3), (2, 6, 0.0, 0.01),
                           (4, 8, 1e-3, 0.0), (4, 16, 1e-2, 0.01), (3, 5, 0.5, 0.0), (2, 6, 1e-7, 0.01),
                           (5, 4, 1e-3, 0.02)]:
        p = np.cumsum(rng.normal(size=(S + 1, 3)), axis=0)
        t = rng.uniform(0.5, 2.0, size=S)
        f6.append(case("pen_o%d_S%d_pw%g_vw%g" % (o, S, pw, vw), o, p, t, pw=pw, vw=vw))
    json.dump({"fixture": "F6", "cases": f6}, open(os.path.join(OUT, "F6_penalties.json"), "w"), indent=1)
    print("wrote fixtures to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
