"""The C++ class shim (TrajectoryGeneratorTool / MinimumSnapConfig / math_util::Bezier) run on
the GPU box through the marshalling harness of SURVEY.md §8a A14 (cs-pathplan_amd/host/shim_selftest.cpp
reproduces UavPathPlanner::Minisnap_3D / Minisnap_EN, uavPathPlanning.cpp:4401-4474), compared
with the oracle's restatement of GenerateTrajectoryMatrix (minimum_snap.cpp:22-206)."""
import importlib.util
import math
import os
import subprocess

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe():
    spec = importlib.util.spec_from_file_location("csp_build", os.path.join(ROOT, "cs-pathplan_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b.build_host_check()


def _run(exe, mode, tmp_path, cfg, pts):
    f = tmp_path / "in.txt"
    with open(f, "w") as fh:
        fh.write("%d %.17g %.17g %.17g %.17g %.17g %d\n" % (cfg["order"], cfg["pw"], cfg["vw"], cfg["V"], cfg["mt"], cfg["sd"], len(pts)))
        for p in pts:
            fh.write("%.17g %.17g %.17g\n" % tuple(p))
    r = subprocess.run([exe, mode, str(f)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    rows = [list(map(float, ln.split())) for ln in r.stdout.strip().splitlines()]
    return np.array(rows).reshape(-1, 3), r.stderr


def test_kat(exe):
    r = subprocess.run([exe, "kat"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)


@pytest.mark.parametrize("cfg", [
    dict(order=2, pw=1e-7, vw=0.01, V=200.0, mt=1.0, sd=300.0),   # the shipped yaml (minimum_snap_config.yaml:5-27)
    dict(order=3, pw=0.0, vw=0.0, V=200.0, mt=1.0, sd=300.0),
    dict(order=2, pw=0.0, vw=0.0, V=30.0, mt=1.0, sd=500.0),      # leader_speed default 30 m/s
])
def test_minisnap_3d_readme_waypoints(exe, oracle_mod, tmp_path, cfg):
    P = synth.README_UAV31_ENU
    got, err = _run(exe, "plan3d", tmp_path, cfg, P)
    ref, info = oracle_mod.generate_trajectory(P, order=cfg["order"], path_weight=cfg["pw"], vel_zero_weight=cfg["vw"],
                                               v_avg=cfg["V"], min_time_s=cfg["mt"], sample_distance=cfg["sd"])
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(got - ref)) < 1e-6 * scale
    climb = float(err.split()[1])
    assert abs(climb - info["max_climb_rate"]) < 1e-6 * max(1.0, info["max_climb_rate"])


def test_minisnap_en_zeroes_height_and_restores_start_up(exe, oracle_mod, tmp_path):
    cfg = dict(order=3, pw=0.0, vw=0.0, V=5.0, mt=0.1, sd=1.0)
    wp, _ = synth.make_batch(1, 6, config_id=11)
    P = wp[0] * 5.0
    got, _ = _run(exe, "planen", tmp_path, cfg, P)
    flat = P.copy()
    flat[:, 2] = 0.0
    ref, _ = oracle_mod.generate_trajectory(flat, order=3, v_avg=5.0, min_time_s=0.1, sample_distance=1.0)
    assert got.shape == ref.shape
    assert np.max(np.abs(got[:, :2] - ref[:, :2])) < 1e-7 * np.max(np.abs(ref))
    assert np.all(got[:, 2] == P[0, 2])


def test_resolve_loop_doubles_vel_zero_weight(exe, oracle_mod, tmp_path):
    """A13: max_dev > 0.2 triggers the <=10x doubling loop (minimum_snap.cpp:80-90)."""
    cfg = dict(order=3, pw=0.5, vw=0.0, V=5.0, mt=0.1, sd=0.5)
    wp, _ = synth.make_batch(1, 5, config_id=12)
    P = wp[0] * 3.0
    ref, info = oracle_mod.generate_trajectory(P, order=3, path_weight=0.5, v_avg=5.0, min_time_s=0.1, sample_distance=0.5)
    assert info["iters"] >= 1, "fixture does not exercise the loop"
    got, _ = _run(exe, "plan3d", tmp_path, cfg, P)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-6 * np.max(np.abs(ref))


def _bezier_py(P, resolution, min_radius):
    n = len(P)
    hd = []
    for i in range(n):
        lo, hi = max(i - 1, 0), min(i + 1, n - 1)
        hd.append(math.atan2(P[hi][1] - P[lo][1], P[hi][0] - P[lo][0]))
    out = []
    for i in range(n - 1):
        a, d = P[i], P[i + 1]
        chord = math.hypot(a[0] - d[0], a[1] - d[1])
        if chord < 1e-1:
            out.append(list(d))
            continue
        k = 1.0 / 3.0

        def ctrl(k):
            b = [a[0] + math.cos(hd[i]) * chord * k, a[1] + math.sin(hd[i]) * chord * k, a[2] + (d[2] - a[2]) / 3.0]
            c = [d[0] - math.cos(hd[i + 1]) * chord * k, d[1] - math.sin(hd[i + 1]) * chord * k, a[2] + (d[2] - a[2]) * 2.0 / 3.0]
            return b, c
        b, c = ctrl(k)   # min_radius <= 1: no curvature iteration
        assert min_radius <= 1.0
        dis = math.hypot(c[0] - b[0], c[1] - b[1]) + chord * 2.0 / 3.0
        step = resolution / dis
        seg = []
        t = 0.0
        while t <= 1.0:
            u = 1.0 - t
            seg.append([u ** 3 * a[j] + 3 * u * u * t * b[j] + 3 * u * t * t * c[j] + t ** 3 * d[j] for j in range(3)])
            t += step
        out.extend(seg if i == 0 else seg[1:])
    return np.array(out)


def test_bezier_surface(exe, tmp_path):
    wp, _ = synth.make_batch(1, 5, config_id=13)
    P = wp[0] * 20.0
    cfg = dict(order=3, pw=0.0, vw=0.0, V=1.0, mt=0.1, sd=2.5)   # V slot = min_radius (harness convention)
    got, _ = _run(exe, "bezier", tmp_path, cfg, P)
    ref = _bezier_py(P.tolist(), 2.5, 1.0)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


def test_f4_fixture_through_the_class_shim(exe, tmp_path):
    """The committed F4 cases (tests/golden/F4_minisnap_en.json: ENU waypoints + MinimumSnapConfig in, sampled ENU
    points out) through the harness that reproduces Minisnap_EN / Minisnap_3D (uavPathPlanning.cpp:4401-4474)."""
    from tests.test_golden_r2 import load_f4
    for c in load_f4():
        e = c["effective"]
        cfg = dict(order=e["order"], pw=e.get("path_weight", 0.0), vw=e.get("vel_zero_weight", 0.0), V=e["V_avg"],
                   mt=e["min_time_s"], sd=e["sample_distance"])
        got, _ = _run(exe, "planen" if c["mode"] == "en" else "plan3d", tmp_path, cfg, c["waypoints_enu"])
        assert got.shape == c["result_enu"].shape, (c["name"], got.shape, c["result_enu"].shape)
        assert np.max(np.abs(got - c["result_enu"])) < 1e-6 * np.max(np.abs(c["result_enu"])), c["name"]
        if c["mode"] == "en":
            assert np.all(got[:, 2] == c["waypoints_enu"][0, 2])
