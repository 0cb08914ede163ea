"""Row N3: batched WGS84 <-> ENU.  The oracle (oracle/geo_oracle.c) is PINNED by the reference's own
README output (readme.md:10-28, 15 decimals), committed as tests/golden/G1_readme_geo.json."""
import json
import os

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "G1_readme_geo.json")))
    lla = np.array(g["wgs84_lon_lat_alt"], dtype=float)
    ref = lla[0].copy()
    ref[2] = 0.0   # origin altitude forced to 0 (uavPathPlanning.cpp:3643-3644)
    return lla, ref, np.array(g["enu_east_north_up"], dtype=float), np.array(g["wgs84_round_trip_lon_lat_alt"], dtype=float)


def test_oracle_reproduces_the_readme():
    lla, ref, enu_gold, back_gold = _golden()
    enu = oracle.wgs84_to_enu(lla, ref)
    assert np.max(np.abs(enu - enu_gold)) < 5e-12          # printed with 15 decimals on 2e4-m values
    back = oracle.enu_to_wgs84(enu, ref)
    assert np.max(np.abs(back[:, :2] - back_gold[:, :2])) < 1e-14
    assert np.max(np.abs(back[:, 2] - back_gold[:, 2])) < 1e-9


@pytest.mark.gpu
def test_hip_reproduces_the_readme_and_the_oracle(csp):
    lla, ref, enu_gold, back_gold = _golden()
    enu = csp.wgs84_to_enu_batch(lla, ref)
    assert np.max(np.abs(enu - enu_gold)) < 1e-8           # metres; device libm differs in the last ulps
    back = csp.enu_to_wgs84_batch(enu, ref)
    assert np.max(np.abs(back[:, :2] - back_gold[:, :2])) < 1e-12
    assert np.max(np.abs(back[:, 2] - back_gold[:, 2])) < 1e-8
    rng = np.random.default_rng(3)
    n = 200000
    pts = np.stack([rng.uniform(-180, 180, n), rng.uniform(-89.9, 89.9, n), rng.uniform(-500, 12000, n)], axis=1)
    ref2 = np.array([109.56, 40.87, 0.0])
    e_gpu, e_cpu = csp.wgs84_to_enu_batch(pts, ref2), oracle.wgs84_to_enu(pts, ref2)
    assert np.max(np.abs(e_gpu - e_cpu)) < 1e-7            # on values up to 1.3e7 m: 1e-14 relative
    near = np.stack([rng.uniform(-5e4, 5e4, n), rng.uniform(-5e4, 5e4, n), rng.uniform(0, 5e3, n)], axis=1)
    l_gpu, l_cpu = csp.enu_to_wgs84_batch(near, ref2), oracle.enu_to_wgs84(near, ref2)
    assert np.max(np.abs(l_gpu[:, :2] - l_cpu[:, :2])) < 1e-12
    assert np.max(np.abs(l_gpu[:, 2] - l_cpu[:, 2])) < 1e-7
    # round trip on the device
    import torch
    d = torch.from_numpy(near).cuda()
    rt = csp.wgs84_to_enu_batch(csp.enu_to_wgs84_batch(d, ref2), ref2).cpu().numpy()
    assert np.max(np.abs(rt - near)) < 1e-6
