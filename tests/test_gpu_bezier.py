"""Batched Bezier sampler (include/csp_bezier.h, SURVEY.md 8f row N4 second half) against the Python restatement of
math_util/bezier.cpp:28-190 that tests/test_dropin_callsites.py also holds the C++ class shim to."""
import numpy as np
import pytest

from tests.test_dropin_callsites import bezier_reference_py

pytestmark = pytest.mark.gpu


def _paths(rng):
    paths = []
    for npts, scale in ((2, 500.0), (7, 1000.0), (3, 40.0), (70, 300.0), (1, 1.0), (130, 120.0), (9, 2500.0)):
        p = np.cumsum(rng.normal(size=(npts, 3)) * scale, axis=0) + rng.uniform(-100, 100, size=3)
        p[:, 2] = 100.0 + np.abs(p[:, 2]) * 0.05
        paths.append(p)
    deg = np.array([[0, 0, 0], [50, 0, 5], [50.01, 0.0, 6], [90, 40, 7]], dtype=np.float64)   # middle leg shorter than 0.1 in the plane
    paths.append(deg)
    return paths


@pytest.mark.parametrize("min_radius,resolution", [(1.0, 25.0), (300.0, 25.0), (300.0, 3.5), (50.0, 100.0)])
def test_batched_bezier_matches_the_restatement(csp, min_radius, resolution):
    import torch
    rng = np.random.default_rng(17)
    paths = _paths(rng)
    wp = np.concatenate(paths)
    off = np.concatenate([[0], np.cumsum([len(p) for p in paths])]).astype(np.int64)
    refs = [bezier_reference_py(p.tolist(), resolution, min_radius)[0] if len(p) >= 2 else np.zeros((0, 3)) for p in paths]
    cap = max(len(r) for r in refs) + 8
    s_h, n_h = csp.bezier_generate_batch(wp, off, resolution, min_radius, cap)
    s_d, n_d = csp.bezier_generate_batch(torch.from_numpy(wp).cuda(), torch.from_numpy(off).cuda(), resolution, min_radius, cap)
    torch.cuda.synchronize()
    assert np.array_equal(n_h, n_d.cpu().numpy())
    for b, ref in enumerate(refs):
        assert n_h[b] == len(ref), (b, n_h[b], len(ref))
        if len(ref):
            assert np.max(np.abs(s_h[b, :len(ref)] - ref)) <= 1e-9 * np.max(np.abs(ref)), b
            assert np.array_equal(s_h[b, :len(ref)], s_d[b, :len(ref)].cpu().numpy())
    # capacity overflow: the true counts come back, only `capacity` rows are written
    s_small, n_small = csp.bezier_generate_batch(wp, off, resolution, min_radius, 5)
    assert np.array_equal(n_small, n_h)
    for b, ref in enumerate(refs):
        k = min(5, len(ref))
        assert np.array_equal(s_small[b, :k], s_h[b, :k])


def test_batched_bezier_rejects_bad_arguments(csp):
    wp = np.zeros((4, 3))
    off = np.array([0, 4], dtype=np.int64)
    for res in (0.0, -1.0, float("nan")):
        with pytest.raises(csp.CspError) as e:
            csp.bezier_generate_batch(wp, off, res, 1.0, 16)
        assert e.value.code == -1
