"""Deterministic synthetic inputs for tests and bench.py (SURVEY.md §8d / BASELINE.md §4).

seed = 20260501 + config_id; counter-based Philox generator so every rank / device / CPU
baseline sees bit-identical inputs.  Waypoints: random walk p_{k+1} = p_k + N(0,1)^3 from
p_0 ~ U(-10,10)^3; segment times ~ U(0.5, 2.0) (well-scaled: cond(R_PP) ~ 5e5 at S=16, o=4).
"""
import json
import os

import numpy as np

BASE_SEED = 20260501

# README `uav31_0` ENU waypoints (reference readme.md:14-20) -- config C1's inputs.
README_UAV31_ENU = np.array([
    [-0.000000000046327, -0.000000000452815, 1669.000000000820137],
    [-22008.910310499257321, 32.799545377501204, 1636.091338242949178],
    [-22009.474804264991690, -2966.281837991115026, 1635.398165184439677],
    [-15007.552345050633448, -2983.825260306681230, 1655.674289593189314],
    [-1003.853909577760191, -2999.001544960936371, 1673.214552272680066],
    [-1003.446472092303907, 0.068179987007966, 1673.921199759593492],
    [-1003.432888336147585, 100.027485618222272, 1673.920415851918733],
])


def make_batch(B, S, config_id=3, offset=0, dtype=np.float64):
    """Rows [offset, offset+B) of config `config_id`'s stream -> (waypoints [B,S+1,3], times [B,S]).
    Row b depends only on (config_id, S, offset+b), so shards generated per rank tile exactly."""
    wp = np.empty((B, S + 1, 3), dtype=np.float64)
    tm = np.empty((B, S), dtype=np.float64)
    chunk = 8192
    done = 0
    while done < B:
        n = min(chunk, B - done)
        # one Philox stream per chunk-aligned block keeps generation O(B) and offset-stable
        first = offset + done
        blk, within = divmod(first, chunk)
        take = min(n, chunk - within)
        g = np.random.Generator(np.random.Philox(key=[BASE_SEED + config_id, (S << 40) | blk]))
        p0 = g.uniform(-10.0, 10.0, size=(chunk, 1, 3))
        steps = g.normal(0.0, 1.0, size=(chunk, S, 3))
        t = g.uniform(0.5, 2.0, size=(chunk, S))
        w = np.concatenate([p0, p0 + np.cumsum(steps, axis=1)], axis=1)
        wp[done:done + take] = w[within:within + take]
        tm[done:done + take] = t[within:within + take]
        done += take
    return wp.astype(dtype), tm.astype(dtype)


def make_ragged(B, config_id=5, smin=4, smax=64, orders=(3, 4, 5)):
    """Config C5: per-trajectory S ~ U{smin..smax}, order ~ U{orders}.  Returns a list of
    (order, waypoints [S+1,3], times [S]) in generation order."""
    g = np.random.Generator(np.random.Philox(key=[BASE_SEED + config_id, 0]))
    out = []
    for _ in range(B):
        S = int(g.integers(smin, smax + 1))
        o = int(orders[int(g.integers(0, len(orders)))])
        p0 = g.uniform(-10.0, 10.0, size=(1, 3))
        w = np.concatenate([p0, p0 + np.cumsum(g.normal(0.0, 1.0, size=(S, 3)), axis=0)], axis=0)
        t = g.uniform(0.5, 2.0, size=S)
        out.append((o, w, t))
    return out


def algorithmic_bytes(S, order, width=8):
    """w*[3(S+1)+S] in + w*3*S*2o out."""
    return width * (3 * (S + 1) + S) + width * 3 * S * 2 * order


def rel_err(got, ref):
    """max over trajectories of ||c - c_ref||_inf / ||c_ref||_inf (the §8d parity gate)."""
    got = np.asarray(got, dtype=np.float64).reshape(got.shape[0], -1)
    ref = np.asarray(ref, dtype=np.float64).reshape(ref.shape[0], -1)
    den = np.max(np.abs(ref), axis=1)
    den[den == 0] = 1.0
    return float(np.max(np.max(np.abs(got - ref), axis=1) / den))


def rel_err_per_power(got, ref):
    """Per-coefficient-power relative error: for every trajectory and every power k the scale is
    max over (segment, axis) of |c_ref[..., k]|, so the t^(2o-1) coefficients -- many orders of
    magnitude below the constant term, which is just the waypoint copied through -- are judged on
    their own size.  `rel_err` above (norm-wise, the section 8d gate) is dominated by the constant term
    and says little about the solve.  got/ref: [B,S,3,m] (or [S,3,m] for one trajectory)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if ref.ndim == 3:
        got, ref = got[None], ref[None]
    got = got.reshape(ref.shape)
    den = np.max(np.abs(ref), axis=(1, 2), keepdims=True)
    den = np.where(den == 0.0, 1.0, den)
    return float(np.max(np.abs(got - ref) / den))


def parity_gate(got, ref, tol, tag=""):
    """THE parity gate of the GPU tests (round 3: everywhere): the PER-POWER relative error must stay below `tol`; the
    norm-wise figure of SURVEY.md section 8d is computed beside it and returned for printing.  got / ref: [B,S,3,m], or
    [S,3,m] for one trajectory (a ragged batch is gated trajectory by trajectory).  With CSP_PARITY_SURVEY=<file> every
    call appends (tag, per-power, norm-wise, tol) to that file and does not assert -- how the tolerances were placed
    (5-10x the measured figure, never above the 1e-6 of BASELINE.json's north star except where a test says why)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.ndim in (3, 4) and got.shape[-2] == 3, got.shape     # the HIP side always carries [.., segment, axis, power]
    ref = ref.reshape(got.shape)                                    # the oracle returns PolyCoeff rows [S, 3m]
    if got.ndim == 3:
        got, ref = got[None], ref[None]
    pp = rel_err_per_power(got, ref)
    nw = rel_err(got, ref)
    survey = os.environ.get("CSP_PARITY_SURVEY")
    if survey:
        with open(survey, "a") as f:
            f.write(json.dumps({"tag": str(tag), "per_power": pp, "norm_wise": nw, "tol": tol}) + "\n")
        return pp, nw
    assert pp < tol, "parity gate %s: per-power rel err %.3e (norm-wise %.3e) >= %.1e" % (tag, pp, nw, tol)
    return pp, nw
