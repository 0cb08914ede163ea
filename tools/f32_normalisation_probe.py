"""Can pure-fp32 arithmetic (CSP_FLAG_F32_ARITH) be made usable by TIME NORMALISATION (per-waypoint scaling
of the free derivatives by powers of a local time scale)?  CPU experiment, numpy only: the block-tridiagonal sweep of
DESIGN.md section 2 in float32 and float64, with the unknowns raw and scaled by tau^(r+1) for three choices of tau (geometric mean,
arithmetic mean, minimum of the adjacent segment times), on BASELINE C5's shape (T ~ U(0.5, 2)).

Result (python tools/f32_normalisation_probe.py; worst of 64 trajectories, per-power relative error against the same sweep
in fp64, which itself matches the 80-bit oracle to 3e-14 / 4e-12 / 9e-9 at orders 3 / 4 / 5): order 3 3e-5 with or without
scaling; order 4 1e-3 norm-wise and 5e-3 .. 1e-2 per power with or without scaling; order 5 O(1) .. O(100) -- the scaling
moves the fp32 error by a factor of 1..3 in either direction (order 5, S = 64: 25x, from 250 to 10, still useless), and fp64
itself loses 2e-8 .. 6e-8 at order 5 between two formulations.  The ill-conditioning is the spline problem's own
(cond(R_PP) ~ 1e8 at order 5), not an artefact of unscaled time: fp32 ARITHMETIC cannot meet 1e-3 at orders 4-5, so
fp32 stays a STORAGE type (fp64 arithmetic, 4e-8) -- DESIGN.md 5.2."""
import importlib.util, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_spec = importlib.util.spec_from_file_location("csp_tablegen", os.path.join(ROOT, "cs-pathplan_amd", "tablegen.py"))
mt = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mt)
from tests import synth

def tabs(o, dt):
    G, Qt1 = mt.tables_float(o)
    return np.array(Qt1, dtype=dt), np.array(G, dtype=dt)

def solve(o, wp, tm, dt, scale="raw"):
    """sequential block-tridiagonal sweep, batch [B,S+1,3],[B,S]; arithmetic in dtype dt. zero bc."""
    B, S = tm.shape
    N, M = o - 1, 2 * o
    Qt1, G = tabs(o, dt)
    wp = wp.astype(dt); T = tm.astype(dt)
    deriv = np.array(list(range(o)) * 2)   # d_a for a in 0..2o-1
    # local scale tau at waypoints 0..S (interior: geometric mean of adjacent segments)
    if scale == "raw":
        tau = np.ones((B, S + 1), dtype=dt)
    else:
        tau = np.ones((B, S + 1), dtype=dt)
        tau[:, 1:S] = np.sqrt(T[:, :-1] * T[:, 1:]) if scale == "geo" else (0.5 * (T[:, :-1] + T[:, 1:]) if scale == "mean" else np.minimum(T[:, :-1], T[:, 1:]))
        tau[:, 0] = T[:, 0]; tau[:, S] = T[:, -1]
    # per segment scaled Qt: Qs[a][b] = Qt1[a][b] * T^(1-2o) * (T/tau_a)^(d_a) (T/tau_b)^(d_b), a<o uses tau at start, a>=o at end
    def seg(k):
        Tk = T[:, k]
        sa = np.concatenate([np.stack([(Tk / tau[:, k]) ** d for d in range(o)], 1), np.stack([(Tk / tau[:, k + 1]) ** d for d in range(o)], 1)], 1).astype(dt)  # [B,2o]
        pref = (Tk ** dt(1 - 2 * o)).astype(dt)
        return (Qt1[None] * sa[:, :, None] * sa[:, None, :] * pref[:, None, None]).astype(dt), sa
    Q = [seg(k) for k in range(S)]
    fr = lambda: slice(1, o)          # start free derivs
    fe = lambda: slice(o + 1, 2 * o)  # end free derivs
    # forward elimination
    W = np.zeros((B, N, N), dtype=dt); z = np.zeros((B, N, 3), dtype=dt)
    Ws, zs = [None] * S, [None] * S
    dP = (wp[:, 1:] - wp[:, :-1]).astype(dt)
    for k in range(1, S):
        Ql, _ = Q[k - 1]; Qr, _ = Q[k]
        Bk = Ql[:, fe(), fe()] + Qr[:, fr(), fr()]
        A = Ql[:, fe(), fr()]        # coupling to previous waypoint's free derivs  (end, start)
        C = Qr[:, fr(), fe()]
        Sm = (Bk - np.einsum('bij,bjk->bik', A, W)).astype(dt)
        # rhs: -R_FP^T d_F : positions enter via dP: -(Ql[end r][start pos]*P_{k-1} + Ql[end r][end pos]*P_k + Qr[start r][start pos] P_k + Qr[start r][end pos] P_{k+1})
        y = -(Ql[:, fe(), 0][:, :, None] * (-dP[:, k - 1][:, None, :]) + Qr[:, fr(), 0][:, :, None] * (-dP[:, k][:, None, :]))
        # note Q[.,start pos] = -Q[.,end pos]; contribution = Q[.,0]*(P_s - P_e) = -Q[.,0]*dP ; y = -that
        y = (y - np.einsum('bij,bjk->bik', A, z)).astype(dt)
        Si = np.linalg.inv(Sm.astype(dt)).astype(dt)
        W = np.einsum('bij,bjk->bik', Si, C).astype(dt); z = np.einsum('bij,bjk->bik', Si, y).astype(dt)
        Ws[k], zs[k] = W, z
    x = [np.zeros((B, N, 3), dtype=dt) for _ in range(S + 1)]
    for k in range(S - 1, 0, -1):
        x[k] = (zs[k] - np.einsum('bij,bjk->bik', Ws[k], x[k + 1])).astype(dt)
    # recovery: d_hat (scaled by T^deriv) = x_scaled * (T/tau)^(r+1)
    co = np.zeros((B, S, 3, M), dtype=dt)
    for k in range(S):
        _, sa = Q[k]
        Tk = T[:, k]
        d = np.zeros((B, 3, M), dtype=dt)
        d[:, :, 0] = wp[:, k]; d[:, :, o] = wp[:, k + 1]
        for r in range(N):
            d[:, :, 1 + r] = x[k][:, r, :] * sa[:, 1 + r][:, None]
            d[:, :, o + 1 + r] = x[k + 1][:, r, :] * sa[:, o + 1 + r][:, None]
        # p_i = T^-pow_i * sum_a G[i][a] d_hat_a ; pow_i = M-1-i
        # positions via dP to avoid cancellation: G[i][0]+G[i][o]=0 for i<o
        acc = np.einsum('ia,bxa->bxi', G[:, 1:o], d[:, :, 1:o]) + np.einsum('ia,bxa->bxi', G[:, o + 1:], d[:, :, o + 1:])
        acc = acc + G[:, o][None, None, :] * dP[:, k][:, :, None] * (np.arange(M) < o)[None, None, :] + (np.arange(M) == M - 1)[None, None, :] * wp[:, k][:, :, None]
        pw = np.stack([Tk ** dt(-(M - 1 - i)) for i in range(M)], 1).astype(dt)
        co[:, k] = (acc * pw[:, None, :]).astype(dt)
    return co

rng = np.random.default_rng(1)
for o in (3, 4, 5):
    for S in (16, 64):
        wp, tm = synth.make_batch(64, S, config_id=5)
        wp32 = wp.astype(np.float32).astype(np.float64); tm32 = tm.astype(np.float32).astype(np.float64)
        ref = solve(o, wp32, tm32, np.float64, "raw")
        chk = solve(o, wp32, tm32, np.float64, "geo")
        line = "o=%d S=%d  f64 raw-vs-geo %.1e |" % (o, S, synth.rel_err_per_power(chk, ref))
        for sc in ("raw", "geo", "mean", "min"):
            got = solve(o, wp32, tm32, np.float32, sc).astype(np.float64)
            line += "  %s: nw %.1e pp %.1e" % (sc, synth.rel_err(got, ref), synth.rel_err_per_power(got, ref))
        print(line, flush=True)
