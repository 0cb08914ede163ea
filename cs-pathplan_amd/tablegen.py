"""Exact-rational constant tables for the structured minimum-snap formulation.

BUILD TOOL of the product (python cs-pathplan_amd/tablegen.py regenerates csrc/minsnap_tables.h; tests/test_tables.py
checks the committed header against it).  This module derives, in exact `fractions.Fraction`
arithmetic, the two per-order constant tables the HIP kernels consume:

  * ``Qt1[o]`` = M(1)^-T Q(1) M(1)^-1  -- the per-segment cost Hessian expressed in
    endpoint-derivative space at unit segment time, and
  * ``G[o]``   = M(1)^-1               -- the endpoint-derivative -> monomial map
    (highest power first, like the reference's PolyCoeff).

M(T) and Q(T) follow the reference's assembly loops
(/root/reference/math_util/minimum_snap.cpp:250-266 for M, :313-330 for Q), including the
``int`` evaluation of the factorial ratios and of the Q prefactor (minimum_snap.cpp:15-20,
:321-323).  For a general segment time T the scaling laws

    M(T)^-1  = diag(T^-pow_i) . G . diag(T^deriv_a)
    Qt(T)    = T^(1-2o) . diag(T^deriv_a) . Qt1 . diag(T^deriv_b)

hold exactly (proved in DESIGN.md; checked numerically in tests/test_tables.py).

`emit_header()` writes cs-pathplan_amd/csrc/minsnap_tables.h.
"""
from fractions import Fraction
import math

SUPPORTED_ORDERS = (1, 2, 3, 4, 5)


def _fact_int32(x):
    """minimum_snap.cpp:15-20 -- factorial in C `int`; asserts no int32 overflow."""
    f = 1
    for i in range(x, 0, -1):
        f *= i
        assert f < 2**31, "int32 overflow in Factorial (reference UB)"
    return f


def m_unit(o):
    """M(1) for one segment: rows = [derivs 0..o-1 at t=0 ; derivs 0..o-1 at t=1],
    columns = monomial coefficients, highest power first (minimum_snap.cpp:255-263)."""
    m = 2 * o
    M0 = [[Fraction(0)] * m for _ in range(m)]
    for j in range(o):
        for k in range(j, m):
            ratio = _fact_int32(k) // _fact_int32(k - j)  # integer division, exact
            assert _fact_int32(k) % _fact_int32(k - j) == 0
            M0[j][m - 1 - k] = Fraction(ratio) * (1 if k == j else 0)  # pow(0,k-j), pow(0,0)=1
            M0[j + o][m - 1 - k] = Fraction(ratio)
    return M0


def q_unit(o):
    """Q(1) for one segment (minimum_snap.cpp:316-326) with the reference's int arithmetic:
    (a!/(a-o)!) * (b!/(b-o)!) / (a+b-(2o-1)) evaluated left-to-right in `int`."""
    m = 2 * o
    p_order = m - 1
    Q = [[Fraction(0)] * m for _ in range(m)]
    for i in range(m):
        for l in range(m):
            if m - i <= o or m - l <= o:
                continue
            fa = _fact_int32(p_order - i) // _fact_int32(p_order - o - i)
            fb = _fact_int32(p_order - l) // _fact_int32(p_order - o - l)
            prod = fa * fb
            assert prod < 2**31, "int32 overflow in Q prefactor (reference UB)"
            den = p_order - i + p_order - l - (2 * o - 1)
            assert prod % den == 0, "reference int division is inexact for this order"
            Q[i][l] = Fraction(prod // den)
    return Q


def _inverse(A):
    n = len(A)
    aug = [list(A[i]) + [Fraction(int(i == j)) for j in range(n)] for i in range(n)]
    for c in range(n):
        piv = next(r for r in range(c, n) if aug[r][c] != 0)
        aug[c], aug[piv] = aug[piv], aug[c]
        inv = 1 / aug[c][c]
        aug[c] = [v * inv for v in aug[c]]
        for r in range(n):
            if r != c and aug[r][c] != 0:
                f = aug[r][c]
                aug[r] = [a - f * b for a, b in zip(aug[r], aug[c])]
    return [row[n:] for row in aug]


def _matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(len(B))) for j in range(len(B[0]))]
            for i in range(len(A))]


def _transpose(A):
    return [list(r) for r in zip(*A)]


_cache = {}


def tables(o):
    """Returns (G, Qt1) as lists of lists of Fraction for derivative order o."""
    if o not in _cache:
        G = _inverse(m_unit(o))
        Qt1 = _matmul(_matmul(_transpose(G), q_unit(o)), G)
        _cache[o] = (G, Qt1)
    return _cache[o]


def hermite_sample_weights(o):
    """HW[s][a], s = 0..16: the value at t* = T*s/16 of the Hermite basis function of endpoint
    derivative a at unit segment time, i.e. HW[s] = M(1)^-T phi(s/16).  The path penalty samples
    exactly these 17 points per segment (minimum_snap.cpp:408-439); for a segment time T the
    weights are HW[s][a] * T^deriv_a."""
    G, _ = tables(o)
    m = 2 * o
    out = []
    for s_ in range(17):
        tau = Fraction(s_, 16)
        out.append([sum(G[i][a] * tau ** (m - 1 - i) for i in range(m)) for a in range(m)])
    return out


def _polydiv_exact(num, den):
    """Polynomials as ascending coefficient lists of Fractions; returns the quotient, asserts a zero remainder."""
    num = list(num)
    q = [Fraction(0)] * (len(num) - len(den) + 1)
    for k in range(len(q) - 1, -1, -1):
        q[k] = num[k + len(den) - 1] / den[-1]
        for j, d in enumerate(den):
            num[k + j] -= q[k] * d
    assert all(v == 0 for v in num), "not divisible"
    return q


def deviation_quotient(o):
    """KQ[t][i] (t = 0..2o-2, i = 0..2o-3) for the path penalty's 17-sample search (minimum_snap.cpp:408-439).

    In normalised time sigma = t/T the pre-solve polynomial of a segment minus its chord L = P0 + sigma*dP is
        e(sigma) = (H_endpos(sigma) - sigma) dP + sum_r H_start,r(sigma) dh_s[r] + H_end,r(sigma) dh_e[r]
    (dh = endpoint derivatives scaled by T^deriv; H_startpos + H_endpos = 1).  Every one of those 2o-1 basis
    functions vanishes at sigma = 0 and 1, so e(sigma) = sigma (1 - sigma) q(sigma) with q of degree 2o-3.
    KQ[t] holds basis function t's quotient in powers of u = sigma - 1/2 (t = 0: the dP term, 1..o-1: start
    derivatives, o..2o-2: end derivatives): the samples s and 16-s are u and -u, so one even/odd Horner pass
    serves both.  Exact rationals."""
    if o < 2:
        return [[Fraction(0)]]
    G, _ = tables(o)
    m = 2 * o
    den = [Fraction(0), Fraction(1), Fraction(-1)]          # sigma - sigma^2
    out = []
    for a in [o] + list(range(1, o)) + list(range(o + 1, m)):
        h = [G[m - 1 - k][a] for k in range(m)]                # ascending powers of sigma
        if a == o:
            h[1] -= 1
        q = _polydiv_exact(h, den)                             # degree 2o-3, ascending in sigma
        # re-centre: sigma = u + 1/2
        c = [Fraction(0)] * len(q)
        for k, qk in enumerate(q):
            for i in range(k + 1):
                c[i] += qk * math.comb(k, i) * Fraction(1, 2) ** (k - i)
        out.append(c)
    return out


def tables_float(o):
    G, Qt1 = tables(o)
    return ([[float(v) for v in r] for r in G], [[float(v) for v in r] for r in Qt1])


def _c_double(fr):
    v = float(fr)
    if fr.denominator == 1 and abs(fr.numerator) < 2**53:
        return "%d.0" % fr.numerator
    return v.hex()


def emit_header(path):
    lines = [
        "// GENERATED by cs-pathplan_amd/tablegen.py -- do not edit.",
        "// Exact-rational constants of the structured minimum-snap formulation, rounded once to",
        "// fp64.  G<o>  = M(1)^-1 (rows: monomial coefficient, highest power first; columns:",
        "// [derivs 0..o-1 at t=0, derivs 0..o-1 at t=T]);  QT<o> = M(1)^-T Q(1) M(1)^-1.",
        "// M and Q follow /root/reference/math_util/minimum_snap.cpp:250-266 and :313-330.",
        "#pragma once",
        "",
        "#if defined(__HIPCC__)",
        "#define CSP_TABLE_QUAL __device__ static constexpr",
        "#else",
        "#define CSP_TABLE_QUAL static constexpr",
        "#endif",
        "",
        "namespace csp { namespace tables {",
        "",
    ]
    for o in SUPPORTED_ORDERS:
        G, Qt1 = tables(o)
        m = 2 * o
        for name, T in (("G", G), ("QT", Qt1)):
            lines.append("CSP_TABLE_QUAL double %s%d[%d][%d] = {" % (name, o, m, m))
            for r in T:
                lines.append("  {" + ", ".join(_c_double(v) for v in r) + "},")
            lines.append("};")
        lines.append("// HW<o>[s][a] = Hermite basis function of endpoint derivative a at t/T = s/16 (path-penalty samples)")
        lines.append("CSP_TABLE_QUAL double HW%d[17][%d] = {" % (o, m))
        for r in hermite_sample_weights(o):
            lines.append("  {" + ", ".join(_c_double(v) for v in r) + "},")
        lines.append("};")
        kq = deviation_quotient(o)
        lines.append("// KQ<o>[t][i]: (pre-solve polynomial - chord)(sigma) = sigma (1 - sigma) * sum_t term_t * sum_i KQ[t][i] (sigma - 1/2)^i,")
        lines.append("// terms: dP, scaled start derivatives 1..o-1, scaled end derivatives 1..o-1 (tablegen.py deviation_quotient)")
        lines.append("CSP_TABLE_QUAL double KQ%d[%d][%d] = {" % (o, len(kq), len(kq[0])))
        for r in kq:
            lines.append("  {" + ", ".join(_c_double(v) for v in r) + "},")
        lines.append("};")
        lines.append("")
    lines.append("}}  // namespace csp::tables")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    import os, sys
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "csrc",
        "minsnap_tables.h")
    emit_header(out)
    G, Qt1 = tables(4)
    print("Qt1[4] row 0:", [str(v) for v in Qt1[0]])
    print("G[4] col(end pos):", [str(r[4]) for r in G])
    print("wrote", out)
