"""Exact-rational constant tables (cs-pathplan_amd/tablegen.py) and the scaling laws the kernels use."""
from fractions import Fraction

import numpy as np
import pytest

import importlib.util as _ilu
import os as _os
_spec = _ilu.spec_from_file_location("csp_tablegen", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cs-pathplan_amd", "tablegen.py"))
mt = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(mt)
from oracle import numpy_ref as nr


@pytest.mark.parametrize("o", [1, 2, 3, 4, 5])
def test_scaling_laws(o):
    """M(T)^-1 = diag(T^-pow) G diag(T^deriv);  Qt(T) = T^(1-2o) diag(T^d) Qt1 diag(T^d)  (K4)."""
    G, Qt1 = mt.tables_float(o)
    G, Qt1 = np.array(G), np.array(Qt1)
    m = 2 * o
    for T in (0.37, 1.0, 2.9):
        M = nr.build_M(o, np.array([T]))
        Q = nr.build_Q(o, np.array([T]))
        Minv = np.linalg.inv(M)
        pw = np.arange(m - 1, -1, -1)
        dv = np.array([a % o for a in range(m)])
        Minv_law = (T ** -pw.astype(float))[:, None] * G * (T ** dv.astype(float))[None, :]
        assert np.max(np.abs(Minv - Minv_law)) <= 1e-11 * np.max(np.abs(Minv))
        Qt = Minv.T @ Q @ Minv
        Qt_law = T ** (1 - 2 * o) * (T ** dv.astype(float))[:, None] * Qt1 * (T ** dv.astype(float))[None, :]
        assert np.max(np.abs(Qt - Qt_law)) <= 1e-9 * np.max(np.abs(Qt))


def test_qt4_is_the_integer_table_of_the_survey():
    _, Qt1 = mt.tables(4)
    assert [int(v) for v in Qt1[0]] == [100800, 50400, 10080, 840, -100800, 50400, -10080, 840]
    assert all(v.denominator == 1 for row in Qt1 for v in row)


@pytest.mark.parametrize("o", [2, 3, 4, 5])
def test_symmetries_used_by_the_fixed_kernel(o):
    """Time-reversal symmetry of Qt1 and the position-column identities (minsnap_fixed.hip)."""
    G, Qt1 = mt.tables(o)
    m = 2 * o
    for a in range(m):
        for b in range(m):
            assert Qt1[a][b] == Qt1[b][a]
    for a in range(o):
        for b in range(o):
            # Qt[end a][end b] = (-1)^(a+b) Qt[start a][start b]
            assert Qt1[o + a][o + b] == (-1) ** (a + b) * Qt1[a][b]
            # Qt[start a][end b] = (-1)^(a+b) Qt[end a][start b]
            assert Qt1[a][o + b] == (-1) ** (a + b) * Qt1[o + a][b]
    for a in range(m):
        assert Qt1[a][0] == -Qt1[a][o]          # a constant polynomial costs nothing
    for i in range(o):                          # high coefficient rows see positions through dP only
        assert G[i][0] == -G[i][o]
    for a in range(1, m):                       # every endpoint quantity enters the highest-power coefficient:
        assert G[0][a] != 0                     # the kernels' status test relies on it (recover(), `if (STATUS)`)
    for i in range(o, m):                       # low rows: c_j = start derivative j / j!
        j = m - 1 - i
        for a in range(m):
            assert G[i][a] == (Fraction(1, nr._fact(j)) if a == j else 0)


def test_header_matches_generator(tmp_path):
    p = tmp_path / "t.h"
    mt.emit_header(str(p))
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = open(os.path.join(here, "cs-pathplan_amd", "csrc", "minsnap_tables.h")).read()
    assert p.read_text() == committed, "regenerate with: python cs-pathplan_amd/tablegen.py"


@pytest.mark.parametrize("o", [2, 3, 4, 5])
def test_deviation_quotient_reproduces_the_hermite_samples(o):
    """KQ (tablegen.deviation_quotient): P(sigma) - L(sigma) = sigma (1 - sigma) q(sigma - 1/2) at the 17 samples of the
    path penalty's search (minimum_snap.cpp:408-439) -- exactly, in rational arithmetic, against the Hermite weights HW
    the generic kernel and the oracle-facing tests use."""
    import random
    random.seed(o)
    HW = mt.hermite_sample_weights(o)
    KQ = mt.deviation_quotient(o)
    m = 2 * o
    assert len(KQ) == m - 1 and all(len(r) == m - 2 for r in KQ)
    d = [Fraction(random.randint(-50, 50), random.randint(1, 9)) for _ in range(m)]   # scaled endpoint derivatives [start.., end..]
    P0, P1 = d[0], d[o]
    terms = [P1 - P0] + [d[1 + r] for r in range(o - 1)] + [d[o + 1 + r] for r in range(o - 1)]
    for s_ in range(17):
        sigma = Fraction(s_, 16)
        direct = sum(HW[s_][a] * d[a] for a in range(m)) - (P0 + sigma * (P1 - P0))
        u = sigma - Fraction(1, 2)
        q = sum(terms[t] * sum(KQ[t][i] * u ** i for i in range(m - 2)) for t in range(m - 1))
        assert direct == sigma * (1 - sigma) * q, (o, s_)
