import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cases(fname):
    """Golden cases with hex floats decoded into numpy arrays."""
    with open(os.path.join(GOLDEN, fname)) as f:
        doc = json.load(f)
    out = []
    for c in doc["cases"]:
        o, S = c["order"], c["segments"]
        d = dict(c)
        unhex = lambda k: np.array([float.fromhex(v) for v in c[k]])
        d["path"] = unhex("path").reshape(S + 1, 3)
        d["time"] = unhex("time")
        d["vel"] = unhex("vel").reshape(2, 3)
        d["acc"] = unhex("acc").reshape(2, 3)
        d["coeff"] = unhex("coeff").reshape(S, 3, 2 * o)
        d["max_dev"] = float.fromhex(c["max_dev"])
        d["bc"] = np.stack([d["vel"][0], d["vel"][1], d["acc"][0], d["acc"][1]])
        out.append(d)
    return out


@pytest.fixture(scope="session")
def csp():
    """The product binding.  Import failure (extension not built) is a hard error, never a skip."""
    return importlib.import_module("cs-pathplan_amd")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
