import json
import os
import sys

import numpy as np
import pytest
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _dec(v):
    if isinstance(v, dict) and v.get("__csc__"):
        d = [float(t) for t in v["data"]]
        return sparse.csc_matrix((d, v["indices"], v["indptr"]), shape=(v["m"], v["n"]))
    if isinstance(v, dict) and v.get("__vec__"):
        return np.array([float(t) for t in v["data"]]).reshape(v["shape"])
    if v in ("inf", "-inf"):
        return float(v)
    return v


def load_golden(name):
    """Fixtures captured from the reference's own generators (tests/golden/make_golden.py)."""
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        g = json.load(f)
    if g["kind"] == "problem":
        pb = {k: _dec(g[k]) for k in "PqAlu"}
        sols = {k: _dec(v) for k, v in g["sols"].items()}
        return pb, sols
    return {k: _dec(v) for k, v in g["data"].items()}


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle.oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library; fails loudly when it is missing or no GPU is visible."""
    import osqp_amd
    L = osqp_amd.lib()
    return L
