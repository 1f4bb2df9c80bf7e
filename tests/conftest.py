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


@pytest.fixture
def pcg_paths(monkeypatch):
    """Tests that are about the PCG kernels themselves (resident launches, the launch-per-step kernels, PCG statistics): problems of up
    to 1024 dense unknowns would otherwise take the dense-direct solve (OSQP_AMD_DENSE_SMALL, csrc/dense_direct_host.h)."""
    monkeypatch.setenv("OSQP_AMD_DENSE_SMALL", "0")


@pytest.fixture(params=["default", "pcg"])
def both_linear_solvers(request, monkeypatch):
    """Module-wide in test_gpu_parity / test_gpu_statuses: every test runs twice -- with the default choice of linear solver (for
    problems of up to 1024 dense unknowns the dense-direct solve) and with that rule off (OSQP_AMD_DENSE_SMALL=0: the resident PCG or the
    launch-per-step PCG kernels, which problems of this size used before round 3's dense-direct solve and larger ones still use)."""
    if request.param == "pcg":
        monkeypatch.setenv("OSQP_AMD_DENSE_SMALL", "0")
    return request.param
