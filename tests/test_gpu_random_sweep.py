"""A slice of tools/stress.py inside the suite: random QPs that mix what the unit tests hold one at a time (short / long / huge rows,
dense diagonal blocks of P with and without stray couplings, equality rows, infinite bounds, scaling on / off, alpha / rho / interval
variations, osqp_update_lin_cost and osqp_update_A between solves) -- each against the CPU oracle: status, iteration count, x and y to
1e-5, no capped linear solve.  Run with the default choice of linear solver and with OSQP_AMD_DENSE_SMALL=0 (the PCG kernels)."""
import os
import sys

import numpy as np
import pytest
from scipy import sparse

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["default", "pcg"])
def test_random_structures_follow_the_oracle(gpu_lib, oracle_mod, monkeypatch, mode):
    import osqp_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stress
    if mode == "pcg":
        monkeypatch.setenv("OSQP_AMD_DENSE_SMALL", "0")
    rng = np.random.default_rng(77)
    kinds = ["sparse", "sparse", "longrows", "blocks", "sparse", "blocks", "sparse", "longrows", "sparse", "blocks", "sparse", "sparse"]
    for case, kind in enumerate(kinds):
        pb = stress.make(rng, kind)
        kw = dict(eps_abs=1e-4, eps_rel=1e-4)
        if rng.random() < 0.3: kw["scaling"] = 0
        if rng.random() < 0.3: kw["alpha"] = float(rng.uniform(1.0, 1.8))
        if rng.random() < 0.3: kw["rho"] = float(10 ** rng.uniform(-2, 1))
        if rng.random() < 0.3: kw["adaptive_rho_interval"] = int(rng.integers(10, 60))
        sg = osqp_amd.OSQP().setup(**pb, **kw)
        so = oracle_mod.OracleOSQP().setup(**pb, **kw)
        for step in ("solve", "update_q", "solve", "update_A", "solve"):
            if step == "update_q":
                q2 = pb["q"] + 0.1 * rng.standard_normal(pb["q"].size); sg.update(q=q2); so.update(q=q2)
            elif step == "update_A":
                A = sparse.csc_matrix(pb["A"]); A.sort_indices()
                Ax2 = A.data * (1.0 + 0.01 * rng.standard_normal(A.nnz)); sg.update(Ax=Ax2); so.update(Ax=Ax2)
            else:
                rg, ro = sg.solve(), so.solve()
                tag = (mode, case, kind, pb["q"].size, kw)
                assert rg.info.status == ro.info.status, tag
                assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates, (tag, rg.info.iter, ro.info.iter)
                if ro.info.status == "solved":
                    assert stress.rel(rg.x, ro.x) < 1e-5 and stress.rel(rg.y, ro.y) < 1e-5, tag
        assert sg.stats()["pcg_forced"] == 0, (mode, case, kind)
