"""one QP row-partitioned over G ranks (osqp_amd/rowpart.py) -- rows of A and the
stored entries of P sharded, one n-vector all-reduce per PCG iteration.  Two gloo ranks; the result must be the
single-process one: same status, iteration count and rho updates as the oracle, x, y within 1e-6, objective 1e-8.

CPU test: the collective logic with scipy SpMVs.  GPU tests: the same with every SpMV in the shard engines' HIP
kernels (both ranks on the one GPU of the test box), on the small portfolio QP against the oracle and on BASELINE
config 5 at full size against the recorded oracle run."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, GOLDEN

WORKER = os.path.join(ROOT, "tests", "_rowpart_worker.py")


def _run(tmp_path, mode, which, world=2):
    from osqp_amd.launch import spawn_ranks
    import io
    out = str(tmp_path / ("res_%s_%s.npz" % (mode, which)))
    rc = spawn_ranks(world, [sys.executable, WORKER, mode, which, out], stdout=io.StringIO())
    assert rc == 0
    return np.load(out)


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("which", ["portfolio_small", "random"])
def test_row_partition_two_ranks_cpu_matches_oracle(tmp_path, oracle_mod, which):
    from osqp_amd.problems import portfolio_qp, random_sparse_qp
    pb, kw = (portfolio_qp(8, 25, sector_rows=5, seed=3), dict(eps_abs=1e-5, eps_rel=1e-5)) if which == "portfolio_small" else (random_sparse_qp(300, 600, seed=5), {})
    ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    r = _run(tmp_path, "cpu", which)
    assert int(r["world"]) == 2 and str(r["status"]) == ro.info.status == "solved"
    assert int(r["iter"]) == ro.info.iter and int(r["rho_updates"]) == ro.info.rho_updates
    assert _rel(r["x"], ro.x) < 1e-6 and _rel(r["y"], ro.y) < 1e-6
    assert abs(float(r["obj"]) - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
    # one all-reduce per operator apply (PCG iterations + one initial residual per ADMM iteration), one per right-hand side,
    # a few per check / rho update: nothing else
    assert int(r["collectives"]) <= int(r["pcg_iters"]) + 2 * int(r["iter"]) + 3 * (int(r["iter"]) // 25 + 1) + 3 * (int(r["rho_updates"]) + 2)


@pytest.mark.gpu
def test_row_partition_two_ranks_gpu_matches_oracle(gpu_lib, tmp_path, oracle_mod):
    from osqp_amd.problems import portfolio_qp
    pb, kw = portfolio_qp(8, 25, sector_rows=5, seed=3), dict(eps_abs=1e-5, eps_rel=1e-5)
    ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    r = _run(tmp_path, "gpu", "portfolio_small")
    assert str(r["status"]) == ro.info.status == "solved" and int(r["iter"]) == ro.info.iter
    assert _rel(r["x"], ro.x) < 1e-6 and _rel(r["y"], ro.y) < 1e-6
    assert abs(float(r["obj"]) - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))


@pytest.mark.gpu
def test_row_partition_config5_full_size_matches_oracle_golden(gpu_lib, tmp_path):
    """BASELINE config 5 (n = 50000, 400 dense blocks, the 50 000-entry budget row) split over two ranks against
    tests/golden/config5_oracle.json: 325 iterations, same rho updates, objective and subsampled x, y to 1e-6."""
    g = json.load(open(os.path.join(GOLDEN, "config5_oracle.json")))
    r = _run(tmp_path, "gpu", "config5")
    gi = g["info"]
    assert str(r["status"]) == gi["status"] == "solved"
    assert int(r["iter"]) == gi["iters"] and int(r["rho_updates"]) == gi["rho_updates"]
    assert abs(float(r["obj"]) - gi["obj"]) <= 1e-6 * abs(gi["obj"])
    assert np.abs(r["x"][::50] - np.array(g["x_sub"])).max() <= 1e-6 * max(1.0, g["x_inf"])
    assert np.abs(r["y"][::50] - np.array(g["y_sub"])).max() <= 1e-6 * max(1.0, g["y_inf"])
