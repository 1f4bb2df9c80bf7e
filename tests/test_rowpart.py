"""one QP row-partitioned over G ranks (osqp_amd/rowpart.py) -- rows of A and the
stored entries of P sharded, one n-vector all-reduce per PCG iteration.  Two gloo ranks; the result must be the
single-process one: same status, iteration count and rho updates as the oracle, x, y within 1e-6, objective 1e-8.

CPU test: the collective logic with scipy SpMVs.  GPU tests: the same with every SpMV in the shard engines' HIP
kernels (both ranks on the one GPU of the test box), on the small portfolio QP against the oracle and on BASELINE
config 5 at full size against the recorded oracle run; and the same solves with the loop driven from C
(include/osqp_amd_rowpart.h: osqp_amd_rp_solve issues kernels and collectives on the shard engine's stream)."""
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


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["portfolio_small", "random"])
def test_native_row_partition_two_ranks_gpu_matches_oracle(gpu_lib, tmp_path, oracle_mod, which):
    """The loop of osqp_amd_rp_solve (C; kernels k_rp_* + the shard engine's k_spmv; the gloo group lent as the collective
    callback), two ranks on the one GPU of the box: same status, iteration count and rho updates as the oracle's direct
    solve, x, y 1e-6, objective 1e-8 -- and the same collective budget as the Python-driven variant plus the iterations
    issued past convergence (the host looks at the flag once per group of PCG iterations)."""
    from osqp_amd.problems import portfolio_qp, random_sparse_qp
    pb, kw = (portfolio_qp(8, 25, sector_rows=5, seed=3), dict(eps_abs=1e-5, eps_rel=1e-5)) if which == "portfolio_small" else (random_sparse_qp(300, 600, seed=5), {})
    ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    r = _run(tmp_path, "native", which)
    assert int(r["world"]) == 2 and str(r["status"]) == ro.info.status == "solved"
    assert int(r["iter"]) == ro.info.iter and int(r["rho_updates"]) == ro.info.rho_updates
    assert _rel(r["x"], ro.x) < 1e-6 and _rel(r["y"], ro.y) < 1e-6
    assert abs(float(r["obj"]) - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
    assert int(r["collectives"]) <= 2 * int(r["pcg_iters"]) + 4 * int(r["iter"]) + 3 * (int(r["iter"]) // 25 + 1) + 3 * (int(r["rho_updates"]) + 2)


@pytest.mark.gpu
def test_native_row_partition_config5_full_size_matches_oracle_golden(gpu_lib, tmp_path):
    """BASELINE config 5 at full size over two ranks through osqp_amd_rp_solve against tests/golden/config5_oracle.json."""
    g = json.load(open(os.path.join(GOLDEN, "config5_oracle.json")))
    r = _run(tmp_path, "native", "config5")
    gi = g["info"]
    assert str(r["status"]) == gi["status"] == "solved"
    assert int(r["iter"]) == gi["iters"] and int(r["rho_updates"]) == gi["rho_updates"]
    assert abs(float(r["obj"]) - gi["obj"]) <= 1e-6 * abs(gi["obj"])
    assert np.abs(r["x"][::50] - np.array(g["x_sub"])).max() <= 1e-6 * max(1.0, g["x_inf"])
    assert np.abs(r["y"][::50] - np.array(g["y_sub"])).max() <= 1e-6 * max(1.0, g["y_inf"])


@pytest.mark.gpu
def test_native_row_partition_one_rank_and_the_rccl_provider(gpu_lib, oracle_mod):
    """One rank (no process group): the C loop alone against the oracle.  Then the built-in RCCL provider on a one-rank
    communicator -- librccl found through dlopen, ncclCommInitRank, ncclAllReduce in place on the engine's stream for every
    collective of the solve -- in a process of its own WITHOUT torch (tools/rccl_world1_probe.py: beside torch's bundled ROCm
    libraries the system librccl finds a second, uninitialised HSA runtime and refuses; a C caller has one ROCm): it must
    change nothing, bit for bit."""
    import subprocess
    from osqp_amd import rowpart
    from osqp_amd.problems import portfolio_qp
    pb, kw = portfolio_qp(8, 40, sector_rows=6, seed=4), dict(eps_abs=1e-5, eps_rel=1e-5)
    ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    scaled = rowpart.scaled_problem_from_engine(**pb)
    s = rowpart.NativeRowPartitionedOSQP(collective="group").setup(scaled, device=0, **kw)
    r = s.solve()
    assert r.info.status == ro.info.status == "solved" and r.info.iter == ro.info.iter and r.info.rho_updates == ro.info.rho_updates
    assert _rel(r.x, ro.x) < 1e-6 and _rel(r.y, ro.y) < 1e-6 and r.info.collectives == 0
    s.cleanup()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_probe.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "use_rccl -> 0" in p.stdout and "identical: True torch loaded: False" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("max_iter", [40, 75])
def test_native_row_partition_stops_like_the_oracle(gpu_lib, oracle_mod, max_iter):
    """osqp_amd_rp_solve at an iteration cap: the same status as the oracle (maximum iterations reached, or solved inaccurate when
    the approximate test passes), the same iterates to 1e-6."""
    from osqp_amd import rowpart
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(300, 600, seed=5)
    kw = dict(max_iter=max_iter, eps_abs=1e-6, eps_rel=1e-6)
    ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    scaled = rowpart.scaled_problem_from_engine(**pb)
    s = rowpart.NativeRowPartitionedOSQP(collective="group").setup(scaled, device=0, **kw)
    r = s.solve()
    assert r.info.status == ro.info.status and r.info.iter == ro.info.iter == max_iter
    assert ro.info.status in ("maximum iterations reached", "solved inaccurate")
    s.cleanup()
