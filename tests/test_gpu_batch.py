"""GPU tests of the batched engine (one workgroup per QP, register-tiled K^-1)
against the per-QP CPU oracle: identical iteration counts, status and rho
updates; x, y within 1e-6 relative; objective within 1e-8 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _oracle_loop(oracle_mod, s, Q, L, U, **kw):
    out = []
    for b in range(Q.shape[0]):
        r = oracle_mod.OracleOSQP().setup(P=s["P"], q=Q[b], A=s["A"], l=L[b], u=U[b], **kw).solve()
        out.append(r)
    return out


@pytest.mark.parametrize("kw", [{}, dict(eps_abs=1e-5, eps_rel=1e-5), dict(scaling=0), dict(alpha=1.0, rho=1.0)])
def test_mpc_batch_matches_oracle(gpu_lib, oracle_mod, kw):
    import osqp_amd
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(batch=24)
    assert (s["n"], s["m"]) == (120, 240)
    Q = Q + 0.05 * np.random.default_rng(0).standard_normal(Q.shape)   # per-QP cost scaling differs
    ref = _oracle_loop(oracle_mod, s, Q, L, U, **kw)
    r = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, **kw).solve()
    for b, ro in enumerate(ref):
        assert r.status_val[b] == ro.info.status_val, b
        assert r.iter[b] == ro.info.iter and r.rho_updates[b] == ro.info.rho_updates, (b, r.iter[b], ro.info.iter)
        if ro.info.status == "solved":
            assert _rel(r.x[b], ro.x) < 1e-6 and _rel(r.y[b], ro.y) < 1e-6, b
            assert abs(r.obj_val[b] - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val)), b
            assert abs(r.pri_res[b] - ro.info.pri_res) <= 1e-4 * ro.info.pri_res + 1e-9
            assert abs(r.dua_res[b] - ro.info.dua_res) <= 1e-4 * ro.info.dua_res + 1e-9


def test_batch_update_and_warm_start(gpu_lib, oracle_mod):
    """MPC loop: solve, shift the initial state (new l, u), warm-started re-solve."""
    import osqp_amd
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(batch=8)
    bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U)
    r1 = bs.solve()
    _, Q2, L2, U2 = mpc_batch(batch=8, seed0=100)
    assert bs.update(L=L2, U=U2) == 0
    r2 = bs.solve()
    for b in range(8):
        so = oracle_mod.OracleOSQP().setup(P=s["P"], q=Q[b], A=s["A"], l=L[b], u=U[b])
        so.solve(); so.update(l=L2[b], u=U2[b]); ro = so.solve()
        assert r2.status_val[b] == ro.info.status_val and r2.iter[b] == ro.info.iter, (b, r2.iter[b], ro.info.iter)
        assert _rel(r2.x[b], ro.x) < 1e-6


def test_batch_small_tile_and_infeasible(gpu_lib, oracle_mod):
    """n <= 64 uses the 4x4 register tile; one QP of the batch is primal infeasible."""
    import osqp_amd
    from osqp_amd import abi
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(40, 60, nnz_per_col=6, seed=4)
    rng = np.random.default_rng(1)
    B = 6
    Q = pb["q"] + 0.1 * rng.standard_normal((B, 40))
    L = np.tile(pb["l"], (B, 1)); U = np.tile(pb["u"], (B, 1))
    # contradictory duplicate rows make QP 3 infeasible
    A = pb["A"].tolil(); A[1, :] = A[0, :]; A = A.tocsc()
    L[3, 0], U[3, 0] = 5.0, 6.0
    L[3, 1], U[3, 1] = -6.0, -5.0
    r = osqp_amd.BatchOSQP().setup(pb["P"], A, Q, L, U, max_iter=4000).solve()
    for b in range(B):
        ro = oracle_mod.OracleOSQP().setup(P=pb["P"], q=Q[b], A=A, l=L[b], u=U[b], max_iter=4000).solve()
        assert r.status_val[b] == ro.info.status_val and r.iter[b] == ro.info.iter, (b, r.status_val[b], ro.info.status)
        if ro.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE:
            assert np.all(r.x[b] == abi.OSQP_NAN)
            assert _rel(r.prim_inf_cert[b], ro.prim_inf_cert) < 1e-5
        else:
            assert _rel(r.x[b], ro.x) < 1e-6


def test_full_batch_1024_properties(gpu_lib):
    """BASELINE config 4 at full size: all 1024 QPs solved; KKT residuals of every
    returned point below tolerance (checked in numpy, independent of the oracle)."""
    import osqp_amd
    from scipy import sparse
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(batch=1024)
    r = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U).solve()
    assert np.all(r.status_val == 1)
    P = (s["P"] + sparse.triu(s["P"], 1).T).toarray(); A = s["A"].toarray()
    AX = r.x @ A.T
    pri = np.abs(AX - np.clip(AX, L, U)).max(axis=1)
    dua = np.abs(r.x @ P + Q + r.y @ A).max(axis=1)
    assert np.all(pri <= 1e-3 + 1e-3 * np.abs(AX).max(axis=1) + 1e-9)
    assert np.all(dua <= 1e-3 + 1e-3 * np.maximum(np.abs(r.x @ P).max(axis=1), np.abs(r.y @ A).max(axis=1)) + 1e-9)


def test_full_batch_1024_matches_oracle(gpu_lib, oracle_mod):
    """BASELINE config 4 at full size, every one of the 1024 QPs against its own oracle solve:
    status, iteration count and rho updates identical; x, y 1e-6 relative; objective 1e-8."""
    import osqp_amd
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(batch=1024)
    r = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U).solve()
    worst_x = worst_y = 0.0
    for b in range(1024):
        ro = oracle_mod.OracleOSQP().setup(P=s["P"], q=Q[b], A=s["A"], l=L[b], u=U[b]).solve()
        assert r.status_val[b] == ro.info.status_val == 1, b
        assert r.iter[b] == ro.info.iter and r.rho_updates[b] == ro.info.rho_updates, (b, r.iter[b], ro.info.iter)
        worst_x = max(worst_x, _rel(r.x[b], ro.x)); worst_y = max(worst_y, _rel(r.y[b], ro.y))
        assert abs(r.obj_val[b] - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val)), b
    assert worst_x < 1e-6 and worst_y < 1e-6, (worst_x, worst_y)


def test_dispatch_order_does_not_change_results(gpu_lib, monkeypatch):
    """From the second solve on, workgroups take the QPs longest-first (by the previous
    solve's iteration counts).  Scheduling only: every solve of a sequence must be
    bit-identical to the same sequence run in index order (OSQP_AMD_BATCH_LPT=0),
    for a batch size that is not a multiple of anything."""
    import osqp_amd
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(batch=333)
    runs = {}
    for lpt in ("0", "1"):
        monkeypatch.setenv("OSQP_AMD_BATCH_LPT", lpt)
        bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, warm_start=0)
        runs[lpt] = [bs.solve() for _ in range(3)]
    assert len(set(runs["0"][0].iter.tolist())) > 1            # a real spread of iteration counts to sort
    for a, b in zip(runs["0"], runs["1"]):
        assert np.array_equal(a.iter, b.iter) and np.array_equal(a.status_val, b.status_val)
        assert np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y)


def test_one_qp_per_stream_matches_sequential(gpu_lib):
    """Several workspaces (each with its own HIP stream) solved concurrently from a
    thread pool give bit-identical results to solving them one after the other."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    pbs = [random_sparse_qp(400, 800, seed=20 + k) for k in range(4)]
    seq = [osqp_amd.OSQP().setup(**pb).solve() for pb in pbs]
    par = osqp_amd.solve_many([osqp_amd.OSQP().setup(**pb) for pb in pbs], max_workers=4)
    for a, b in zip(seq, par):
        assert a.info.iter == b.info.iter and a.info.status == b.info.status == "solved"
        assert np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y)


def test_batch_members_above_128_variables_go_one_per_stream(gpu_lib, oracle_mod):
    """n > 128 does not fit the register-tiled batch kernel: BatchOSQP then drives one single-QP engine per member
    (one HIP stream each) behind the same interface; results per QP equal the oracle's."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    n, m, B = 150, 220, 5
    base = random_sparse_qp(n, m, nnz_per_col=8, seed=31)
    rng = np.random.default_rng(7)
    Q = np.array([base["q"] + 0.3 * rng.standard_normal(n) for _ in range(B)])
    L = np.array([base["l"] - rng.uniform(0, 0.2, m) for _ in range(B)]); U = np.array([base["u"] + rng.uniform(0, 0.2, m) for _ in range(B)])
    bs = osqp_amd.BatchOSQP().setup(base["P"], base["A"], Q, L, U, eps_abs=1e-5, eps_rel=1e-5)
    r = bs.solve()
    assert r.x.shape == (B, n) and r.y.shape == (B, m)
    for b in range(B):
        ro = oracle_mod.OracleOSQP().setup(P=base["P"], q=Q[b], A=base["A"], l=L[b], u=U[b], eps_abs=1e-5, eps_rel=1e-5).solve()
        assert r.status_val[b] == ro.info.status_val == 1 and r.iter[b] == ro.info.iter
        assert np.abs(r.x[b] - ro.x).max() <= 1e-6 * max(1.0, np.abs(ro.x).max())
        assert np.abs(r.y[b] - ro.y).max() <= 1e-6 * max(1.0, np.abs(ro.y).max())
    assert bs.update(Q=Q[::-1].copy()) == 0
    r2 = bs.solve()
    ro = oracle_mod.OracleOSQP().setup(P=base["P"], q=Q[B - 1], A=base["A"], l=L[0], u=U[0], eps_abs=1e-5, eps_rel=1e-5).solve()
    assert np.abs(r2.x[0] - ro.x).max() <= 1e-5 * max(1.0, np.abs(ro.x).max())
    with pytest.raises(ValueError):
        bs.update(Q=Q[:, :10])
