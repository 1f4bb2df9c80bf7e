"""Slack-like variables eliminated from the linear system (engine.hip: k_elim_refresh; launch-per-step engines).
A variable with one entry in its column of A and no coupling in P is taken out of the PCG system exactly (a block elimination
with a diagonal block); the ADMM iterates must not notice.  Checks, through the C ABI, against the oracle's direct solve and
against the same engine with OSQP_AMD_ELIM=0: Lasso (docs/examples/lasso.rst:41-63: the residual variables y), with
osqp_update_lin_cost, osqp_update_A, osqp_update_rho, warm starts; a factor-model portfolio
(docs/examples/portfolio.rst:51-62: y = F'x) with polish; the plugin boundary (LinSysSolver.solve) against a dense solve."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


class _env:
    def __init__(self, **kw): self.kw = kw
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items(): os.environ[k] = str(v)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _elim_count(s):
    import osqp_amd
    L = osqp_amd.lib()
    L.hipeng_elim_count.restype = C.c_longlong
    L.hipeng_elim_count.argtypes = [C.c_void_p]
    return int(L.hipeng_elim_count(s.engine()))


def _lasso(nf, md, seed):
    from osqp_amd.problems import lasso_qp
    pb = lasso_qp(nf, md, density=0.15, seed=seed)
    return {k: pb[k] for k in "PqAlu"}, pb


@pytest.mark.parametrize("resident", [0, 1])
def test_lasso_with_eliminated_residual_variables(gpu_lib, oracle_mod, resident, pcg_paths):
    """resident = 0: launch-per-step kernels; 1: the resident PCG, whose k_form_K forms the reduced operator (n = 600 here)."""
    import osqp_amd
    data, pb = _lasso(150, 300, 3)
    nf, md = 150, 300
    kw = dict(eps_abs=1e-5, eps_rel=1e-5, adaptive_rho_interval=50)
    with _env(OSQP_AMD_RESIDENT=resident):
        sg = osqp_amd.OSQP().setup(**data, **kw)
        with _env(OSQP_AMD_ELIM=0):
            s0 = osqp_amd.OSQP().setup(**data, **kw)
    so = oracle_mod.OracleOSQP().setup(**data, **kw)
    assert _elim_count(sg) == md and _elim_count(s0) == 0          # y_1..y_md: one row each (y = Ad x - b), P_yy = 1
    assert sg.stats()["resident"] == resident == s0.stats()["resident"]

    def same(rg, ro, r0, txy=1e-6):
        assert rg.info.status == ro.info.status == r0.info.status == "solved"
        assert rg.info.iter == ro.info.iter == r0.info.iter and rg.info.rho_updates == ro.info.rho_updates
        assert _rel(rg.x, ro.x) < txy and _rel(rg.y, ro.y) < txy
        assert _rel(rg.x, r0.x) < txy and _rel(rg.y, r0.y) < txy
        assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
        # residuals: 1e-4 relative, or 1e-3 of the requested eps where they sit far below it (there the last digits are the
        # linear solver's: 5e-8 against eps = 1e-5 in the last solve below)
        assert abs(rg.info.pri_res - ro.info.pri_res) <= 1e-4 * abs(ro.info.pri_res) + 1e-8
        assert abs(rg.info.dua_res - ro.info.dua_res) <= 1e-4 * abs(ro.info.dua_res) + 1e-8

    p0 = sg.stats()["pcg_iters_total"], s0.stats()["pcg_iters_total"]
    same(sg.solve(), so.solve(), s0.solve())
    # the point of it: far fewer PCG iterations for the same ADMM trajectory
    assert (sg.stats()["pcg_iters_total"] - p0[0]) * 2 < s0.stats()["pcg_iters_total"] - p0[1]
    assert sg.stats()["pcg_forced"] == 0
    # osqp_update_lin_cost (gamma), warm-started
    q = np.concatenate([np.zeros(nf + md), 3.0 * np.ones(nf)])
    for s in (sg, so, s0): s.update(q=q)
    same(sg.solve(), so.solve(), s0.solve())
    # osqp_update_A (new data, same pattern): values of the eliminated columns' entries change with the re-equilibration
    A = sparse.csc_matrix(data["A"]); A.sort_indices()
    Ax = A.data * (1.0 + 0.01 * np.random.default_rng(5).standard_normal(A.nnz))
    for s in (sg, so, s0): assert s.update(Ax=Ax) == 0
    same(sg.solve(), so.solve(), s0.solve())
    # osqp_update_rho, then a warm start from given x, y (osqp_warm_start: the eliminated variables' x~ come from x too)
    rng = np.random.default_rng(9)
    x0 = rng.standard_normal(sg.n) * 0.1; y0 = rng.standard_normal(sg.m) * 0.1
    for s in (sg, so, s0):
        s.update_rho(0.3); s.warm_start(x=x0, y=y0)
    same(sg.solve(), so.solve(), s0.solve())
    # bounds of the rows that carry the eliminated variables
    l2 = data["l"].copy(); u2 = data["u"].copy(); l2[:md] -= 0.05; u2[:md] += 0.05      # equalities become ranges: rho of those rows changes type
    for s in (sg, so, s0): s.update(l=l2, u=u2)
    same(sg.solve(), so.solve(), s0.solve())


def test_factor_portfolio_with_polish(gpu_lib, oracle_mod):
    """minimize x'Dx + y'y - mu'x / gamma  s.t. y = F'x, 1'x = 1, x >= 0 (docs/examples/portfolio.rst:41-65): the k factor
    variables y are slack-like.  Also polish=1: the second plugin instance (reduced A, sigma = delta) eliminates them too."""
    import osqp_amd
    rng = np.random.default_rng(4)
    n, k = 400, 40
    F = sparse.random(n, k, density=0.5, random_state=5, data_rvs=rng.standard_normal, format="csc")
    D = sparse.diags(rng.uniform(0.1, 1.0, n) * np.sqrt(k), format="csc")
    mu = rng.standard_normal(n)
    P = sparse.block_diag([2 * D, 2 * sparse.eye(k)], format="csc")
    q = np.concatenate([-mu, np.zeros(k)])
    A = sparse.vstack([sparse.hstack([F.T, -sparse.eye(k)]), sparse.hstack([sparse.csc_matrix(np.ones((1, n))), sparse.csc_matrix((1, k))]),
                       sparse.hstack([sparse.eye(n), sparse.csc_matrix((n, k))])], format="csc")
    l = np.concatenate([np.zeros(k), [1.0], np.zeros(n)]); u = np.concatenate([np.zeros(k), [1.0], np.ones(n)])
    pb = dict(P=P, q=q, A=A, l=l, u=u)
    kw = dict(eps_abs=1e-5, eps_rel=1e-5, adaptive_rho_interval=25, polish=1)
    with _env(OSQP_AMD_RESIDENT=0):
        sg = osqp_amd.OSQP().setup(**pb, **kw)
    so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    assert _elim_count(sg) == k
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status == "solved" and rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
    assert rg.info.status_polish == ro.info.status_polish == 1
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))


def test_plugin_solve_with_eliminated_variables(gpu_lib):
    """LinSysSolver.solve (include/types.h:300-301; qdldl_interface.c:350-376) on a KKT system whose last variables are
    slack-like: [x~; z~] against a dense solve of the reduced system, before and after update_rho_vec."""
    from osqp_amd import abi
    rng = np.random.default_rng(12)
    nx, ny, mi = 120, 60, 90
    n, m = nx + ny, ny + mi
    B = sparse.random(nx, nx, density=0.05, random_state=1, data_rvs=rng.standard_normal, format="csc")
    P = sparse.block_diag([B @ B.T + 0.1 * sparse.eye(nx), sparse.diags(rng.uniform(0.0, 2.0, ny))], format="csc")   # P_yy >= 0, some near 0
    G = sparse.random(ny, nx, density=0.2, random_state=2, data_rvs=rng.standard_normal, format="csc")
    H = sparse.random(mi, nx, density=0.1, random_state=3, data_rvs=rng.standard_normal, format="csc")
    A = sparse.vstack([sparse.hstack([G, sparse.diags(rng.uniform(0.5, 2.0, ny) * rng.choice([-1.0, 1.0], ny))]),
                       sparse.hstack([H, sparse.csc_matrix((mi, ny))])], format="csc")
    sigma = 1e-6
    rho = np.concatenate([100.0 * np.ones(ny), 0.1 * np.ones(mi)])
    L = gpu_lib
    S = C.POINTER(abi.LinSysSolver)
    L.init_linsys_solver_hip_pcg.restype = abi.c_int
    L.init_linsys_solver_hip_pcg.argtypes = [C.POINTER(S), C.POINTER(abi.csc), C.POINTER(abi.csc), abi.c_float, abi.c_float_p, abi.c_int]
    hp, ha = abi.CscHolder(sparse.triu(P, format="csc")), abi.CscHolder(A)
    with _env(OSQP_AMD_RESIDENT=0):
        s = S()
        r = abi.as_f64(rho)
        assert L.init_linsys_solver_hip_pcg(C.byref(s), C.byref(hp.struct), C.byref(ha.struct), sigma, abi.fptr(r), 0) == 0
    for rr in (rho, 3.0 * rho):
        r = abi.as_f64(rr)
        assert s.contents.update_rho_vec(s, abi.fptr(r)) == 0
        rhs = rng.standard_normal(n + m)
        b = abi.as_f64(rhs).copy()
        assert s.contents.solve(s, abi.fptr(b)) == 0
        K = (P + sigma * sparse.eye(n) + A.T @ sparse.diags(rr) @ A).toarray()
        xt = np.linalg.solve(K, rhs[:n] + A.T @ (rr * rhs[n:]))
        assert _rel(b[:n], xt) < 1e-8 and _rel(b[n:], A @ xt) < 1e-8
    s.contents.free(s)


def test_random_qps_with_slack_variables(gpu_lib, oracle_mod):
    """Random QPs in the form  min 1/2 x'Px + q'x + 1/2 s'Ws  s.t.  l <= Gx + c.s <= u  plus boxes on part of x: the slack
    variables s (one row each; W_ii = 0 for some: only sigma holds them, some rows equalities, some one-sided) are eliminated;
    settings vary (scaling on/off, rho, alpha).  Same iteration count, status and rho updates as the oracle, x and y to 1e-6."""
    import osqp_amd
    for case in range(8):
        rng = np.random.default_rng(100 + case)
        nx = int(rng.integers(40, 260)); ns = int(rng.integers(5, nx)); nb = int(rng.integers(0, nx // 2))
        B = sparse.random(nx, nx, density=min(1.0, 3.0 / nx), random_state=case, data_rvs=rng.standard_normal, format="csc")
        W = rng.uniform(0.0, 2.0, ns) * (rng.random(ns) < 0.7)          # 30 % of the slack variables have no cost at all
        P = sparse.block_diag([B @ B.T + sparse.diags(rng.uniform(0.01, 1.0, nx)), sparse.diags(W)], format="csc")
        G = sparse.random(ns, nx, density=min(1.0, 6.0 / nx), random_state=case + 50, data_rvs=rng.standard_normal, format="csc")
        cs = rng.uniform(0.3, 3.0, ns) * rng.choice([-1.0, 1.0], ns)
        rows = [sparse.hstack([G, sparse.diags(cs)])]
        if nb: rows.append(sparse.hstack([sparse.eye(nx, format="csc")[:nb], sparse.csc_matrix((nb, ns))]))
        A = sparse.vstack(rows, format="csc")
        x0 = rng.standard_normal(nx + ns)
        Ax = A @ x0
        l = Ax - rng.uniform(0.0, 1.0, A.shape[0]); u = Ax + rng.uniform(0.0, 1.0, A.shape[0])
        eq = rng.random(A.shape[0]) < 0.3; l[eq] = u[eq] = Ax[eq]
        lo = rng.random(A.shape[0]) < 0.15; l[lo & ~eq] = -np.inf
        pb = dict(P=P, q=rng.standard_normal(nx + ns), A=A, l=l, u=u)
        kw = [dict(), dict(scaling=0), dict(rho=1.5, alpha=1.2), dict(eps_abs=1e-6, eps_rel=1e-6, adaptive_rho_interval=15)][case % 4]
        with _env(OSQP_AMD_RESIDENT=0):
            sg = osqp_amd.OSQP().setup(**pb, **kw)
        so = oracle_mod.OracleOSQP().setup(**pb, **kw)
        assert _elim_count(sg) >= ns, (case, _elim_count(sg), ns)      # (an x_j with no coupling in P and a single row of its own qualifies too)
        rg, ro = sg.solve(), so.solve()
        assert rg.info.status == ro.info.status, (case, rg.info.status, ro.info.status)
        assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates, (case, rg.info.iter, ro.info.iter)
        if ro.info.status == "solved":
            assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6, (case, _rel(rg.x, ro.x), _rel(rg.y, ro.y))
        assert sg.stats()["pcg_forced"] == 0
