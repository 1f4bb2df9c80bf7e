"""Dense-direct solve (engine.hip res_kind 4, csrc/dense_direct.h): the reduced matrix formed densely on the matrix cores
(TN GEMM over the dense rows of A + scattered short rows, P, sigma + a private sparse Schur complement), inverted explicitly by
blocked Gauss-Jordan, one pass over the inverse per ADMM iteration.  Parity through the C ABI against the CPU oracle's direct
LDL^T solve: status, iteration count, rho updates, x, y 1e-6 -- on the Lasso family it is built for (BASELINE config 3 shape) and,
forced with OSQP_AMD_DENSE_DIRECT=2, on QPs with coupling in P and rows of every length."""
import contextlib
import ctypes as C
import os

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _info(solver):
    import osqp_amd
    L = osqp_amd.lib()
    L.hipeng_resident_info.restype = C.c_int
    L.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    out = (C.c_longlong * 16)()
    assert L.hipeng_resident_info(solver.engine(), out) == 0
    return dict(built=out[0], in_use=out[1], dense_unknowns=out[3], cholesky=out[5], form=out[9], sparse_unknowns=out[15])


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def test_blocked_inversion_against_numpy(gpu_lib):
    """hipeng_dense_invert_selftest: explicit inverse of a symmetric positive definite matrix by blocked Gauss-Jordan, every
    flop outside the 128 x 128 pivot blocks in the MFMA GEMM kernel."""
    import osqp_amd
    f = osqp_amd.lib().hipeng_dense_invert_selftest
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for n in (128, 384, 1280):
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n, n + 50))
        A = G @ G.T / n + 0.1 * np.eye(n)
        Ainv = np.zeros((n, n)); ms = (C.c_double * 2)()
        assert f(n, A.ctypes.data_as(C.c_void_p), Ainv.ctypes.data_as(C.c_void_p), ms) == 0
        assert np.abs(Ainv @ A - np.eye(n)).max() < 1e-10
        assert np.abs(Ainv - np.linalg.inv(A)).max() <= 1e-10 * np.abs(Ainv).max()
    A = np.eye(256); A[200, 200] = -1.0                       # a pivot that is not positive is reported
    Ainv = np.zeros((256, 256))
    assert f(256, A.ctypes.data_as(C.c_void_p), Ainv.ctypes.data_as(C.c_void_p), None) == 1


@pytest.mark.parametrize("nf,md", [(60, 150), (300, 700), (700, 1500)])
def test_lasso_family_matches_oracle(gpu_lib, oracle_mod, nf, md):
    """docs/examples/lasso.rst shape: the residual variables leave by the engine's elimination, the bound variables t by the
    solver's own Schur complement, the features are the dense unknowns; the gamma sweep (osqp_update_lin_cost), an
    osqp_update_rho, and new values of A (osqp_update_A) follow the oracle too."""
    import osqp_amd
    from osqp_amd.problems import lasso_qp
    full = lasso_qp(nf, md, seed=nf)
    pb = {k: v for k, v in full.items() if k in "PqAlu"}
    with _env(OSQP_AMD_DENSE_DIRECT=2, OSQP_AMD_RESIDENT=0):        # (small members of the family fit the resident PCG, which comes first)
        sg = osqp_amd.OSQP().setup(**pb)
    so = oracle_mod.OracleOSQP().setup(**pb)
    inf = _info(sg)
    assert inf["built"] and inf["form"] == 4 and inf["dense_unknowns"] == nf and inf["sparse_unknowns"] == nf, inf
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status == "solved"
    assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-6 * max(1.0, abs(ro.info.obj_val))
    st = sg.stats()
    assert st["pcg_iters_total"] == rg.info.iter and st["pcg_forced"] == 0          # one application of the inverse per ADMM iteration
    for gamma in (0.5, 2.0):
        q = np.concatenate([np.zeros(nf + md), gamma * np.ones(nf)])
        sg.update(q=q); so.update(q=q)
        r1, r2 = sg.solve(), so.solve()
        assert r1.info.iter == r2.info.iter and _rel(r1.x, r2.x) < 1e-6 and _rel(r1.y, r2.y) < 1e-6
    sg.update_rho(0.3); so.update_rho(0.3)
    r1, r2 = sg.solve(), so.solve()
    assert r1.info.iter == r2.info.iter and _rel(r1.x, r2.x) < 1e-6 and _rel(r1.y, r2.y) < 1e-6
    A = sparse.csc_matrix(pb["A"]); A.sort_indices()
    Ax = A.data * (1.0 + 0.01 * np.random.default_rng(1).standard_normal(A.nnz))
    sg.update(Ax=Ax); so.update(Ax=Ax)
    r1, r2 = sg.solve(), so.solve()
    assert r1.info.iter == r2.info.iter and _rel(r1.x, r2.x) < 1e-6 and _rel(r1.y, r2.y) < 1e-6
    # the same problem on the launch-per-step PCG kernels
    with _env(OSQP_AMD_DENSE_DIRECT=0, OSQP_AMD_RESIDENT=0):
        s2 = osqp_amd.OSQP().setup(**pb)
    assert _info(s2)["form"] != 4
    r3 = s2.solve()
    assert r3.info.iter == rg.info.iter and _rel(r3.x, rg.x) < 1e-6 and _rel(r3.y, rg.y) < 1e-6


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_general_qps_forced_dense_match_oracle(gpu_lib, oracle_mod, seed):
    """Forced (OSQP_AMD_DENSE_DIRECT=2) on QPs it would not choose: P with off-diagonal coupling, rows of A from one entry to
    dense, equality rows, slack-like variables (eliminated by the engine) and box rows (the solver's Schur complement) together."""
    import osqp_amd
    rng = np.random.default_rng(seed)
    n, md = 260, 400
    Ad = sparse.random(md, n, density=0.3, random_state=seed, data_rvs=rng.standard_normal, format="csc")       # dense rows (78 entries)
    As = sparse.random(150, n, density=0.01, random_state=seed + 7, data_rvs=rng.standard_normal, format="csc")  # short rows
    box = sparse.eye(n, format="csc")
    ns = 40                                                     # slack variables: one equality row each
    slack_rows = sparse.hstack([sparse.random(ns, n, density=0.05, random_state=seed + 3, data_rvs=rng.standard_normal, format="csc"), -sparse.eye(ns)], format="csc")
    A = sparse.vstack([sparse.hstack([Ad, sparse.csc_matrix((md, ns))]), sparse.hstack([As, sparse.csc_matrix((150, ns))]),
                       sparse.hstack([box, sparse.csc_matrix((n, ns))]), slack_rows], format="csc")
    G = sparse.random(n, n, density=0.02, random_state=seed + 11, data_rvs=rng.standard_normal, format="csc")
    Pxx = (G @ G.T + 0.05 * sparse.eye(n)).tocsc()
    P = sparse.block_diag([Pxx, 0.5 * sparse.eye(ns)], format="csc")
    q = rng.standard_normal(n + ns)
    l = np.concatenate([-1.0 - rng.random(md), -0.5 * np.ones(150), -np.ones(n), np.zeros(ns)])
    u = np.concatenate([1.0 + rng.random(md), 0.5 * np.ones(150), np.ones(n), np.zeros(ns)])
    l[:20] = u[:20] = 0.1                                       # some equality rows among the dense ones
    pb = dict(P=sparse.triu(P, format="csc"), q=q, A=A, l=l, u=u)
    kw = dict(eps_abs=1e-5, eps_rel=1e-5)
    with _env(OSQP_AMD_DENSE_DIRECT=2, OSQP_AMD_RESIDENT=0):
        sg = osqp_amd.OSQP().setup(**pb, **kw)
    so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    assert _info(sg)["form"] == 4
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status and rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    Pu = pb["P"].copy(); Pu.data = Pu.data * 1.05
    sg.update(Px=Pu.data); so.update(Px=Pu.data)                # new values of P: the dense matrix is formed again
    r1, r2 = sg.solve(), so.solve()
    assert r1.info.iter == r2.info.iter and _rel(r1.x, r2.x) < 1e-6 and _rel(r1.y, r2.y) < 1e-6


def test_an_inverse_that_fails_its_check_is_computed_again_by_cholesky(gpu_lib, oracle_mod):
    """The block sweeps are Gauss-Jordan in blocks: their error grows like cond^2 eps.  Every fresh inverse is checked against the
    matrix as formed (dd_refresh); on a reduced matrix of condition 1e7 (eigenvalues of P from 1e-5 up, equality rows at 1e3 rho)
    the sweeps' inverse fails the check and the blocked Cholesky route (cond eps) takes over: the engine keeps the dense solve, no
    false non-convexity report, same trajectory as the oracle."""
    import osqp_amd
    rng = np.random.default_rng(4)
    n, m = 200, 60
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    P = Q @ np.diag(10.0 ** rng.uniform(-5, 1, n)) @ Q.T
    P = sparse.csc_matrix(0.5 * (P + P.T))
    A = sparse.random(m, n, density=0.2, format="csc", random_state=rng)
    x0 = rng.standard_normal(n); Ax = A @ x0
    l = Ax - rng.uniform(0, 1, m); u = Ax + rng.uniform(0, 1, m)
    l[:20] = u[:20] = Ax[:20]
    pb = dict(P=sparse.triu(P, format="csc"), q=rng.standard_normal(n), A=A, l=l, u=u)
    with _env(OSQP_AMD_DENSE_DIRECT=2, OSQP_AMD_RESIDENT=0):
        sg = osqp_amd.OSQP().setup(**pb)
    inf = _info(sg)
    assert inf["form"] == 4 and inf["cholesky"] == 1, inf
    rg, ro = sg.solve(), oracle_mod.OracleOSQP().setup(**pb).solve()
    assert rg.info.status == ro.info.status == "solved" and rg.info.iter == ro.info.iter
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert _info(sg)["form"] == 4


def test_blocked_cholesky_inversion_on_ill_conditioned_matrices(gpu_lib):
    """hipeng_dense_invert_selftest with OSQP_AMD_DENSE_CHOL=1: eigenvalues from 1e-6 to 1e3; |A^-1 A - I| within a small multiple
    of numpy's own (the block sweeps are off by 2.4 on this matrix)."""
    import osqp_amd
    f = osqp_amd.lib().hipeng_dense_invert_selftest
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(4)
    n = 512
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = Q @ np.diag(10.0 ** rng.uniform(-6, 3, n)) @ Q.T
    A = 0.5 * (A + A.T)
    Ainv = np.zeros((n, n))
    with _env(OSQP_AMD_DENSE_CHOL=1):
        assert f(n, A.ctypes.data_as(C.c_void_p), Ainv.ctypes.data_as(C.c_void_p), None) == 0
    ref = np.abs(np.linalg.inv(A) @ A - np.eye(n)).max()
    assert np.abs(Ainv @ A - np.eye(n)).max() <= 10 * ref + 1e-10


def test_infeasibility_certificates_and_polish_through_the_dense_path(gpu_lib, oracle_mod):
    """The reference's primal / dual infeasibility problems (tests/primal_infeasibility, tests/primal_dual_infeasibility) and a
    polished Lasso with every linear solve on the dense-direct path: statuses, iteration counts and certificates as the oracle's;
    polish (its own plugin instance on the reduced KKT system) ends with the same status_polish and solution."""
    import osqp_amd
    from osqp_amd import abi
    from conftest import load_golden
    from osqp_amd.problems import lasso_qp
    pb, _ = load_golden("primal_infeasibility")
    with _env(OSQP_AMD_DENSE_DIRECT=2, OSQP_AMD_RESIDENT=0):
        sg = osqp_amd.OSQP().setup(**pb, max_iter=10000)
    assert _info(sg)["form"] == 4
    rg, ro = sg.solve(), oracle_mod.OracleOSQP().setup(**pb, max_iter=10000).solve()
    assert rg.info.status_val == ro.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE and rg.info.iter == ro.info.iter
    assert np.abs(rg.prim_inf_cert - ro.prim_inf_cert).max() < 1e-5
    d = load_golden("primal_dual_infeasibility")
    for A, u, want in (("A12", "u2", abi.OSQP_PRIMAL_INFEASIBLE), ("A34", "u3", abi.OSQP_DUAL_INFEASIBLE)):
        kw = dict(max_iter=50, alpha=1.6)
        with _env(OSQP_AMD_DENSE_DIRECT=2, OSQP_AMD_RESIDENT=0):
            sg = osqp_amd.OSQP().setup(d["P"], d["q"], d[A], d["l"], d[u], **kw)
        rg, ro = sg.solve(), oracle_mod.OracleOSQP().setup(d["P"], d["q"], d[A], d["l"], d[u], **kw).solve()
        assert rg.info.status_val == ro.info.status_val == want and rg.info.iter == ro.info.iter, (A, u, _info(sg))
    full = lasso_qp(200, 500, seed=9)
    pb = {k: v for k, v in full.items() if k in "PqAlu"}
    with _env(OSQP_AMD_DENSE_DIRECT=2, OSQP_AMD_RESIDENT=0):
        sg = osqp_amd.OSQP().setup(**pb, polish=1)
    assert _info(sg)["form"] == 4
    rg, ro = sg.solve(), oracle_mod.OracleOSQP().setup(**pb, polish=1).solve()
    assert rg.info.status == ro.info.status == "solved" and rg.info.iter == ro.info.iter
    assert rg.info.status_polish == ro.info.status_polish
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
