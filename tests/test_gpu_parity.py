"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine behind the C ABI
against the CPU oracle and the reference's golden fixtures.

Bars: SpMV kernels bit-exact vs the oracle's reference-order loops; everything
that goes through the indirect (PCG, stop 1e-5 x eps of the request, 1e-8 x eps with equality rows) solve within the stated
fp64 tolerances: iterates/solutions 1e-6 relative (inf norm), objective 1e-8
relative, identical iteration counts and status.  The reference's own tests use
1e-4 absolute (tests/osqp_tester.h:9)."""
import ctypes as C

import numpy as np
import pytest
from scipy import sparse

from conftest import load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_linear_solvers")]
TOL = 1e-4


def _engine_spmv(solver, which, x, outlen):
    from osqp_amd import abi
    L = solver._lib
    L.hipeng_spmv.restype = C.c_int
    L.hipeng_spmv.argtypes = [C.c_void_p, C.c_int, abi.c_float_p, abi.c_float_p]
    x = abi.as_f64(x)
    y = np.zeros(max(outlen, 1))
    assert L.hipeng_spmv(solver.engine(), which, abi.fptr(x), abi.fptr(y)) == 0
    return y[:outlen]


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def test_library_loaded_is_in_tree(gpu_lib):
    import osqp_amd, os
    assert os.path.exists(osqp_amd.LIB_PATH)


@pytest.mark.parametrize("n,m,seed", [(60, 90, 0), (1000, 2000, 1), (3000, 500, 2)])
def test_spmv_bit_exact_vs_oracle(gpu_lib, oracle_mod, n, m, seed):
    """k_spmv (CSR-stream, LDS-staged) == reference-order CPU loops, bit for bit
    (lin_alg.c:241-322), on the unscaled matrices (scaling=0 keeps them as given)."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(n, m, nnz_per_col=min(20, m), seed=seed)
    s = osqp_amd.OSQP().setup(**pb, scaling=0)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n); y = rng.standard_normal(m)
    assert np.array_equal(_engine_spmv(s, 0, x, m), oracle_mod.mat_vec(pb["A"], x))
    assert np.array_equal(_engine_spmv(s, 1, y, n), oracle_mod.mat_tpose_vec(pb["A"], y))
    assert np.array_equal(_engine_spmv(s, 2, x, n), oracle_mod.sym_mat_vec(pb["P"], x))


def test_spmv_golden_lin_alg(gpu_lib):
    """The reference's own SpMV vectors (tests/lin_alg/generate_problem.py)."""
    import osqp_amd
    d = load_golden("lin_alg")
    A, Pu, x, y = d["test_mat_vec_A"], d["test_mat_vec_Pu"], d["test_mat_vec_x"], d["test_mat_vec_y"]
    m, n = A.shape
    # the fixture's P is a random symmetric (indefinite) matrix: a large sigma keeps the
    # reduced matrix positive definite so that setup's convexity probe accepts it
    s = osqp_amd.OSQP().setup(P=Pu, q=np.zeros(n), A=A, l=-np.ones(m), u=np.ones(m), scaling=0, sigma=100.0)
    assert np.abs(_engine_spmv(s, 0, x, m) - d["test_mat_vec_Ax"]).max() < 1e-14
    assert np.abs(_engine_spmv(s, 1, y, n) - d["test_mat_vec_ATy"]).max() < 1e-14
    assert np.abs(_engine_spmv(s, 2, x, n) - d["test_mat_vec_Px"]).max() < 1e-14


def test_long_rows_take_the_workgroup_path(gpu_lib, oracle_mod):
    """A dense row/column longer than one LDS chunk (2048 products)."""
    import osqp_amd
    n, m = 5000, 40
    rng = np.random.default_rng(3)
    A = sparse.vstack([sparse.csc_matrix(rng.standard_normal((1, n))),
                       sparse.random(m - 1, n, density=0.01, format="csc", random_state=rng)], format="csc")
    P = sparse.diags(rng.uniform(1, 2, n), format="csc")
    s = osqp_amd.OSQP().setup(P=P, q=np.zeros(n), A=A, l=-np.ones(m), u=np.ones(m), scaling=0)
    x = rng.standard_normal(n); y = rng.standard_normal(m)
    assert _rel(_engine_spmv(s, 0, x, m), oracle_mod.mat_vec(A, x)) < 1e-13
    assert _rel(_engine_spmv(s, 1, y, n), oracle_mod.mat_tpose_vec(A, y)) < 1e-13


def test_plugin_boundary_known_answer(gpu_lib):
    """The reference's plugin KAT (tests/solve_linsys/test_solve_linsys.h:12-46) through
    init_linsys_solver_hip_pcg + vtable solve on a host vector."""
    import osqp_amd
    from osqp_amd import abi
    d = load_golden("solve_linsys")
    L = gpu_lib
    S = C.POINTER(abi.LinSysSolver)
    L.init_linsys_solver_hip_pcg.restype = abi.c_int
    L.init_linsys_solver_hip_pcg.argtypes = [C.POINTER(S), C.POINTER(abi.csc), C.POINTER(abi.csc),
                                             abi.c_float, abi.c_float_p, abi.c_int]
    hp, ha = abi.CscHolder(d["test_solve_KKT_Pu"]), abi.CscHolder(d["test_solve_KKT_A"])
    rho = abi.as_f64(d["test_solve_KKT_rho"] * np.ones(d["test_solve_KKT_m"]))
    s = S()
    assert L.init_linsys_solver_hip_pcg(C.byref(s), C.byref(hp.struct), C.byref(ha.struct),
                                        d["test_solve_KKT_sigma"], abi.fptr(rho), 0) == 0
    assert s.contents.type == abi.HIP_PCG_SOLVER
    b = abi.as_f64(d["test_solve_KKT_rhs"]).copy()
    assert s.contents.solve(s, abi.fptr(b)) == 0
    assert np.abs(b - d["test_solve_KKT_x"]).max() < 1e-9     # reference test: 1e-4
    # update_rho_vec then solve again still solves the (new) system
    rho2 = abi.as_f64(2.0 * rho)
    assert s.contents.update_rho_vec(s, abi.fptr(rho2)) == 0
    b2 = abi.as_f64(d["test_solve_KKT_rhs"]).copy()
    assert s.contents.solve(s, abi.fptr(b2)) == 0
    P = sparse.csc_matrix(d["test_solve_KKT_Pu"]); P = P + sparse.triu(P, 1).T
    A = d["test_solve_KKT_A"]; n = 3
    K = P + d["test_solve_KKT_sigma"] * sparse.eye(n) + A.T @ sparse.diags(rho2) @ A
    rhs = d["test_solve_KKT_rhs"]
    xt = np.linalg.solve(K.toarray(), rhs[:n] + A.T @ (rho2 * rhs[n:]))
    assert np.abs(b2[:n] - xt).max() < 1e-9 and np.abs(b2[n:] - A @ xt).max() < 1e-9
    s.contents.free(s)


def test_demo_matches_oracle(gpu_lib, oracle_mod):
    import osqp_amd
    from osqp_amd.problems import demo_qp
    ro = oracle_mod.OracleOSQP().setup(**demo_qp()).solve()
    rg = osqp_amd.OSQP().setup(**demo_qp()).solve()
    assert (rg.info.iter, rg.info.status) == (ro.info.iter, ro.info.status) == (25, "solved")
    assert abs(rg.info.obj_val - ro.info.obj_val) < 1e-8 * max(1, abs(ro.info.obj_val))
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert abs(rg.info.pri_res - ro.info.pri_res) < 1e-8 and abs(rg.info.dua_res - ro.info.dua_res) < 1e-8
    assert abs(rg.info.rho_estimate - ro.info.rho_estimate) < 1e-6


def test_basic_qp_golden(gpu_lib):
    """tests/basic_qp/test_basic_qp.h:10-90 (solve; polish through the plugin)."""
    import osqp_amd
    pb, sol = load_golden("basic_qp")
    r = osqp_amd.OSQP().setup(**pb, max_iter=2000, alpha=1.6, polish=1, scaling=0, warm_start=1).solve()
    assert r.info.status == "solved" and r.info.iter == 75
    assert np.abs(r.x - sol["x_test"]).max() < TOL and np.abs(r.y - sol["y_test"]).max() < TOL
    assert abs(r.info.obj_val - sol["obj_value_test"]) < TOL


def test_basic_qp_updates_warm_start_rho(gpu_lib):
    """tests/basic_qp/test_basic_qp.h:461-904."""
    import osqp_amd
    pb, sol = load_golden("basic_qp")
    s = osqp_amd.OSQP().setup(**pb, max_iter=200, alpha=1.6, scaling=0)
    assert s.update(q=sol["q_new"]) == 0
    q = np.ctypeslib.as_array(s.work.data.contents.q, shape=(2,))
    assert np.abs(q - sol["q_new"]).max() < TOL
    assert s.update(l=sol["l_new"], u=sol["u_new"]) == 0
    assert s.update(l=sol["u_new"] + 1, u=sol["u_new"]) == 1
    kw = dict(max_iter=200, alpha=1.6, scaling=0, eps_abs=1e-4, eps_rel=1e-4, check_termination=1,
              adaptive_rho=0)
    s2 = osqp_amd.OSQP().setup(**pb, **kw)
    r0 = s2.solve(); it = r0.info.iter
    s2.warm_start(x=np.zeros(2), y=np.zeros(4))
    assert s2.solve().info.iter == it
    s2.warm_start(x=r0.x, y=r0.y)
    assert s2.solve().info.iter == 1
    # update_rho == fresh setup with that rho (:643-769)
    kw = dict(max_iter=2000, alpha=1.6, scaling=0, adaptive_rho=0, eps_abs=5e-5, eps_rel=5e-5, check_termination=1)
    a = osqp_amd.OSQP().setup(**pb, rho=0.7, **kw).solve()
    s3 = osqp_amd.OSQP().setup(**pb, rho=0.1, **kw); s3.solve(); s3.update_rho(0.7)
    s3.update_settings(warm_start=0)
    assert s3.solve().info.iter == a.info.iter
    # check_termination = 0 => iter == max_iter (:570-641)
    r = osqp_amd.OSQP().setup(**pb, max_iter=200, alpha=1.6, scaling=0, check_termination=0).solve()
    assert r.info.iter == 200 and r.info.status == "solved"


def test_time_limit(gpu_lib):
    """tests/basic_qp/test_basic_qp.h:772-841: tiny time limit => OSQP_TIME_LIMIT_REACHED."""
    import osqp_amd
    from osqp_amd import abi
    pb, _ = load_golden("basic_qp")
    s = osqp_amd.OSQP().setup(**pb, rho=20.0, adaptive_rho=0, max_iter=2000000000, check_termination=0,
                              time_limit=1e-5, scaling=0)
    r = s.solve()
    assert r.info.status_val == abi.OSQP_TIME_LIMIT_REACHED


def test_basic_qp2_unconstrained_golden(gpu_lib, oracle_mod):
    """tests/basic_qp2 (solve, update q/u) against the oracle with identical settings,
    and tests/unconstrained (m = 0) against the generator's solution."""
    import osqp_amd
    pb, sol = load_golden("basic_qp2")
    kw = dict(alpha=1.6, rho=0.1, scaling=0)
    s = osqp_amd.OSQP().setup(**pb, **kw); so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    r, ro = s.solve(), so.solve()
    assert r.info.status == ro.info.status == "solved" and r.info.iter == ro.info.iter
    assert _rel(r.x, ro.x) < 1e-6 and _rel(r.y, ro.y) < 1e-6
    for h in (s, so):
        h.update(q=sol["q_new"]); h.update(u=sol["u_new"])
    r, ro = s.solve(), so.solve()
    assert r.info.status == ro.info.status and r.info.iter == ro.info.iter
    assert _rel(r.x, ro.x) < 1e-6 and _rel(r.y, ro.y) < 1e-6
    assert np.abs(r.x - sol["x_test_new"]).max() < 0.5
    pb, sol = load_golden("unconstrained")
    r = osqp_amd.OSQP().setup(**pb).solve()
    assert r.info.status == "solved" and r.info.iter == 25
    assert np.abs(r.x - sol["x_test"]).max() < TOL and abs(r.info.obj_val - sol["obj_value_test"]) < TOL


def test_polish_through_plugin(gpu_lib):
    """polish=1 (src/polish.c) runs through a second plugin instance with polish=1;
    on the reference's small problems it must reach the generator's exact solutions."""
    import osqp_amd
    pb, sol = load_golden("basic_qp2")
    r = osqp_amd.OSQP().setup(**pb, alpha=1.6, rho=0.1, polish=1, scaling=0).solve()
    assert r.info.status == "solved" and r.info.status_polish == 1
    assert np.abs(r.x - sol["x_test"]).max() < TOL and np.abs(r.y - sol["y_test"]).max() < TOL
    assert abs(r.info.obj_val - sol["obj_value_test"]) < TOL
    pb, sol = load_golden("basic_qp")
    r = osqp_amd.OSQP().setup(**pb, max_iter=2000, alpha=1.6, polish=1, scaling=0).solve()
    assert r.info.status_polish == 1
    assert np.abs(r.x - sol["x_test"]).max() < 1e-6 and np.abs(r.y - sol["y_test"]).max() < 1e-5


@pytest.mark.parametrize("n,m,seed,kw", [
    (300, 600, 5, {}),
    (800, 1600, 6, dict(eps_abs=1e-5, eps_rel=1e-5)),
    (500, 200, 7, dict(scaling=0)),
    (600, 1200, 9, dict(polish_refine_iter=1)),
])
def test_polish_midsize_matches_oracle(gpu_lib, oracle_mod, n, m, seed, kw):
    """polish=1 on random QPs with hundreds of active constraints (src/polish.c:212-350): same
    status_polish as the oracle's direct-factorisation polish, and the polished x, y, objective and
    residuals agree within the reference's test tolerance (1e-4 absolute, tests/osqp_tester.h:9) --
    in fact to 1e-6 relative, both being the solution of the same unregularised KKT system."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(n, m, nnz_per_col=min(20, m), seed=seed)
    ro = oracle_mod.OracleOSQP().setup(**pb, polish=1, **kw).solve()
    rg = osqp_amd.OSQP().setup(**pb, polish=1, **kw).solve()
    assert rg.info.status == ro.info.status == "solved" and rg.info.iter == ro.info.iter
    assert rg.info.status_polish == ro.info.status_polish == 1
    assert np.abs(rg.x - ro.x).max() < TOL and np.abs(rg.y - ro.y).max() < TOL and abs(rg.info.obj_val - ro.info.obj_val) < TOL
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
    assert rg.info.pri_res < 1e-8 and rg.info.dua_res < 1e-8


def test_infeasibility_statuses(gpu_lib):
    """tests/primal_dual_infeasibility + tests/primal_infeasibility statuses and iteration counts."""
    import osqp_amd
    from osqp_amd import abi
    d = load_golden("primal_dual_infeasibility")
    kw = dict(max_iter=2000, alpha=1.6, scaling=0, polish=1)   # the reference test polishes
    r = osqp_amd.OSQP().setup(d["P"], d["q"], d["A12"], d["l"], d["u1"], **kw).solve()
    assert r.info.status_val == abi.OSQP_SOLVED and r.info.iter == 50
    assert np.abs(r.x - d["x1"]).max() < TOL and np.abs(r.y - d["y1"]).max() < TOL
    r = osqp_amd.OSQP().setup(d["P"], d["q"], d["A12"], d["l"], d["u2"], **kw).solve()
    assert r.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE and r.info.iter == 50
    assert r.info.obj_val == abi.OSQP_INFTY and np.all(r.x == abi.OSQP_NAN)
    assert abs(np.abs(r.prim_inf_cert).max() - 1.0) < 1e-12
    r = osqp_amd.OSQP().setup(d["P"], d["q"], d["A34"], d["l"], d["u3"], **kw).solve()
    assert r.info.status_val == abi.OSQP_DUAL_INFEASIBLE and r.info.iter == 50
    assert abs(np.abs(r.dual_inf_cert).max() - 1.0) < 1e-12
    r = osqp_amd.OSQP().setup(d["P"], d["q"], d["A34"], d["l"], d["u4"], **kw).solve()
    assert r.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE and r.info.iter == 25
    pb, _ = load_golden("primal_infeasibility")
    r = osqp_amd.OSQP().setup(**pb, max_iter=10000, alpha=1.6, scaling=0).solve()
    assert r.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE


def test_update_matrices_golden(gpu_lib):
    """tests/update_matrices/test_update_matrices.h:73-311."""
    import osqp_amd
    d = load_golden("update_matrices")
    pb = dict(P=d["test_solve_Pu"], q=d["test_solve_q"], A=d["test_solve_A"], l=d["test_solve_l"],
              u=d["test_solve_u"])
    fresh = lambda: osqp_amd.OSQP().setup(**pb, eps_abs=1e-5, eps_rel=1e-5)
    r = fresh().solve()
    assert r.info.status == "solved" and r.info.iter == 25
    assert np.abs(r.x - d["test_solve_x"]).max() < TOL and abs(r.info.obj_val - d["test_solve_obj_value"]) < TOL
    Pn = sparse.csc_matrix(d["test_solve_Pu_new"]); An = sparse.csc_matrix(d["test_solve_A_new"])
    Pn.sort_indices(); An.sort_indices()
    s = fresh(); s.solve(); assert s.update(Px=Pn.data) == 0
    r = s.solve()
    assert np.abs(r.x - d["test_solve_P_new_x"]).max() < TOL and abs(r.info.obj_val - d["test_solve_P_new_obj_value"]) < TOL
    s = fresh(); s.solve(); assert s.update(Ax=An.data, Ax_idx=np.arange(An.nnz)) == 0
    assert np.abs(s.solve().x - d["test_solve_A_new_x"]).max() < TOL
    s = fresh(); s.solve(); assert s.update(Px=Pn.data, Ax=An.data) == 0
    r = s.solve()
    assert np.abs(r.x - d["test_solve_P_A_new_x"]).max() < TOL
    assert abs(r.info.obj_val - d["test_solve_P_A_new_obj_value"]) < TOL


def test_invalid_inputs_rejected(gpu_lib):
    import osqp_amd
    pb, _ = load_golden("basic_qp")
    for bad in (dict(rho=-1.0), dict(alpha=2.5), dict(max_iter=0), dict(linsys_solver=7)):
        with pytest.raises(ValueError, match="error 2"):
            osqp_amd.OSQP().setup(**pb, **bad)
    b = dict(pb); b["l"] = pb["u"] + 1.0; b["l"][3] = 0.0
    with pytest.raises(ValueError, match="error 1"):
        osqp_amd.OSQP().setup(**b)


@pytest.mark.parametrize("n,m,seed,kw", [
    (300, 600, 5, {}),
    (800, 1600, 6, dict(eps_abs=1e-5, eps_rel=1e-5)),
    (500, 200, 7, dict(scaling=0, adaptive_rho_interval=50)),
    (400, 800, 8, dict(scaled_termination=1, alpha=1.0)),
])
def test_random_qp_matches_oracle(gpu_lib, oracle_mod, n, m, seed, kw):
    """Same seeded problem through the oracle (direct LDL^T) and the HIP engine
    (PCG at its default stop): identical iteration count, status and rho updates;
    x, y within 1e-6 relative; objective within 1e-8 relative; residuals 1e-4 rel + 1e-9."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(n, m, nnz_per_col=min(20, m), seed=seed)
    ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    sg = osqp_amd.OSQP().setup(**pb, **kw)
    rg = sg.solve()
    assert rg.info.status == ro.info.status == "solved"
    assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    # residuals are differences of nearly equal vectors: 1e-9 absolute + 1e-4 relative
    assert abs(rg.info.pri_res - ro.info.pri_res) <= 1e-4 * ro.info.pri_res + 1e-9
    assert abs(rg.info.dua_res - ro.info.dua_res) <= 1e-4 * ro.info.dua_res + 1e-9
    assert sg.stats()["pcg_forced"] == 0


def test_trajectory_matches_oracle_stepwise(gpu_lib, oracle_mod):
    """Iterates after k = 1, 2, 5, 30 ADMM iterations agree with the oracle's
    (max_iter = k, no termination checks)."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(200, 400, seed=11)
    for k in (1, 2, 5, 30):
        kw = dict(max_iter=k, check_termination=0, adaptive_rho=0)
        so = oracle_mod.OracleOSQP().setup(**pb, **kw); so.solve()
        sg = osqp_amd.OSQP().setup(**pb, **kw); sg.solve()
        xo, zo, yo = so.iterates()
        w = sg.work
        xg, zg, yg = sg._vec(w.x, 200), sg._vec(w.z, 400), sg._vec(w.y, 400)
        assert _rel(xg, xo) < 1e-8 and _rel(zg, zo) < 1e-8 and _rel(yg, yo) < 1e-8


def test_full_size_config2_properties(gpu_lib):
    """BASELINE config 2 at full size (n=10000, m=20000): the oracle's direct
    factorisation takes minutes here, so parity is checked through
    size-independent properties evaluated in numpy/scipy on the returned point:
    KKT residuals below the requested tolerances, complementary slackness sign
    pattern, and objective == 1/2 x'Px + q'x."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp()
    eps = 1e-4
    s = osqp_amd.OSQP().setup(**pb, eps_abs=eps, eps_rel=eps, adaptive_rho_interval=100)
    r = s.solve()
    assert r.info.status == "solved" and r.info.iter <= 500
    P = pb["P"] + sparse.triu(pb["P"], 1).T
    A = pb["A"]
    x, y = r.x, r.y
    Ax = A @ x
    z = np.clip(Ax, pb["l"], pb["u"])
    pri = np.abs(Ax - z).max()
    dua = np.abs(P @ x + pb["q"] + A.T @ y).max()
    assert pri <= eps + eps * max(np.abs(Ax).max(), np.abs(z).max()) + 1e-9
    assert dua <= eps + eps * max(np.abs(P @ x).max(), np.abs(A.T @ y).max(), np.abs(pb["q"]).max()) + 1e-9
    assert abs(r.info.obj_val - (0.5 * x @ (P @ x) + pb["q"] @ x)) < 1e-8 * abs(r.info.obj_val)
    # multipliers push outwards only at (nearly) active bounds
    slack_lo, slack_hi = Ax - pb["l"], pb["u"] - Ax
    assert np.all(y[slack_hi > 1e-2] <= 1e-6) and np.all(y[slack_lo > 1e-2] >= -1e-6)
    assert s.stats()["pcg_forced"] == 0


def test_full_size_config2_matches_oracle_golden(gpu_lib):
    """BASELINE config 2 at full size against the CPU oracle's result, captured once in the
    build container (tests/golden/config2_oracle.json, 10 minutes of direct LDL^T; generator
    tools/make_config2_golden.py): same 125 iterations and rho update, objective 1e-8
    relative, x and y 1e-6 relative on the stored subsample."""
    import json, os
    import osqp_amd
    from conftest import GOLDEN
    from osqp_amd.problems import random_sparse_qp
    g = json.load(open(os.path.join(GOLDEN, "config2_oracle.json")))
    pb = random_sparse_qp()
    r = osqp_amd.OSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100).solve()
    gi = g["info"]
    assert r.info.status == gi["status"] == "solved"
    assert r.info.iter == gi["iters"] == 125 and r.info.rho_updates == gi["rho_updates"]
    assert abs(r.info.obj_val - gi["obj"]) <= 1e-8 * abs(gi["obj"])
    xs, ys = np.array(g["x_sub"]), np.array(g["y_sub"])
    assert np.abs(r.x[::10] - xs).max() <= 1e-6 * g["x_inf"]
    assert np.abs(r.y[::20] - ys).max() <= 1e-6 * g["y_inf"]
    assert abs(r.info.pri_res - gi["pri"]) <= 1e-4 * gi["pri"] + 1e-9
    assert abs(r.info.dua_res - gi["dua"]) <= 1e-4 * gi["dua"] + 1e-9


@pytest.mark.parametrize("cfg", ["config5", "config5s", "config3"])
def test_full_size_config3_config5_match_oracle_golden(gpu_lib, cfg):
    """BASELINE configs 5 (portfolio, n = 50000: 400 dense 125x125 blocks of P through the dense
    block kernels, a 50 000-entry budget row through the sliced huge-row path) and 3 (Lasso,
    7.5 M non-zeros, rows of 751 entries) at full size against the CPU oracle's results
    (tests/golden/config{5,3}_oracle.json; generators tools/make_config{5,3}_golden.py; the
    Lasso run took the oracle 77 s to factorise and 282 s to solve): same iteration count
    (325 / 1875) and rho updates, x, y and the objective within 1e-6 relative -- at the default
    engine options (both families carry equality rows, for which osqp_solve tightens the PCG stop
    to 1e-12 by itself; at a plain 1e-10 they agreed to 1e-4 only).  config5s: config 5 + 500 sparse sector rows
    (`tools/make_config5_golden.py 500`: 2525 iterations, 103 s on the CPU), solved by the coupled
    block-direct form."""
    import json, os
    import osqp_amd
    from conftest import GOLDEN
    from osqp_amd.problems import portfolio_qp, lasso_qp
    g = json.load(open(os.path.join(GOLDEN, cfg + "_oracle.json")))
    if cfg in ("config5", "config5s"):
        pb, kw, step = portfolio_qp(sector_rows=500 if cfg == "config5s" else 0), dict(adaptive_rho_interval=100), 50
    else:
        pb, kw, step = {k: v for k, v in lasso_qp().items() if k in "PqAlu"}, {}, 20
    assert osqp_amd.engine_options()["pcg_eps_rel"] == 1e-9
    s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, **kw)
    r = s.solve()
    gi = g["info"]
    assert r.info.status == gi["status"] == "solved"
    assert r.info.iter == gi["iters"] and r.info.rho_updates == gi["rho_updates"]
    assert abs(r.info.obj_val - gi["obj"]) <= 1e-6 * abs(gi["obj"])
    xs, ys = np.array(g["x_sub"]), np.array(g["y_sub"])
    assert np.abs(r.x[::step] - xs).max() <= 1e-6 * max(1.0, g["x_inf"])
    assert np.abs(r.y[::step] - ys).max() <= 1e-6 * max(1.0, g["y_inf"])
    assert s.stats()["pcg_forced"] == 0
    # config 5 runs its linear solves as block-direct solves (k_blk_apply / k_blk_finish), config 3 as dense-direct solves
    # (csrc/dense_direct.h: one application of an explicit 5 120 x 5 120 inverse per ADMM iteration)
    assert s.stats()["resident"] == 1
    assert s.stats()["pcg_iters_total"] == r.info.iter


def test_non_cvx_golden(gpu_lib):
    """tests/non_cvx/test_non_cvx.h:26-58: indefinite P.  With sigma = 1e-6 the reduced
    matrix is not positive definite and setup must fail with OSQP_NONCVX_ERROR (5) -- the
    reference gets it from the LDL^T inertia, this engine from negative curvature met by a
    short CG probe; with sigma = 5 setup succeeds and the solve must end OSQP_NON_CVX with
    obj_val == OSQP_NAN."""
    import osqp_amd
    from osqp_amd import abi
    pb, sol = load_golden("non_cvx")
    with pytest.raises(ValueError, match="error 5"):
        osqp_amd.OSQP().setup(**pb, adaptive_rho=0, sigma=1e-6)
    r = osqp_amd.OSQP().setup(**pb, adaptive_rho=0, sigma=float(sol["sigma_new"])).solve()
    assert r.info.status_val == abi.OSQP_NON_CVX and r.info.obj_val == abi.OSQP_NAN


def test_plain_c_caller_links_and_matches_oracle(gpu_lib, oracle_mod, tmp_path):
    """examples/c_caller.c is compiled with gcc against include/osqp_amd.h, linked with
    libosqp_amd.so and run as its own process: the drop-in boundary exercised from C, the
    way a program written for the reference would use it (setup, solve, update q/l/u, solve)."""
    import os, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_caller"
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_caller.c"),
                           "-L", os.path.join(root, "osqp_amd"), "-losqp_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(root, "osqp_amd"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("solve")]
    assert len(lines) == 2
    from osqp_amd.problems import demo_qp
    so = oracle_mod.OracleOSQP().setup(**demo_qp(), alpha=1.0)
    refs = [so.solve()]
    so.update(q=np.array([2.0, 3.0]), l=np.array([2.0, -1.0, -1.0]), u=np.array([2.0, 2.5, 2.5]))
    refs.append(so.solve())
    for line, ro in zip(lines, refs):
        f = dict(re.findall(r"(\w+)=([^ ]+)", line))
        assert int(f["rc"]) == 0 and int(f["status"]) == ro.info.status_val and int(f["iter"]) == ro.info.iter
        assert abs(float(f["obj"]) - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
        x = np.array([float(v) for v in f["x"].split(",")]); y = np.array([float(v) for v in f["y"].split(",")])
        assert _rel(x, ro.x) < 1e-6 and _rel(y, ro.y) < 1e-6
