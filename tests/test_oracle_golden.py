"""CPU tests: the oracle (plain-C restatement of the reference CPU path) against
the reference's own fixtures (tests/golden, captured from the reference's
generators) and the known answers recorded in SURVEY.md App. B-D.  Tolerance is
the reference's TESTS_TOL = 1e-4 absolute (tests/osqp_tester.h:9) unless a
tighter one is stated."""
import numpy as np
import pytest
from scipy import sparse

from conftest import load_golden

TOL = 1e-4


def test_lin_alg_spmv_matches_reference_vectors(oracle_mod):
    """tests/lin_alg/test_lin_alg.h:168-249 (mat_vec / mat_tpose_vec incl. += forms)."""
    d = load_golden("lin_alg")
    A, Pu, x, y = d["test_mat_vec_A"], d["test_mat_vec_Pu"], d["test_mat_vec_x"], d["test_mat_vec_y"]
    assert np.abs(oracle_mod.mat_vec(A, x) - d["test_mat_vec_Ax"]).max() < TOL
    assert np.abs(oracle_mod.mat_vec(A, x, y, plus_eq=1) - d["test_mat_vec_Ax_cum"]).max() < TOL
    assert np.abs(oracle_mod.mat_tpose_vec(A, y) - d["test_mat_vec_ATy"]).max() < TOL
    assert np.abs(oracle_mod.mat_tpose_vec(A, y, x, plus_eq=1) - d["test_mat_vec_ATy_cum"]).max() < TOL
    assert np.abs(oracle_mod.sym_mat_vec(Pu, x) - d["test_mat_vec_Px"]).max() < TOL
    # tighter: these are plain double sums
    assert np.abs(oracle_mod.sym_mat_vec(Pu, x) - d["test_mat_vec_Px"]).max() < 1e-14


def test_lin_alg_norms_and_quadform(oracle_mod):
    """tests/lin_alg/test_lin_alg.h:112-166, 251-300."""
    d = load_golden("lin_alg")
    A = d["test_mat_ops_A"]
    assert np.abs(oracle_mod.col_norms(A, "cols") - d["test_mat_ops_inf_norm_cols"]).max() < TOL
    assert np.abs(oracle_mod.col_norms(A, "rows") - d["test_mat_ops_inf_norm_rows"]).max() < TOL
    assert np.abs(oracle_mod.col_norms(d["test_mat_extr_triu_Pu"], "sym")
                  - d["test_mat_extr_triu_P_inf_norm_cols"]).max() < TOL
    assert abs(oracle_mod.quad_form(d["test_qpform_Pu"], d["test_qpform_x"]) - d["test_qpform_value"]) < TOL


def test_form_KKT_entrywise(oracle_mod):
    """tests/update_matrices/test_update_matrices.h:13-71 (entry-by-entry vs scipy bmat)."""
    d = load_golden("update_matrices")
    m = d["test_form_KKT_m"]
    for Pu, A, ref in ((d["test_form_KKT_Pu"], d["test_form_KKT_A"], d["test_form_KKT_KKTu"]),
                       (d["test_form_KKT_Pu_new"], d["test_form_KKT_A_new"], d["test_form_KKT_KKTu_new"])):
        K = oracle_mod.form_KKT(Pu, A, d["test_form_KKT_sigma"], np.ones(m) / d["test_form_KKT_rho"])
        ref = sparse.csc_matrix(ref)
        ref.sort_indices()
        assert np.array_equal(K.indptr, ref.indptr)
        assert np.array_equal(K.indices, ref.indices)
        assert np.abs(K.data - ref.data).max() < 1e-14


def test_solve_KKT_known_answer(oracle_mod):
    """tests/solve_linsys/test_solve_linsys.h:12-46: init + solve vs scipy splu."""
    d = load_golden("solve_linsys")
    m = d["test_solve_KKT_m"]
    rc, b = oracle_mod.kkt_solve(d["test_solve_KKT_Pu"], d["test_solve_KKT_A"], d["test_solve_KKT_sigma"],
                                 d["test_solve_KKT_rho"] * np.ones(m), d["test_solve_KKT_rhs"])
    assert rc == 0
    assert np.abs(b - d["test_solve_KKT_x"]).max() < 1e-12


def test_demo_known_answer(oracle_mod):
    """25 iterations, obj 1.8797, pri 1.60e-3, dua 9.48e-4, rho est 0.214."""
    from osqp_amd.problems import demo_qp
    r = oracle_mod.OracleOSQP().setup(**demo_qp()).solve()
    assert (r.info.iter, r.info.status) == (25, "solved")
    assert abs(r.info.obj_val - 1.8797) < 1e-4
    assert abs(r.info.pri_res - 1.60e-3) < 1e-5 and abs(r.info.dua_res - 9.48e-4) < 1e-6
    assert abs(r.info.rho_estimate - 0.214) < 1e-3


def test_basic_qp_solve_with_polish(oracle_mod):
    """tests/basic_qp/test_basic_qp.h:10-90 + App. B (75 iterations, polished obj 1.8800)."""
    pb, sol = load_golden("basic_qp")
    s = oracle_mod.OracleOSQP().setup(**pb, max_iter=2000, alpha=1.6, polish=1, scaling=0, warm_start=1)
    r = s.solve()
    assert r.info.status == "solved" and r.info.iter == 75 and r.info.status_polish == 1
    assert np.abs(r.x - sol["x_test"]).max() < TOL and np.abs(r.y - sol["y_test"]).max() < TOL
    assert abs(r.info.obj_val - sol["obj_value_test"]) < TOL
    assert abs(r.info.rho_estimate - 3.40) < 1e-2


def test_basic_qp_update_and_warm_start(oracle_mod):
    """tests/basic_qp/test_basic_qp.h:461-568 (updates), :845-904 (warm start: iter==1 from optimum)."""
    pb, sol = load_golden("basic_qp")
    s = oracle_mod.OracleOSQP().setup(**pb, max_iter=200, alpha=1.6, polish=1, scaling=0, warm_start=1)
    assert s.update(q=sol["q_new"]) == 0
    q = np.ctypeslib.as_array(s.work.data.contents.q, shape=(2,))
    assert np.abs(q - sol["q_new"]).max() < TOL
    assert s.update(l=sol["l_new"], u=sol["u_new"]) == 0
    assert s.update(l=sol["u_new"] + 1, u=sol["u_new"]) == 1        # l > u rejected
    # warm start
    s2 = oracle_mod.OracleOSQP().setup(**pb, max_iter=200, alpha=1.6, polish=0, scaling=0, warm_start=1,
                                       eps_abs=1e-4, eps_rel=1e-4, check_termination=1, adaptive_rho=0)
    r0 = s2.solve()
    it = r0.info.iter
    s2.warm_start(x=np.zeros(2), y=np.zeros(4))
    assert s2.solve().info.iter == it
    s2.warm_start(x=r0.x, y=r0.y)
    assert s2.solve().info.iter == 1


def test_basic_qp_update_rho_same_iterations(oracle_mod):
    """tests/basic_qp/test_basic_qp.h:643-769: osqp_update_rho == fresh setup with that rho."""
    pb, _ = load_golden("basic_qp")
    kw = dict(max_iter=2000, alpha=1.6, polish=0, scaling=0, adaptive_rho=0, eps_abs=5e-5, eps_rel=5e-5,
              check_termination=1)
    a = oracle_mod.OracleOSQP().setup(**pb, rho=0.7, **kw).solve()
    s = oracle_mod.OracleOSQP().setup(**pb, rho=0.1, **kw)
    s.solve()
    s.update_rho(0.7)
    s.work.settings.contents.warm_start = 0
    b = s.solve()
    assert a.info.iter == b.info.iter


def test_basic_qp_check_termination_off(oracle_mod):
    """tests/basic_qp/test_basic_qp.h:570-641: check_termination=0 => iter == max_iter."""
    pb, _ = load_golden("basic_qp")
    r = oracle_mod.OracleOSQP().setup(**pb, max_iter=200, alpha=1.6, scaling=0, check_termination=0).solve()
    assert r.info.iter == 200 and r.info.status == "solved"


def test_basic_qp2(oracle_mod):
    """tests/basic_qp2/test_basic_qp2.h: solve, then update q/u."""
    pb, sol = load_golden("basic_qp2")
    s = oracle_mod.OracleOSQP().setup(**pb, alpha=1.6, rho=0.1, polish=1, scaling=0)
    r = s.solve()
    assert r.info.status == "solved"
    assert np.abs(r.x - sol["x_test"]).max() < TOL * 10 and np.abs(r.y - sol["y_test"]).max() < TOL * 100
    assert abs(r.info.obj_val - sol["obj_value_test"]) < TOL * 100
    s.update(q=sol["q_new"]); s.update(u=sol["u_new"])
    r = s.solve()
    assert r.info.status == "solved"
    assert np.abs(r.x - sol["x_test_new"]).max() < TOL * 10
    assert abs(r.info.obj_val - sol["obj_value_test_new"]) < TOL * 100


def test_unconstrained(oracle_mod):
    """tests/unconstrained/test_unconstrained.h (m = 0)."""
    pb, sol = load_golden("unconstrained")
    r = oracle_mod.OracleOSQP().setup(**pb).solve()
    assert r.info.status == "solved" and r.info.iter == 25
    assert np.abs(r.x - sol["x_test"]).max() < TOL
    assert abs(r.info.obj_val - sol["obj_value_test"]) < TOL


def test_non_cvx(oracle_mod):
    """tests/non_cvx/test_non_cvx.h:26-58: setup fails with OSQP_NONCVX_ERROR (5) at
    sigma=1e-6; with sigma=5 the solve ends OSQP_NON_CVX with obj == OSQP_NAN."""
    from osqp_amd import abi
    pb, sol = load_golden("non_cvx")
    with pytest.raises(ValueError, match="error 5"):
        oracle_mod.OracleOSQP().setup(**pb, adaptive_rho=0, sigma=1e-6)
    r = oracle_mod.OracleOSQP().setup(**pb, adaptive_rho=0, sigma=float(sol["sigma_new"])).solve()
    assert r.info.status_val == abi.OSQP_NON_CVX and r.info.obj_val == abi.OSQP_NAN


def test_primal_dual_infeasibility(oracle_mod):
    """tests/primal_dual_infeasibility/*: optimal / primal inf / dual inf / both (App. D:
    50 / 50 / 50 / 25 iterations)."""
    from osqp_amd import abi
    d = load_golden("primal_dual_infeasibility")
    kw = dict(max_iter=2000, alpha=1.6, polish=1, scaling=0)
    r = oracle_mod.OracleOSQP().setup(d["P"], d["q"], d["A12"], d["l"], d["u1"], **kw).solve()
    assert r.info.status_val == abi.OSQP_SOLVED and r.info.iter == 50
    assert np.abs(r.x - d["x1"]).max() < TOL and np.abs(r.y - d["y1"]).max() < TOL
    assert abs(r.info.obj_val - d["obj_value1"]) < TOL
    r = oracle_mod.OracleOSQP().setup(d["P"], d["q"], d["A12"], d["l"], d["u2"], **kw).solve()
    assert r.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE and r.info.iter == 50
    r = oracle_mod.OracleOSQP().setup(d["P"], d["q"], d["A34"], d["l"], d["u3"], **kw).solve()
    assert r.info.status_val == abi.OSQP_DUAL_INFEASIBLE and r.info.iter == 50
    r = oracle_mod.OracleOSQP().setup(d["P"], d["q"], d["A34"], d["l"], d["u4"], **kw).solve()
    assert r.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE and r.info.iter == 25


def test_primal_infeasible_random(oracle_mod):
    """tests/primal_infeasibility/test_primal_infeasibility.h (n=50, m=150)."""
    from osqp_amd import abi
    pb, _ = load_golden("primal_infeasibility")
    r = oracle_mod.OracleOSQP().setup(**pb, max_iter=10000, alpha=1.6, polish=1, scaling=0).solve()
    assert r.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE


def test_update_matrices(oracle_mod):
    """tests/update_matrices/test_update_matrices.h:73-311: osqp_update_P / _A / _P_A,
    whole and indexed, against the generator's hard-coded solutions."""
    d = load_golden("update_matrices")
    pb = dict(P=d["test_solve_Pu"], q=d["test_solve_q"], A=d["test_solve_A"], l=d["test_solve_l"],
              u=d["test_solve_u"])

    def fresh():
        return oracle_mod.OracleOSQP().setup(**pb, eps_abs=1e-5, eps_rel=1e-5)

    r = fresh().solve()
    assert r.info.status == "solved" and r.info.iter == 25
    assert np.abs(r.x - d["test_solve_x"]).max() < TOL and np.abs(r.y - d["test_solve_y"]).max() < TOL
    assert abs(r.info.obj_val - d["test_solve_obj_value"]) < TOL
    Pn = sparse.csc_matrix(d["test_solve_Pu_new"]); An = sparse.csc_matrix(d["test_solve_A_new"])
    Pn.sort_indices(); An.sort_indices()
    for idx in (False, True):
        s = fresh(); s.solve()
        kw = dict(Px=Pn.data)
        if idx:
            kw["Px_idx"] = np.arange(Pn.nnz)
        assert s.update(**kw) == 0
        r = s.solve()
        assert np.abs(r.x - d["test_solve_P_new_x"]).max() < TOL
        assert abs(r.info.obj_val - d["test_solve_P_new_obj_value"]) < TOL
        s = fresh(); s.solve()
        kw = dict(Ax=An.data)
        if idx:
            kw["Ax_idx"] = np.arange(An.nnz)
        assert s.update(**kw) == 0
        r = s.solve()
        assert np.abs(r.x - d["test_solve_A_new_x"]).max() < TOL
        s = fresh(); s.solve()
        assert s.update(Px=Pn.data, Ax=An.data) == 0
        r = s.solve()
        assert np.abs(r.x - d["test_solve_P_A_new_x"]).max() < TOL
        assert abs(r.info.obj_val - d["test_solve_P_A_new_obj_value"]) < TOL
    # too many indexed elements -> error code 1 / 2 (osqp.c:1035, 1225)
    s = fresh()
    assert s.update(Px=np.zeros(Pn.nnz + 1), Px_idx=np.zeros(Pn.nnz + 1, dtype=np.int64)) == 1


def test_invalid_data_and_settings(oracle_mod):
    """tests/basic_qp/test_basic_qp.h:92-389 (subset): validation error codes 1 and 2."""
    pb, _ = load_golden("basic_qp")
    for bad in (dict(rho=-1.0), dict(alpha=2.5), dict(max_iter=0), dict(eps_abs=0.0, eps_rel=0.0),
                dict(scaling=-1), dict(adaptive_rho_tolerance=0.5), dict(warm_start=5)):
        with pytest.raises(ValueError, match="error 2"):
            oracle_mod.OracleOSQP().setup(**pb, **bad)
    bad = dict(pb); bad["l"] = pb["u"] + 1.0
    bad["l"][3] = 0.0
    with pytest.raises(ValueError, match="error 1"):
        oracle_mod.OracleOSQP().setup(**bad)
    Pfull = sparse.csc_matrix([[4., 1.], [1., 2.]])
    # a non-upper-triangular P handed straight to the C API is rejected
    import ctypes as C
    from osqp_amd import abi
    h = oracle_mod.OracleOSQP()
    Ph, Ah = abi.CscHolder(Pfull), abi.CscHolder(pb["A"])
    q, l, u = abi.as_f64(pb["q"]), np.maximum(abi.as_f64(pb["l"]), -1e30), np.minimum(abi.as_f64(pb["u"]), 1e30)
    data = abi.OSQPData(2, 4, C.pointer(Ph.struct), C.pointer(Ah.struct), abi.fptr(q), abi.fptr(l), abi.fptr(u))
    w = C.POINTER(abi.OSQPWorkspace)()
    st = h.default_settings()
    assert h._api["setup"](C.byref(w), C.byref(data), C.byref(st)) == 1


def test_min_degree_is_a_permutation_and_reduces_fill(oracle_mod):
    """The ordering must be a valid permutation and beat the natural order on an arrow matrix."""
    import ctypes as C
    from osqp_amd import abi
    L = oracle_mod.lib()
    n = 200
    M = sparse.lil_matrix((n, n))
    M.setdiag(4.0)
    M[0, :] = 1.0            # arrow: natural order fills completely
    M = sparse.triu(M.tocsc(), format="csc")
    h = abi.CscHolder(M)
    perm = np.zeros(n, dtype=np.int64)
    L.orc_min_degree_order.restype = abi.c_int
    L.orc_min_degree_order.argtypes = [abi.c_int, abi.c_int_p, abi.c_int_p, abi.c_int_p]
    assert L.orc_min_degree_order(n, abi.iptr(h.p), abi.iptr(h.i), abi.iptr(perm)) == 0
    assert sorted(perm.tolist()) == list(range(n))
    assert perm[-1] == 0 or perm[-2] == 0   # the hub is eliminated (almost) last
