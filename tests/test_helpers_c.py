"""The helper symbols of the drop-in boundary, called from plain C (examples/helpers_caller.c) the way the
reference's own test translation units use them (include/cs.h, lin_alg.h, kkt.h; tests/*/test_*.h call
vec_norm_inf_diff 63 times, csc_spalloc/csc_spfree/csc_to_triu, mat_vec, form_KKT, update_KKT_*).

CPU part (no GPU needed: the helpers are host C): the program is compiled with gcc against include/,
linked with libosqp_amd.so, run on the reference's fixtures (tests/golden: lin_alg `mat_vec`, update_matrices
`form_KKT`) and every printed result is compared with scipy and with the fixtures' own expected values --
form_KKT entry by entry including the row order inside columns (tests/osqp_tester.h:36-54 does the same).
It also swaps the allocator for a counting one (the reference's custom-memory build,
tests/custom_memory/custom_memory.c:7-35) and checks allocations == frees.

GPU part: the basic_qp data-update sequence (tests/basic_qp/test_basic_qp.h:461-568) through the C API,
checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest
from scipy import sparse

from conftest import load_golden, ROOT


def _build(tmp_path):
    import osqp_amd
    osqp_amd.build()
    exe = tmp_path / "helpers_caller"
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "helpers_caller.c"),
                           "-L", os.path.join(ROOT, "osqp_amd"), "-losqp_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "osqp_amd"), "-o", str(exe)])
    return str(exe)


def _run(exe, path, *extra):
    out = subprocess.run([exe, path, *extra], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = {}
    for line in out.stdout.splitlines():
        if ": " in line or line.endswith(":"):
            k, _, v = line.partition(":")
            res[k.strip()] = v.split()
    return res


def _f(v):
    return np.array([float(t) for t in v])


def _i(v):
    return np.array([int(t) for t in v], dtype=np.int64)


def _mat(res, key, shape, by_row=False):
    p, i, x = _i(res[key + ".p"]), _i(res[key + ".i"]), _f(res[key + ".x"])
    if by_row:
        return sparse.csr_matrix((x, i, p), shape=shape), p, i
    return sparse.csc_matrix((x, i, p), shape=shape), p, i


def _problem(tmp_path, P, A, x, y, name):
    from osqp_amd.io import save_problem
    path = str(tmp_path / name)
    save_problem(path, P, x, A, y, y + 0.75)
    return path


def test_helpers_match_scipy_and_reference_fixtures(tmp_path):
    exe = _build(tmp_path)
    d = load_golden("lin_alg")
    A, Pu, x, y = d["test_mat_vec_A"], sparse.triu(d["test_mat_vec_Pu"], format="csc"), d["test_mat_vec_x"], d["test_mat_vec_y"]
    m, n = A.shape
    r = _run(exe, _problem(tmp_path, Pu, A, x, y, "linalg.bin"))
    u = y + 0.75
    Pf = (Pu + sparse.triu(Pu, 1).T).tocsc()
    T = 1e-13
    # the reference's own expected vectors (tests/lin_alg/generate_problem.py)
    assert np.abs(_f(r["mat_vec"]) - d["test_mat_vec_Ax"]).max() < T
    assert np.abs(_f(r["mat_tpose_vec"]) - d["test_mat_vec_ATy"]).max() < T
    assert np.abs(_f(r["sym_mat_vec"]) - d["test_mat_vec_Px"]).max() < T
    assert np.abs(_f(r["mat_vec_pluseq"]) - 2 * (A @ x)).max() < T and np.abs(_f(r["mat_vec_minuseq"])).max() < T
    assert np.abs(_f(r["mat_tpose_vec_pluseq"]) - 2 * (A.T @ y)).max() < T
    assert abs(_f(r["quad_form"])[0] - 0.5 * x @ (Pf @ x)) < T
    assert np.allclose(_f(r["vec_norms"]), [np.abs(x).max(), np.abs(y - u).max(), np.abs(x * x).max()], rtol=0, atol=T)
    assert np.allclose(_f(r["vec_mean_prod"]), [x.mean(), y @ u], rtol=0, atol=T)
    assert np.allclose(_f(r["vec_add_mult_scalar"]), -2.0 * (x + 0.5), atol=T) and np.allclose(_f(r["vec_add_scaled"]), 4.0 * x, atol=T)
    assert np.allclose(_f(r["vec_ew_recipr"]), 1.0 / x, rtol=1e-15) and np.allclose(_f(r["vec_ew_prod_sqrt"]), np.abs(x), atol=T)
    assert np.allclose(_f(r["vec_ew_max_min"]), np.clip(x, 0.1, 0.4), atol=0)
    assert np.allclose(_f(r["vec_ew_max_vec"]), np.maximum(x, 0.25), atol=0) and np.allclose(_f(r["vec_ew_min_vec"]), np.minimum(x, 0.25), atol=0)
    assert list(_i(r["int_vec"])) == [7, 7, 7]
    assert np.allclose(_f(r["mat_inf_norm_cols"]), np.abs(A).max(axis=0).toarray().ravel(), atol=0)
    assert np.allclose(_f(r["mat_inf_norm_rows"]), np.abs(A).max(axis=1).toarray().ravel(), atol=0)
    assert np.allclose(_f(r["mat_inf_norm_cols_sym_triu"]), np.abs(Pf).max(axis=0).toarray().ravel(), atol=0)
    B, _, _ = _mat(r, "scaled_copy", (m, n))
    assert abs(B - sparse.diags(u) @ (2.0 * A) @ sparse.diags(x)).max() < T
    C, p, i = _mat(r, "triplet_to_csc", (m, n))
    Ac = sparse.csc_matrix(A); Ac.sort_indices()
    assert np.array_equal(p, Ac.indptr) and np.array_equal(i, Ac.indices) and abs(C - A).max() == 0
    mp = _i(r["triplet_to_csc.map"])            # triplet k -> compressed slot: the values must land on themselves
    trip = np.concatenate([Ac.data[Ac.indptr[j]:Ac.indptr[j + 1]] for j in range(n - 1, -1, -1)])
    assert np.array_equal(Ac.data[mp], trip)
    R, _, _ = _mat(r, "triplet_to_csr", (m, n), by_row=True)
    assert abs(R - A).max() == 0
    assert np.array_equal(_f(r["csc_to_dns"]), A.toarray().ravel(order="F"))
    U, p, i = _mat(r, "csc_to_triu", (n, n))
    Pc = sparse.csc_matrix(Pu); Pc.sort_indices()
    assert np.array_equal(p, Pc.indptr) and np.array_equal(i, Pc.indices) and abs(U - Pu).max() == 0
    assert list(_i(r["csc_pinv"])) == list(range(n - 1, -1, -1))
    S, _, _ = _mat(r, "csc_symperm", (n, n))
    Sf = (S + sparse.triu(S, 1).T).toarray()
    assert np.array_equal(Sf, Pf.toarray()[::-1, ::-1]) and abs(sparse.tril(S, -1)).sum() == 0
    assert np.array_equal(S.data[_i(r["csc_symperm.map"])], Pc.data)
    assert list(_i(r["csc_cumsum"])) == [0, 3, 3, 5, 10] and _i(r["csc_cumsum.total"])[0] == 10
    a, f = r["allocator"]
    assert a.split("=")[1] == f.split("=")[1] and int(a.split("=")[1]) > 20


def test_form_kkt_matches_reference_fixture(tmp_path):
    """tests/update_matrices/test_update_matrices.h:13-71 (test_form_KKT): form_KKT and update_KKT_P/A on the
    reference's matrices, compared entry by entry (incl. row order) with the fixture's scipy KKT."""
    exe = _build(tmp_path)
    d = load_golden("update_matrices")
    P, A = sparse.triu(d["test_form_KKT_Pu"], format="csc"), sparse.csc_matrix(d["test_form_KKT_A"])
    n, m = d["test_form_KKT_n"], d["test_form_KKT_m"]
    r = _run(exe, _problem(tmp_path, P, A, np.arange(1.0, n + 1), np.arange(1.0, m + 1), "kkt.bin"))
    sigma, p2 = 0.5, 1.0 / (1.6 + 0.1 * np.arange(m))
    Pf = P + sparse.triu(P, 1).T
    ref = sparse.triu(sparse.bmat([[Pf + sigma * sparse.eye(n), A.T], [A, -sparse.diags(p2)]]), format="csc")
    ref.sort_indices()
    K, p, i = _mat(r, "form_KKT", (n + m, n + m))
    assert np.array_equal(p, ref.indptr) and np.array_equal(i, ref.indices) and np.abs(K.data - ref.data).max() < 1e-15
    Kr, _, _ = _mat(r, "form_KKT_csr", (n + m, n + m), by_row=True)
    assert abs(Kr - ref).max() < 1e-15
    Pc, Ac = sparse.csc_matrix(P), sparse.csc_matrix(A); Pc.sort_indices(); Ac.sort_indices()
    # the index maps point at the entries they claim to
    diag = Pc.indices == np.repeat(np.arange(n), np.diff(Pc.indptr))
    assert np.allclose(K.data[_i(r["form_KKT.PtoKKT"])], Pc.data + sigma * diag, atol=1e-15)
    assert np.array_equal(K.data[_i(r["form_KKT.AtoKKT"])], Ac.data)
    assert np.allclose(K.data[_i(r["form_KKT.param2toKKT"])], -p2, atol=0)
    assert np.array_equal(_i(r["form_KKT.Pdiag_idx"]), np.nonzero(diag)[0])
    ref2 = sparse.triu(sparse.bmat([[2 * Pf + sigma * sparse.eye(n), -A.T], [-A, -sparse.diags(3 * p2)]]), format="csc")
    ref2.sort_indices()
    assert np.abs(_f(r["update_KKT.x"]) - ref2.data).max() < 1e-15
    # and the fixture's own KKT (rho = 1.6 on every row, sigma = 0.1) through the same entry point in Python:
    # the exported form_KKT is what tests/osqp_tester.h:36-54 would compare with test_form_KKT_KKTu
    import ctypes as C
    import osqp_amd
    from osqp_amd import abi
    L = osqp_amd.lib()
    L.form_KKT.restype = C.POINTER(abi.csc)
    L.form_KKT.argtypes = [C.POINTER(abi.csc), C.POINTER(abi.csc), abi.c_int, abi.c_float, abi.c_float_p,
                           abi.c_int_p, abi.c_int_p, C.POINTER(abi.c_int_p), abi.c_int_p, abi.c_int_p]
    L.csc_spfree.restype = None; L.csc_spfree.argtypes = [C.POINTER(abi.csc)]
    hp, ha = abi.CscHolder(P), abi.CscHolder(A)
    rinv = abi.as_f64(np.full(m, 1.0 / d["test_form_KKT_rho"]))
    nul = C.cast(None, abi.c_int_p)
    Kp = L.form_KKT(C.byref(hp.struct), C.byref(ha.struct), 0, d["test_form_KKT_sigma"], abi.fptr(rinv), nul, nul, None, nul, nul)
    k = Kp.contents
    N = n + m
    kp = np.ctypeslib.as_array(k.p, shape=(N + 1,)).copy(); nnz = int(kp[-1])
    ki = np.ctypeslib.as_array(k.i, shape=(nnz,)).copy(); kx = np.ctypeslib.as_array(k.x, shape=(nnz,)).copy()
    L.csc_spfree(Kp)
    gold = sparse.csc_matrix(d["test_form_KKT_KKTu"]); gold.sort_indices()
    assert np.array_equal(kp, gold.indptr) and np.array_equal(ki, gold.indices) and np.abs(kx - gold.data).max() < 1e-15


@pytest.mark.gpu
def test_basic_qp_update_sequence_from_c(gpu_lib, oracle_mod, tmp_path):
    """tests/basic_qp/test_basic_qp.h:461-568 (test_basic_qp_update) from plain C: update_lin_cost / update_bounds
    (valid and l > u) / update_lower_bound / update_upper_bound leave the workspace's copies as given, the refused
    ones return 1, and the solves before and after agree with the oracle driven through the same sequence."""
    exe = _build(tmp_path)
    pb, sol = load_golden("basic_qp")
    from osqp_amd.io import save_problem
    path = str(tmp_path / "basic_qp.bin")
    save_problem(path, pb["P"], pb["q"], pb["A"], pb["l"], pb["u"])
    r = _run(exe, path, "gpu")
    kw = dict(max_iter=200, alpha=1.6, polish=1, scaling=0, warm_start=0)
    so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    r0 = so.solve()
    assert np.abs(_f(r["solve0.x"]) - sol["x_test"]).max() < 1e-4 and np.abs(_f(r["solve0.y"]) - sol["y_test"]).max() < 1e-4
    assert np.abs(_f(r["solve0.x"]) - r0.x).max() < 1e-6 and np.abs(_f(r["solve0.y"]) - r0.y).max() < 1e-6
    for key in ("update_lin_cost", "update_bounds", "update_lower_bound", "update_upper_bound"):
        f = dict(t.split("=") for t in r[key][:1]); assert int(f["rc"]) == 0
        assert all(float(t.split("=")[-1]) == 0.0 for t in r[key][1:])
    assert r["update_bounds_bad"] == ["rc=1"] and r["update_lower_bound_bad"] == ["rc=1"]
    q2, l2, u2 = _f(r["solve1.q"]), _f(r["solve1.l"]), _f(r["solve1.u"])
    so.update(q=q2); so.update(l=l2, u=u2)
    r1 = so.solve()
    s1 = dict(t.split("=") for t in r["solve1"])
    assert int(s1["status"]) == r1.info.status_val and int(s1["iter"]) == r1.info.iter
    assert abs(float(s1["obj"]) - r1.info.obj_val) < 1e-8 * max(1.0, abs(r1.info.obj_val))
    assert np.abs(_f(r["solve1.x"]) - r1.x).max() < 1e-6 and np.abs(_f(r["solve1.y"]) - r1.y).max() < 1e-6
