"""GPU parity on the status branches of check_termination / osqp_solve that the plain
solves never take (reference src/osqp.c:537-598, src/auxil.c:681-786): the approximate
statuses (`*_INACCURATE`, tolerances x 10 at max_iter), MAX_ITER_REACHED, the time-limit
branch with its approximate check, the verbose reporting cadence (update_info at iteration 1
and every 200, src/osqp.c:411-420), infeasibility certificates against the oracle's vectors,
and the setup-time convexity test on mid-size problems.

Every expected value comes from the CPU oracle run on the same problem and settings in the
same test; the max_iter windows were located with the oracle (a scan over max_iter) and the
tests assert that the oracle really takes the branch before comparing."""
import numpy as np
import pytest
from scipy import sparse

from conftest import load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_linear_solvers")]


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _same_solve(rg, ro, xy=1e-6):
    assert rg.info.status_val == ro.info.status_val and rg.info.status == ro.info.status
    assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
    assert _rel(rg.x, ro.x) < xy and _rel(rg.y, ro.y) < xy
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
    assert abs(rg.info.pri_res - ro.info.pri_res) <= 1e-4 * ro.info.pri_res + 1e-9
    assert abs(rg.info.dua_res - ro.info.dua_res) <= 1e-4 * ro.info.dua_res + 1e-9


@pytest.mark.parametrize("max_iter, expect", [(150, "MAX_ITER_REACHED"), (176, "MAX_ITER_REACHED"), (177, "SOLVED_INACCURATE"),
                                              (200, "SOLVED_INACCURATE"), (212, "SOLVED_INACCURATE"), (213, "SOLVED")])
def test_solved_inaccurate_and_max_iter_match_oracle(gpu_lib, oracle_mod, max_iter, expect):
    """Random QP (n=200, m=400) at eps = 1e-5, which the oracle solves in 225 iterations: stopped at
    max_iter it ends MAX_ITER_REACHED up to 176, SOLVED_INACCURATE from 177 to 212 (residuals below
    10 x the tolerances, osqp.c:576-581 + auxil.c:709-714) and SOLVED from 213 on (post-loop check)."""
    import osqp_amd
    from osqp_amd import abi
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(200, 400, seed=11)
    kw = dict(max_iter=max_iter, eps_abs=1e-5, eps_rel=1e-5)
    ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    assert ro.info.status_val == getattr(abi, "OSQP_" + expect)       # the oracle takes the branch under test
    rg = osqp_amd.OSQP().setup(**pb, **kw).solve()
    _same_solve(rg, ro)
    assert rg.info.iter == max_iter


@pytest.mark.parametrize("scaling", [0, 10])
def test_infeasible_inaccurate_statuses_and_certificates_match_oracle(gpu_lib, oracle_mod, scaling):
    """tests/primal_dual_infeasibility problems stopped early: PRIMAL_INFEASIBLE_INACCURATE (max_iter 28..35),
    DUAL_INFEASIBLE_INACCURATE (22..31) and the exact statuses, each with its certificate compared with the
    oracle's normalised delta_y / delta_x (1e-5)."""
    import osqp_amd
    from osqp_amd import abi
    d = load_golden("primal_dual_infeasibility")
    cases = [("A12", "u2", 30, abi.OSQP_PRIMAL_INFEASIBLE_INACCURATE), ("A12", "u2", 36, abi.OSQP_PRIMAL_INFEASIBLE),
             ("A12", "u2", 50, abi.OSQP_PRIMAL_INFEASIBLE), ("A34", "u3", 24, abi.OSQP_DUAL_INFEASIBLE_INACCURATE),
             ("A34", "u3", 50, abi.OSQP_DUAL_INFEASIBLE), ("A34", "u4", 10, abi.OSQP_DUAL_INFEASIBLE_INACCURATE),
             ("A34", "u4", 25, abi.OSQP_PRIMAL_INFEASIBLE)]
    for A, u, mi, want in cases:
        kw = dict(max_iter=mi, alpha=1.6, scaling=scaling)
        ro = oracle_mod.OracleOSQP().setup(d["P"], d["q"], d[A], d["l"], d[u], **kw).solve()
        assert ro.info.status_val == want, (A, u, mi, ro.info.status)
        rg = osqp_amd.OSQP().setup(d["P"], d["q"], d[A], d["l"], d[u], **kw).solve()
        assert rg.info.status_val == want and rg.info.iter == ro.info.iter, (A, u, mi, rg.info.status, rg.info.iter, ro.info.iter)
        assert rg.info.obj_val == ro.info.obj_val                         # +-OSQP_INFTY
        assert np.all(rg.x == abi.OSQP_NAN) and np.all(rg.y == abi.OSQP_NAN)
        if want in (abi.OSQP_PRIMAL_INFEASIBLE, abi.OSQP_PRIMAL_INFEASIBLE_INACCURATE):
            assert np.abs(rg.prim_inf_cert - ro.prim_inf_cert).max() < 1e-5
        else:
            assert np.abs(rg.dual_inf_cert - ro.dual_inf_cert).max() < 1e-5


def test_primal_infeasible_certificate_midsize_matches_oracle(gpu_lib, oracle_mod):
    """tests/primal_infeasibility (n=50, m=150, random): status, iteration count and the certificate
    delta_y against the oracle's vector, with and without scaling."""
    import osqp_amd
    from osqp_amd import abi
    pb, _ = load_golden("primal_infeasibility")
    for kw in (dict(scaling=0, alpha=1.6, max_iter=10000), dict(max_iter=10000)):
        ro = oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
        rg = osqp_amd.OSQP().setup(**pb, **kw).solve()
        assert rg.info.status_val == ro.info.status_val == abi.OSQP_PRIMAL_INFEASIBLE
        assert rg.info.iter == ro.info.iter
        assert abs(np.abs(rg.prim_inf_cert).max() - 1.0) < 1e-12
        assert np.abs(rg.prim_inf_cert - ro.prim_inf_cert).max() < 1e-5


def test_verbose_run_equals_quiet_run(gpu_lib, capfd):
    """verbose=1 evaluates update_info at iteration 1 and every 200 iterations besides the termination
    cadence (osqp.c:411-420) and prints; the iterates, counts and results must not change."""
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(300, 600, seed=5)
    kw = dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=450, adaptive_rho_interval=75)
    rq = osqp_amd.OSQP().setup(**pb, verbose=0, **kw).solve()
    rv = osqp_amd.OSQP().setup(**pb, verbose=1, **kw).solve()
    out = capfd.readouterr().out
    assert (rv.info.iter, rv.info.status, rv.info.rho_updates) == (rq.info.iter, rq.info.status, rq.info.rho_updates)
    assert np.array_equal(rv.x, rq.x) and np.array_equal(rv.y, rq.y)
    assert rv.info.obj_val == rq.info.obj_val and rv.info.pri_res == rq.info.pri_res and rv.info.dua_res == rq.info.dua_res
    lines = [l.split() for l in out.splitlines() if l[:5].strip().isdigit()]
    its = [int(l[0]) for l in lines]
    assert its[0] == 1 and 200 in its and (rq.info.iter <= 400 or 400 in its) and its[-1] == rq.info.iter


def test_time_limit_takes_the_approximate_branch(gpu_lib, oracle_mod):
    """osqp.c:583-598: on OSQP_TIME_LIMIT_REACHED the approximate check runs.  Warm-started at a 1e-7 solution
    and asked for eps = 1e-9, the iterate after the first window (8 iterations; the time limit is polled
    between windows) has residuals between eps and 10 eps: the oracle stopped at max_iter = 8 says
    `solved inaccurate` through the same approximate check, and so must the time-limited run, which reports the
    iterations it completed -- like the reference, whose `iter - 1` after its break at the top of iteration k is that count
    (osqp.c:404, 545).  Far from the optimum the status stays `run time limit reached`."""
    import osqp_amd
    from osqp_amd import abi
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(200, 400, seed=11)
    r0 = oracle_mod.OracleOSQP().setup(**pb, eps_abs=1e-7, eps_rel=1e-7).solve()
    assert r0.info.status == "solved"
    so = oracle_mod.OracleOSQP().setup(**pb, eps_abs=1e-9, eps_rel=1e-9, check_termination=0, max_iter=8)
    so.warm_start(x=r0.x, y=r0.y)
    ro = so.solve()
    assert ro.info.status_val == abi.OSQP_SOLVED_INACCURATE
    s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-9, eps_rel=1e-9, check_termination=0, time_limit=1e-6, max_iter=100000)
    s.warm_start(x=r0.x, y=r0.y)
    r = s.solve()
    assert r.info.status_val == abi.OSQP_SOLVED_INACCURATE and r.info.iter == 8 == ro.info.iter
    assert _rel(r.x, ro.x) < 1e-6 and _rel(r.y, ro.y) < 1e-6
    assert abs(r.info.pri_res - ro.info.pri_res) <= 1e-3 * ro.info.pri_res and abs(r.info.dua_res - ro.info.dua_res) <= 1e-3 * ro.info.dua_res
    s2 = osqp_amd.OSQP().setup(**pb, eps_abs=1e-9, eps_rel=1e-9, check_termination=0, time_limit=1e-6, max_iter=100000)
    r2 = s2.solve()
    assert r2.info.status_val == abi.OSQP_TIME_LIMIT_REACHED and r2.info.iter == 8


def _indefinite_qp(n, m, neg, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = rng.uniform(0.5, 2.0, n)
    ev[:neg] = -rng.uniform(0.5, 1.0, neg)
    P = Q @ np.diag(ev) @ Q.T
    P = sparse.csc_matrix(0.5 * (P + P.T))
    # A touches only a few variables, so A' rho A cannot lift the negative directions of P
    A = sparse.hstack([sparse.eye(m, format="csc"), sparse.csc_matrix((m, n - m))], format="csc")
    return dict(P=sparse.triu(P, format="csc"), q=rng.standard_normal(n), A=A, l=-np.ones(m), u=np.ones(m))


@pytest.mark.parametrize("n, neg", [(120, 1), (300, 3), (500, 1)])
def test_indefinite_p_rejected_at_setup(gpu_lib, oracle_mod, n, neg):
    """A P with `neg` negative eigenvalues (of n) and sigma = 1e-6: the reduced matrix is indefinite, the
    reference's LDL^T inertia test fails (qdldl_interface.c:93-99) and so does the oracle's; the HIP engine's
    CG probe must meet the negative curvature and return the same OSQP_NONCVX_ERROR (5)."""
    import osqp_amd
    pb = _indefinite_qp(n, 10, neg, seed=n + neg)
    with pytest.raises(ValueError, match="error 5"):
        oracle_mod.OracleOSQP().setup(**pb)
    with pytest.raises(ValueError, match="error 5"):
        osqp_amd.OSQP().setup(**pb)


def test_ill_conditioned_convex_accepted_at_setup(gpu_lib, oracle_mod):
    """Convex but badly conditioned (eigenvalues of P from 1e-5 to 1e1 and a few equality rows at rho_eq = 1e3 rho:
    cond(K) ~ 1e7): the probe must not mistake the rounding floor of CG for negative curvature, and the solve must
    agree with the oracle (x to about cond * pcg_eps_rel)."""
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
    ro = oracle_mod.OracleOSQP().setup(**pb).solve()
    sg = osqp_amd.OSQP().setup(**pb)
    rg = sg.solve()
    assert rg.info.status == ro.info.status == "solved" and rg.info.iter == ro.info.iter
    assert _rel(rg.x, ro.x) < 1e-4 and abs(rg.info.obj_val - ro.info.obj_val) <= 1e-6 * max(1.0, abs(ro.info.obj_val))


def test_engine_options_are_per_workspace(gpu_lib):
    """The side-channel knobs (include/osqp_amd.h) are copied into a workspace at setup: changing the defaults
    or another workspace's copy afterwards does not reach it (no process-global state is read at solve time)."""
    import osqp_amd
    from osqp_amd.problems import demo_qp
    base = osqp_amd.engine_options()
    s1 = osqp_amd.OSQP().setup(**demo_qp())
    try:
        osqp_amd.set_engine_options(pcg_adaptive=1, pcg_eps_rel=1e-9)
        s2 = osqp_amd.OSQP().setup(**demo_qp())
    finally:
        osqp_amd.set_engine_options(**{k: base[k] for k in ("pcg_adaptive", "pcg_eps_rel")})
    assert s1.options()["pcg_adaptive"] == 0 and s1.options()["pcg_eps_rel"] == base["pcg_eps_rel"]
    assert s2.options()["pcg_adaptive"] == 1 and s2.options()["pcg_eps_rel"] == 1e-9
    s1.set_options(pcg_eps_rel=1e-12)
    assert s1.options()["pcg_eps_rel"] == 1e-12 and s2.options()["pcg_eps_rel"] == 1e-9
    assert osqp_amd.engine_options() == base
    with pytest.raises(ValueError):
        s1.set_options(device=3)
    assert s1.solve().info.status == "solved" and s2.solve().info.status == "solved"


def test_wrapper_rejects_out_of_bounds_updates(gpu_lib):
    """The C entry points trust lengths and indices like the reference's (osqp.c:1012-1169); the Python wrapper
    refuses what would read or write outside the matrices, and the C side refuses an out-of-range index."""
    import ctypes as C
    import osqp_amd
    from osqp_amd import abi
    from osqp_amd.problems import random_sparse_qp
    pb = random_sparse_qp(60, 90, seed=2)
    s = osqp_amd.OSQP().setup(**pb)
    with pytest.raises(ValueError):
        s.update(Px=np.ones(3))                                  # short value array without an index array
    with pytest.raises(ValueError):
        s.update(Ax=np.ones(2), Ax_idx=np.array([0, s.nnzA]))    # index past the end
    with pytest.raises(ValueError):
        s.update(Ax=np.ones(2), Ax_idx=np.array([0]))            # lengths differ
    with pytest.raises(ValueError):
        s.update(q=np.ones(59))
    with pytest.raises(ValueError):
        s.warm_start(x=np.ones(61))
    bad = abi.as_i64(np.array([0, s.nnzP + 5])); v = abi.as_f64(np.ones(2))
    assert s._api["update_P"](s._work, abi.fptr(v), abi.iptr(bad), 2) == 1
    Pu = sparse.triu(pb["P"], format="csc"); Pu.sort_indices()
    assert s.update(Px=1.01 * Pu.data[:2], Px_idx=np.array([0, 1])) == 0
    assert s.solve().info.status == "solved"
    with pytest.raises(ValueError):
        osqp_amd.BatchOSQP().setup(pb["P"], pb["A"], np.zeros((2, 60)), np.ones((2, 90)), np.zeros((2, 90)))   # l > u


def test_a_capped_linear_solve_is_reported_without_verbose(gpu_lib, pcg_paths):
    """An indirect solve that stops at its iteration cap is accepted as it stands; OSQPInfo (ABI) cannot say so.  osqp_solve
    therefore prints ONE warning per workspace on stderr even with verbose = 0, and osqp_amd_get_stats counts the solves."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import osqp_amd\n"
            "from osqp_amd.problems import random_sparse_qp\n"
            "osqp_amd.set_engine_options(pcg_max_iter=3)\n"
            "s = osqp_amd.OSQP().setup(**random_sparse_qp(200, 400, seed=2), max_iter=50)\n"
            "s.solve(); s.solve()\n"
            "print('forced', s.stats()['pcg_forced'])\n") % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert int(p.stdout.split("forced")[1]) > 0
    assert p.stderr.count("osqp_amd warning:") == 1 and "pcg_forced" in p.stderr
