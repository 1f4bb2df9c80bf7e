"""Resident PCG (engine.hip: k_form_K, k_pcg_resident): the reduced matrix K = P + sigma I + A' diag(rho) A held in
registers and the whole linear solve of an ADMM iteration in one launch.  Checks: K itself against scipy, the three
modes of the kernel (pipelined, pipelined handed over after a failed true-residual check, Chronopoulos-Gear only)
against the oracle through the C ABI, problems far below the size the mode is meant for, and two processes sharing
the GPU (a resident launch that finds CUs taken gives up and the engine continues with the launch-per-step kernels)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy import sparse

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pcg_paths")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


class _env:
    def __init__(self, **kw): self.kw = kw
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = str(v)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _info(solver):
    import osqp_amd
    L = osqp_amd.lib()
    L.hipeng_resident_info.restype = C.c_int
    L.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    out = (C.c_longlong * 16)()
    assert L.hipeng_resident_info(solver.engine(), out) == 0
    return dict(built=out[0], in_use=out[1], E=out[2], nwg=out[3], nnzK=out[4], lds=out[5], last_iters=out[6], pipe_off=out[7], checks_failed=out[8], form=out[9], gave_up=out[10],
                slow_waits=out[11], slow_max_ticks=out[12], republished=out[13], strikes=out[14], coupling_rows=out[15])


def _qp(n, m, seed, eq=0, dens=0.02):
    rng = np.random.default_rng(seed)
    A = sparse.random(m, n, density=dens, random_state=seed, data_rvs=rng.standard_normal, format="csc")
    A = (A + sparse.csc_matrix((np.ones(min(m, n)), (np.arange(min(m, n)), np.arange(min(m, n)))), shape=(m, n))).tocsc()
    B = sparse.random(n, n, density=dens, random_state=seed + 1, data_rvs=rng.standard_normal, format="csc")
    P = (B @ B.T + 0.05 * sparse.eye(n)).tocsc()
    q = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    l = A @ x0 - rng.uniform(0.1, 1, m); u = A @ x0 + rng.uniform(0.1, 1, m)
    if eq: l[:eq] = u[:eq]
    return dict(P=P, q=q, A=A, l=l, u=u)


def test_resident_K_is_P_plus_sigma_plus_AtRhoA(gpu_lib):
    """k_form_K: every entry of K against scipy (unscaled problem, so the engine's matrices are the caller's), the
    pattern complete, and K symmetric to the bit (both triangles are summed in the same order)."""
    import osqp_amd
    pb = _qp(700, 1100, 3, eq=150)
    s = osqp_amd.OSQP().setup(**pb, scaling=0, rho=0.3, sigma=1e-6, adaptive_rho=0)
    inf = _info(s)
    assert inf["built"] and inf["in_use"] and inf["nwg"] > 0 and inf["E"] >= 8
    L = osqp_amd.lib()
    L.hipeng_resident_dump.restype = C.c_longlong
    L.hipeng_resident_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
    nnz = int(inf["nnzK"])
    row = np.zeros(nnz, dtype=np.int32); col = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
    assert L.hipeng_resident_dump(s.engine(), row.ctypes.data, col.ctypes.data, val.ctypes.data, nnz) == nnz
    n, m = 700, 1100
    K = sparse.coo_matrix((val, (row, col)), shape=(n, n)).toarray()
    w = s.work
    rho = np.ctypeslib.as_array(w.rho_vec, shape=(m,)).copy()
    assert set(np.round(rho / 0.3, 6)) == {1.0, 1000.0}                  # equality rows carry 1e3 rho
    Pf = pb["P"].toarray()
    Pf = np.triu(Pf) + np.triu(Pf, 1).T
    A = pb["A"].toarray()
    Kref = Pf + 1e-6 * np.eye(n) + A.T @ (rho[:, None] * A)
    assert np.abs(K - Kref).max() <= 1e-13 * np.abs(Kref).max()
    assert np.array_equal(K, K.T)
    # every structural non-zero of the reference has a slot
    pat = sparse.coo_matrix((np.ones(nnz), (row, col)), shape=(n, n)).toarray() > 0
    assert not ((np.abs(Kref) > 0) & ~pat).any()
    # and the triplets are unique
    assert len(set(zip(row.tolist(), col.tolist()))) == nnz


@pytest.mark.parametrize("pipe", [1, 0])
def test_resident_modes_match_oracle(gpu_lib, oracle_mod, pipe):
    """Mid-size QPs with equality rows, with and without the pipelined phase: same iteration count, rho updates and
    status as the oracle's direct solve; x, y to 1e-6, objective to 1e-8 (the bars of the launch-per-step path)."""
    import osqp_amd
    for seed, (n, m, eq) in enumerate([(400, 600, 0), (900, 700, 200), (1500, 300, 40)]):
        pb = _qp(n, m, 10 + seed, eq=eq)
        kw = dict(eps_abs=1e-5, eps_rel=1e-5, adaptive_rho_interval=25)
        with _env(OSQP_AMD_RESIDENT_PIPE=pipe):
            sg = osqp_amd.OSQP().setup(**pb, **kw)
        so = oracle_mod.OracleOSQP().setup(**pb, **kw)
        assert _info(sg)["in_use"]
        rg, ro = sg.solve(), so.solve()
        assert rg.info.status == ro.info.status == "solved"
        assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
        assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
        assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-8 * max(1.0, abs(ro.info.obj_val))
        # new q, bounds, then a warm-started solve
        q2 = pb["q"] * 1.1
        sg.update(q=q2); so.update(q=q2)
        rg, ro = sg.solve(), so.solve()
        assert rg.info.iter == ro.info.iter and _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
        inf = _info(sg)
        assert inf["in_use"] and inf["gave_up"] == 0, inf


def test_resident_hands_ill_conditioned_solves_to_the_robust_recurrences(gpu_lib, oracle_mod):
    """cond(K) ~ 1e7 at a tight tolerance: the pipelined recurrences stall above the stop there (their residual drifts
    from the true one).  Every pipelined solve is checked against r0 - K (x - x0); the failed check continues the solve
    with Chronopoulos-Gear and switches the pipelined phase off for this K.  Result: the oracle's trajectory."""
    import osqp_amd
    rng = np.random.default_rng(7)
    n, m = 320, 200
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    P = sparse.csc_matrix(Q @ np.diag(np.logspace(-5, 1, n)) @ Q.T)
    P = ((P + P.T) * 0.5).tocsc()
    A = sparse.random(m, n, density=0.05, random_state=2, data_rvs=rng.standard_normal, format="csc")
    x0 = rng.standard_normal(n)
    pb = dict(P=P, q=rng.standard_normal(n), A=A, l=A @ x0 - 0.5, u=A @ x0 + 0.5)
    kw = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=4000)
    sg = osqp_amd.OSQP().setup(**pb, **kw); so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    assert _info(sg)["in_use"]
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status == "solved"
    assert rg.info.iter == ro.info.iter
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-5
    inf = _info(sg)
    assert inf["in_use"] and inf["gave_up"] == 0, inf            # still resident: the hand-over happens inside the launch


def test_resident_on_tiny_problems(gpu_lib, oracle_mod):
    """Far below the size the mode is meant for (most workgroups own no row): OSQP_AMD_RESIDENT_MIN_N=1."""
    import osqp_amd
    for seed, (n, m) in enumerate([(2, 3), (7, 5), (40, 90), (130, 61)]):
        pb = _qp(n, m, 30 + seed, eq=min(2, m), dens=0.3)
        with _env(OSQP_AMD_RESIDENT_MIN_N=1):
            sg = osqp_amd.OSQP().setup(**pb, eps_abs=1e-6, eps_rel=1e-6)
        assert _info(sg)["in_use"]
        so = oracle_mod.OracleOSQP().setup(**pb, eps_abs=1e-6, eps_rel=1e-6)
        rg, ro = sg.solve(), so.solve()
        assert rg.info.status == ro.info.status and rg.info.iter == ro.info.iter
        assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6


def test_resident_with_many_entries_per_thread(gpu_lib, oracle_mod):
    """A dense K (n = 1800: 3.2 M entries, 48 per thread): the instantiations with 32 and more entries per thread keep part of
    K in scratch memory; same results."""
    import osqp_amd
    pb = _qp(1800, 300, 60, eq=30, dens=0.03)
    kw = dict(eps_abs=1e-5, eps_rel=1e-5)
    sg = osqp_amd.OSQP().setup(**pb, **kw); so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    inf = _info(sg)
    assert inf["in_use"] and inf["E"] >= 32, inf
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status == "solved" and rg.info.iter == ro.info.iter
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert _info(sg)["in_use"]


@pytest.mark.parametrize("direct", [1, 0])
def test_block_forms_of_the_portfolio_match_oracle(gpu_lib, oracle_mod, direct):
    """P = dense diagonal blocks, rows of A single-entry + one huge budget row (72 blocks of 125, the budget row has 9000
    entries).  direct=1: the block-direct solve (k_blk_invert / k_blk_apply / k_blk_finish: explicit inverse blocks + the budget
    row as a Woodbury term, ONE "PCG iteration" per linear solve); direct=0: the block-resident PCG (k_pcg_blockres).  Same
    trajectory as the oracle's direct solve; then new q and a warm-started solve, an osqp_update_rho; and the same problem on
    the launch-per-step kernels (OSQP_AMD_RESIDENT_BLOCKS=0): one trajectory, three linear solvers."""
    import osqp_amd
    from osqp_amd.problems import portfolio_qp
    pb = portfolio_qp(72, 125, seed=5)
    kw = dict(eps_abs=1e-4, eps_rel=1e-4)
    with _env(OSQP_AMD_BLOCK_DIRECT=direct):
        sg = osqp_amd.OSQP().setup(**pb, **kw)
    so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    inf = _info(sg)
    assert inf["built"] and inf["form"] == (3 if direct else 2), inf
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status == "solved"
    assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-6 * max(1.0, abs(ro.info.obj_val))     # (the bar of the full-size config-5 test)
    if direct:
        st = sg.stats()
        assert st["pcg_iters_total"] == rg.info.iter and st["pcg_forced"] == 0          # one application of K^-1 per ADMM iteration
    q2 = pb["q"] * 0.9
    sg.update(q=q2); so.update(q=q2)
    rg2, ro2 = sg.solve(), so.solve()
    assert rg2.info.iter == ro2.info.iter and _rel(rg2.x, ro2.x) < 1e-6 and _rel(rg2.y, ro2.y) < 1e-6
    sg.update_rho(0.4); so.update_rho(0.4)           # the inverse blocks and the capacitance matrix are formed again
    rg3, ro3 = sg.solve(), so.solve()
    assert rg3.info.iter == ro3.info.iter and _rel(rg3.x, ro3.x) < 1e-6 and _rel(rg3.y, ro3.y) < 1e-6
    with _env(OSQP_AMD_RESIDENT_BLOCKS=0):
        s2 = osqp_amd.OSQP().setup(**pb, **kw)
    assert _info(s2)["form"] not in (2, 3)
    r2 = s2.solve()
    assert r2.info.iter == rg.info.iter and _rel(r2.x, rg.x) < 1e-7 and _rel(r2.y, rg.y) < 1e-7
    assert _info(sg)["gave_up"] == 0


@pytest.mark.parametrize("nb,b,rows", [(8, 40, 5), (72, 125, 40), (24, 100, 300)])
def test_block_direct_with_sector_rows_matches_oracle(gpu_lib, oracle_mod, nb, b, rows):
    """SURVEY C5 "+ sparse sector rows": rows of A that tie variables of different blocks together.  The block-direct solve
    carries EVERY multi-entry row of A (the sector rows and the budget row) as the low-rank term of a Woodbury solve whose
    capacitance matrix is formed and inverted on the device (k_cpl_dot / k_cpl_solve / k_blk_apply_back / k_cap_invert).  Same
    trajectory as the oracle's direct solve, also after new bounds, an osqp_update_rho and new values of A; and one application of
    K^-1 per ADMM iteration."""
    import osqp_amd
    from osqp_amd.problems import portfolio_qp
    pb = portfolio_qp(nb, b, sector_rows=rows, seed=7)
    kw = dict(eps_abs=1e-4, eps_rel=1e-4)
    sg = osqp_amd.OSQP().setup(**pb, **kw)
    so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    inf = _info(sg)
    multi = int((np.diff(pb["A"].tocsr().indptr) >= 2).sum())
    assert multi >= 2 and inf["built"] and inf["form"] == 3 and inf["coupling_rows"] == multi, inf
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status == "solved"
    assert rg.info.iter == ro.info.iter and rg.info.rho_updates == ro.info.rho_updates
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    assert abs(rg.info.obj_val - ro.info.obj_val) <= 1e-6 * max(1.0, abs(ro.info.obj_val))
    st = sg.stats()
    assert st["pcg_iters_total"] == rg.info.iter and st["pcg_forced"] == 0
    u2 = pb["u"].copy(); u2[-rows:] *= 0.8                       # tighter sector caps
    sg.update(u=u2); so.update(u=u2)
    rg2, ro2 = sg.solve(), so.solve()
    assert rg2.info.iter == ro2.info.iter and _rel(rg2.x, ro2.x) < 1e-6 and _rel(rg2.y, ro2.y) < 1e-6
    sg.update_rho(0.4); so.update_rho(0.4)                      # blocks and capacitance matrix are formed again
    rg3, ro3 = sg.solve(), so.solve()
    assert rg3.info.iter == ro3.info.iter and _rel(rg3.x, ro3.x) < 1e-6 and _rel(rg3.y, ro3.y) < 1e-6
    Ax = pb["A"].data.copy(); Ax[pb["A"].indices > pb["A"].shape[0] - rows - 1] *= 1.1     # new values in the sector rows
    sg.update(Ax=Ax); so.update(Ax=Ax)
    rg4, ro4 = sg.solve(), so.solve()
    assert rg4.info.iter == ro4.info.iter and _rel(rg4.x, ro4.x) < 1e-6 and _rel(rg4.y, ro4.y) < 1e-6
    with _env(OSQP_AMD_BLOCK_COUPLED_MAX=0):                    # the same problem on the launch-per-step kernels
        s2 = osqp_amd.OSQP().setup(**pb, **kw)
    assert _info(s2)["form"] != 3
    r2 = s2.solve()
    assert r2.info.iter == rg.info.iter and _rel(r2.x, rg.x) < 1e-6 and _rel(r2.y, rg.y) < 1e-6


def test_block_direct_with_ill_conditioned_blocks(gpu_lib, oracle_mod):
    """Blocks with eigenvalues from 1e-6 to 1 whose variables have no row of their own (only sigma = 1e-6 regularises them:
    cond(B_b) = 5e5).  The inverse blocks come from element-wise Gauss-Jordan in LDS (error ~ cond eps on a positive definite block --
    it is the BLOCKED sweeps of the dense-direct solve that square the condition number) and are checked against the blocks as
    formed (k_blk_check) at every refresh; here the check passes, the engine keeps the block-direct solve and follows the oracle."""
    import osqp_amd
    rng = np.random.default_rng(3)
    nb, b = 72, 125
    n = nb * b
    blocks = []
    for _ in range(nb):
        Q, _ = np.linalg.qr(rng.standard_normal((b, b)))
        Pb = Q @ np.diag(10.0 ** rng.uniform(-6, 0, b)) @ Q.T
        blocks.append(sparse.csc_matrix(np.triu(0.5 * (Pb + Pb.T))))
    P = sparse.block_diag(blocks, format="csc")
    half = np.arange(0, 50)                               # box rows on fifty variables of the first block only
    A = sparse.vstack([sparse.csc_matrix(np.ones((1, n))), sparse.eye(n, format="csc")[half]], format="csc")
    pb = dict(P=P, q=-1e-4 * rng.standard_normal(n), A=A, l=np.concatenate([[1.0], -np.ones(half.size)]), u=np.concatenate([[1.0], np.ones(half.size)]))
    kw = dict(eps_abs=1e-4, eps_rel=1e-4, scaling=0)
    sg = osqp_amd.OSQP().setup(**pb, **kw)
    assert _info(sg)["form"] == 3
    rg, ro = sg.solve(), oracle_mod.OracleOSQP().setup(**pb, **kw).solve()
    assert rg.info.status == ro.info.status and rg.info.iter == ro.info.iter
    assert _rel(rg.x, ro.x) < 1e-5 and _rel(rg.y, ro.y) < 1e-5
    assert _info(sg)["form"] == 3


def test_block_direct_reports_an_indefinite_block(gpu_lib):
    """A dense block of P with a negative eigenvalue: the Gauss-Jordan inversion meets a non-positive pivot, the setup-time
    convexity probe sees it as negative curvature and osqp_setup returns OSQP_NONCVX_ERROR (reference: the LDL' inertia test,
    qdldl_interface.c:93-99)."""
    import osqp_amd
    from osqp_amd.problems import portfolio_qp
    pb = portfolio_qp(72, 125, seed=2)
    P = sparse.csc_matrix(pb["P"] + sparse.triu(pb["P"], 1).T).tolil()
    w, V = np.linalg.eigh(P[:125, :125].toarray())
    w[0] = -0.5
    P[:125, :125] = (V * w) @ V.T
    P = sparse.csc_matrix(P); P = ((P + P.T) * 0.5)
    pb2 = dict(pb, P=sparse.triu(P, format="csc"))
    ok = osqp_amd.OSQP().setup(**{k: pb[k] for k in "PqAlu"})
    assert _info(ok)["form"] == 3
    with pytest.raises(ValueError, match="error 5"):
        osqp_amd.OSQP().setup(**{k: pb2[k] for k in "PqAlu"})


def test_a_launch_that_gives_up_costs_one_window(gpu_lib, oracle_mod):
    """The give-up path on demand (OSQP_AMD_RESIDENT_INJECT: in the launch of ADMM iteration 40 one workgroup walks away).
    That call finishes on the launch-per-step kernels, the next one is resident again -- with an epoch beyond the tags the
    abandoned launch left in the exchange buffers -- and every result is the oracle's."""
    import osqp_amd
    pb = _qp(700, 500, 70, eq=60)
    kw = dict(eps_abs=1e-5, eps_rel=1e-5, adaptive_rho_interval=25)
    with _env(OSQP_AMD_RESIDENT_INJECT=40):
        sg = osqp_amd.OSQP().setup(**pb, **kw)
    so = oracle_mod.OracleOSQP().setup(**pb, **kw)
    assert _info(sg)["in_use"]
    rg, ro = sg.solve(), so.solve()
    assert rg.info.status == ro.info.status == "solved" and rg.info.iter == ro.info.iter > 40
    assert _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    inf = _info(sg)
    assert inf["gave_up"] == 1 and inf["in_use"] == 1, inf
    for k in range(3):          # resident again: same trajectories as the oracle
        q2 = pb["q"] * (1.0 + 0.1 * (k + 1))
        sg.update(q=q2); so.update(q=q2)
        rg, ro = sg.solve(), so.solve()
        assert rg.info.iter == ro.info.iter and _rel(rg.x, ro.x) < 1e-6 and _rel(rg.y, ro.y) < 1e-6
    inf = _info(sg)
    assert inf["gave_up"] == 1 and inf["strikes"] == 0, inf      # clean calls since: the strike count started over


def test_resident_off_by_environment_and_for_large_n(gpu_lib):
    import osqp_amd
    pb = _qp(400, 300, 50)
    with _env(OSQP_AMD_RESIDENT=0):
        s = osqp_amd.OSQP().setup(**pb)
    assert not _info(s)["built"]
    r0 = s.solve()
    s1 = osqp_amd.OSQP().setup(**pb)
    r1 = s1.solve()
    assert _info(s1)["in_use"]
    # two different linear solvers, one ADMM trajectory
    assert r0.info.iter == r1.info.iter and _rel(r0.x, r1.x) < 1e-7 and _rel(r0.y, r1.y) < 1e-7


_WORKER = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import osqp_amd
from osqp_amd.problems import random_sparse_qp
pb = random_sparse_qp(2000, 4000, seed=4)
s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-5, eps_rel=1e-5)
out = []
for k in range(12):
    s.update(q=pb["q"] * (1.0 + 0.05 * k))
    r = s.solve()
    out.append(dict(status=r.info.status, iter=int(r.info.iter), obj=float(r.info.obj_val), x0=float(r.x[0]), xs=float(np.abs(r.x).sum())))
print(json.dumps(out))
"""


def test_two_processes_share_the_gpu(gpu_lib):
    """A resident launch needs every CU; a second process's launches take some.  Whatever the interleaving does --
    launches that wait for each other's workgroups give up after 20 ms and the engine carries on with the
    launch-per-step kernels -- both processes must return the same solutions as a process running alone."""
    code = _WORKER % ROOT
    alone = json.loads(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, check=True).stdout.strip().splitlines()[-1])
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        got = json.loads(so.strip().splitlines()[-1])
        for a, b in zip(alone, got):
            assert b["status"] == a["status"] == "solved" and b["iter"] == a["iter"]
            assert abs(b["obj"] - a["obj"]) <= 1e-8 * max(1.0, abs(a["obj"]))
            assert abs(b["xs"] - a["xs"]) <= 1e-7 * a["xs"]
