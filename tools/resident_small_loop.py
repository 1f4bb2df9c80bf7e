"""Resident launches on SMALL problems, where round 2 saw launches give up: Chronopoulos-Gear mode forced
(two exchanges per PCG iteration, hundreds of exchanges per launch), a bounded number of seconds per case, once with the grid
sized to the problem (the default) and once with round 2's geometry (every CU takes part: OSQP_AMD_RESIDENT_NWG=256).
Prints per case: grid, time per ADMM iteration and per PCG iteration, and what the waits inside the launches reported --
launches that gave up, waits over 50 us that ended well, the longest of them, re-publications.
usage: python tools/resident_small_loop.py [seconds per case]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy import sparse
import osqp_amd
from osqp_amd.problems import random_sparse_qp
from test_gpu_resident import _qp, _info

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
grids = sys.argv[2].split(",") if len(sys.argv) > 2 else ["sized", "256"]      # e.g. "256" with OSQP_AMD_TRACE=1: every window that saw a slow wait says so on stderr


def ill(n=320, m=200):
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    P = sparse.csc_matrix(Q @ np.diag(np.logspace(-5, 1, n)) @ Q.T); P = ((P + P.T) * 0.5).tocsc()
    A = sparse.random(m, n, density=0.05, random_state=2, data_rvs=rng.standard_normal, format="csc")
    x0 = rng.standard_normal(n)
    return dict(P=P, q=rng.standard_normal(n), A=A, l=A @ x0 - 0.5, u=A @ x0 + 0.5)


cases = [("n=400 m=600", _qp(400, 600, 10), dict(eps_abs=1e-5, eps_rel=1e-5)),
         ("n=900 m=700 eq=200", _qp(900, 700, 11, eq=200), dict(eps_abs=1e-5, eps_rel=1e-5)),
         ("n=1500 m=300 eq=40", _qp(1500, 300, 12, eq=40), dict(eps_abs=1e-5, eps_rel=1e-5)),
         ("n=320 cond 1e7", ill(), dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=4000)),
         ("n=2000 m=4000 (config-2 recipe)", random_sparse_qp(2000, 4000, seed=1), dict(eps_abs=1e-4, eps_rel=1e-4)),
         ("n=10000 m=20000 (config 2)", random_sparse_qp(10000, 20000, seed=1), dict(eps_abs=1e-4, eps_rel=1e-4))]
for label, pb, kw in cases:
    for grid in grids:
        for pipe in ((0,) if "config" not in label else (0, 1)):
            os.environ["OSQP_AMD_RESIDENT_PIPE"] = str(pipe)
            if grid == "256": os.environ["OSQP_AMD_RESIDENT_NWG"] = "256"
            else: os.environ.pop("OSQP_AMD_RESIDENT_NWG", None)
            s = osqp_amd.OSQP().setup(**pb, **kw, warm_start=0, adaptive_rho_interval=25)
            inf = _info(s)
            if not inf["in_use"]:
                print("%-34s grid %-5s: not resident" % (label, grid)); continue
            s.solve()                                   # (graphs instantiated, counts calibrated)
            st0 = s.stats(); t0 = time.perf_counter(); solves = 0; iters = 0
            while time.perf_counter() - t0 < budget:
                r = s.solve(); solves += 1; iters += r.info.iter
            dt = time.perf_counter() - t0
            st = s.stats(); pcg = st["pcg_iters_total"] - st0["pcg_iters_total"]
            inf = _info(s)
            print("%-34s grid %-5s (%3d workgroups, E=%2d) %s: %5d solves, %8d PCG iterations, %6.1f us per ADMM iteration, %5.2f us per PCG iteration "
                  "(whole solve / count); gave up %d, waits > 50 us %d (longest %.1f us), re-publications %d" % (
                      label, grid, inf["nwg"], inf["E"], "pipelined" if pipe else "Chronopoulos-Gear", solves, pcg, 1e6 * dt / max(iters, 1), 1e6 * dt / max(pcg, 1),
                      inf["gave_up"], inf["slow_waits"], inf["slow_max_ticks"] * 0.01, inf["republished"]))
            sys.stdout.flush()
            del s
