import sys, time
sys.path.insert(0, '.')
import numpy as np
import osqp_amd
import oracle.oracle as orc
from osqp_amd.problems import random_sparse_qp, lasso_qp, portfolio_qp, mpc_batch
def rel(a, b): return np.abs(a - b).max() / max(1.0, np.abs(b).max())
cases = []
for (n, m, seed, kw) in [(300, 600, 5, {}), (800, 1600, 6, dict(eps_abs=1e-5, eps_rel=1e-5)), (500, 200, 7, dict(scaling=0, adaptive_rho_interval=50)), (2000, 4000, 1, dict(eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100))]:
    cases.append(("rand %dx%d" % (n, m), random_sparse_qp(n, m, nnz_per_col=min(20, m), seed=seed), kw))
pb = lasso_qp(200, 400, density=0.15, seed=2); cases.append(("lasso200", {k: pb[k] for k in "PqAlu"}, dict(eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=50)))
cases.append(("portfolio8x25", portfolio_qp(8, 25, sector_rows=5, seed=3), dict(eps_abs=1e-5, eps_rel=1e-5)))
s, Q, L, U = mpc_batch(3)
for b in range(3): cases.append(("mpc%d" % b, dict(P=s["P"], q=Q[b], A=s["A"], l=L[b], u=U[b]), {}))
refs = [orc.OracleOSQP().setup(**pb, **kw).solve() for _, pb, kw in cases]
for eps in (1e-7, 1e-8, 1e-9, 1e-10, 1e-12):
    osqp_amd.set_engine_options(pcg_eps_rel=eps)
    row = []
    for (name, pb, kw), ro in zip(cases, refs):
        sg = osqp_amd.OSQP().setup(**pb, **kw); rg = sg.solve(); st = sg.stats()
        row.append("%s it%+d x%.0e y%.0e pcg%.0f" % (name, rg.info.iter - ro.info.iter, rel(rg.x, ro.x), rel(rg.y, ro.y), st["pcg_iters_total"] / max(1, rg.info.iter)))
    print("eps %.0e | " % eps + " | ".join(row), flush=True)
