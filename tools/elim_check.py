"""Slack elimination (engine.hip, k_elim_refresh) against the oracle and against OSQP_AMD_ELIM=0 on Lasso QPs:
iteration counts, x, y, PCG iterations per ADMM iteration.   usage: python tools/elim_check.py [n_feat m_data ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import osqp_amd
from osqp_amd.problems import lasso_qp
import oracle.oracle as orc

sizes = [(int(sys.argv[k]), int(sys.argv[k + 1])) for k in range(1, len(sys.argv) - 1, 2)] or [(300, 600), (1000, 2000)]
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
for nf, md in sizes:
    pb = lasso_qp(nf, md, 0.15, 1.0, seed=1); pb = {k: pb[k] for k in ("P", "q", "A", "l", "u")}
    kw = dict(eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100)
    ro = orc.OracleOSQP().setup(**pb, **kw).solve() if nf <= 1000 else None
    for elim in (1, 0):
        os.environ["OSQP_AMD_ELIM"] = str(elim); os.environ["OSQP_AMD_RESIDENT"] = "0"
        s = osqp_amd.OSQP().setup(**pb, **kw)
        t0 = time.perf_counter(); r = s.solve(); dt = time.perf_counter() - t0
        st = s.stats()
        line = "lasso %d x %d  ELIM=%d: %s in %d iterations, %.1f PCG iterations each, %.1f it/s, forced %d" % (
            nf, md, elim, r.info.status, r.info.iter, st["pcg_iters_total"] / max(r.info.iter, 1), r.info.iter / dt, st["pcg_forced"])
        if ro is not None:
            line += "; oracle %d iterations, x %.1e y %.1e obj %.1e" % (ro.info.iter, rel(r.x, ro.x), rel(r.y, ro.y), abs(r.info.obj_val - ro.info.obj_val) / max(1, abs(ro.info.obj_val)))
        print(line); sys.stdout.flush()
        # new gamma (q) and a warm-started solve
        q2 = pb["q"].copy(); q2[-nf:] *= 2.0
        s.update(q=q2); r2 = s.solve()
        print("   after update_lin_cost: %s in %d iterations, %.1f PCG iterations each" % (r2.info.status, r2.info.iter, (s.stats()["pcg_iters_total"]) / max(r2.info.iter, 1)))
