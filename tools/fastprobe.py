import sys, time
sys.path.insert(0, '.')
import numpy as np
from scipy import sparse
import osqp_amd
from osqp_amd.problems import random_sparse_qp, lasso_qp, portfolio_qp
def kkt(pb, r):
    P = pb["P"] + sparse.triu(pb["P"], 1).T; A = pb["A"]
    l = np.maximum(pb["l"], -1e30); u = np.minimum(pb["u"], 1e30)
    Ax = A @ r.x
    return np.abs(Ax - np.clip(Ax, l, u)).max(), np.abs(P @ r.x + pb["q"] + A.T @ r.y).max()
for name, pb, kw in (("c2", random_sparse_qp(), dict(eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100)),
                     ("lasso1000", {k: v for k, v in lasso_qp(1000, 2000).items() if k in "PqAlu"}, dict(eps_abs=1e-4, eps_rel=1e-4)),
                     ("portfolio40", portfolio_qp(40, 125), dict(eps_abs=1e-4, eps_rel=1e-4))):
    for ad in (0, 1):
        osqp_amd.set_engine_options(pcg_adaptive=ad)
        s = osqp_amd.OSQP().setup(**pb, **kw)
        t = time.time(); r = s.solve(); dt = time.time() - t
        st = s.stats()
        print("%s adaptive=%d: %s iters %d in %.3fs (%.0f it/s) pcg/it %.1f obj %.8f kkt %.2e %.2e" % (name, ad, r.info.status, r.info.iter, dt, r.info.iter / dt, st["pcg_iters_total"] / r.info.iter, r.info.obj_val, *kkt(pb, r)), flush=True)
