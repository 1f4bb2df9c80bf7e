import sys, time
sys.path.insert(0, '.')
import numpy as np
from scipy import sparse
import osqp_amd
from osqp_amd.problems import lasso_qp, portfolio_qp
def kkt(pb, r, eps):
    P = pb["P"] + sparse.triu(pb["P"], 1).T; A = pb["A"]
    x, y = r.x, r.y; Ax = A @ x
    l = np.maximum(pb["l"], -1e30); u = np.minimum(pb["u"], 1e30)
    pri = np.abs(Ax - np.clip(Ax, l, u)).max(); dua = np.abs(P @ x + pb["q"] + A.T @ y).max()
    return pri, dua
for name, pb in (("lasso 500x1000 d=.15", lasso_qp(500, 1000)), ("lasso 5000x10000 d=.15", lasso_qp()),
                 ("portfolio 40x125", portfolio_qp(40, 125)), ("portfolio 400x125", portfolio_qp())):
    t = time.time(); s = osqp_amd.OSQP().setup(**{k: pb[k] for k in "PqAlu"}, eps_abs=1e-4, eps_rel=1e-4); ts = time.time() - t
    t = time.time(); r = s.solve(); tv = time.time() - t
    st = s.stats()
    print(name, "n=%d m=%d nnzA=%d nnzP=%d | setup %.2fs solve %.3fs iters %d status %s obj %.6f | kkt pri %.2e dua %.2e | pcg/it %.1f forced %d" % (
        s.n, s.m, s.nnzA, s.nnzP, ts, tv, r.info.iter, r.info.status, r.info.obj_val, *kkt(pb, r, 1e-4), st["pcg_iters_total"] / max(1, r.info.iter), st["pcg_forced"]), flush=True)
