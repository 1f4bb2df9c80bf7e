import sys, time
sys.path.insert(0, '.')
import numpy as np
from scipy import sparse
import osqp_amd
from osqp_amd.problems import lasso_qp
pb = lasso_qp()
t = time.time(); s = osqp_amd.OSQP().setup(**{k: pb[k] for k in "PqAlu"}, eps_abs=1e-4, eps_rel=1e-4); print("setup %.3fs" % (time.time() - t))
t = time.time(); r = s.solve(); print("cold solve %.3fs iters %d" % (time.time() - t, r.info.iter))
A = sparse.csc_matrix(pb["A"]); A.sort_indices()
Ax_new = A.data * (1.0 + 0.01 * np.random.default_rng(0).standard_normal(A.nnz))
t = time.time(); rc = s.update(Ax=Ax_new); tu = time.time() - t
print("update_A rc=%d %.3fs (info.update_time %.3f)" % (rc, tu, s.work.info.contents.update_time))
t = time.time(); r = s.solve(); print("warm solve %.3fs iters %d status %s" % (time.time() - t, r.info.iter, r.info.status))
nf, md = pb["n_feat"], pb["m_data"]
for g in (2.0, 4.0):
    q = np.concatenate([np.zeros(nf + md), g * np.ones(nf)])
    t = time.time(); s.update(q=q); tq = time.time() - t
    t = time.time(); r = s.solve(); print("gamma %.1f: update_q %.4fs solve %.3fs iters %d" % (g, tq, time.time() - t, r.info.iter))
