import sys, time
sys.path.insert(0, '.')
import numpy as np
import osqp_amd
from osqp_amd.problems import lasso_qp
pb = lasso_qp()
s = osqp_amd.OSQP().setup(**{k: pb[k] for k in "PqAlu"}, eps_abs=1e-4, eps_rel=1e-4, max_iter=100)
t = time.time(); r = s.solve(); tv = time.time() - t
st = s.stats()
print("lasso full: 100 iters in %.3fs -> %.1f it/s; pcg/it %.1f => %.1f us per PCG iteration" % (tv, 100 / tv, st["pcg_iters_total"] / 100, 1e6 * tv / st["pcg_iters_total"]))
