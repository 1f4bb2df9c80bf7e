"""One-off generator of tests/golden/config3_oracle.json: BASELINE config 3 (Lasso, n_feat=5000,
m_data=10000, 7.5 M non-zeros) at full size through the CPU oracle (direct LDL^T; ~75 s per
factorisation, ~12 ADMM it/s).  The JSON keeps info + subsampled x, y."""
import time, sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osqp_amd.problems import lasso_qp
import oracle.oracle as orc
import numpy as np
pb = {k: v for k, v in lasso_qp().items() if k in "PqAlu"}
t = time.time(); s = orc.OracleOSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4); ts = time.time() - t
print("setup", ts, flush=True)
t = time.time(); r = s.solve(); tv = time.time() - t
out = dict(setup_s=ts, solve_s=tv, iters=r.info.iter, its=r.info.iter / tv, rho_updates=r.info.rho_updates,
           obj=r.info.obj_val, pri=r.info.pri_res, dua=r.info.dua_res, status=r.info.status)
print(json.dumps(out), flush=True)
g = dict(info=out, x_sub=r.x[::20].tolist(), y_sub=r.y[::20].tolist(), x_inf=float(np.abs(r.x).max()),
         y_inf=float(np.abs(r.y).max()), x_sum=float(r.x.sum()), y_sum=float(r.y.sum()))
json.dump(g, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config3_oracle.json"), "w"))
