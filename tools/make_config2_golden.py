"""One-off generator of tests/golden/config2_oracle.json: BASELINE config 2 at full size through the
CPU oracle (direct LDL^T).  Takes ~10 minutes single-threaded (ordering + factorisation 300 s,
solve incl. one re-factorisation 300 s).  The JSON keeps info + subsampled x, y."""
import time, sys, json
sys.path.insert(0,'/root/repo')
from osqp_amd.problems import random_sparse_qp
import oracle.oracle as orc
import ctypes as C
pb = random_sparse_qp()
t=time.time(); s=orc.OracleOSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100); ts=time.time()-t
print("setup", ts, flush=True)
L=orc.lib(); L.orc_linsys_nnzL.restype=C.c_longlong; L.orc_linsys_nnzL.argtypes=[C.c_void_p]
nnzL=L.orc_linsys_nnzL(C.cast(s.work.linsys_solver, C.c_void_p))
print("nnzL", nnzL, flush=True)
t=time.time(); r=s.solve(); tv=time.time()-t
out=dict(setup_s=ts, solve_s=tv, iters=r.info.iter, its=r.info.iter/tv, nnzL=nnzL, rho_updates=r.info.rho_updates, obj=r.info.obj_val, pri=r.info.pri_res, dua=r.info.dua_res, status=r.info.status)
print(json.dumps(out), flush=True)
import numpy as np
g = dict(info=out, x_sub=r.x[::10].tolist(), y_sub=r.y[::20].tolist(), x_inf=float(np.abs(r.x).max()),
         y_inf=float(np.abs(r.y).max()), x_sum=float(r.x.sum()), y_sum=float(r.y.sum()))
json.dump(g, open('/root/repo/tests/golden/config2_oracle.json', 'w'))
