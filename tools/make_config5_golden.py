"""One-off generator of tests/golden/config5_oracle.json: BASELINE config 5 (portfolio, n=50000,
400 dense 125x125 blocks, budget row with 50 000 entries) at full size through the CPU oracle
(direct LDL^T).  The JSON keeps info + subsampled x, y.
With an argument k: the same with k sparse sector rows (SURVEY C5 "optionally + 500 sparse sector rows") into
tests/golden/config5s_oracle.json."""
import time, sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osqp_amd.problems import portfolio_qp
import oracle.oracle as orc
import ctypes as C
import numpy as np
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pb = portfolio_qp(sector_rows=rows)
t = time.time(); s = orc.OracleOSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100); ts = time.time() - t
print("setup", ts, flush=True)
L = orc.lib(); L.orc_linsys_nnzL.restype = C.c_longlong; L.orc_linsys_nnzL.argtypes = [C.c_void_p]
nnzL = L.orc_linsys_nnzL(C.cast(s.work.linsys_solver, C.c_void_p))
print("nnzL", nnzL, flush=True)
t = time.time(); r = s.solve(); tv = time.time() - t
out = dict(setup_s=ts, solve_s=tv, iters=r.info.iter, its=r.info.iter / tv, nnzL=nnzL, rho_updates=r.info.rho_updates,
           sector_rows=rows, obj=r.info.obj_val, pri=r.info.pri_res, dua=r.info.dua_res, status=r.info.status)
print(json.dumps(out), flush=True)
g = dict(info=out, x_sub=r.x[::50].tolist(), y_sub=r.y[::50].tolist(), x_inf=float(np.abs(r.x).max()),
         y_inf=float(np.abs(r.y).max()), x_sum=float(r.x.sum()), y_sum=float(r.y.sum()))
json.dump(g, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config5s_oracle.json" if rows else "config5_oracle.json"), "w"))
