import time, sys, os
sys.path.insert(0, '.')
import numpy as np, torch
import osqp_amd
from osqp_amd.problems import mpc_batch
def run(B, **kw):
    s,Q,L,U = mpc_batch(B)
    bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, warm_start=0, **kw)
    bs.solve(fetch=False)
    t=time.perf_counter()
    for _ in range(5): bs.solve(fetch=False)
    return (time.perf_counter()-t)/5*1e3
for B in (256, 1024):
    a = run(B, max_iter=1, check_termination=0, adaptive_rho=0)
    b = run(B, max_iter=101, check_termination=0, adaptive_rho=0)
    c = run(B, max_iter=101, check_termination=25, adaptive_rho=0, eps_abs=1e-12, eps_rel=1e-12)
    d = run(B, max_iter=1, check_termination=0, adaptive_rho=0, scaling=0)
    print("B=%d refine=%s: setup+1it %.3f ms (scaling=0: %.3f) ; +100 it %.3f ms -> %.2f us/it ; with 4 checks %.3f ms" % (B, os.environ.get("OSQP_AMD_BATCH_REFINE","1"), a, d, b, (b-a)*10, c))
