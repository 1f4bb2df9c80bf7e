import time, sys, os
sys.path.insert(0, '.')
os.environ["OSQP_AMD_BATCH_PROFILE"]="1"
import numpy as np
import osqp_amd
from osqp_amd.problems import mpc_batch
s,Q,L,U = mpc_batch(256)
for kw in (dict(max_iter=1, check_termination=0, adaptive_rho=0), dict(max_iter=101, check_termination=0, adaptive_rho=0), {}):
    bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, warm_start=0, **kw)
    bs.solve(fetch=False); r = bs.solve()
    t = r.dual_inf_cert[:, :8] / 100.0   # us (100 MHz)
    print(kw, "median us: load %.1f scale %.1f rho %.1f formK %.1f invert %.1f loop %.1f store %.1f | iters %.1f" % tuple(list(np.median(np.diff(t, axis=1), axis=0)) + [r.iter.mean()]))
    pa = r.dual_inf_cert[:, 8:12] / 100.0 / r.iter[:, None]
    print("   per-iteration us: rhs %.2f gemv %.2f refine %.2f update %.2f" % tuple(np.median(pa, axis=0)))
    mhz = r.dual_inf_cert[:, 12] / (r.dual_inf_cert[:, 7] / 100.0)
    print("   shader clock (clock64 / wall): median %.0f MHz" % np.median(mhz))
for rep in (1, 20, 200):
    bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, warm_start=0)
    for _ in range(rep): bs.solve(fetch=False)
    r = bs.solve()
    print("after %d back-to-back solves: clock %.0f MHz, kernel total %.1f us" % (rep, np.median(r.dual_inf_cert[:, 12] / (r.dual_inf_cert[:, 7] / 100.0)), np.median(r.dual_inf_cert[:, 7]) / 100))
