"""Steady-state statistics of the 1024-QP batch leg: iteration histogram, rho updates, time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, osqp_amd
from osqp_amd.problems import mpc_batch
s, Q, L, U = mpc_batch(1024)
for kw in ({}, dict(adaptive_rho=0), dict(check_termination=10), dict(check_termination=5)):
    bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, warm_start=0, **kw)
    for _ in range(3): bs.solve(fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): bs.solve(fetch=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    r = bs.solve()
    it = r.iter
    print(kw, "%.3f ms/batch; iters mean %.1f max %d sum %d; hist(25s) %s; rho_updates sum %d; solved %d" % (
        dt * 1e3, it.mean(), it.max(), it.sum(), np.bincount((it // 25).astype(int)).tolist(), int(r.rho_updates.sum()), int((r.status_val == 1).sum())))
    print("   bound sum/256*2.22us = %.3f ms, longest = %.3f ms" % (it.sum() / 256 * 2.22e-3, it.max() * 2.22e-3))
