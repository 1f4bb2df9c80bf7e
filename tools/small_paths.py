"""Which linear solver for small and mid-size QPs?  Launch-per-step PCG, resident PCG and the dense-direct solve (forced) on random
sparse QPs and MPC-like QPs: whole-solve time (cold start, default settings) and the cost of an osqp_update_rho."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_amd
from osqp_amd.problems import random_sparse_qp
def run(pb, env):
    os.environ.update(env)
    try:
        t0 = time.perf_counter(); s = osqp_amd.OSQP().setup(**pb); ts = time.perf_counter() - t0
    finally:
        for k in env: os.environ.pop(k)
    r = s.solve()
    best = 1e9
    for _ in range(3):
        s.warm_start(x=np.zeros(pb["P"].shape[0]), y=np.zeros(pb["A"].shape[0])); s.update_rho(0.1)
        t0 = time.perf_counter(); r = s.solve(); best = min(best, time.perf_counter() - t0)
    t0 = time.perf_counter(); s.update_rho(0.2); tr = time.perf_counter() - t0
    st = s.stats(); s.cleanup()
    return "%6.2f ms (%d it, %d rho upd, setup %.0f ms, update_rho %.2f ms)" % (1e3 * best, r.info.iter, r.info.rho_updates, 1e3 * ts, 1e3 * tr)
for n, m in ((150, 300), (300, 600), (600, 1200), (1000, 2000), (2000, 4000), (4000, 8000)):
    pb = random_sparse_qp(n, m, seed=n)
    print("n=%d m=%d:" % (n, m), flush=True)
    print("   per-step PCG :", run(pb, dict(OSQP_AMD_RESIDENT="0", OSQP_AMD_DENSE_DIRECT="0")), flush=True)
    print("   resident PCG :", run(pb, dict(OSQP_AMD_DENSE_DIRECT="0")), flush=True)
    print("   dense-direct :", run(pb, dict(OSQP_AMD_RESIDENT="0", OSQP_AMD_DENSE_DIRECT="2")), flush=True)
