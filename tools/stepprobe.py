"""Three bench steps of config 2 and nothing else (for kernel traces)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import osqp_amd
from osqp_amd.problems import random_sparse_qp
if len(sys.argv) > 2: osqp_amd.set_engine_options(pcg_adaptive=int(sys.argv[2]))
pb = random_sparse_qp(10000, 20000)
s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100, warm_start=0)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter(); s.update_rho(0.1); r = s.solve(); dt = time.perf_counter() - t0
    print("step %d: %d iters %.2f ms (%.0f it/s) solve_time %.2f ms" % (i, r.info.iter, dt * 1e3, r.info.iter / dt, r.info.solve_time * 1e3), s.stats())
