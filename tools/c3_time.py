"""Config 3 (Lasso 5000 x 10000) on the dense-direct solve and on the launch-per-step PCG: setup, cold solve, warm re-solves.
usage: python tools/c3_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_amd
from osqp_amd.problems import lasso_qp
full = lasso_qp()
pb = {k: v for k, v in full.items() if k in "PqAlu"}
nf, md = full["n_feat"], full["m_data"]
for dd in (["1", "0"] if len(sys.argv) < 2 else sys.argv[1:]):
    os.environ["OSQP_AMD_DENSE_DIRECT"] = dd
    t0 = time.perf_counter(); s = osqp_amd.OSQP().setup(**pb); ts = time.perf_counter() - t0
    t0 = time.perf_counter(); r = s.solve(); tv = time.perf_counter() - t0
    t0 = time.perf_counter(); s.update_rho(0.2); tr = time.perf_counter() - t0
    q = np.concatenate([np.zeros(nf + md), 4.0 * np.ones(nf)])
    s.update(q=q); t0 = time.perf_counter(); r2 = s.solve(); t2 = time.perf_counter() - t0
    print("OSQP_AMD_DENSE_DIRECT=%s: setup %.3f s; cold solve %d iterations (%d rho updates) in %.3f s = %.0f it/s; osqp_update_rho %.1f ms; warm re-solve %d iterations = %.0f it/s" % (
        dd, ts, r.info.iter, r.info.rho_updates, tv, r.info.iter / tv, 1e3 * tr, r2.info.iter, r2.info.iter / t2), flush=True)
    s.cleanup()
