"""Setup time of config 2 with and without the resident-PCG structures (symbolic K on the host, layout, upload).
usage: python tools/setup_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import osqp_amd
from osqp_amd.problems import random_sparse_qp
pb = random_sparse_qp(10000, 20000, seed=1)
for res in ("1", "0", "1", "0"):
    os.environ["OSQP_AMD_RESIDENT"] = res
    t0 = time.perf_counter(); s = osqp_amd.OSQP().setup(**pb); t1 = time.perf_counter()
    r = s.solve(); t2 = time.perf_counter()
    print("OSQP_AMD_RESIDENT=%s: setup %.3f s, first solve %.1f ms (%d iterations, resident=%d)" % (res, t1 - t0, 1e3 * (t2 - t1), r.info.iter, s.stats()["resident"]))
    s.cleanup()
