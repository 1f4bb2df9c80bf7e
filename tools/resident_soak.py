"""Soak test of the resident PCG launch: many cold-started solves of one equality-constrained QP (long linear solves:
hundreds of in-launch exchanges each); reports whether any launch timed out (the engine then leaves the resident path)
and whether every solve returned the same bits.   usage: python tools/resident_soak.py [solves] [n m eq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, osqp_amd
from test_gpu_resident import _qp
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, m, eq = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (900, 700, 200)
pb = _qp(n, m, 11, eq=eq)
s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-5, eps_rel=1e-5, adaptive_rho=0, warm_start=0)   # every solve the same work
ref = None; bad = 0; t0 = time.perf_counter(); pcg = 0
for k in range(solves):
    r = s.solve()
    st = s.stats()
    if ref is None: ref = (r.x.copy(), r.y.copy(), r.info.iter)
    same = np.array_equal(r.x, ref[0]) and np.array_equal(r.y, ref[1]) and r.info.iter == ref[2]
    if not st["resident"] or not same:
        bad += 1
        print("solve %d: resident=%d identical=%s iter=%d" % (k, st["resident"], same, r.info.iter)); sys.stdout.flush()
        if not st["resident"]: break
import ctypes as C
L = osqp_amd.lib(); L.hipeng_resident_info.restype = C.c_int; L.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
info = (C.c_longlong * 16)(); L.hipeng_resident_info(s.engine(), info)
print("%d solves of %d ADMM iterations, %d PCG iterations in all, %.1f s: %d anomalies, %d failed true-residual checks, %d launches that gave up waiting (PIPE=%s)" % (
    k + 1, ref[2], s.stats()["pcg_iters_total"], time.perf_counter() - t0, bad, info[8], info[10], os.environ.get("OSQP_AMD_RESIDENT_PIPE", "1")))
