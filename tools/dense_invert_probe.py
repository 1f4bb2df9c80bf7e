"""Blocked inversion of the dense-direct solve (hipeng_dense_invert_selftest) against numpy, and the rate of its GEMM kernel.
usage: python tools/dense_invert_probe.py [n ...]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_amd
L = osqp_amd.lib()
f = L.hipeng_dense_invert_selftest
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
ok = True
for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 5120]:
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, 2 * n))
    A = G @ G.T / (2 * n) + 0.05 * np.eye(n)
    Ainv = np.zeros((n, n)); ms = (C.c_double * 2)()
    rc = f(n, A.ctypes.data_as(C.c_void_p), Ainv.ctypes.data_as(C.c_void_p), ms)
    err = np.abs(Ainv @ A - np.eye(n)).max()
    ref = np.abs(np.linalg.inv(A) - Ainv).max() / np.abs(Ainv).max()
    print("n=%d rc=%d: max|Ainv A - I| = %.2e, vs numpy %.2e rel; inversion %.2f ms = %.1f TFLOP/s (2 n^3), GEMM n^3: %.2f ms = %.1f TFLOP/s" % (
        n, rc, err, ref, ms[0], 2.0 * n ** 3 / ms[0] / 1e9, ms[1], 2.0 * n ** 3 / ms[1] / 1e9), flush=True)
    ok = ok and rc == 0 and err < 1e-8
sys.exit(0 if ok else 1)
