"""Dense inversion hook on ill-conditioned SPD matrices (is a pivot ever reported non-positive?)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_amd
f = osqp_amd.lib().hipeng_dense_invert_selftest
f.restype = C.c_int; f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(4)
for n, lo in ((256, -5), (256, -7), (512, -6)):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = Q @ np.diag(10.0 ** rng.uniform(lo, 3, n)) @ Q.T
    A = 0.5 * (A + A.T)
    Ainv = np.zeros((n, n))
    rc = f(n, A.ctypes.data_as(C.c_void_p), Ainv.ctypes.data_as(C.c_void_p), None)
    print("n=%d eig 1e%d..1e3: rc=%d, |Ainv A - I| = %.2e (numpy: %.2e)" % (n, lo, rc, np.abs(Ainv @ A - np.eye(n)).max(), np.abs(np.linalg.inv(A) @ A - np.eye(n)).max()))
