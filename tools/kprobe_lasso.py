import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, osqp_amd
from osqp_amd.problems import lasso_qp
pb = lasso_qp()
s = osqp_amd.OSQP().setup(**{k: pb[k] for k in "PqAlu"}, eps_abs=1e-4, eps_rel=1e-4, max_iter=10)
s.solve()
L = osqp_amd.lib()
L.hipeng_time_kernel.restype = C.c_int
L.hipeng_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
L.hipeng_kernel_bytes.restype = C.c_int
L.hipeng_kernel_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
for which, name in ((0, "k_cg_A"), (1, "k_cg_B")):
    us = C.c_double(); by = C.c_double()
    assert L.hipeng_time_kernel(s.engine(), which, 50, C.byref(us)) == 0
    assert L.hipeng_kernel_bytes(s.engine(), which, C.byref(by)) == 0
    print("%s: %.1f us, %.1f MB algorithmic -> %.0f GB/s" % (name, us.value, by.value / 1e6, by.value / us.value / 1e3))
for which, name in ((3, "k_cg_A update-only"), (4, "k_cg_A apply-only")):
    us = C.c_double()
    assert L.hipeng_time_kernel(s.engine(), which, 50, C.byref(us)) == 0
    print("%s: %.1f us" % (name, us.value))
