import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, osqp_amd
from osqp_amd.problems import random_sparse_qp
pb = random_sparse_qp()
s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100)
s.solve()
L = osqp_amd.lib()
L.hipeng_time_kernel.restype = C.c_int
L.hipeng_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
def t(which):
    us = C.c_double(); assert L.hipeng_time_kernel(s.engine(), which, 500, C.byref(us)) == 0; return us.value
print("k_cg_A full        %.2f us" % t(0))
print("k_cg_B full        %.2f us" % t(1))
print("k_cg_B no blocksum %.2f us" % t(1 | (32 << 8)))
print("k_cg_B prefetch    %.2f us" % t(1 | (16 << 8)))
print("k_cg_B empty       %.2f us" % t(1 | (64 << 8)))
print("k_cg_B known desc  %.2f us" % t(1 | (128 << 8)))
