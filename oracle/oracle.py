"""CPU ORACLE loader (test infrastructure, NOT the product).

Loads oracle/liborc_osqp.so (plain-C restatement of the reference CPU path,
see oracle/orc_osqp.h) and exposes it through the same Python handle class the
product uses, with the orc_ symbol prefix.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from osqp_amd import _abi as abi
from osqp_amd.interface import SolverHandle

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc_osqp.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc_osqp.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


class OracleOSQP(SolverHandle):
    """Reference CPU path (direct LDL^T) restated in C."""

    def __init__(self):
        super().__init__(lib(), "orc_")

    def iterate(self, k=1):
        f = self._lib.orc_admm_iterate
        f.restype = None
        f.argtypes = [C.POINTER(abi.OSQPWorkspace)]
        for _ in range(k):
            f(self._work)

    def iterates(self):
        w = self.work
        return (self._vec(w.x, self.n), self._vec(w.z, self.m), self._vec(w.y, self.m))


# ---- kernel-level entry points for known-answer tests ----------------------
def _csc(M):
    return abi.CscHolder(M)


def mat_vec(A, x, y=None, plus_eq=0):
    L = lib()
    L.orc_mat_vec.restype = None
    L.orc_mat_vec.argtypes = [C.POINTER(abi.csc), abi.c_float_p, abi.c_float_p, abi.c_int]
    h = _csc(A)
    x = abi.as_f64(x)
    out = np.zeros(h.m) if y is None else abi.as_f64(y).copy()
    L.orc_mat_vec(C.byref(h.struct), abi.fptr(x), abi.fptr(out), plus_eq)
    return out


def mat_tpose_vec(A, x, y=None, plus_eq=0, skip_diag=0):
    L = lib()
    L.orc_mat_tpose_vec.restype = None
    L.orc_mat_tpose_vec.argtypes = [C.POINTER(abi.csc), abi.c_float_p, abi.c_float_p,
                                    abi.c_int, abi.c_int]
    h = _csc(A)
    x = abi.as_f64(x)
    out = np.zeros(h.n) if y is None else abi.as_f64(y).copy()
    L.orc_mat_tpose_vec(C.byref(h.struct), abi.fptr(x), abi.fptr(out), plus_eq, skip_diag)
    return out


def sym_mat_vec(Pu, x):
    """P x from the upper triangle, in the reference's two-pass order
    (auxil.c:299-302: mat_vec then mat_tpose_vec with skip_diag)."""
    y = mat_vec(Pu, x)
    return mat_tpose_vec(Pu, x, y, plus_eq=1, skip_diag=1)


def quad_form(Pu, x):
    L = lib()
    L.orc_quad_form.restype = abi.c_float
    L.orc_quad_form.argtypes = [C.POINTER(abi.csc), abi.c_float_p]
    h = _csc(Pu)
    x = abi.as_f64(x)
    return float(L.orc_quad_form(C.byref(h.struct), abi.fptr(x)))


def col_norms(M, kind):
    L = lib()
    f = {"cols": L.orc_mat_inf_norm_cols, "rows": L.orc_mat_inf_norm_rows,
         "sym": L.orc_mat_inf_norm_cols_sym_triu}[kind]
    f.restype = None
    f.argtypes = [C.POINTER(abi.csc), abi.c_float_p]
    h = _csc(M)
    out = np.zeros(h.m if kind == "rows" else h.n)
    f(C.byref(h.struct), abi.fptr(out))
    return out


def form_KKT(Pu, A, sigma, rho_inv):
    """Upper-triangular KKT as scipy CSC (reference kkt.c:6-177)."""
    from scipy import sparse
    L = lib()
    L.orc_form_KKT.restype = C.POINTER(abi.csc)
    L.orc_form_KKT.argtypes = [C.POINTER(abi.csc), C.POINTER(abi.csc), abi.c_float,
                               abi.c_float_p, abi.c_int_p, abi.c_int_p, abi.c_int_p]
    L.orc_csc_free.restype = None
    L.orc_csc_free.argtypes = [C.POINTER(abi.csc)]
    hp, ha = _csc(Pu), _csc(A)
    r = abi.as_f64(rho_inv)
    nul = C.cast(None, abi.c_int_p)
    K = L.orc_form_KKT(C.byref(hp.struct), C.byref(ha.struct), sigma, abi.fptr(r), nul, nul, nul)
    k = K.contents
    N = k.n
    p = np.ctypeslib.as_array(k.p, shape=(N + 1,)).copy()
    nnz = int(p[-1])
    i = np.ctypeslib.as_array(k.i, shape=(nnz,)).copy()
    x = np.ctypeslib.as_array(k.x, shape=(nnz,)).copy()
    L.orc_csc_free(K)
    return sparse.csc_matrix((x, i, p), shape=(N, N))


def kkt_solve(Pu, A, sigma, rho_vec, rhs, polish=0):
    """init_linsys_solver + solve on a host vector (the plugin boundary KAT)."""
    L = lib()
    S = C.POINTER(abi.LinSysSolver)
    L.orc_init_linsys_solver.restype = abi.c_int
    L.orc_init_linsys_solver.argtypes = [C.POINTER(S), C.POINTER(abi.csc), C.POINTER(abi.csc),
                                         abi.c_float, abi.c_float_p, abi.c_int]
    hp, ha = _csc(Pu), _csc(A)
    rv = abi.as_f64(rho_vec)
    s = S()
    rc = L.orc_init_linsys_solver(C.byref(s), C.byref(hp.struct), C.byref(ha.struct), sigma,
                                  abi.fptr(rv), polish)
    if rc:
        return rc, None
    b = abi.as_f64(rhs).copy()
    s.contents.solve(s, abi.fptr(b))
    s.contents.free(s)
    return 0, b
