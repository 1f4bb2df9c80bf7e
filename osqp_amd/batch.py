"""Batched solves of many small QPs with a shared sparsity pattern (MPC-style,
BASELINE config 4) -- ctypes plumbing over include/osqp_amd_batch.h.  One
workgroup per QP on the GPU; see osqp_amd/csrc/batch.hip."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
from scipy import sparse

from . import _abi as abi
from ._lib import lib

BATCH_MAX_N = 128     # register-tiled K^-1 of the batch kernel (batch.hip)
INFO_FIELDS = ["iter", "status_val", "obj_val", "pri_res", "dua_res", "rho_updates", "rho_estimate", "rho"]


def _bind(L):
    H = C.c_void_p
    L.osqp_amd_batch_setup.restype = abi.c_int
    L.osqp_amd_batch_setup.argtypes = [C.POINTER(H), abi.c_int, C.POINTER(abi.csc), C.POINTER(abi.csc),
                                       abi.c_float_p, abi.c_float_p, abi.c_float_p, abi.c_float_p,
                                       abi.c_float_p, C.POINTER(abi.OSQPSettings), abi.c_int]
    L.osqp_amd_batch_update.restype = abi.c_int
    L.osqp_amd_batch_update.argtypes = [H, abi.c_float_p, abi.c_float_p, abi.c_float_p]
    L.osqp_amd_batch_solve.restype = abi.c_int
    L.osqp_amd_batch_solve.argtypes = [H]
    L.osqp_amd_batch_get.restype = abi.c_int
    L.osqp_amd_batch_get.argtypes = [H, abi.c_float_p, abi.c_float_p, abi.c_float_p, abi.c_float_p, abi.c_float_p]
    L.osqp_amd_batch_cleanup.restype = None
    L.osqp_amd_batch_cleanup.argtypes = [H]


def _p(a):
    return C.cast(None, abi.c_float_p) if a is None else abi.fptr(a)


class BatchOSQP:
    def __init__(self):
        self._lib = lib()
        _bind(self._lib)
        self._h = None

    def setup(self, P, A, Q, L, U, Px_all=None, Ax_all=None, device=None, **settings):
        """P (n x n, any triangle content; upper triangle is used), A (m x n): shared
        pattern/values.  Q [B, n], L, U [B, m].  Px_all / Ax_all [B, nnz] optional
        per-QP values in CSC order of triu(P) / A."""
        from . import engine_options
        self.Pu = abi.CscHolder(sparse.triu(P, format="csc"))
        self.Ah = abi.CscHolder(A)
        Q = abi.as_f64(Q)
        self.B, self.n = Q.shape
        self.m = self.Ah.m
        L = np.maximum(abi.as_f64(L), -abi.OSQP_INFTY)
        U = np.minimum(abi.as_f64(U), abi.OSQP_INFTY)
        if L.shape != (self.B, self.m) or U.shape != (self.B, self.m) or self.Pu.n != self.n or self.Ah.n != self.n:
            raise ValueError("dimension mismatch: Q [B, n], L, U [B, m], P n x n, A m x n")
        if np.any(L > U):
            raise ValueError("lower bound greater than upper bound")          # validate_data, src/auxil.c:868-875
        for nm, V, nnz in (("Px_all", Px_all, self.Pu.nnz), ("Ax_all", Ax_all, self.Ah.nnz)):
            if V is not None and np.shape(V) != (self.B, nnz):
                raise ValueError("%s must be [B, %d]" % (nm, nnz))
        st = abi.OSQPSettings()
        self._lib.osqp_set_default_settings.restype = None
        self._lib.osqp_set_default_settings.argtypes = [C.POINTER(abi.OSQPSettings)]
        self._lib.osqp_set_default_settings(C.byref(st))
        st.verbose = 0
        for k, v in settings.items():
            if k not in abi.SETTING_NAMES:
                raise ValueError("unknown setting %r" % k)
            setattr(st, k, v)
        if Px_all is not None:
            Px_all = abi.as_f64(Px_all)
        if Ax_all is not None:
            Ax_all = abi.as_f64(Ax_all)
        if device is None:
            device = engine_options()["device"]
        self._many = None
        if self.n > BATCH_MAX_N:
            # The one-workgroup-per-QP kernel keeps K^-1 in registers (n <= 128).  Larger members of a batch go
            # one QP per HIP stream through the single-QP engine (osqp_amd/multi.py): same results, the PCG path.
            from . import OSQP, set_engine_options
            old = engine_options()["device"]
            set_engine_options(device=device)
            try:
                Pf, Af = sparse.csc_matrix(P), sparse.csc_matrix(A)
                self._many = []
                for b in range(self.B):
                    Pb, Ab = Pf.copy(), Af.copy()
                    if Px_all is not None:
                        Pb = sparse.triu(Pf, format="csc"); Pb.sort_indices(); Pb.data = Px_all[b].copy()
                    if Ax_all is not None:
                        Ab.sort_indices(); Ab.data = Ax_all[b].copy()
                    self._many.append(OSQP().setup(P=Pb, q=Q[b], A=Ab, l=L[b], u=U[b], **settings))
            finally:
                set_engine_options(device=old)
            self._last = None
            return self
        h = C.c_void_p()
        rc = self._lib.osqp_amd_batch_setup(C.byref(h), self.B, C.byref(self.Pu.struct), C.byref(self.Ah.struct),
                                            _p(Px_all), _p(Ax_all), abi.fptr(Q), _p(L if self.m else None),
                                            _p(U if self.m else None), C.byref(st), device)
        if rc:
            raise ValueError("osqp_amd_batch_setup failed with error %d" % rc)
        self._h = h
        return self

    def _many_update(self, Q, L, U):
        # the first failing member's code comes back (and which member it was stays readable in `last_update_failed`)
        self.last_update_failed = None
        for b, s in enumerate(self._many):
            rc = s.update(q=None if Q is None else Q[b], l=None if L is None else L[b], u=None if U is None else U[b])
            if rc:
                self.last_update_failed = b
                return rc
        return 0

    def _many_results(self):
        rs = self._last
        X = np.array([r.x for r in rs]); Y = np.array([r.y for r in rs]).reshape(self.B, self.m)
        info = np.array([[r.info.iter, r.info.status_val, r.info.obj_val, r.info.pri_res, r.info.dua_res, r.info.rho_updates,
                          r.info.rho_estimate, s.settings().rho] for r, s in zip(rs, self._many)], dtype=np.float64)
        out = SimpleNamespace(x=X, y=Y, dual_inf_cert=np.array([r.dual_inf_cert for r in rs]),
                              prim_inf_cert=np.array([r.prim_inf_cert for r in rs]).reshape(self.B, self.m), info_raw=info)
        for k, name in enumerate(INFO_FIELDS):
            col = info[:, k]
            setattr(out, name, col.astype(np.int64) if name in ("iter", "status_val", "rho_updates") else col)
        return out

    def update(self, Q=None, L=None, U=None):
        Q = None if Q is None else abi.as_f64(Q)
        L = None if L is None else np.maximum(abi.as_f64(L), -abi.OSQP_INFTY)
        U = None if U is None else np.minimum(abi.as_f64(U), abi.OSQP_INFTY)
        if (Q is not None and Q.shape != (self.B, self.n)) or (L is not None and L.shape != (self.B, self.m)) or \
           (U is not None and U.shape != (self.B, self.m)):
            raise ValueError("update arrays must be Q [B, n], L [B, m], U [B, m]")
        if L is not None and U is not None and np.any(L > U):
            raise ValueError("lower bound greater than upper bound")
        if self._many is not None:
            return self._many_update(Q, L, U)
        return int(self._lib.osqp_amd_batch_update(self._h, _p(Q), _p(L), _p(U)))

    def solve(self, fetch=True):
        if self._many is not None:
            from .multi import solve_many
            self._last = solve_many(self._many, max_workers=8)
            return self._many_results() if fetch else None
        rc = self._lib.osqp_amd_batch_solve(self._h)
        if rc:
            raise RuntimeError("osqp_amd_batch_solve failed (%d)" % rc)
        return self.results() if fetch else None

    def results(self):
        if self._many is not None:
            return self._many_results()
        X = np.zeros((self.B, self.n)); Y = np.zeros((self.B, max(self.m, 1)))
        info = np.zeros((self.B, 8)); DX = np.zeros((self.B, self.n)); DY = np.zeros((self.B, max(self.m, 1)))
        rc = self._lib.osqp_amd_batch_get(self._h, abi.fptr(X), abi.fptr(Y), abi.fptr(info), abi.fptr(DX), abi.fptr(DY))
        if rc:
            raise RuntimeError("osqp_amd_batch_get failed (%d)" % rc)
        out = SimpleNamespace(x=X, y=Y[:, :self.m], dual_inf_cert=DX, prim_inf_cert=DY[:, :self.m], info_raw=info)
        for k, name in enumerate(INFO_FIELDS):
            col = info[:, k]
            setattr(out, name, col.astype(np.int64) if name in ("iter", "status_val", "rho_updates") else col)
        return out

    def device_arrays(self):
        """The result arrays as they sit in HBM -- X [B, n], Y [B, m], info8 [B, 8] -- as objects carrying
        `__cuda_array_interface__` (no copy; `torch.as_tensor(a, device="cuda")` wraps them for a device-side
        gather).  Valid until the next solve / cleanup."""
        if self._many is not None:
            raise RuntimeError("this batch runs one single-QP engine per member (n > %d): its results are host arrays "
                               "(results()); there is no packed device image to wrap" % BATCH_MAX_N)
        f = self._lib.osqp_amd_batch_device_ptrs
        f.restype = abi.c_int
        f.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        px, py, pi = C.c_void_p(), C.c_void_p(), C.c_void_p()
        if f(self._h, C.byref(px), C.byref(py), C.byref(pi)):
            raise RuntimeError("osqp_amd_batch_device_ptrs failed")

        class _Dev:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = dict(shape=shape, typestr="<f8", data=(int(ptr), False), version=2, strides=None)
        return _Dev(px.value, (self.B, self.n)), _Dev(py.value, (self.B, max(self.m, 1))), _Dev(pi.value, (self.B, 8))

    def cleanup(self):
        if getattr(self, "_many", None):
            for s in self._many:
                s.cleanup()
            self._many = None
        if self._h is not None:
            self._lib.osqp_amd_batch_cleanup(self._h)
            self._h = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
