"""osqp_amd -- MI355X-native drop-in for the OSQP ADMM hot path.

Python here is plumbing only (ctypes over the C ABI of libosqp_amd.so).  The
interface mirrors the reference's Python wrapper: `OSQP().setup(P, q, A, l, u,
**settings)`, `.solve()`, `.update(...)`, `.warm_start(...)`.
"""
import ctypes as C

from . import _abi as abi
from ._lib import lib, build, LIB_PATH
from .interface import SolverHandle, Results
from .batch import BatchOSQP
from .multi import solve_many

__all__ = ["OSQP", "BatchOSQP", "solve_many", "abi", "lib", "build", "engine_options", "set_engine_options"]


class _Options(C.Structure):
    _fields_ = [("pcg_eps_rel", abi.c_float), ("pcg_eps_abs", abi.c_float),
                ("pcg_max_iter", abi.c_int), ("device", abi.c_int), ("pcg_adaptive", abi.c_int)]


class _Stats(C.Structure):
    _fields_ = [("pcg_iters_total", abi.c_int), ("pcg_iters_last", abi.c_int),
                ("pcg_forced", abi.c_int), ("graph_launches", abi.c_int),
                ("host_syncs", abi.c_int), ("resident", abi.c_int)]


def engine_options():
    o = _Options()
    f = lib().osqp_amd_get_options
    f.restype = None
    f.argtypes = [C.POINTER(_Options)]
    f(C.byref(o))
    return {k: getattr(o, k) for k, _ in _Options._fields_}


def set_engine_options(**kw):
    """Defaults for workspaces set up from now on (existing ones keep their own copy: `OSQP.set_options`)."""
    cur = engine_options()
    cur.update(kw)
    o = _Options(**cur)
    f = lib().osqp_amd_set_options
    f.restype = None
    f.argtypes = [C.POINTER(_Options)]
    f(C.byref(o))


class OSQP(SolverHandle):
    """One QP resident on one MI355X (HIP PCG engine behind the reference API)."""

    def __init__(self):
        super().__init__(lib(), "")

    def stats(self):
        s = _Stats()
        f = self._lib.osqp_amd_get_stats
        f.restype = abi.c_int
        f.argtypes = [C.POINTER(abi.OSQPWorkspace), C.POINTER(_Stats)]
        if f(self._work, C.byref(s)):
            raise RuntimeError("no engine")
        return {k: getattr(s, k) for k, _ in _Stats._fields_}

    def options(self):
        """This workspace's engine options (a copy of the defaults taken at setup; `engine_options()` reads the defaults)."""
        o = _Options()
        f = self._lib.osqp_amd_get_workspace_options
        f.restype = abi.c_int
        f.argtypes = [C.POINTER(abi.OSQPWorkspace), C.POINTER(_Options)]
        if f(self._work, C.byref(o)):
            raise RuntimeError("no engine")
        return {k: getattr(o, k) for k, _ in _Options._fields_}

    def set_options(self, **kw):
        """Change engine options of THIS workspace only (pcg_eps_rel, pcg_eps_abs, pcg_max_iter, pcg_adaptive)."""
        cur = self.options()
        for k in kw:
            if k not in cur or k == "device":
                raise ValueError("unknown or fixed option %r" % k)
        cur.update(kw)
        o = _Options(**cur)
        f = self._lib.osqp_amd_set_workspace_options
        f.restype = abi.c_int
        f.argtypes = [C.POINTER(abi.OSQPWorkspace), C.POINTER(_Options)]
        if f(self._work, C.byref(o)):
            raise RuntimeError("no engine")

    def engine(self):
        f = self._lib.osqp_amd_engine
        f.restype = C.c_void_p
        f.argtypes = [C.POINTER(abi.OSQPWorkspace)]
        return f(self._work)
