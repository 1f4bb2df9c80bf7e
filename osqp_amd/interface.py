"""Host-side object interface over the C ABI, shaped like the reference's Python
wrapper (docs/interfaces/python.rst in the reference: `OSQP().setup(P, q, A, l,
u, **settings)`, `.solve()`, `.update(...)`, `.warm_start(...)`,
`.update_settings(...)`).  All numerics happen behind the C ABI in
libosqp_amd.so (HIP); this file only marshals arrays.

`SolverHandle` is library-agnostic (library + symbol prefix) so the tests can
drive a second library with the same ABI through exactly the same Python code path.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np
from scipy import sparse

from . import _abi as abi


class Results(SimpleNamespace):
    pass


class SolverHandle:
    def __init__(self, lib, prefix=""):
        self._lib = lib
        self._prefix = prefix
        self._api = abi.bind_api(lib, prefix)
        self._work = None
        self._keep = None
        self.n = self.m = 0

    # ------------------------------------------------------------------ setup
    def default_settings(self):
        s = abi.OSQPSettings()
        self._api["set_default_settings"](C.byref(s))
        s.verbose = 0
        return s

    def setup(self, P=None, q=None, A=None, l=None, u=None, **settings):
        if self._work is not None:
            raise ValueError("solver already set up")
        if P is None and q is None:
            raise ValueError("P and q cannot both be missing")
        n = P.shape[0] if P is not None else len(q)
        if P is None:
            P = sparse.csc_matrix((n, n))
        if q is None:
            q = np.zeros(n)
        if A is None:
            A = sparse.csc_matrix((0, n))
            l = np.zeros(0)
            u = np.zeros(0)
        m = A.shape[0]
        if l is None:
            l = -np.inf * np.ones(m)
        if u is None:
            u = np.inf * np.ones(m)
        Pu = abi.CscHolder(sparse.triu(P, format="csc"))
        Ah = abi.CscHolder(A)
        qv = abi.as_f64(q)
        lv = np.maximum(abi.as_f64(l), -abi.OSQP_INFTY)
        uv = np.minimum(abi.as_f64(u), abi.OSQP_INFTY)
        if qv.shape != (n,) or lv.shape != (m,) or uv.shape != (m,):
            raise ValueError("dimension mismatch")
        if m == 0:  # keep pointers non-NULL like malloc(0) callers do
            lv = np.zeros(1)
            uv = np.zeros(1)
        data = abi.OSQPData(n, m, C.pointer(Pu.struct), C.pointer(Ah.struct),
                            abi.fptr(qv), abi.fptr(lv), abi.fptr(uv))
        st = self.default_settings()
        for k, v in settings.items():
            if k not in abi.SETTING_NAMES:
                raise ValueError("unknown setting %r" % k)
            setattr(st, k, v)
        work = C.POINTER(abi.OSQPWorkspace)()
        rc = self._api["setup"](C.byref(work), C.byref(data), C.byref(st))
        self._keep = (Pu, Ah, qv, lv, uv)
        if rc != 0:
            if work:
                self._api["cleanup"](work)
            raise ValueError("osqp_setup failed with error %d" % rc)
        self._work = work
        self.n, self.m = n, m
        self.nnzP, self.nnzA = Pu.nnz, Ah.nnz
        return self

    # ------------------------------------------------------------------ solve
    def _vec(self, ptr, k):
        if k == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(ptr, shape=(k,)).copy()

    def solve(self):
        w = self._work
        rc = self._api["solve"](w)
        info = w.contents.info.contents
        sol = w.contents.solution.contents
        fields = {f[0]: getattr(info, f[0]) for f in abi.OSQPInfo._fields_}
        fields["status"] = info.status.decode()
        res = Results(x=self._vec(sol.x, self.n), y=self._vec(sol.y, self.m),
                      info=SimpleNamespace(**fields), exitflag=int(rc),
                      prim_inf_cert=self._vec(w.contents.delta_y, self.m),
                      dual_inf_cert=self._vec(w.contents.delta_x, self.n))
        return res

    # ---------------------------------------------------------------- updates
    def update(self, q=None, l=None, u=None, Px=None, Px_idx=None, Ax=None, Ax_idx=None):
        w = self._work
        rc = 0
        if q is not None:
            q = abi.as_f64(q)
            if q.shape != (self.n,):
                raise ValueError("q must have %d entries" % self.n)
            rc |= self._api["update_lin_cost"](w, abi.fptr(q))
        if l is not None:
            l = np.maximum(abi.as_f64(l), -abi.OSQP_INFTY)
            if l.shape != (self.m,):
                raise ValueError("l must have %d entries" % self.m)
        if u is not None:
            u = np.minimum(abi.as_f64(u), abi.OSQP_INFTY)
            if u.shape != (self.m,):
                raise ValueError("u must have %d entries" % self.m)
        if l is not None and u is not None:
            rc |= self._api["update_bounds"](w, abi.fptr(l), abi.fptr(u))
        elif l is not None:
            rc |= self._api["update_lower_bound"](w, abi.fptr(l))
        elif u is not None:
            rc |= self._api["update_upper_bound"](w, abi.fptr(u))

        def idx(a):
            if a is None:
                return None, C.cast(None, abi.c_int_p)
            a = abi.as_i64(a)
            return a, abi.iptr(a)

        def check(vals, ind, nnz, name):
            # the C entry points trust their arguments like the reference's (osqp.c:1012-1169 reads nnz values when no
            # index array is given and never range-checks the indices): refuse what would read or write out of bounds
            if vals.ndim != 1:
                raise ValueError("%sx must be a vector" % name)
            if ind is None:
                if vals.size != nnz:
                    raise ValueError("%sx has %d values, %s has %d non-zeros (pass %sx_idx for a partial update)" % (name, vals.size, name, nnz, name))
            else:
                if ind.shape != vals.shape:
                    raise ValueError("%sx and %sx_idx differ in length" % (name, name))
                if ind.size and (ind.min() < 0 or ind.max() >= nnz):
                    raise ValueError("%sx_idx out of range [0, %d)" % (name, nnz))

        if Px is not None and Ax is not None:
            Px = abi.as_f64(Px); Ax = abi.as_f64(Ax)
            pi, pip = idx(Px_idx); ai, aip = idx(Ax_idx)
            check(Px, pi, self.nnzP, "P"); check(Ax, ai, self.nnzA, "A")
            rc |= self._api["update_P_A"](w, abi.fptr(Px), pip, len(Px), abi.fptr(Ax), aip, len(Ax))
        elif Px is not None:
            Px = abi.as_f64(Px)
            pi, pip = idx(Px_idx)
            check(Px, pi, self.nnzP, "P")
            rc |= self._api["update_P"](w, abi.fptr(Px), pip, len(Px))
        elif Ax is not None:
            Ax = abi.as_f64(Ax)
            ai, aip = idx(Ax_idx)
            check(Ax, ai, self.nnzA, "A")
            rc |= self._api["update_A"](w, abi.fptr(Ax), aip, len(Ax))
        return int(rc)

    def update_rho(self, rho):
        return int(self._api["update_rho"](self._work, float(rho)))

    def update_settings(self, **kw):
        """Post-setup setting changes (reference osqp.c:1339-1617 setters)."""
        st = self._work.contents.settings.contents
        for k, v in kw.items():
            if k == "rho":
                self.update_rho(v)
                continue
            if k in ("sigma", "scaling", "linsys_solver", "adaptive_rho",
                     "adaptive_rho_interval", "adaptive_rho_tolerance", "adaptive_rho_fraction"):
                raise ValueError("%s cannot be changed after setup" % k)
            setter = getattr(self._lib, self._prefix + "osqp_update_" + k, None)
            if setter is not None:
                tp = dict(abi.OSQPSettings._fields_)[k]
                setter.restype = abi.c_int
                setter.argtypes = [C.POINTER(abi.OSQPWorkspace), tp]
                if setter(self._work, v) != 0:
                    raise ValueError("invalid value for %s" % k)
            else:
                setattr(st, k, v)

    def warm_start(self, x=None, y=None):
        w = self._work
        if (x is not None and np.shape(x) != (self.n,)) or (y is not None and np.shape(y) != (self.m,)):
            raise ValueError("warm start vectors must have %d (x) and %d (y) entries" % (self.n, self.m))
        if x is not None and y is not None:
            x = abi.as_f64(x); y = abi.as_f64(y)
            return int(self._api["warm_start"](w, abi.fptr(x), abi.fptr(y)))
        if x is not None:
            x = abi.as_f64(x)
            return int(self._api["warm_start_x"](w, abi.fptr(x)))
        if y is not None:
            y = abi.as_f64(y)
            return int(self._api["warm_start_y"](w, abi.fptr(y)))
        return 0

    # -------------------------------------------------------------- accessors
    @property
    def work(self):
        return self._work.contents

    def settings(self):
        return self._work.contents.settings.contents

    def cleanup(self):
        if self._work is not None:
            self._api["cleanup"](self._work)
            self._work = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
