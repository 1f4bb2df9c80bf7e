"""ctypes mirror of include/osqp_amd_types.h (the C ABI of the drop-in boundary).

Field order follows the reference's default desktop build
(/root/reference/include/types.h:21-319; c_int = long long, c_float = double,
include/glob_opts.h:79-90).  Nothing here computes anything: it only describes
memory layouts and wraps a shared library that exports the osqp_* entry points
(optionally behind a symbol prefix).
"""
import ctypes as C

import numpy as np

c_int = C.c_longlong
c_float = C.c_double
c_int_p = C.POINTER(c_int)
c_float_p = C.POINTER(c_float)

OSQP_INFTY = 1e30
OSQP_NAN = float(0x7FC00000)  # the reference's "NaN" is this number (constants.h:94-96)

STATUS = {
    4: "dual infeasible inaccurate", 3: "primal infeasible inaccurate",
    2: "solved inaccurate", 1: "solved", -2: "maximum iterations reached",
    -3: "primal infeasible", -4: "dual infeasible", -5: "interrupted",
    -6: "run time limit reached", -7: "problem non convex", -10: "unsolved",
}
OSQP_SOLVED = 1
OSQP_SOLVED_INACCURATE = 2
OSQP_PRIMAL_INFEASIBLE_INACCURATE = 3
OSQP_DUAL_INFEASIBLE_INACCURATE = 4
OSQP_MAX_ITER_REACHED = -2
OSQP_PRIMAL_INFEASIBLE = -3
OSQP_DUAL_INFEASIBLE = -4
OSQP_TIME_LIMIT_REACHED = -6
OSQP_NON_CVX = -7
OSQP_UNSOLVED = -10

QDLDL_SOLVER = 0
MKL_PARDISO_SOLVER = 1
HIP_PCG_SOLVER = 2


class csc(C.Structure):
    _fields_ = [("nzmax", c_int), ("m", c_int), ("n", c_int), ("p", c_int_p),
                ("i", c_int_p), ("x", c_float_p), ("nz", c_int)]


class OSQPScaling(C.Structure):
    _fields_ = [("c", c_float), ("D", c_float_p), ("E", c_float_p),
                ("cinv", c_float), ("Dinv", c_float_p), ("Einv", c_float_p)]


class OSQPSolution(C.Structure):
    _fields_ = [("x", c_float_p), ("y", c_float_p)]


class OSQPInfo(C.Structure):
    _fields_ = [("iter", c_int), ("status", C.c_char * 32), ("status_val", c_int),
                ("status_polish", c_int), ("obj_val", c_float), ("pri_res", c_float),
                ("dua_res", c_float), ("setup_time", c_float), ("solve_time", c_float),
                ("update_time", c_float), ("polish_time", c_float), ("run_time", c_float),
                ("rho_updates", c_int), ("rho_estimate", c_float)]


class OSQPData(C.Structure):
    _fields_ = [("n", c_int), ("m", c_int), ("P", C.POINTER(csc)), ("A", C.POINTER(csc)),
                ("q", c_float_p), ("l", c_float_p), ("u", c_float_p)]


class OSQPSettings(C.Structure):
    _fields_ = [("rho", c_float), ("sigma", c_float), ("scaling", c_int),
                ("adaptive_rho", c_int), ("adaptive_rho_interval", c_int),
                ("adaptive_rho_tolerance", c_float), ("adaptive_rho_fraction", c_float),
                ("max_iter", c_int), ("eps_abs", c_float), ("eps_rel", c_float),
                ("eps_prim_inf", c_float), ("eps_dual_inf", c_float), ("alpha", c_float),
                ("linsys_solver", C.c_int), ("delta", c_float), ("polish", c_int),
                ("polish_refine_iter", c_int), ("verbose", c_int),
                ("scaled_termination", c_int), ("check_termination", c_int),
                ("warm_start", c_int), ("time_limit", c_float)]


class OSQPPolish(C.Structure):
    _fields_ = [("Ared", C.POINTER(csc)), ("n_low", c_int), ("n_upp", c_int),
                ("A_to_Alow", c_int_p), ("A_to_Aupp", c_int_p), ("Alow_to_A", c_int_p),
                ("Aupp_to_A", c_int_p), ("x", c_float_p), ("z", c_float_p), ("y", c_float_p),
                ("obj_val", c_float), ("pri_res", c_float), ("dua_res", c_float)]


class LinSysSolver(C.Structure):
    pass


LinSysSolver._fields_ = [
    ("type", C.c_int),
    ("solve", C.CFUNCTYPE(c_int, C.POINTER(LinSysSolver), c_float_p)),
    ("free", C.CFUNCTYPE(None, C.POINTER(LinSysSolver))),
    ("update_matrices", C.CFUNCTYPE(c_int, C.POINTER(LinSysSolver), C.POINTER(csc), C.POINTER(csc))),
    ("update_rho_vec", C.CFUNCTYPE(c_int, C.POINTER(LinSysSolver), c_float_p)),
    ("nthreads", c_int),
]


class OSQPWorkspace(C.Structure):
    _fields_ = [("data", C.POINTER(OSQPData)), ("linsys_solver", C.POINTER(LinSysSolver)),
                ("pol", C.POINTER(OSQPPolish)),
                ("rho_vec", c_float_p), ("rho_inv_vec", c_float_p), ("constr_type", c_int_p),
                ("x", c_float_p), ("y", c_float_p), ("z", c_float_p), ("xz_tilde", c_float_p),
                ("x_prev", c_float_p), ("z_prev", c_float_p),
                ("Ax", c_float_p), ("Px", c_float_p), ("Aty", c_float_p),
                ("delta_y", c_float_p), ("Atdelta_y", c_float_p),
                ("delta_x", c_float_p), ("Pdelta_x", c_float_p), ("Adelta_x", c_float_p),
                ("D_temp", c_float_p), ("D_temp_A", c_float_p), ("E_temp", c_float_p),
                ("settings", C.POINTER(OSQPSettings)), ("scaling", C.POINTER(OSQPScaling)),
                ("solution", C.POINTER(OSQPSolution)), ("info", C.POINTER(OSQPInfo)),
                ("timer", C.c_void_p), ("first_run", c_int), ("clear_update_time", c_int),
                ("rho_update_from_solve", c_int), ("summary_printed", c_int)]


SETTING_NAMES = [f[0] for f in OSQPSettings._fields_]


def as_f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def as_i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    return a.ctypes.data_as(c_int_p)


class CscHolder:
    """Owns numpy arrays + a csc struct pointing at them."""

    def __init__(self, M):
        from scipy import sparse
        M = sparse.csc_matrix(M)
        M.sort_indices()
        self.m, self.n = M.shape
        self.p = as_i64(M.indptr)
        self.i = as_i64(M.indices)
        self.x = as_f64(M.data)
        # keep 1-element backing so pointers stay valid for empty matrices
        if self.i.size == 0:
            self.i = np.zeros(1, np.int64)
            self.x = np.zeros(1, np.float64)
        nnz = int(self.p[-1])
        self.struct = csc(max(nnz, 1), self.m, self.n, iptr(self.p), iptr(self.i),
                          fptr(self.x), -1)
        self.nnz = nnz


def bind_api(lib, prefix=""):
    """Declare argtypes/restypes of the osqp_* entry points on `lib`."""
    W = C.POINTER(OSQPWorkspace)

    def fn(name, res, *args):
        f = getattr(lib, prefix + name)
        f.restype = res
        f.argtypes = list(args)
        return f

    api = {}
    api["set_default_settings"] = fn("osqp_set_default_settings", None, C.POINTER(OSQPSettings))
    api["setup"] = fn("osqp_setup", c_int, C.POINTER(W), C.POINTER(OSQPData), C.POINTER(OSQPSettings))
    api["solve"] = fn("osqp_solve", c_int, W)
    api["cleanup"] = fn("osqp_cleanup", c_int, W)
    api["update_lin_cost"] = fn("osqp_update_lin_cost", c_int, W, c_float_p)
    api["update_bounds"] = fn("osqp_update_bounds", c_int, W, c_float_p, c_float_p)
    api["update_lower_bound"] = fn("osqp_update_lower_bound", c_int, W, c_float_p)
    api["update_upper_bound"] = fn("osqp_update_upper_bound", c_int, W, c_float_p)
    api["warm_start"] = fn("osqp_warm_start", c_int, W, c_float_p, c_float_p)
    api["warm_start_x"] = fn("osqp_warm_start_x", c_int, W, c_float_p)
    api["warm_start_y"] = fn("osqp_warm_start_y", c_int, W, c_float_p)
    api["update_P"] = fn("osqp_update_P", c_int, W, c_float_p, c_int_p, c_int)
    api["update_A"] = fn("osqp_update_A", c_int, W, c_float_p, c_int_p, c_int)
    api["update_P_A"] = fn("osqp_update_P_A", c_int, W, c_float_p, c_int_p, c_int,
                           c_float_p, c_int_p, c_int)
    api["update_rho"] = fn("osqp_update_rho", c_int, W, c_float)
    return api
