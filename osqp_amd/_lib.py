"""Loader of the product library libosqp_amd.so (HIP engine + C host side).

There is no CPU fallback: if the library is missing, loading raises; if no HIP
device is present, osqp_setup returns OSQP_LINSYS_SOLVER_LOAD_ERROR.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libosqp_amd.so")
_LIB = None


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("libosqp_amd.so is not built (run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C osqp_amd/csrc`); there is no CPU fallback")
        _LIB = C.CDLL(LIB_PATH)
    return _LIB
