"""Fill of the CPU baseline's ordering: nnz(L) of the oracle's LDL^T (own approximate-minimum-degree ordering,
oracle/orc_ldl.c) against an independent minimum-degree implementation on the same KKT matrices.

The reference orders with its vendored SuiteSparse AMD (lin_sys/direct/qdldl/qdldl_interface.c:106-173).  Those
sources cannot be compiled in this image without writing a stand-in for the cmake-generated osqp_configure.h
that every one of them includes through glob_opts.h (amd/include/SuiteSparse_config.h:45, amd_internal.h:40),
so the yardstick here is SuperLU's MMD on A'+A (scipy.sparse.linalg.splu, permc_spec="MMD_AT_PLUS_A", symmetric
mode, no pivoting): the multiple-minimum-degree code AMD descends from, an implementation independent of the oracle.
A ratio near 1 says the oracle's factorisation -- and with it the timed CPU baseline -- is not a strawman.

usage: python tools/nnzL_check.py [--full]      (CPU only; --full adds config 2 at n = 10000: minutes and ~6 GB)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import sparse
from scipy.sparse.linalg import splu

import oracle.oracle as orc
from osqp_amd import _abi as abi
from osqp_amd.problems import random_sparse_qp, lasso_qp, portfolio_qp


def oracle_nnzL(K):
    """nnz(L) (strictly lower part) of the oracle's ordering + elimination tree on upper-triangular K."""
    L = orc.lib()
    n = K.shape[0]
    Ku = sparse.triu(K, format="csc"); Ku.sort_indices()
    p, i = abi.as_i64(Ku.indptr), abi.as_i64(Ku.indices)
    perm = np.zeros(n, dtype=np.int64)
    L.orc_min_degree_order.restype = abi.c_int
    L.orc_min_degree_order.argtypes = [abi.c_int, abi.c_int_p, abi.c_int_p, abi.c_int_p]
    t0 = time.perf_counter()
    assert L.orc_min_degree_order(n, abi.iptr(p), abi.iptr(i), abi.iptr(perm)) == 0
    t_ord = time.perf_counter() - t0
    Pm = sparse.eye(n, format="csc")[:, perm]          # column j of Pm = e_perm[j]
    Kp = (Pm.T @ (Ku + sparse.triu(Ku, 1).T) @ Pm).tocsc()
    Kpu = sparse.triu(Kp, format="csc"); Kpu.sort_indices()
    pp, ii = abi.as_i64(Kpu.indptr), abi.as_i64(Kpu.indices)
    work = np.zeros(n, dtype=np.int64); Lnz = np.zeros(n, dtype=np.int64); et = np.zeros(n, dtype=np.int64)
    L.orc_ldl_etree.restype = abi.c_int
    L.orc_ldl_etree.argtypes = [abi.c_int, abi.c_int_p, abi.c_int_p, abi.c_int_p, abi.c_int_p, abi.c_int_p]
    tot = L.orc_ldl_etree(n, abi.iptr(pp), abi.iptr(ii), abi.iptr(work), abi.iptr(Lnz), abi.iptr(et))
    assert tot >= 0
    return int(tot), t_ord


def superlu_nnzL(K):
    Kf = (sparse.triu(K) + sparse.triu(K, 1).T).tocsc()
    t0 = time.perf_counter()
    lu = splu(Kf, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    return int(lu.L.nnz - Kf.shape[0]), time.perf_counter() - t0


def kkt(pb, sigma=1e-6, rho=0.1):
    l, u = pb["l"], pb["u"]
    rv = np.where(u - l < 1e-4, 1e3 * rho, rho)
    return orc.form_KKT(sparse.triu(pb["P"], format="csc"), pb["A"], sigma, 1.0 / rv)


def main():
    full = "--full" in sys.argv
    orc.build()
    cases = [("config2 recipe n=2000 m=4000", random_sparse_qp(2000, 4000, seed=1)),
             ("config2 recipe n=4000 m=8000", random_sparse_qp(4000, 8000, seed=1)),
             ("config3 Lasso 500x1000 (n=2000, m=2000)", {k: v for k, v in lasso_qp(500, 1000, density=0.15).items() if k in "PqAlu"}),
             ("config3 Lasso 1500x3000 (n=6000, m=6000)", {k: v for k, v in lasso_qp(1500, 3000, density=0.15).items() if k in "PqAlu"}),
             ("config5 portfolio n=50000 (full size)", portfolio_qp())]
    if full:
        cases.insert(2, ("config2 n=10000 m=20000 (full size)", random_sparse_qp()))
    print("%-44s %12s %12s %7s   (ordering s: oracle / SuperLU factor s)" % ("KKT matrix", "nnzL oracle", "nnzL MMD", "ratio"))
    for name, pb in cases:
        K = kkt(pb)
        a, ta = oracle_nnzL(K)
        b, tb = superlu_nnzL(K)
        print("%-44s %12d %12d %7.3f   (%.1f / %.1f)" % (name, a, b, a / max(1, b), ta, tb), flush=True)


if __name__ == "__main__":
    main()
