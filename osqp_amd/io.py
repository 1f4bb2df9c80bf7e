"""Binary problem files: the wire format of include/osqp_amd.h
(osqp_amd_write_problem / osqp_amd_read_problem) read and written from numpy."""
import numpy as np
from scipy import sparse

MAGIC = b"OSQPAMD1"


def save_problem(path, P, q, A, l, u):
    Pu = sparse.triu(sparse.csc_matrix(P), format="csc"); Pu.sort_indices()
    Ac = sparse.csc_matrix(A); Ac.sort_indices()
    n, m = Pu.shape[0], Ac.shape[0]
    with open(path, "wb") as f:
        f.write(MAGIC)
        np.array([n, m, Pu.nnz, Ac.nnz], dtype="<i8").tofile(f)
        for M in (Pu, Ac):
            M.indptr.astype("<i8").tofile(f); M.indices.astype("<i8").tofile(f); M.data.astype("<f8").tofile(f)
        np.asarray(q, "<f8").tofile(f)
        np.clip(np.asarray(l, "<f8"), -1e30, 1e30).tofile(f)
        np.clip(np.asarray(u, "<f8"), -1e30, 1e30).tofile(f)


def load_problem(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not an osqp_amd problem file")
        n, m, nnzP, nnzA = (int(t) for t in np.fromfile(f, "<i8", 4))
        mats = []
        for rows, nnz in ((n, nnzP), (m, nnzA)):
            p = np.fromfile(f, "<i8", n + 1); i = np.fromfile(f, "<i8", nnz); x = np.fromfile(f, "<f8", nnz)
            mats.append(sparse.csc_matrix((x, i, p), shape=(rows, n)))
        q = np.fromfile(f, "<f8", n); l = np.fromfile(f, "<f8", m); u = np.fromfile(f, "<f8", m)
    return dict(P=mats[0], q=q, A=mats[1], l=l, u=u)
