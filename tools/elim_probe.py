"""Where the engine's slack elimination leaves the oracle's trajectory on the forced general QP of tools/dd_check.py (one iteration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import sparse
import osqp_amd
import oracle.oracle as orc
seed = 1
rng = np.random.default_rng(seed)
n, md, ns = 260, 400, 40
Ad = sparse.random(md, n, density=0.3, random_state=seed, data_rvs=rng.standard_normal, format="csc")
As = sparse.random(150, n, density=0.01, random_state=seed + 7, data_rvs=rng.standard_normal, format="csc")
box = sparse.eye(n, format="csc")
slack_rows = sparse.hstack([sparse.random(ns, n, density=0.05, random_state=seed + 3, data_rvs=rng.standard_normal, format="csc"), -sparse.eye(ns)], format="csc")
G = sparse.random(n, n, density=0.02, random_state=seed + 11, data_rvs=rng.standard_normal, format="csc")
q = rng.standard_normal(n + ns)
l0 = np.concatenate([-1.0 - rng.random(md), -0.5 * np.ones(150), -np.ones(n), np.zeros(ns)])
u0 = np.concatenate([1.0 + rng.random(md), 0.5 * np.ones(150), np.ones(n), np.zeros(ns)])
l0[:20] = u0[:20] = 0.1
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
def run(tag, use_dense=True, use_short=True, use_box=True, eqslack=True, scaling=10, pss=0.5):
    rows = []
    keep = []
    off = 0
    for on, blk, cnt in ((use_dense, Ad, md), (use_short, As, 150), (use_box, box, n)):
        if on: rows.append(sparse.hstack([blk, sparse.csc_matrix((cnt, ns))])); keep += list(range(off, off + cnt))
        off += cnt
    rows.append(slack_rows); keep += list(range(off, off + ns))
    A = sparse.vstack(rows, format="csc")
    l, u = l0[keep].copy(), u0[keep].copy()
    if not eqslack: l[-ns:] = -0.3; u[-ns:] = 0.4
    P = sparse.block_diag([(G @ G.T + 0.05 * sparse.eye(n)).tocsc(), pss * sparse.eye(ns)], format="csc")
    pb = dict(P=sparse.triu(P, format="csc"), q=q, A=A, l=l, u=u)
    kw = dict(max_iter=1, scaling=scaling)
    ro = orc.OracleOSQP().setup(**pb, **kw).solve()
    os.environ["OSQP_AMD_RESIDENT"] = "0"; os.environ["OSQP_AMD_DENSE_DIRECT"] = "0"
    s = osqp_amd.OSQP().setup(**pb, **kw); r = s.solve()
    L = osqp_amd.lib(); L.hipeng_elim_count.restype = int; L.hipeng_elim_count.argtypes = [C.c_void_p]
    print("%-40s eliminated %d: x_main %.2e x_slack %.2e y %.2e" % (tag, L.hipeng_elim_count(s.engine()), rel(r.x[:n], ro.x[:n]), np.abs(r.x[n:] - ro.x[n:]).max(), rel(r.y, ro.y)), flush=True)
import ctypes as C
run("all rows")
run("no scaling", scaling=0)
run("no dense rows", use_dense=False)
run("no short rows", use_short=False)
run("no box rows", use_box=False)
run("slack rows not equalities", eqslack=False)
run("only slack rows", use_dense=False, use_short=False, use_box=False)
run("P_ss = 0", pss=0.0)
