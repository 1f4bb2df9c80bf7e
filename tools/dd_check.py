"""Dense-direct solve against the launch-per-step PCG and the oracle on the forced general QP of tests/test_gpu_dense_direct.py.
usage: python tools/dd_check.py [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import sparse
import osqp_amd
import oracle.oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
n, md, ns = 260, 400, 40
Ad = sparse.random(md, n, density=0.3, random_state=seed, data_rvs=rng.standard_normal, format="csc")
As = sparse.random(150, n, density=0.01, random_state=seed + 7, data_rvs=rng.standard_normal, format="csc")
box = sparse.eye(n, format="csc")
slack_rows = sparse.hstack([sparse.random(ns, n, density=0.05, random_state=seed + 3, data_rvs=rng.standard_normal, format="csc"), -sparse.eye(ns)], format="csc")
A = sparse.vstack([sparse.hstack([Ad, sparse.csc_matrix((md, ns))]), sparse.hstack([As, sparse.csc_matrix((150, ns))]),
                   sparse.hstack([box, sparse.csc_matrix((n, ns))]), slack_rows], format="csc")
G = sparse.random(n, n, density=0.02, random_state=seed + 11, data_rvs=rng.standard_normal, format="csc")
P = sparse.block_diag([(G @ G.T + 0.05 * sparse.eye(n)).tocsc(), 0.5 * sparse.eye(ns)], format="csc")
q = rng.standard_normal(n + ns)
l = np.concatenate([-1.0 - rng.random(md), -0.5 * np.ones(150), -np.ones(n), np.zeros(ns)])
u = np.concatenate([1.0 + rng.random(md), 0.5 * np.ones(150), np.ones(n), np.zeros(ns)])
l[:20] = u[:20] = 0.1
pb = dict(P=sparse.triu(P, format="csc"), q=q, A=A, l=l, u=u)
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
for eps in (1e-5, 1e-7):
    kw = dict(eps_abs=eps, eps_rel=eps)
    ro = orc.OracleOSQP().setup(**pb, **kw).solve()
    for name, env in (("dense", dict(OSQP_AMD_DENSE_DIRECT="2", OSQP_AMD_RESIDENT="0")), ("pcg", dict(OSQP_AMD_DENSE_DIRECT="0", OSQP_AMD_RESIDENT="0")),
                      ("resident", dict(OSQP_AMD_DENSE_DIRECT="0", OSQP_AMD_RESIDENT="1")), ("dense, no elimination", dict(OSQP_AMD_DENSE_DIRECT="2", OSQP_AMD_RESIDENT="0", OSQP_AMD_ELIM="0"))):
        os.environ.update(env)
        s = osqp_amd.OSQP().setup(**pb, **kw)
        for k in env: os.environ.pop(k)
        r = s.solve()
        print("eps %.0e %-22s iter %d (oracle %d) x %.2e y %.2e obj %.2e pcg/admm %.1f" % (eps, name, r.info.iter, ro.info.iter, rel(r.x, ro.x), rel(r.y, ro.y),
              abs(r.info.obj_val - ro.info.obj_val), s.stats()["pcg_iters_total"] / max(1, r.info.iter)), flush=True)

# where does the engine's slack elimination leave the oracle's trajectory?  iterates after k iterations, elimination on
print("iterates after k ADMM iterations, elimination on (dense-direct), against the oracle:")
for k in (1, 2, 3, 10, 24, 25, 26, 50, 51, 75, 99):
    kw = dict(eps_abs=1e-5, eps_rel=1e-5, max_iter=k)
    ro = orc.OracleOSQP().setup(**pb, **kw).solve()
    os.environ.update(dict(OSQP_AMD_DENSE_DIRECT="2", OSQP_AMD_RESIDENT="0"))
    s = osqp_amd.OSQP().setup(**pb, **kw)
    r = s.solve()
    os.environ.update(dict(OSQP_AMD_ELIM="0"))
    s0 = osqp_amd.OSQP().setup(**pb, **kw)
    r0 = s0.solve()
    for kk in ("OSQP_AMD_DENSE_DIRECT", "OSQP_AMD_RESIDENT", "OSQP_AMD_ELIM"): os.environ.pop(kk)
    print("k=%3d: elim x %.2e y %.2e (slack part of x %.2e, rho updates %d/%d) | no elim x %.2e y %.2e" % (
        k, rel(r.x, ro.x), rel(r.y, ro.y), np.abs(r.x[n:] - ro.x[n:]).max(), r.info.rho_updates, ro.info.rho_updates, rel(r0.x, ro.x), rel(r0.y, ro.y)), flush=True)
