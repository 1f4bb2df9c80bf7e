"""Block-direct solve: right-hand-side form (OSQP_AMD_BLOCK_PLAIN=1) against the residual form (=0) on config 5, iterate by iterate."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_amd
from osqp_amd.problems import portfolio_qp
pb = portfolio_qp()
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config5_oracle.json")))
for k in (1, 2, 5, 50, 99, 100, 101, 110, 150, 325):
    res = {}
    for plain in ("0", "1"):
        os.environ["OSQP_AMD_BLOCK_PLAIN"] = plain
        s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100, max_iter=k)
        res[plain] = s.solve(); s.cleanup()
    a, b = res["0"], res["1"]
    print("k=%3d: |x1 - x0| = %.2e (max |x| %.2e), |y1 - y0| = %.2e, obj %.10f vs %.10f, rho updates %d/%d%s" % (
        k, np.abs(a.x - b.x).max(), np.abs(a.x).max(), np.abs(a.y - b.y).max(), a.info.obj_val, b.info.obj_val, a.info.rho_updates, b.info.rho_updates,
        ("; golden obj %.10f" % g["info"]["obj"]) if k == 325 else ""), flush=True)
