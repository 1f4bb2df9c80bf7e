"""How tight must the PCG stop be for config 2?  Solves the full-size QP at several pcg_eps_rel and compares with the
CPU oracle's recorded run (tests/golden/config2_oracle.json): iterations, objective, x, y, residuals, time.
usage: for f in 1e-6 1e-5 1e-4; do OSQP_AMD_PCG_EPS_FACTOR=$f python tools/pcg_tol_sweep.py; done"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, osqp_amd
from osqp_amd.problems import random_sparse_qp
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config2_oracle.json")))
pb = random_sparse_qp()
xs, ys, gi = np.array(g["x_sub"]), np.array(g["y_sub"]), g["info"]
s = osqp_amd.OSQP().setup(**pb, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100, warm_start=0)
tol = float(os.environ.get("OSQP_AMD_PCG_EPS_FACTOR", "1e-6")) * 1e-4
s.set_options(pcg_eps_rel=1e-3)          # the cap; the stop in force is OSQP_AMD_PCG_EPS_FACTOR x eps (read once per process)
for _ in range(1):
    s.update_rho(0.1); s.solve()
    st0 = s.stats(); s.update_rho(0.1)
    t0 = time.perf_counter(); r = s.solve(); dt = time.perf_counter() - t0
    st = s.stats()
    print("pcg_eps_rel %.0e: iter %d (oracle %d) rho_updates %d  PCG/it %.2f  %.2f ms  obj rel %.1e  x %.1e  y %.1e  pri %.1e dua %.1e" % (
        tol, r.info.iter, gi["iters"], r.info.rho_updates, (st["pcg_iters_total"] - st0["pcg_iters_total"]) / r.info.iter, 1e3 * dt,
        abs(r.info.obj_val - gi["obj"]) / abs(gi["obj"]), np.abs(r.x[::10] - xs).max() / g["x_inf"], np.abs(r.y[::20] - ys).max() / g["y_inf"],
        abs(r.info.pri_res - gi["pri"]) / gi["pri"], abs(r.info.dua_res - gi["dua"]) / gi["dua"]))
