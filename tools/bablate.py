"""Timing ablation of the batch kernel's per-iteration phases (results are garbage when phases are skipped)."""
import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch, osqp_amd
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(256)
    T = {}
    for its in (101, 501):
        bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U, warm_start=0, max_iter=its, check_termination=0, adaptive_rho=0)
        for _ in range(3): bs.solve(fetch=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): bs.solve(fetch=False)
        torch.cuda.synchronize(); T[its] = (time.perf_counter() - t0) / 20
    slope = (T[501] - T[101]) / 400
    print("ablate=%s: %.3f us per iteration, fixed %.1f us per solve" % (os.environ.get("OSQP_AMD_BATCH_ABLATE", "0"), slope * 1e6, (T[101] - 101 * slope) * 1e6))
else:
    for a in (0, 1, 2, 4, 8, 3, 7, 15):
        env = dict(os.environ, OSQP_AMD_BATCH_ABLATE=str(a))
        subprocess.run([sys.executable, __file__, "x"], env=env)
