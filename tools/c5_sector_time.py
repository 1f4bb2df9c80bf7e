"""Config 5 with and without 500 sparse sector rows: ADMM it/s of the block-direct solve (plain / coupled form) against the
launch-per-step kernels (OSQP_AMD_RESIDENT_BLOCKS=0).  usage: python tools/c5_sector_time.py [sector_rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_amd
from osqp_amd.problems import portfolio_qp
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for sr in (0, rows):
    pb = portfolio_qp(400, 125, sector_rows=sr, seed=1)
    ref = None
    for blocks in ("1", "0"):
        os.environ["OSQP_AMD_RESIDENT_BLOCKS"] = blocks
        t0 = time.perf_counter(); s = osqp_amd.OSQP().setup(**pb); t1 = time.perf_counter()
        r = s.solve()
        best = rbest = 1e9
        for _ in range(3):
            s.warm_start(x=np.zeros(pb["P"].shape[0]), y=np.zeros(pb["A"].shape[0]))
            t3 = time.perf_counter(); s.update_rho(0.1); rbest = min(rbest, time.perf_counter() - t3)
            t2 = time.perf_counter(); r = s.solve(); best = min(best, time.perf_counter() - t2)
        st = s.stats()
        if ref is None: ref = r
        print("sector rows %d, RESIDENT_BLOCKS=%s: setup %.3f s, %d iterations (%d rho updates) in %.2f ms = %.0f it/s; osqp_update_rho %.2f ms; %s; PCG iterations %d; x vs first %.2e" % (
            sr, blocks, t1 - t0, r.info.iter, r.info.rho_updates, 1e3 * best, r.info.iter / best, 1e3 * rbest, r.info.status, st["pcg_iters_total"],
            np.abs(r.x - ref.x).max()), flush=True)
        s.cleanup()
