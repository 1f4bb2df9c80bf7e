"""Tight ADMM tolerances on an ill-conditioned reduced system (dense rows, no scaling):
does the PCG stop (eps_rel) leave a residual floor above the requested accuracy?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import osqp_amd, oracle.oracle as orc
sys.argv = sys.argv[:1]
import tools.stress as st
rng = np.random.default_rng(7)
pb = st.make(rng, "longrows")
for eps in (1e-5, 1e-7, 1e-9):
    kw = dict(eps_abs=eps, eps_rel=eps, scaling=0, max_iter=4000)
    ro = orc.OracleOSQP().setup(**pb, **kw).solve()
    for rel in (None, 1e-13):
        if rel: osqp_amd.set_engine_options(pcg_eps_rel=rel)
        s = osqp_amd.OSQP().setup(**pb, **kw); rg = s.solve()
        osqp_amd.set_engine_options(pcg_eps_rel=1e-10)
        print("eps %.0e pcg_eps_rel %s: gpu %s it %d pri %.2e dua %.2e | cpu %s it %d pri %.2e dua %.2e | pcg/it %.0f forced %d" % (
            eps, rel or "default", rg.info.status, rg.info.iter, rg.info.pri_res, rg.info.dua_res, ro.info.status, ro.info.iter, ro.info.pri_res, ro.info.dua_res,
            s.stats()["pcg_iters_total"] / max(1, rg.info.iter), s.stats()["pcg_forced"]))
