"""One rank of the row-partitioned solve (started by tests/test_rowpart.py through osqp_amd.launch.spawn_ranks).
usage: _rowpart_worker.py <cpu|gpu|native> <problem> <out.npz>   (native: the loop driven from C, osqp_amd_rp_solve; the gloo group as its collective)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

mode, which, out = sys.argv[1], sys.argv[2], sys.argv[3]
dist.init_process_group("gloo")
from osqp_amd.problems import portfolio_qp, random_sparse_qp
from osqp_amd import rowpart

if which == "portfolio_small":
    pb, kw = portfolio_qp(8, 25, sector_rows=5, seed=3), dict(eps_abs=1e-5, eps_rel=1e-5)
elif which == "random":
    pb, kw = random_sparse_qp(300, 600, seed=5), {}
else:
    pb, kw = portfolio_qp(), dict(eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=100)
if mode == "cpu":     # CPU rehearsal of the collective logic: scipy SpMVs, scaling taken from the oracle's workspace (test infrastructure)
    import oracle.oracle as orc
    so = orc.OracleOSQP().setup(**pb)
    scaled = rowpart.scaled_problem_from_handle(so)
    ops = rowpart.ScipyOps
else:
    scaled = rowpart.scaled_problem_from_engine(**pb)
    ops = rowpart.HipOps
if mode == "native":
    r = rowpart.NativeRowPartitionedOSQP(collective="group").setup(scaled, device=0, **kw).solve()
else:
    r = rowpart.RowPartitionedOSQP().setup(scaled, ops, device=0, **kw).solve()
if dist.get_rank() == 0:
    np.savez(out, x=r.x, y=r.y, iter=r.info.iter, status=r.info.status, obj=r.info.obj_val, rho_updates=r.info.rho_updates,
             pcg_iters=r.info.pcg_iters, collectives=r.info.collectives, world=dist.get_world_size())
dist.barrier()
dist.destroy_process_group()
