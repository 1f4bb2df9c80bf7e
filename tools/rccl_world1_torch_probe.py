"""The built-in RCCL provider inside a torch process whose NCCL backend is up (bench.py --gpus N's situation), one rank:
torch first, init_process_group("nccl"), one torch all-reduce, then osqp_amd_rp_use_rccl on the already loaded librccl.
usage: python tools/rccl_world1_torch_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", world_size=1, rank=0)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
from osqp_amd import rowpart
from osqp_amd.problems import portfolio_qp
pb = portfolio_qp(8, 40, sector_rows=6, seed=4)
scaled = rowpart.scaled_problem_from_engine(**pb)
kw = dict(eps_abs=1e-5, eps_rel=1e-5)
a = rowpart.NativeRowPartitionedOSQP(collective="group").setup(scaled, device=0, **kw)
ra = a.solve()
b = rowpart.NativeRowPartitionedOSQP(collective="group").setup(scaled, device=0, **kw)
rc = b.use_rccl_world1()
print("use_rccl ->", rc, flush=True)
ok = False
if rc == 0:
    rb = b.solve()
    ok = bool(np.array_equal(ra.x, rb.x) and np.array_equal(ra.y, rb.y))
    print("plain: iter %d pcg %d collectives %d; rccl: iter %d pcg %d collectives %d; identical: %s" % (
        ra.info.iter, ra.info.pcg_iters, ra.info.collectives, rb.info.iter, rb.info.pcg_iters, rb.info.collectives, ok))
dist.destroy_process_group()
sys.exit(0 if ok else 1)
