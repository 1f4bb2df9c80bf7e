"""The built-in RCCL provider of the row-partitioned solve (osqp_amd_rp_use_rccl: librccl through dlopen, ncclCommInitRank,
ncclAllReduce in place on the engine's stream) on a one-rank communicator, in a process WITHOUT torch -- the situation of a plain C
caller, one ROCm in the address space.  Prints the solve with the callback-free one-rank loop and with every collective of the loop going
through RCCL; the two must agree bit for bit.   usage: [NCCL_DEBUG=INFO] python tools/rccl_world1_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osqp_amd import rowpart
from osqp_amd.problems import portfolio_qp
assert "torch" not in sys.modules
pb = portfolio_qp(8, 40, sector_rows=6, seed=4)
n, m = pb["P"].shape[0], pb["A"].shape[0]
scaled = dict(pb, D=np.ones(n), E=np.ones(m), c=1.0)            # (no scaling: nothing of it needs another engine here)
kw = dict(eps_abs=1e-5, eps_rel=1e-5)
a = rowpart.NativeRowPartitionedOSQP(world=1).setup(scaled, device=0, **kw)
ra = a.solve()
b = rowpart.NativeRowPartitionedOSQP(world=1).setup(scaled, device=0, **kw)
rc = b.use_rccl_world1()
print("use_rccl ->", rc, flush=True)
rb = b.solve()
print("plain: %s iter %d pcg %d collectives %d" % (ra.info.status, ra.info.iter, ra.info.pcg_iters, ra.info.collectives))
print("rccl : %s iter %d pcg %d collectives %d" % (rb.info.status, rb.info.iter, rb.info.pcg_iters, rb.info.collectives))
same = bool(np.array_equal(ra.x, rb.x) and np.array_equal(ra.y, rb.y))
print("identical:", same, "torch loaded:", "torch" in sys.modules)
sys.exit(0 if (rc == 0 and same and rb.info.collectives > rb.info.pcg_iters) else 1)
