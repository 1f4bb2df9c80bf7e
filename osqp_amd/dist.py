"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The only path that shards is the batch of independent QPs: contiguous blocks of
ceil(B / world) problems per rank, no communication during the solves, and ONE
all_gather of the per-QP records [x | y | info8] at the end.
A single QP does not shard; `bench.py --gpus N` runs replicas."""
import numpy as np


def shard_range(batch, rank, world):
    per = (batch + world - 1) // world
    lo = min(batch, rank * per)
    hi = min(batch, lo + per)
    return lo, hi, per


def sharded_batch_solve(local_solve, Q, L, U, group=None, device=None):
    """local_solve(Q_shard, L_shard, U_shard) -> (X, Y, info8) numpy arrays for the
    rank's shard.  Returns the gathered (X, Y, info8) of the whole batch on every
    rank.  `device` = torch device for the gather buffers (cuda for nccl)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, n = Q.shape
    m = L.shape[1]
    lo, hi, per = shard_range(B, rank, world)
    X, Y, info = local_solve(Q[lo:hi], L[lo:hi], U[lo:hi])
    rec = np.zeros((per, n + m + 8))
    rec[:hi - lo, :n] = X
    rec[:hi - lo, n:n + m] = Y
    rec[:hi - lo, n + m:] = info
    if world == 1:
        full = rec
    else:
        t = torch.from_numpy(rec)
        if device is not None:
            t = t.to(device)
        out = torch.empty((world * per, n + m + 8), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        full = out.cpu().numpy()
    full = full[:B]
    return full[:, :n], full[:, n:n + m], full[:, n + m:]


def gather_batch_records(bs, per, world, coll_dev):
    """The one collective of the batch path, device to device: the rank's result arrays are wrapped where they sit in
    HBM (BatchOSQP.device_arrays), packed into `per` records [x | y | info8] by a device-side concatenation and
    gathered with ONE all_gather_into_tensor (RCCL over xGMI when `coll_dev` is a cuda device).  With a CPU collective
    device (gloo rehearsal) the packed records take one D2H copy first.  Returns the [world * per, n + m + 8] tensor."""
    import torch
    import torch.distributed as dist
    if getattr(bs, "_many", None) is not None:
        # members above the batch kernel's size (one single-QP engine each): the records are built from the host results
        r = bs.results()
        rec = torch.zeros((per, bs.n + bs.m + 8), dtype=torch.float64)
        rec[:bs.B, :bs.n] = torch.from_numpy(np.ascontiguousarray(r.x))
        if bs.m:
            rec[:bs.B, bs.n:bs.n + bs.m] = torch.from_numpy(np.ascontiguousarray(r.y))
        rec[:bs.B, bs.n + bs.m:] = torch.from_numpy(np.ascontiguousarray(r.info_raw))
        if coll_dev.type == "cuda":
            rec = rec.to(coll_dev)
    else:
        torch.cuda.synchronize()          # the batch engine writes the arrays on its own stream
        X, Y, I = (torch.as_tensor(a, device="cuda") for a in bs.device_arrays())
        rec = torch.zeros((per, bs.n + bs.m + 8), dtype=torch.float64, device=X.device)
        rec[:bs.B, :bs.n] = X
        if bs.m:
            rec[:bs.B, bs.n:bs.n + bs.m] = Y[:, :bs.m]
        rec[:bs.B, bs.n + bs.m:] = I
        if coll_dev.type != "cuda":
            rec = rec.cpu()
    out = torch.empty((world * per, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec)
    return out
