"""Row-partitioned single-QP solve over G GPUs (BASELINE config 5 "1 -> 8 MI355X").

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" for rehearsals).  The reference has
no counterpart -- its linear solve is one thread (lin_sys/direct/qdldl/qdldl_interface.c:216) -- so the contract is
the survey's: the rows of A and the (block-diagonal) entries of P are sharded, the n-vectors are replicated, and the
only data-path traffic is

  * ONE all-reduce of an n-vector per PCG iteration:  K u = sigma u + sum_g [ P_g u + A_g' (rho_g . (A_g u)) ],
  * ONE all-reduce of an n-vector per ADMM iteration for the right-hand side  A'(rho z - y) = sum_g A_g'(.)_g,
  * two n-vectors and a handful of scalars at each termination check.

Because every rank holds the same x, r, p, ... after the all-reduce (a ring all-reduce leaves identical bits on every
rank), the PCG dot products are computed redundantly and need NO scalar collective, and all ranks take the same
decisions.  The m-vectors z, y, l, u, rho live only on the rank that owns their rows.

Algorithm = the reference's ADMM (src/osqp.c:354-532, src/auxil.c:161-225) in the scaled space of scale_data
(src/scaling.c:44-156) with the indirect KKT solve of engine.hip (Jacobi-PCG on P + sigma I + A' rho A), termination
and rho adaptation as src/auxil.c:13-74, 240-359, 681-740.  Scope: statuses solved / solved inaccurate / maximum
iterations reached (no infeasibility certificates, no polish in this variant).

The SpMVs -- the part that touches the matrices -- run in the HIP kernels of the rank's shard engine
(hipeng_spmv_dev); vectors are torch tensors in HBM and the O(n) vector arithmetic between collectives is
torch element-wise code: with a collective after every operator apply this variant is bound by all-reduce latency,
not by those passes.  Expected break-even: an all-reduce of 400 KB over xGMI costs ~20-30 us
against ~26 us for a whole single-GPU PCG iteration at config 5 -- the partition pays only when the local operator
apply is well above that, i.e. for P blocks / A shards of several hundred MB per GPU.
"""
import ctypes as C

import numpy as np
from scipy import sparse

RHO_MIN, RHO_MAX, RHO_TOL, RHO_EQ = 1e-6, 1e6, 1e-4, 1e3
INF_BOUND = 1e30 * 1e-4
DIV_TOL = 1e-30


def shard_rows(A, world):
    """Contiguous row ranges of A with about equal numbers of non-zeros."""
    Ar = sparse.csr_matrix(A)
    nnz = Ar.indptr
    cuts = [0]
    for g in range(1, world):
        cuts.append(int(np.searchsorted(nnz, nnz[-1] * g / world)))
    cuts.append(Ar.shape[0])
    return [(cuts[g], max(cuts[g], cuts[g + 1])) for g in range(world)]


def shard_triu(Pu, world):
    """Split the stored upper triangle of P by contiguous column ranges: P = sum_g sym(P_g)."""
    Pc = sparse.csc_matrix(Pu)
    n = Pc.shape[0]
    nnz = Pc.indptr
    cuts = [0] + [int(np.searchsorted(nnz, nnz[-1] * g / world)) for g in range(1, world)] + [n]
    out = []
    for g in range(world):
        j0, j1 = cuts[g], max(cuts[g], cuts[g + 1])
        part = sparse.csc_matrix(Pc[:, j0:j1])
        M = sparse.hstack([sparse.csc_matrix((n, j0)), part, sparse.csc_matrix((n, n - j1))], format="csc")
        out.append(sparse.triu(M, format="csc"))
    return out


class ScipyOps:
    """SpMV back end on the CPU (rehearsals of the collective logic with gloo): same interface as HipOps."""

    def __init__(self, Pu_g, A_g, device=None):
        import torch
        self.torch = torch
        self.P = (Pu_g + sparse.triu(Pu_g, 1).T).tocsr()
        self.A = sparse.csr_matrix(A_g)
        self.At = self.A.T.tocsr()
        self.device = torch.device("cpu")

    def vec(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).clone()

    def A_mul(self, x):
        return self.torch.from_numpy(self.A @ x.numpy())

    def At_mul(self, y):
        return self.torch.from_numpy(self.At @ y.numpy())

    def P_mul(self, x):
        return self.torch.from_numpy(self.P @ x.numpy())

    def diag_terms(self):
        return self.P.diagonal(), self.A.multiply(self.A).tocsc()


class HipOps:
    """SpMV back end on the rank's GPU: the shard (P_g, A_g) is resident in a HIP engine (set up with scaling = 0 on
    already scaled data) and the products run in its k_spmv kernels on device pointers."""

    def __init__(self, Pu_g, A_g, device=0):
        import torch
        import osqp_amd
        self.torch = torch
        self.device = torch.device("cuda", device)
        n, m = Pu_g.shape[0], A_g.shape[0]
        self.n, self.m = n, m
        self._A = sparse.csr_matrix(A_g)
        self._P = Pu_g
        # a shard with no rows still needs an engine for its part of P
        old = osqp_amd.engine_options()["device"]
        osqp_amd.set_engine_options(device=device)
        try:
            self.solver = osqp_amd.OSQP().setup(P=Pu_g, q=np.zeros(n), A=A_g, l=-np.ones(m), u=np.ones(m), scaling=0, sigma=1.0)
        finally:
            osqp_amd.set_engine_options(device=old)
        L = self.solver._lib
        L.hipeng_spmv_dev.restype = C.c_int
        L.hipeng_spmv_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.hipeng_sync.restype = C.c_int
        L.hipeng_sync.argtypes = [C.c_void_p]
        self._L, self._e = L, self.solver.engine()

    def vec(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(self.device)

    def _spmv(self, which, x, outlen):
        t = self.torch
        x = x.contiguous()
        y = t.zeros(max(outlen, 1), dtype=t.float64, device=self.device)
        t.cuda.current_stream(self.device).synchronize()          # x written by torch before the engine's stream reads it
        rc = self._L.hipeng_spmv_dev(self._e, which, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()))
        if rc or self._L.hipeng_sync(self._e):
            raise RuntimeError("hipeng_spmv_dev failed (%d)" % rc)
        return y[:outlen]

    def A_mul(self, x):
        return self._spmv(0, x, self.m)

    def At_mul(self, y):
        if self.m == 0:
            return self.torch.zeros(self.n, dtype=self.torch.float64, device=self.device)
        return self._spmv(1, y, self.n)

    def P_mul(self, x):
        return self._spmv(2, x, self.n)

    def diag_terms(self):
        Pf = self._P + sparse.triu(self._P, 1).T
        return Pf.diagonal(), self._A.multiply(self._A).tocsc()


def scaled_problem_from_handle(s):
    """The scaled problem (scale_data, src/scaling.c:44-156) read back from the mirrors of a set-up workspace:
    P (upper triangle), q, A, l, u and D, E, c."""
    w = s.work
    n, m = s.n, s.m

    def mat(cp, rows):
        c = cp.contents
        p = np.ctypeslib.as_array(c.p, shape=(n + 1,)).copy(); nnz = int(p[-1])
        i = np.ctypeslib.as_array(c.i, shape=(max(nnz, 1),))[:nnz].copy(); x = np.ctypeslib.as_array(c.x, shape=(max(nnz, 1),))[:nnz].copy()
        return sparse.csc_matrix((x, i, p), shape=(rows, n))
    d = w.data.contents
    out = dict(P=mat(d.P, n), A=mat(d.A, m), q=s._vec(d.q, n), l=s._vec(d.l, m), u=s._vec(d.u, m))
    if w.settings.contents.scaling:
        sc = w.scaling.contents
        out.update(D=s._vec(sc.D, n), E=s._vec(sc.E, m), c=float(sc.c))
    else:
        out.update(D=np.ones(n), E=np.ones(m), c=1.0)
    return out


def scaled_problem_from_engine(P, q, A, l, u, scaling=10, device=0):
    """scale_data as the single-GPU engine performs it (hipeng_ruiz_scale), so that the row-partitioned variant
    iterates in exactly the scaled space of the single-GPU path."""
    import osqp_amd
    old = osqp_amd.engine_options()["device"]
    osqp_amd.set_engine_options(device=device)
    try:
        s = osqp_amd.OSQP().setup(P=P, q=q, A=A, l=l, u=u, scaling=scaling)
    finally:
        osqp_amd.set_engine_options(device=old)
    out = scaled_problem_from_handle(s)
    s.cleanup()
    return out


class RowPartitionedOSQP:
    """osqp_setup / osqp_solve for ONE QP whose matrices are sharded over the ranks of a torch.distributed group."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collectives = 0

    # ---- collectives (device to device with nccl; staged through the host for a gloo rehearsal) ----
    def _allreduce(self, t, op="sum"):
        if self.world == 1:
            return t
        d = self.dist
        self.collectives += 1
        rop = d.ReduceOp.SUM if op == "sum" else d.ReduceOp.MAX
        if t.is_cuda and d.get_backend(self.group) != "nccl":
            h = t.cpu(); d.all_reduce(h, op=rop, group=self.group); t.copy_(h)
        else:
            d.all_reduce(t, op=rop, group=self.group)
        return t

    def setup(self, scaled, ops_factory, device=0, **settings):
        """`scaled`: dict(P (triu), q, A, l, u, D, E, c) -- the scaled problem, identical on every rank;
        `ops_factory(Pu_g, A_g, device)` builds the rank's SpMV back end (HipOps on a GPU)."""
        st = dict(rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, check_termination=25,
                  adaptive_rho=1, adaptive_rho_interval=0, adaptive_rho_tolerance=5.0, scaled_termination=0,
                  pcg_eps_rel=1e-9, pcg_max_iter=0)
        for k, v in settings.items():
            if k not in st:
                raise ValueError("unsupported setting %r in the row-partitioned variant" % k)
            st[k] = v
        self.st = st
        n, m = scaled["P"].shape[0], scaled["A"].shape[0]
        self.n, self.m = n, m
        self.rows = shard_rows(scaled["A"], self.world)
        r0, r1 = self.rows[self.rank]
        self.r0, self.r1 = r0, r1
        Pg = shard_triu(scaled["P"], self.world)[self.rank]
        Ag = sparse.csr_matrix(scaled["A"])[r0:r1]
        self.ops = ops_factory(Pg, Ag, device)
        o = self.ops
        self.q = o.vec(scaled["q"]); self.l = o.vec(scaled["l"][r0:r1]); self.u = o.vec(scaled["u"][r0:r1])
        self.D = o.vec(scaled["D"]); self.Dinv = 1.0 / self.D
        self.E = o.vec(scaled["E"][r0:r1]); self.Einv = 1.0 / self.E
        self.c = float(scaled["c"]); self.cinv = 1.0 / self.c
        self.scaled_data = bool(np.any(scaled["D"] != 1.0) or np.any(scaled["E"] != 1.0) or scaled["c"] != 1.0)
        lg, ug = scaled["l"][r0:r1], scaled["u"][r0:r1]
        self.cls = np.where((lg < -INF_BOUND) & (ug > INF_BOUND), -1, np.where(ug - lg < RHO_TOL, 1, 0))
        self.has_eq = bool(self._allreduce(o.vec([float((self.cls == 1).any())]), "max").item() > 0)
        pd, a2 = o.diag_terms()
        self._pdiag, self._a2 = pd, a2                   # this rank's share of diag(P) and of A.^2 (column sums weighted by rho)
        t = self.torch
        self.x = t.zeros(n, dtype=t.float64, device=o.device); self.xt = self.x.clone()
        self.z = t.zeros(r1 - r0, dtype=t.float64, device=o.device); self.y = self.z.clone()
        self._set_rho(st["rho"])
        self.rho_updates = 0
        return self

    def _set_rho(self, rho):
        rho = min(max(rho, RHO_MIN), RHO_MAX)
        self.rho = rho
        rv = np.where(self.cls == -1, RHO_MIN, np.where(self.cls == 1, RHO_EQ * rho, rho))
        self.rho_vec = self.ops.vec(rv)
        # Jacobi preconditioner: diag(P) + sigma + sum_i rho_i A_ij^2, the shard sums meet in one all-reduce
        local = self._pdiag + (np.asarray(self._a2.T @ rv).ravel() if self.m and rv.size else 0.0)
        diag = self._allreduce(self.ops.vec(local))
        self.minv = 1.0 / (diag + self.st["sigma"])

    # ---- operator and PCG -------------------------------------------------------------------------
    def _K(self, u):
        o = self.ops
        part = o.P_mul(u)
        if self.r1 > self.r0:
            part = part + o.At_mul(self.rho_vec * o.A_mul(u))
        return self._allreduce(part) + self.st["sigma"] * u       # the one n-vector all-reduce of a PCG iteration

    def _pcg(self, b, x0, eps_rel):
        n = self.n
        cap = self.st["pcg_max_iter"] or max(20000, 10 * n)
        x = x0.clone()
        r = b - self._K(x)
        tol2 = max(eps_rel * eps_rel * float(b @ b), 1e-30)
        z = self.minv * r
        p = z.clone()
        rz = float(r @ z)
        it = 0
        while float(r @ r) > tol2 and it < cap:
            Kp = self._K(p)
            a = rz / float(p @ Kp)
            x += a * p
            r -= a * Kp
            z = self.minv * r
            rz2 = float(r @ z)
            p = z + (rz2 / rz) * p
            rz = rz2
            it += 1
        self.pcg_iters += it
        return x

    # ---- residuals, termination, rho ---------------------------------------------------------------
    def _info(self):
        t, o = self.torch, self.ops
        Ax = o.A_mul(self.x) if self.r1 > self.r0 else self.z
        pri = Ax - self.z
        loc = t.stack([(self.Einv * pri).abs().max() if pri.numel() else t.zeros((), dtype=t.float64, device=o.device),
                       (self.Einv * self.z).abs().max() if pri.numel() else t.zeros((), dtype=t.float64, device=o.device),
                       (self.Einv * Ax).abs().max() if pri.numel() else t.zeros((), dtype=t.float64, device=o.device),
                       pri.abs().max() if pri.numel() else t.zeros((), dtype=t.float64, device=o.device),
                       self.z.abs().max() if pri.numel() else t.zeros((), dtype=t.float64, device=o.device),
                       Ax.abs().max() if pri.numel() else t.zeros((), dtype=t.float64, device=o.device)])
        pri_u, z_u, Ax_u, pri_s, z_s, Ax_s = (float(v) for v in self._allreduce(loc, "max"))
        both = t.cat([o.P_mul(self.x), o.At_mul(self.y) if self.r1 > self.r0 else t.zeros_like(self.x)])
        both = self._allreduce(both)
        Px, Aty = both[:self.n], both[self.n:]
        dua = Px + self.q + Aty
        f = lambda v, S=None: float((v if S is None else S * v).abs().max())
        s = dict(pri_u=pri_u, z_u=z_u, Ax_u=Ax_u, pri_s=pri_s, z_s=z_s, Ax_s=Ax_s,
                 dua_u=f(dua, self.Dinv), dua_s=f(dua), q_u=f(self.q, self.Dinv), q_s=f(self.q), Aty_u=f(Aty, self.Dinv), Aty_s=f(Aty),
                 Px_u=f(Px, self.Dinv), Px_s=f(Px), obj=float(self.x @ (0.5 * Px + self.q)) * self.cinv)
        self.sc = s
        un = self.scaled_data and not self.st["scaled_termination"]
        self.pri_res = 0.0 if self.m == 0 else (s["pri_u"] if un else s["pri_s"])
        self.dua_res = self.cinv * s["dua_u"] if un else s["dua_s"]
        return un

    def _terminated(self, approximate=False):
        un = self.scaled_data and not self.st["scaled_termination"]
        s, st = self.sc, self.st
        k = 10.0 if approximate else 1.0
        eps_abs, eps_rel = k * st["eps_abs"], k * st["eps_rel"]
        prim_ok = self.m == 0 or self.pri_res < eps_abs + eps_rel * (max(s["z_u"], s["Ax_u"]) if un else max(s["z_s"], s["Ax_s"]))
        nrm = self.cinv * max(s["q_u"], s["Aty_u"], s["Px_u"]) if un else max(s["q_s"], s["Aty_s"], s["Px_s"])
        dual_ok = self.dua_res < eps_abs + eps_rel * nrm
        return prim_ok and dual_ok

    def _rho_estimate(self):
        s = self.sc
        pri = (s["pri_s"] if self.m else 0.0) / (max(s["z_s"], s["Ax_s"]) + DIV_TOL)
        dua = s["dua_s"] / (max(s["q_s"], s["Aty_s"], s["Px_s"]) + DIV_TOL)
        return min(max(self.rho * np.sqrt(pri / dua), RHO_MIN), RHO_MAX)

    # ---- solve ---------------------------------------------------------------------------------------
    def solve(self):
        from types import SimpleNamespace
        st, o, t = self.st, self.ops, self.torch
        e = min(st["eps_abs"] or st["eps_rel"], st["eps_rel"] or st["eps_abs"])
        eps_pcg = max(1e-13, min(st["pcg_eps_rel"], 1e-5 * e))      # the rule of osqp_solve (osqp_host.c)
        if self.has_eq:
            eps_pcg = max(1e-13, 1e-3 * eps_pcg)
        interval = st["adaptive_rho_interval"] or (4 * st["check_termination"] if st["check_termination"] else 100)
        self.pcg_iters = 0
        alpha, sigma = st["alpha"], st["sigma"]
        status, it, checked = "unsolved", 0, False
        for it in range(1, st["max_iter"] + 1):
            w = self.rho_vec * self.z - self.y
            b = sigma * self.x - self.q + self._allreduce(o.At_mul(w) if self.r1 > self.r0 else t.zeros_like(self.x))
            self.xt = self._pcg(b, self.xt, eps_pcg)
            zt = o.A_mul(self.xt) if self.r1 > self.r0 else self.z
            self.x = alpha * self.xt + (1.0 - alpha) * self.x
            v = alpha * zt + (1.0 - alpha) * self.z
            zn = t.minimum(t.maximum(v + self.y / self.rho_vec, self.l), self.u)
            self.y = self.y + self.rho_vec * (v - zn)
            self.z = zn
            checked = bool(st["check_termination"]) and it % st["check_termination"] == 0
            if checked:
                self._info()
                if self._terminated():
                    status = "solved"
                    break
            if st["adaptive_rho"] and it % interval == 0:
                if not checked:
                    self._info()
                new = self._rho_estimate()
                if new > self.rho * st["adaptive_rho_tolerance"] or new < self.rho / st["adaptive_rho_tolerance"]:
                    self._set_rho(new)
                    self.rho_updates += 1
        if not checked:
            self._info()
            if self._terminated():
                status = "solved"
        if status == "unsolved":
            status = "solved inaccurate" if self._terminated(approximate=True) else "maximum iterations reached"
        # unscale; y is gathered from the row shards
        x = (self.D * self.x).cpu().numpy()
        yl = (self.E * self.y * self.cinv)
        if self.world > 1:
            per = max(r1 - r0 for r0, r1 in self.rows)
            pad = t.zeros(per, dtype=t.float64, device=yl.device); pad[:yl.numel()] = yl
            if pad.is_cuda and self.dist.get_backend(self.group) != "nccl":
                pad = pad.cpu()
            out = t.empty(self.world * per, dtype=t.float64, device=pad.device)
            self.dist.all_gather_into_tensor(out, pad, group=self.group)
            out = out.cpu().numpy()
            y = np.concatenate([out[g * per:g * per + (r1 - r0)] for g, (r0, r1) in enumerate(self.rows)])
        else:
            y = yl.cpu().numpy()
        info = SimpleNamespace(status=status, iter=it, obj_val=self.sc["obj"], pri_res=self.pri_res, dua_res=self.dua_res,
                               rho_updates=self.rho_updates, rho_estimate=self._rho_estimate(), pcg_iters=self.pcg_iters,
                               collectives=self.collectives)
        return SimpleNamespace(x=x, y=y, info=info)


# ------------------------------------------------------------------------------------------------------------------
# The same solve driven from C (include/osqp_amd_rowpart.h, csrc/rowpart_native.h): kernels, collectives and the PCG's
# stopping decisions are issued by osqp_amd_rp_solve on the shard engine's stream.  Python only shards the matrices,
# creates the engine and -- for a gloo rehearsal -- lends its process group as the collective.
# ------------------------------------------------------------------------------------------------------------------
class _RpSettings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "adaptive_rho_tolerance", "pcg_eps_rel")] + \
               [(k, C.c_int) for k in ("max_iter", "check_termination", "adaptive_rho", "adaptive_rho_interval", "scaled_termination", "pcg_max_iter")]


class _RpInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("rho_updates", C.c_int), ("pcg_iters", C.c_longlong), ("collectives", C.c_longlong),
                ("obj_val", C.c_double), ("pri_res", C.c_double), ("dua_res", C.c_double), ("rho_estimate", C.c_double)]


_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p)
_STATUS = {1: "solved", 2: "solved inaccurate", -2: "maximum iterations reached"}


class _DevView:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = dict(shape=(int(count),), typestr="<f8", data=(int(ptr), False), version=2, strides=None)


class NativeRowPartitionedOSQP:
    """osqp_setup / osqp_solve of ONE row-partitioned QP through the C entry points osqp_amd_rp_*.
    collective = "rccl": ncclAllReduce of librccl on the engine's stream (the production path, one GPU per rank);
                 "group": the torch.distributed group as a callback (any backend; gloo stages through the host);
                 "auto": rccl when the group's backend is nccl, else group."""

    def __init__(self, group=None, collective="auto", world=None):
        # world=1: no process group and no torch at all (a plain C caller's situation; tools/rccl_world1_probe.py)
        self.torch = self.dist = None
        self.group, self.world, self.rank = group, 1, 0
        if world != 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            if collective == "auto":
                collective = "rccl" if (dist.is_initialized() and dist.get_backend(group) == "nccl") else "group"
        self.collective = "group" if collective == "auto" else collective
        self._rp = None

    def _bind(self, L):
        L.osqp_amd_rp_create.restype = C.c_void_p
        L.osqp_amd_rp_create.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_double, C.c_int, C.c_int, C.POINTER(_RpSettings), C.c_int, C.c_int, _ALLREDUCE_FN, C.c_void_p]
        L.osqp_amd_rp_solve.restype = C.c_int; L.osqp_amd_rp_solve.argtypes = [C.c_void_p, C.POINTER(_RpInfo)]
        L.osqp_amd_rp_get_solution.restype = C.c_int; L.osqp_amd_rp_get_solution.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.osqp_amd_rp_free.restype = None; L.osqp_amd_rp_free.argtypes = [C.c_void_p]
        L.osqp_amd_rp_rccl_unique_id.restype = C.c_int; L.osqp_amd_rp_rccl_unique_id.argtypes = [C.c_void_p, C.c_int]
        L.osqp_amd_rp_use_rccl.restype = C.c_int; L.osqp_amd_rp_use_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hipeng_sync.restype = C.c_int; L.hipeng_sync.argtypes = [C.c_void_p]

    def _group_allreduce(self, user, buf, count, op, stream):
        # the callback form of the collective: everything queued on the engine's stream first, then the group's all-reduce on
        # the wrapped device memory, complete before this returns (osqp_amd_rowpart.h: "a synchronous one")
        try:
            t, d = self.torch, self.dist
            if self._L.hipeng_sync(self._e):
                return 1
            v = t.as_tensor(_DevView(buf, count), device=self.dev)
            rop = d.ReduceOp.MAX if op else d.ReduceOp.SUM
            if d.get_backend(self.group) == "nccl":
                d.all_reduce(v, op=rop, group=self.group)
            else:
                h = v.cpu(); d.all_reduce(h, op=rop, group=self.group); v.copy_(h)
            t.cuda.synchronize(self.dev)
            return 0
        except Exception as ex:                          # an exception must not unwind through the C frames
            import sys
            print("osqp_amd.rowpart: collective failed: %r" % (ex,), file=sys.stderr)
            return 1

    def setup(self, scaled, device=0, **settings):
        t, d = self.torch, self.dist
        st = dict(rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, check_termination=25,
                  adaptive_rho=1, adaptive_rho_interval=0, adaptive_rho_tolerance=5.0, scaled_termination=0,
                  pcg_eps_rel=1e-9, pcg_max_iter=0)
        for k, v in settings.items():
            if k not in st:
                raise ValueError("unsupported setting %r in the row-partitioned variant" % k)
            st[k] = v
        n, m = scaled["P"].shape[0], scaled["A"].shape[0]
        self.n, self.m = n, m
        self.rows = shard_rows(scaled["A"], self.world)
        r0, r1 = self.rows[self.rank]
        self.r0, self.r1 = r0, r1
        # the rank's shard (P_g, A_g) in an ordinary engine: scaling = 0 on already scaled data
        import osqp_amd
        Pg, Ag = shard_triu(scaled["P"], self.world)[self.rank], sparse.csr_matrix(scaled["A"])[r0:r1]
        old = osqp_amd.engine_options()["device"]
        osqp_amd.set_engine_options(device=device)
        try:
            self.shard = osqp_amd.OSQP().setup(P=Pg, q=np.zeros(n), A=Ag, l=-np.ones(r1 - r0), u=np.ones(r1 - r0), scaling=0, sigma=1.0)
        finally:
            osqp_amd.set_engine_options(device=old)
        self.dev = t.device("cuda", device) if t is not None else None
        L = self.shard._lib
        self._bind(L)
        self._L, self._e = L, self.shard.engine()
        lg, ug = np.ascontiguousarray(scaled["l"][r0:r1], dtype=np.float64), np.ascontiguousarray(scaled["u"][r0:r1], dtype=np.float64)
        has_eq = float(np.any(ug - lg < RHO_TOL))
        if self.world > 1:
            flag = t.tensor([has_eq], dtype=t.float64, device=self.dev if d.get_backend(self.group) == "nccl" else "cpu")
            d.all_reduce(flag, op=d.ReduceOp.MAX, group=self.group)
            has_eq = float(flag.item())
        kinds = dict(_RpSettings._fields_)
        s = _RpSettings(**{k: (float(v) if kinds[k] is C.c_double else int(v)) for k, v in st.items()})
        self._cb = _ALLREDUCE_FN(self._group_allreduce)              # kept alive with the object
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        q, D, E = f64(scaled["q"]), f64(scaled["D"]), f64(scaled["E"][r0:r1])
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
        self._rp = L.osqp_amd_rp_create(self._e, ptr(q), ptr(lg), ptr(ug), ptr(D), ptr(E), float(scaled["c"]), int(m), int(has_eq > 0),
                                        C.byref(s), self.world, self.rank, self._cb, None)
        if not self._rp:
            raise RuntimeError("osqp_amd_rp_create failed")
        if self.collective == "rccl" and self.world > 1:
            idb = t.zeros(128, dtype=t.uint8)
            if self.rank == 0:
                raw = (C.c_char * 128)()
                if L.osqp_amd_rp_rccl_unique_id(raw, 128) != 128:
                    raise RuntimeError("osqp_amd_rp_rccl_unique_id failed")
                idb = t.frombuffer(bytearray(raw.raw), dtype=t.uint8).clone()
            on = self.dev if d.get_backend(self.group) == "nccl" else "cpu"
            idb = idb.to(on)
            d.broadcast(idb, src=0, group=self.group)
            raw = bytes(idb.cpu().numpy().tobytes())
            bad = t.tensor([float(L.osqp_amd_rp_use_rccl(self._rp, raw, 128) != 0)], dtype=t.float64, device=on)
            d.all_reduce(bad, op=d.ReduceOp.MAX, group=self.group)
            if bad.item() > 0:                         # every rank takes the same decision: the group serves as the collective
                import sys
                print("osqp_amd.rowpart: RCCL could not be attached on some rank; the process group is the collective", file=sys.stderr)
                L.osqp_amd_rp_use_rccl(self._rp, None, 0)
                self.collective = "group"
        return self

    def use_rccl_world1(self):
        """Test hook: the built-in RCCL provider on a one-rank communicator (dlopen, ncclCommInitRank, the stream-ordered call)."""
        raw = (C.c_char * 128)()
        if self._L.osqp_amd_rp_rccl_unique_id(raw, 128) != 128:
            raise RuntimeError("osqp_amd_rp_rccl_unique_id failed")
        return self._L.osqp_amd_rp_use_rccl(self._rp, raw.raw, 128)

    def solve(self):
        from types import SimpleNamespace
        t, d = self.torch, self.dist
        info = _RpInfo()
        rc = self._L.osqp_amd_rp_solve(self._rp, C.byref(info))
        if rc:
            raise RuntimeError("osqp_amd_rp_solve failed (%d)" % rc)
        x = np.zeros(self.n); yl = np.zeros(max(self.r1 - self.r0, 1))
        if self._L.osqp_amd_rp_get_solution(self._rp, x.ctypes.data_as(C.c_void_p), yl.ctypes.data_as(C.c_void_p)):
            raise RuntimeError("osqp_amd_rp_get_solution failed")
        yl = yl[:self.r1 - self.r0]
        if self.world > 1:                                  # the duals are gathered from the row shards (once per solve, off the data path)
            per = max(b - a for a, b in self.rows)
            pad = t.zeros(per, dtype=t.float64); pad[:yl.size] = t.from_numpy(yl)
            if d.get_backend(self.group) == "nccl":
                pad = pad.to(self.dev)
            out = t.empty(self.world * per, dtype=t.float64, device=pad.device)
            d.all_gather_into_tensor(out, pad, group=self.group)
            out = out.cpu().numpy()
            y = np.concatenate([out[g * per:g * per + (b - a)] for g, (a, b) in enumerate(self.rows)])
        else:
            y = yl
        ns = SimpleNamespace(status=_STATUS.get(info.status, "unsolved"), iter=info.iter, obj_val=info.obj_val, pri_res=info.pri_res, dua_res=info.dua_res,
                             rho_updates=info.rho_updates, rho_estimate=info.rho_estimate, pcg_iters=info.pcg_iters, collectives=info.collectives)
        return SimpleNamespace(x=x, y=y, info=ns)

    def cleanup(self):
        if self._rp:
            self._L.osqp_amd_rp_free(self._rp)
            self._rp = None
