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
