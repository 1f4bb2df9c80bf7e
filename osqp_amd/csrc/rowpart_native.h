// rowpart_native.h -- included at the end of engine.hip (it uses the engine's k_spmv, DevMat and allocation helpers).
//
// ONE QP over several GPUs by rows, the whole loop driven from C: include/osqp_amd_rowpart.h has the contract.  Per rank:
// the shard (P_g, A_g) in an ordinary engine, n-vectors replicated, m-vectors of the rank's rows.  All kernels and
// collectives go to the shard engine's stream; the PCG scalars live on the device (every rank computes the same ones),
// kernels of iterations issued past convergence return at once, and the host looks at one flag per group of iterations.
//
//   ADMM iteration (src/osqp.c:387-424):  w = rho z - y;  buf = A_g' w;  ALL-REDUCE(buf);  b = sigma x - q + buf
//                                         x~ = PCG(b, x~);  z~ = A_g x~;  update_x / update_z / update_y on the rank's rows
//   PCG iteration:                        t = rho (A_g p);  part = [P_g | A_g'] [p; t];  ALL-REDUCE(part);  Kp = part + sigma p
//                                         alpha = rz / p'Kp;  x~ += alpha p;  r -= alpha Kp;  z = Minv r;  p = z + (rz'/rz) p
//   termination check (src/auxil.c:681-740): six maxima (ALL-REDUCE max), [P x; A' y] (ALL-REDUCE of 2n), norms on the device.
#include <dlfcn.h>
#include "../../include/osqp_amd_rowpart.h"

#define RP_G 128                 // workgroups (= partials per dot product) of the n- and m-vector kernels
struct RpS { double rz[2]; double rr, tol2, bb; int done, iters, cap, bad; };

// sum of RP_G partials, the same order in every workgroup and on every rank
__device__ __forceinline__ double rp_total(const double *part, double *red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < RP_G; i += TB) s += part[i];
  return block_sum(s, red);
}
__global__ void __launch_bounds__(TB) k_rp_w(int m, const double *rho, const double *z, const double *y, double *out) {
  for (int i = blockIdx.x * TB + threadIdx.x; i < m; i += gridDim.x * TB) out[i] = rho[i] * z[i] - y[i];
}
__global__ void __launch_bounds__(TB) k_rp_scale(int m, const double *rho, double *t) {
  for (int i = blockIdx.x * TB + threadIdx.x; i < m; i += gridDim.x * TB) t[i] *= rho[i];
}
// b = sigma x - q + A'(rho z - y); partials of b'b
__global__ void __launch_bounds__(TB) k_rp_b(int n, double sigma, const double *x, const double *q, const double *buf, double *b, double *pbb) {
  __shared__ double red[16];
  double s = 0.0;
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) { const double v = sigma * x[j] - q[j] + buf[j]; b[j] = v; s += v * v; }
  s = block_sum(s, red);
  if (threadIdx.x == 0) pbb[blockIdx.x] = s;
}
// r = b - (part + sigma x~);  z = Minv r;  p = z;  partials of r'z and r'r
__global__ void __launch_bounds__(TB) k_rp_r0(int n, double sigma, const double *b, const double *part, const double *xt, const double *minv,
                                              double *r, double *zz, double *p, double *prz, double *prr) {
  __shared__ double red[16];
  double s0 = 0.0, s1 = 0.0;
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) {
    const double rj = b[j] - (part[j] + sigma * xt[j]), zj = minv[j] * rj;
    r[j] = rj; zz[j] = zj; p[j] = zj; s0 += rj * zj; s1 += rj * rj;
  }
  s0 = block_sum(s0, red); s1 = block_sum(s1, red);
  if (threadIdx.x == 0) { prz[blockIdx.x] = s0; prr[blockIdx.x] = s1; }
}
__global__ void __launch_bounds__(TB) k_rp_s0(RpS *S, const double *pbb, const double *prz, const double *prr, double eps, int cap) {
  __shared__ double red[16];
  const double bb = rp_total(pbb, red), rz = rp_total(prz, red), rr = rp_total(prr, red);
  if (threadIdx.x == 0) {
    S->bb = bb; S->rz[0] = rz; S->rr = rr; S->tol2 = fmax(eps * eps * bb, 1e-30);
    S->iters = 0; S->cap = cap; S->bad = 0; S->done = !(rr > S->tol2);
  }
}
// Kp = part + sigma p; partials of p'Kp
__global__ void __launch_bounds__(TB) k_rp_pkp(int n, double sigma, const RpS *S, const double *part, const double *p, double *Kp, double *ppkp) {
  if (S->done) return;
  __shared__ double red[16];
  double s = 0.0;
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) { const double v = part[j] + sigma * p[j]; Kp[j] = v; s += p[j] * v; }
  s = block_sum(s, red);
  if (threadIdx.x == 0) ppkp[blockIdx.x] = s;
}
__global__ void __launch_bounds__(TB) k_rp_upd(int n, int par, const RpS *S, const double *ppkp, const double *p, const double *Kp, const double *minv,
                                               double *xt, double *r, double *zz, double *prz, double *prr) {
  if (S->done) return;
  __shared__ double red[16];
  const double pkp = rp_total(ppkp, red);
  const double a = pkp > 0.0 ? S->rz[par] / pkp : 0.0;          // (p'Kp <= 0: K is not positive definite; k_rp_dir ends the solve and reports it)
  double s0 = 0.0, s1 = 0.0;
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) {
    xt[j] += a * p[j];
    const double rj = r[j] - a * Kp[j], zj = minv[j] * rj;
    r[j] = rj; zz[j] = zj; s0 += rj * zj; s1 += rj * rj;
  }
  s0 = block_sum(s0, red); s1 = block_sum(s1, red);
  if (threadIdx.x == 0) { prz[blockIdx.x] = s0; prr[blockIdx.x] = s1; }
}
// p = z + (rz'/rz) p; workgroup 0 closes the iteration (new r'z into the other slot: the others still read the old one)
__global__ void __launch_bounds__(TB) k_rp_dir(int n, int par, RpS *S, const double *ppkp, const double *prz, const double *prr, const double *zz, double *p) {
  if (S->done) return;
  __shared__ double red[16];
  const double rz2 = rp_total(prz, red), rr = rp_total(prr, red), pkp = rp_total(ppkp, red);
  const double beta = rz2 / S->rz[par];
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) p[j] = zz[j] + beta * p[j];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    S->rz[par ^ 1] = rz2; S->rr = rr;
    const int it = S->iters + 1;
    S->iters = it;
    if (!(pkp > 0.0)) S->bad = 1;
    if (!(rr > S->tol2) || it >= S->cap || !(pkp > 0.0)) S->done = 1;      // (written after every workgroup of this launch has passed its own test or not: both are fine, p is not used again)
  }
}
__global__ void __launch_bounds__(TB) k_rp_admm_x(int n, double alpha, const double *xt, double *x) {
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) x[j] = alpha * xt[j] + (1.0 - alpha) * x[j];
}
// update_z / update_y (src/auxil.c:190-225) on the rank's rows
__global__ void __launch_bounds__(TB) k_rp_admm_z(int m, double alpha, const double *zt, const double *rho, const double *l, const double *u, double *z, double *y) {
  for (int i = blockIdx.x * TB + threadIdx.x; i < m; i += gridDim.x * TB) {
    const double v = alpha * zt[i] + (1.0 - alpha) * z[i];
    const double zn = fmin(fmax(v + y[i] / rho[i], l[i]), u[i]);
    y[i] += rho[i] * (v - zn);
    z[i] = zn;
  }
}
// rho per row from its bounds (set_rho_vec, src/auxil.c:28-52)
__global__ void __launch_bounds__(TB) k_rp_rho(int m, double rho, const double *l, const double *u, double *rv) {
  for (int i = blockIdx.x * TB + threadIdx.x; i < m; i += gridDim.x * TB) {
    const double lo = l[i], hi = u[i];
    rv[i] = (lo < -1e26 && hi > 1e26) ? 1e-6 : ((hi - lo < 1e-4) ? 1e3 * rho : rho);
  }
}
// this rank's share of diag(P) + sum_i rho_i A_ij^2 (the Jacobi preconditioner before the all-reduce), from the rows of [P_g | A_g']
__global__ void __launch_bounds__(TB) k_rp_diag(DevMat M, int n, const double *rho, double *out) {
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) {
    double d = 0.0;
    for (int k = M.rowptr[j]; k < M.split[j]; ++k) if (M.col[k] == j) d += M.val[k];
    for (int k = M.split[j]; k < M.rowptr[j + 1]; ++k) { const double a = M.val[k]; d += rho[M.col[k] - n] * a * a; }
    out[j] = d;
  }
}
__global__ void __launch_bounds__(TB) k_rp_minv(int n, double sigma, const double *diag, double *minv) {
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) minv[j] = 1.0 / (diag[j] + sigma);
}
__device__ __forceinline__ double block_max(double v, double *red) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// primal side of a check: maxima over the rank's rows of |Einv (Ax - z)|, |Einv z|, |Einv Ax|, |Ax - z|, |z|, |Ax|  (one workgroup)
__global__ void __launch_bounds__(TB) k_rp_pri(int m, const double *ax, const double *z, const double *Einv, double *out6) {
  __shared__ double red[16];
  double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < m; i += TB) {
    const double a = ax[i], zi = z[i], e = Einv[i], d = a - zi;
    v[0] = fmax(v[0], fabs(e * d)); v[1] = fmax(v[1], fabs(e * zi)); v[2] = fmax(v[2], fabs(e * a));
    v[3] = fmax(v[3], fabs(d)); v[4] = fmax(v[4], fabs(zi)); v[5] = fmax(v[5], fabs(a));
  }
  for (int k = 0; k < 6; ++k) { const double t = block_max(v[k], red); if (threadIdx.x == 0) out6[k] = t; __syncthreads(); }
}
// dual side: both = [P x ; A' y] summed over the ranks.  out: dua_u, dua_s, q_u, q_s, Aty_u, Aty_s, Px_u, Px_s, x'(0.5 P x + q)  (one workgroup)
__global__ void __launch_bounds__(TB) k_rp_dua(int n, const double *both, const double *q, const double *Dinv, const double *x, double *out9) {
  __shared__ double red[16];
  double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, obj = 0.0;
  for (int j = threadIdx.x; j < n; j += TB) {
    const double px = both[j], aty = both[n + j], qj = q[j], di = Dinv[j], d = px + qj + aty;
    v[0] = fmax(v[0], fabs(di * d)); v[1] = fmax(v[1], fabs(d)); v[2] = fmax(v[2], fabs(di * qj)); v[3] = fmax(v[3], fabs(qj));
    v[4] = fmax(v[4], fabs(di * aty)); v[5] = fmax(v[5], fabs(aty)); v[6] = fmax(v[6], fabs(di * px)); v[7] = fmax(v[7], fabs(px));
    obj += x[j] * (0.5 * px + qj);
  }
  for (int k = 0; k < 8; ++k) { const double t = block_max(v[k], red); if (threadIdx.x == 0) out9[k] = t; __syncthreads(); }
  obj = block_sum(obj, red);
  if (threadIdx.x == 0) out9[8] = obj;
}
__global__ void __launch_bounds__(TB) k_rp_unscale(int n, const double *s, const double *v, double f, double *out) {
  for (int j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) out[j] = s[j] * v[j] * f;
}

// ---- RCCL through dlopen (no link-time dependency; a process that already loaded librccl gets that copy) ----
struct RpNcclId { char b[128]; };
struct RpRccl {
  void *lib = nullptr, *comm = nullptr;
  int (*GetUniqueId)(RpNcclId *) = nullptr;
  int (*CommInitRank)(void **, int, RpNcclId, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
};
static int rp_rccl_open(RpRccl &R) {
  if (R.lib) return 0;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char *nm : names) if ((R.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!R.lib) for (const char *nm : names) if ((R.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
  if (!R.lib) { fprintf(stderr, "osqp_amd: librccl.so could not be loaded (%s)\n", dlerror()); return HIPENG_ERR_HIP; }
  R.GetUniqueId = reinterpret_cast<int (*)(RpNcclId *)>(dlsym(R.lib, "ncclGetUniqueId"));
  R.CommInitRank = reinterpret_cast<int (*)(void **, int, RpNcclId, int)>(dlsym(R.lib, "ncclCommInitRank"));
  R.AllReduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(R.lib, "ncclAllReduce"));
  R.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(R.lib, "ncclCommDestroy"));
  if (!R.GetUniqueId || !R.CommInitRank || !R.AllReduce || !R.CommDestroy) { fprintf(stderr, "osqp_amd: librccl.so lacks the nccl entry points\n"); return HIPENG_ERR_HIP; }
  return 0;
}

struct osqp_amd_rp {
  hipeng *e = nullptr;
  int n = 0, m = 0, m_total = 0, world = 1, rank = 0, has_eq = 0;
  osqp_amd_rp_settings st{};
  osqp_amd_rp_allreduce_fn ar = nullptr; void *user = nullptr;
  RpRccl rccl;
  double c = 1.0, rho = 0.1;
  bool scaled = false;
  double *x, *xt, *q, *D, *Dinv, *minv, *r, *zz, *Kp, *b, *part, *both, *stage;      // n (both: 2n, stage: n + m)
  double *z, *y, *l, *u, *E, *Einv, *rv, *zt, *ax;                                   // m
  double *pa, *pb, *pc, *sc15;                                                      // RP_G partials x 3; 6 + 9 check scalars
  RpS *S = nullptr, *hS = nullptr;                                                  // device / pinned host
  double *h15 = nullptr;
  long long collectives = 0, pcg_iters = 0;
  int last_iters = 4, rho_updates = 0;
  double sc[15];
};

static int rp_allreduce(osqp_amd_rp *rp, double *buf, long long count, int op) {
  if (rp->world <= 1 && !rp->rccl.comm) return 0;
  rp->collectives++;
  if (rp->rccl.comm) return rp->rccl.AllReduce(buf, buf, (size_t)count, 8 /* ncclFloat64 */, op ? 2 /* ncclMax */ : 0 /* ncclSum */, rp->rccl.comm, rp->e->stream) ? HIPENG_ERR_HIP : 0;
  return rp->ar ? rp->ar(rp->user, buf, count, op, (void *)rp->e->stream) : HIPENG_ERR_ARG;
}
static inline dim3 rp_grid() { return dim3(RP_G); }
// part = [P_g | A_g'] [u ; rho (A_g u)] for the u that sits in stage[0, n)
static void rp_apply_local(osqp_amd_rp *rp) {
  hipeng *e = rp->e;
  const int n = rp->n, m = rp->m;
  if (m > 0) {
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), dim3(TB), 0, e->stream, e->c.A, (const double *)rp->stage, rp->stage + n, 0);
    hipLaunchKernelGGL(k_rp_scale, rp_grid(), dim3(TB), 0, e->stream, m, (const double *)rp->rv, rp->stage + n);
  }
  hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.M.nblk))), dim3(TB), 0, e->stream, e->c.M, (const double *)rp->stage, rp->part, 0);
}
static int rp_set_rho(osqp_amd_rp *rp, double rho) {
  hipeng *e = rp->e;
  rho = std::min(std::max(rho, 1e-6), 1e6);
  rp->rho = rho;
  if (rp->m > 0) hipLaunchKernelGGL(k_rp_rho, rp_grid(), dim3(TB), 0, e->stream, rp->m, rho, (const double *)rp->l, (const double *)rp->u, rp->rv);
  hipLaunchKernelGGL(k_rp_diag, rp_grid(), dim3(TB), 0, e->stream, e->c.M, rp->n, (const double *)rp->rv, rp->part);
  if (int rc = rp_allreduce(rp, rp->part, rp->n, 0)) return rc;
  hipLaunchKernelGGL(k_rp_minv, rp_grid(), dim3(TB), 0, e->stream, rp->n, rp->st.sigma, (const double *)rp->part, rp->minv);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" osqp_amd_rp *osqp_amd_rp_create(void *shard_engine, const double *q, const double *l_loc, const double *u_loc, const double *D, const double *E_loc,
                                           double c, int m_total, int has_eq_any, const osqp_amd_rp_settings *settings, int world, int rank,
                                           osqp_amd_rp_allreduce_fn allreduce, void *user) {
  hipeng *e = static_cast<hipeng *>(shard_engine);
  if (!e || !q || !D || !settings || world < 1 || rank < 0 || rank >= world || (e->m > 0 && (!l_loc || !u_loc || !E_loc))) return nullptr;
  if (hipSetDevice(e->device) != hipSuccess) return nullptr;
  osqp_amd_rp *rp = new osqp_amd_rp();
  rp->e = e; rp->n = e->n; rp->m = e->m; rp->m_total = m_total; rp->world = world; rp->rank = rank; rp->has_eq = has_eq_any;
  rp->st = *settings; rp->ar = allreduce; rp->user = user; rp->c = c;
  const size_t n = (size_t)e->n, m = (size_t)e->m;
  double **nv[] = {&rp->x, &rp->xt, &rp->q, &rp->D, &rp->Dinv, &rp->minv, &rp->r, &rp->zz, &rp->Kp, &rp->b, &rp->part};
  double **mv[] = {&rp->z, &rp->y, &rp->l, &rp->u, &rp->E, &rp->Einv, &rp->rv, &rp->zt, &rp->ax};
  bool bad = false;
  for (double **p : nv) bad = bad || dev_alloc(e, p, n);
  for (double **p : mv) bad = bad || dev_alloc(e, p, m);
  bad = bad || dev_alloc(e, &rp->both, 2 * n) || dev_alloc(e, &rp->stage, n + m) || dev_alloc(e, &rp->pa, (size_t)RP_G) || dev_alloc(e, &rp->pb, (size_t)RP_G) ||
        dev_alloc(e, &rp->pc, (size_t)RP_G) || dev_alloc(e, &rp->sc15, (size_t)16) || dev_alloc(e, &rp->S, (size_t)1);
  if (!bad) bad = hipHostMalloc((void **)&rp->hS, sizeof(RpS)) != hipSuccess || hipHostMalloc((void **)&rp->h15, 16 * sizeof(double)) != hipSuccess;
  if (bad) { delete rp; return nullptr; }
  std::vector<double> Dinv(n), Einv(m);
  bool scaled = c != 1.0;
  for (size_t j = 0; j < n; j++) { Dinv[j] = 1.0 / D[j]; scaled = scaled || D[j] != 1.0; }
  for (size_t i = 0; i < m; i++) { Einv[i] = 1.0 / E_loc[i]; scaled = scaled || E_loc[i] != 1.0; }
  rp->scaled = scaled;
  auto up = [&](double *dst, const double *src, size_t k) { return k == 0 || hipMemcpyAsync(dst, src, k * sizeof(double), hipMemcpyHostToDevice, e->stream) == hipSuccess; };
  bool ok = up(rp->q, q, n) && up(rp->D, D, n) && up(rp->Dinv, Dinv.data(), n) && up(rp->l, l_loc, m) && up(rp->u, u_loc, m) && up(rp->E, E_loc, m) && up(rp->Einv, Einv.data(), m);
  ok = ok && hipStreamSynchronize(e->stream) == hipSuccess;          // (the inverses are locals)
  if (!ok) { osqp_amd_rp_free(rp); return nullptr; }
  return rp;
}

extern "C" int osqp_amd_rp_rccl_unique_id(void *out, int cap_bytes) {
  if (!out || cap_bytes < 128) return -1;
  RpRccl R;
  if (rp_rccl_open(R)) return -1;
  RpNcclId id;
  if (R.GetUniqueId(&id)) return -1;
  memcpy(out, id.b, 128);
  return 128;
}
extern "C" int osqp_amd_rp_use_rccl(osqp_amd_rp *rp, const void *unique_id, int id_bytes) {
  if (rp && !unique_id && id_bytes == 0) {           // detach: back to the callback
    if (rp->rccl.comm) rp->rccl.CommDestroy(rp->rccl.comm);
    rp->rccl.comm = nullptr;
    return 0;
  }
  if (!rp || !unique_id || id_bytes != 128) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(rp->e->device));
  if (int rc = rp_rccl_open(rp->rccl)) return rc;
  RpNcclId id;
  memcpy(id.b, unique_id, 128);
  if (rp->rccl.CommInitRank(&rp->rccl.comm, rp->world, id, rp->rank)) { rp->rccl.comm = nullptr; return HIPENG_ERR_HIP; }
  return 0;
}

// one termination check: fills rp->sc (the fifteen scalars) and pri_res / dua_res
static int rp_check(osqp_amd_rp *rp, double *pri_res, double *dua_res) {
  hipeng *e = rp->e;
  const int n = rp->n, m = rp->m;
  HIPCHK(hipMemsetAsync(rp->sc15, 0, 16 * sizeof(double), e->stream));
  if (m > 0) {
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), dim3(TB), 0, e->stream, e->c.A, (const double *)rp->x, rp->ax, 0);
    hipLaunchKernelGGL(k_rp_pri, dim3(1), dim3(TB), 0, e->stream, m, (const double *)rp->ax, (const double *)rp->z, (const double *)rp->Einv, rp->sc15);
  }
  if (int rc = rp_allreduce(rp, rp->sc15, 6, 1)) return rc;
  HIPCHK(hipMemcpyAsync(rp->stage, rp->x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  if (m > 0) HIPCHK(hipMemcpyAsync(rp->stage + n, rp->y, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  const dim3 gm(std::min(MAX_PARTS, std::max(1, e->c.M.nblk)));
  hipLaunchKernelGGL(k_spmv, gm, dim3(TB), 0, e->stream, e->c.M, (const double *)rp->stage, rp->both, 1);            // P_g x
  hipLaunchKernelGGL(k_spmv, gm, dim3(TB), 0, e->stream, e->c.M, (const double *)rp->stage, rp->both + n, 2);        // A_g' y
  if (int rc = rp_allreduce(rp, rp->both, 2ll * n, 0)) return rc;
  hipLaunchKernelGGL(k_rp_dua, dim3(1), dim3(TB), 0, e->stream, n, (const double *)rp->both, (const double *)rp->q, (const double *)rp->Dinv, (const double *)rp->x, rp->sc15 + 6);
  HIPCHK(hipMemcpyAsync(rp->h15, rp->sc15, 15 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  memcpy(rp->sc, rp->h15, sizeof(rp->sc));
  const bool un = rp->scaled && !rp->st.scaled_termination;
  const double *s = rp->sc;             // 0 pri_u 1 z_u 2 Ax_u 3 pri_s 4 z_s 5 Ax_s | 6 dua_u 7 dua_s 8 q_u 9 q_s 10 Aty_u 11 Aty_s 12 Px_u 13 Px_s 14 obj
  *pri_res = rp->m_total == 0 ? 0.0 : (un ? s[0] : s[3]);
  *dua_res = un ? s[6] / rp->c : s[7];
  return 0;
}
static bool rp_terminated(const osqp_amd_rp *rp, double pri_res, double dua_res, bool approximate) {
  const bool un = rp->scaled && !rp->st.scaled_termination;
  const double *s = rp->sc, k = approximate ? 10.0 : 1.0, ea = k * rp->st.eps_abs, er = k * rp->st.eps_rel;
  const bool prim_ok = rp->m_total == 0 || pri_res < ea + er * (un ? std::max(s[1], s[2]) : std::max(s[4], s[5]));
  const double nrm = un ? std::max(s[8], std::max(s[10], s[12])) / rp->c : std::max(s[9], std::max(s[11], s[13]));
  return prim_ok && dua_res < ea + er * nrm;
}
static double rp_rho_estimate(const osqp_amd_rp *rp) {
  const double *s = rp->sc;
  const double pri = (rp->m_total ? s[3] : 0.0) / (std::max(s[4], s[5]) + 1e-30);
  const double dua = s[7] / (std::max(s[9], std::max(s[11], s[13])) + 1e-30);
  return std::min(std::max(rp->rho * std::sqrt(pri / dua), 1e-6), 1e6);
}

extern "C" int osqp_amd_rp_solve(osqp_amd_rp *rp, osqp_amd_rp_info *info) {
  if (!rp || !info) return HIPENG_ERR_ARG;
  hipeng *e = rp->e;
  HIPCHK(hipSetDevice(e->device));
  const osqp_amd_rp_settings &st = rp->st;
  const int n = rp->n, m = rp->m;
  const dim3 g = rp_grid(), tb(TB);
  const double ee = std::min(st.eps_abs > 0 ? st.eps_abs : st.eps_rel, st.eps_rel > 0 ? st.eps_rel : st.eps_abs);
  double eps_pcg = std::max(1e-13, std::min(st.pcg_eps_rel, 1e-5 * ee));            // the rule of osqp_solve (osqp_host.c)
  if (rp->has_eq) eps_pcg = std::max(1e-13, 1e-3 * eps_pcg);
  const int interval = st.adaptive_rho_interval ? st.adaptive_rho_interval : (st.check_termination ? 4 * st.check_termination : 100);
  const int cap = st.pcg_max_iter ? st.pcg_max_iter : std::max(20000, 10 * n);
  rp->pcg_iters = 0; rp->collectives = 0; rp->rho_updates = 0;
  if (int rc = rp_set_rho(rp, st.rho)) return rc;
  int status = 0, it = 0;
  bool checked = false;
  double pri_res = 0.0, dua_res = 0.0;
  for (it = 1; it <= st.max_iter; it++) {
    // right-hand side
    HIPCHK(hipMemsetAsync(rp->stage, 0, (size_t)n * sizeof(double), e->stream));
    if (m > 0) hipLaunchKernelGGL(k_rp_w, g, tb, 0, e->stream, m, (const double *)rp->rv, (const double *)rp->z, (const double *)rp->y, rp->stage + n);
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.M.nblk))), tb, 0, e->stream, e->c.M, (const double *)rp->stage, rp->part, 2);   // A_g' w
    if (int rc = rp_allreduce(rp, rp->part, n, 0)) return rc;
    hipLaunchKernelGGL(k_rp_b, g, tb, 0, e->stream, n, st.sigma, (const double *)rp->x, (const double *)rp->q, (const double *)rp->part, rp->b, rp->pa);
    // PCG from the previous x~
    HIPCHK(hipMemcpyAsync(rp->stage, rp->xt, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    rp_apply_local(rp);
    if (int rc = rp_allreduce(rp, rp->part, n, 0)) return rc;
    hipLaunchKernelGGL(k_rp_r0, g, tb, 0, e->stream, n, st.sigma, (const double *)rp->b, (const double *)rp->part, (const double *)rp->xt, (const double *)rp->minv,
                       rp->r, rp->zz, rp->stage, rp->pb, rp->pc);
    hipLaunchKernelGGL(k_rp_s0, dim3(1), tb, 0, e->stream, rp->S, (const double *)rp->pa, (const double *)rp->pb, (const double *)rp->pc, eps_pcg, cap);
    int issued = 0;
    for (;;) {
      // a group of iterations, then one look at the flag: the first group is as long as the previous solve was, the next ones two
      const int group = issued == 0 ? std::max(1, rp->last_iters) : 2;
      for (int k = 0; k < group; k++, issued++) {
        const int par = issued & 1;
        rp_apply_local(rp);                                   // p sits in stage[0, n)
        if (int rc = rp_allreduce(rp, rp->part, n, 0)) return rc;
        hipLaunchKernelGGL(k_rp_pkp, g, tb, 0, e->stream, n, st.sigma, (const RpS *)rp->S, (const double *)rp->part, (const double *)rp->stage, rp->Kp, rp->pa);
        hipLaunchKernelGGL(k_rp_upd, g, tb, 0, e->stream, n, par, (const RpS *)rp->S, (const double *)rp->pa, (const double *)rp->stage, (const double *)rp->Kp,
                           (const double *)rp->minv, rp->xt, rp->r, rp->zz, rp->pb, rp->pc);
        hipLaunchKernelGGL(k_rp_dir, g, tb, 0, e->stream, n, par, rp->S, (const double *)rp->pa, (const double *)rp->pb, (const double *)rp->pc, (const double *)rp->zz, rp->stage);
      }
      HIPCHK(hipMemcpyAsync(rp->hS, rp->S, sizeof(RpS), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      if (rp->hS->done || issued >= cap) break;
    }
    if (rp->hS->bad) { fprintf(stderr, "osqp_amd: row-partitioned PCG met p'Kp <= 0 (the problem is not convex)\n"); return HIPENG_ERR_ARG; }
    rp->last_iters = rp->hS->iters;
    rp->pcg_iters += rp->hS->iters;
    // x, z, y
    if (m > 0) hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), tb, 0, e->stream, e->c.A, (const double *)rp->xt, rp->zt, 0);
    hipLaunchKernelGGL(k_rp_admm_x, g, tb, 0, e->stream, n, st.alpha, (const double *)rp->xt, rp->x);
    if (m > 0) hipLaunchKernelGGL(k_rp_admm_z, g, tb, 0, e->stream, m, st.alpha, (const double *)rp->zt, (const double *)rp->rv, (const double *)rp->l, (const double *)rp->u, rp->z, rp->y);
    HIPCHK(hipGetLastError());
    checked = st.check_termination && it % st.check_termination == 0;
    if (checked) {
      if (int rc = rp_check(rp, &pri_res, &dua_res)) return rc;
      if (rp_terminated(rp, pri_res, dua_res, false)) { status = 1; break; }
    }
    if (st.adaptive_rho && it % interval == 0) {
      if (!checked) if (int rc = rp_check(rp, &pri_res, &dua_res)) return rc;
      const double nw = rp_rho_estimate(rp);
      if (nw > rp->rho * st.adaptive_rho_tolerance || nw < rp->rho / st.adaptive_rho_tolerance) {
        if (int rc = rp_set_rho(rp, nw)) return rc;
        rp->rho_updates++;
      }
    }
  }
  if (it > st.max_iter) it = st.max_iter;
  if (!checked) {
    if (int rc = rp_check(rp, &pri_res, &dua_res)) return rc;
    if (rp_terminated(rp, pri_res, dua_res, false)) status = 1;
  }
  if (!status) status = rp_terminated(rp, pri_res, dua_res, true) ? 2 : -2;
  info->status = status; info->iter = it; info->rho_updates = rp->rho_updates; info->pcg_iters = rp->pcg_iters; info->collectives = rp->collectives;
  info->obj_val = rp->sc[14] / rp->c; info->pri_res = pri_res; info->dua_res = dua_res; info->rho_estimate = rp_rho_estimate(rp);
  return 0;
}

extern "C" int osqp_amd_rp_get_solution(osqp_amd_rp *rp, double *x, double *y_loc) {
  if (!rp || !x) return HIPENG_ERR_ARG;
  hipeng *e = rp->e;
  HIPCHK(hipSetDevice(e->device));
  hipLaunchKernelGGL(k_rp_unscale, rp_grid(), dim3(TB), 0, e->stream, rp->n, (const double *)rp->D, (const double *)rp->x, 1.0, rp->part);
  HIPCHK(hipMemcpyAsync(x, rp->part, (size_t)rp->n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (rp->m > 0 && y_loc) {
    hipLaunchKernelGGL(k_rp_unscale, rp_grid(), dim3(TB), 0, e->stream, rp->m, (const double *)rp->E, (const double *)rp->y, 1.0 / rp->c, rp->zt);
    HIPCHK(hipMemcpyAsync(y_loc, rp->zt, (size_t)rp->m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" void osqp_amd_rp_free(osqp_amd_rp *rp) {
  if (!rp) return;
  if (rp->rccl.comm) rp->rccl.CommDestroy(rp->rccl.comm);
  if (rp->hS) (void)hipHostFree(rp->hS);
  if (rp->h15) (void)hipHostFree(rp->h15);
  delete rp;                 // (the device vectors belong to the shard engine's allocation list and go with it)
}
