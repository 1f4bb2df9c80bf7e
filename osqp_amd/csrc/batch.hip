// batch.hip -- batched OSQP engine for many small QPs with one sparsity pattern
// (MPC-style: BASELINE config 4, 1024 x (n=120, m=240)), gfx950 / MI355X.
//
// One 256-thread workgroup solves one QP from raw data to unscaled solution in
// a single kernel launch; the grid is the batch.  Nothing leaves the CU during
// the ADMM loop:
//   * problem vectors and the (per-QP) scaled matrix values live in LDS;
//   * the KKT solve of update_xz_tilde (reference src/auxil.c:177-183) is a
//     direct one: K = P + sigma I + A' diag(rho) A (n x n, SPD) is formed and
//     inverted in place by Gauss-Jordan with the matrix held in REGISTERS,
//     tiled TILE x TILE over a 16 x 16 thread grid (n <= 16*TILE); every ADMM
//     iteration is then one register-tile GEMV (+ one step of iterative
//     refinement through the sparse operator) -- the per-QP analogue of the
//     reference's factor-once / solve-many LDL^T (qdldl_interface.c:341-376),
//     re-done on every rho update exactly like its re-factorisation (:396-410);
//   * Ruiz equilibration (src/scaling.c:44-156), rho classification
//     (src/auxil.c:76-98), the iteration (src/osqp.c:356-370), residuals and
//     termination incl. infeasibility tests (src/auxil.c:227-512, 681-786), rho
//     adaptation (src/auxil.c:13-74) and solution unscaling (src/scaling.c:177)
//     follow the reference statement by statement, per QP.
// Sparsity patterns (shared by the batch) are read from global memory and stay
// L1/L2 resident; per-QP data is read once and written once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <algorithm>
#include "../../include/osqp_amd.h"
#include "../../include/osqp_amd_batch.h"

#define BT 256
#define BINF 1e26

#define BCHK(call)                                                              \
  do {                                                                          \
    hipError_t _e = (call);                                                     \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "osqp_amd batch: HIP error %s at %s:%d (%s)\n",           \
              hipGetErrorString(_e), __FILE__, __LINE__, #call);                \
      return -102;                                                              \
    }                                                                           \
  } while (0)

struct BPattern {          // shared sparsity (device pointers)
  int n, m, nnzP, nnzA, nnzPf;
  const int *Pp, *Pi, *Pc;       // triu(P): col ptr, row idx, column of each entry
  const int *Fp, *Fi, *Fk;       // full symmetric P by columns: ptr, row, triu slot
  const int *Ap, *Ai, *Ac;       // A CSC: col ptr, row idx, column of each entry
  const int *Rp, *Rj, *Rk;       // A CSR: row ptr, col idx, CSC slot
};

struct BSettings {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_pinf, eps_dinf, rho_tol, adapt_tol;
  int scaling, adaptive_rho, rho_interval, max_iter, check_termination, scaled_termination,
      warm_start, refine;
};

struct BIO {               // per-batch arrays (device)
  const double *Px, *Ax;   // shared values, or per-QP values when strideP/strideA != 0
  long long strideP, strideA;
  const double *Q, *L, *U; // [B][n], [B][m], [B][m]
  double *Xs, *Zs, *Ys;    // scaled iterates kept between solves (warm start) [B][..]
  double *Xo, *Yo;         // unscaled solution out
  double *DXo, *DYo;       // certificates out
  double *rho_io;          // current rho per QP (persists between solves)
  double *info;            // [B][8]: iter, status, obj, pri, dua, rho_updates, rho_estimate, rho
};

// ---------------------------------------------------------------------------
// workgroup helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double b_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double b_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double b_sum(double v, double *red) {
  v = b_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double b_max(double v, double *red) {
  v = b_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__device__ __forceinline__ double clip_scale(double v) {
  if (v < 1e-4) v = 1.0;
  if (v > 1e4) v = 1e4;
  return v;
}

// LDS working set of one QP
struct BL {
  double *Pv, *Av;                                  // scaled matrix values (triu P, CSC A)
  double *q, *l, *u, *rho, *rinv, *x, *z, *y, *xt, *zt, *w, *dx, *dy, *D, *E, *tn, *tm, *b;
  double *rowk, *colk;                              // Gauss-Jordan exchange (2 x 2 x NP)
  double *red;
  int *ctype;
};

// y_i = sum_j A_ij v_j  (row gather through the CSR view of the CSC values)
__device__ __forceinline__ double a_row_dot(const BPattern &p, const double *Av, const double *v, int i) {
  double s = 0.0;
  for (int k = p.Rp[i]; k < p.Rp[i + 1]; ++k) s += Av[p.Rk[k]] * v[p.Rj[k]];
  return s;
}
// (A' v)_j  (column gather)
__device__ __forceinline__ double a_col_dot(const BPattern &p, const double *Av, const double *v, int j) {
  double s = 0.0;
  for (int k = p.Ap[j]; k < p.Ap[j + 1]; ++k) s += Av[k] * v[p.Ai[k]];
  return s;
}
// (P v)_j from the full symmetric pattern
__device__ __forceinline__ double p_row_dot(const BPattern &p, const double *Pv, const double *v, int j) {
  double s = 0.0;
  for (int k = p.Fp[j]; k < p.Fp[j + 1]; ++k) s += Pv[p.Fk[k]] * v[p.Fi[k]];
  return s;
}

// ---------------------------------------------------------------------------
// register-tiled K^-1: thread (tr, tc) of a 16 x 16 grid owns rows tr*T.. and
// columns tc*T.. of the (padded) NP x NP matrix, NP = 16*T.
// ---------------------------------------------------------------------------
template <int T>
__device__ void form_K(double (&a)[T][T], const BPattern &p, const BL &s, double sigma) {
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
#pragma unroll
  for (int r = 0; r < T; ++r)
#pragma unroll
    for (int c = 0; c < T; ++c) {
      const int i = tr * T + r, j = tc * T + c;
      double v = 0.0;
      if (i < p.n && j < p.n) {
        // P_ij from the upper triangle: column max(i,j), row min(i,j)
        const int cj = i > j ? i : j, ri = i > j ? j : i;
        for (int k = p.Pp[cj]; k < p.Pp[cj + 1]; ++k) if (p.Pi[k] == ri) v += s.Pv[k];
        if (i == j) v += sigma;
        // sum_t rho_t A_ti A_tj : merge of the two sorted columns
        int ka = p.Ap[i], kb = p.Ap[j];
        const int ea = p.Ap[i + 1], eb = p.Ap[j + 1];
        while (ka < ea && kb < eb) {
          const int ra = p.Ai[ka], rb = p.Ai[kb];
          if (ra == rb) { v += s.rho[ra] * s.Av[ka] * s.Av[kb]; ++ka; ++kb; }
          else if (ra < rb) ++ka; else ++kb;
        }
      } else if (i == j) v = 1.0;      // identity padding keeps the inverse well defined
      a[r][c] = v;
    }
}

// In-place Gauss-Jordan inversion without pivoting (K is SPD).  One barrier per
// pivot: the pivot row / column are exchanged through double-buffered LDS.
template <int T>
__device__ void invert_tiles(double (&a)[T][T], const BL &s) {
  constexpr int NP = 16 * T;
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
  for (int k = 0; k < NP; ++k) {
    double *rowk = s.rowk + (k & 1) * NP, *colk = s.colk + (k & 1) * NP;
    const int kb = k / T, ko = k % T;
    if (tr == kb) {
#pragma unroll
      for (int c = 0; c < T; ++c) {
        double v = 0;
#pragma unroll
        for (int r = 0; r < T; ++r) if (r == ko) v = a[r][c];
        rowk[tc * T + c] = v;
      }
    }
    if (tc == kb) {
#pragma unroll
      for (int r = 0; r < T; ++r) {
        double v = 0;
#pragma unroll
        for (int c = 0; c < T; ++c) if (c == ko) v = a[r][c];
        colk[tr * T + r] = v;
      }
    }
    __syncthreads();
    const double piv = 1.0 / rowk[k];
#pragma unroll
    for (int r = 0; r < T; ++r) {
      const int i = tr * T + r;
      const double ci = colk[i];
#pragma unroll
      for (int c = 0; c < T; ++c) {
        const int j = tc * T + c;
        const double rkj = (j == k) ? piv : rowk[j] * piv;
        if (i == k) a[r][c] = rkj;
        else a[r][c] = ((j == k) ? 0.0 : a[r][c]) - ci * rkj;
      }
    }
  }
  __syncthreads();
}

// out_i = sum_j Kinv_ij in_j ; in / out are LDS vectors of length >= NP
template <int T>
__device__ void tile_gemv(const double (&a)[T][T], const double *in, double *out) {
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
  double bj[T], acc[T];
#pragma unroll
  for (int c = 0; c < T; ++c) bj[c] = in[tc * T + c];
#pragma unroll
  for (int r = 0; r < T; ++r) {
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < T; ++c) v += a[r][c] * bj[c];
    acc[r] = v;
  }
  // fixed xor tree over the 16 lanes that share a row block
#pragma unroll
  for (int r = 0; r < T; ++r) {
    double v = acc[r];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    acc[r] = v;
  }
  __syncthreads();            // readers of `out`'s previous contents are done
  if (tc == 0) {
#pragma unroll
    for (int r = 0; r < T; ++r) out[tr * T + r] = acc[r];
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(BT) k_batch_solve(BPattern p, BSettings st, BIO io, int first_solve) {
  constexpr int NP = 16 * T;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int n = p.n, m = p.m, tid = threadIdx.x;
  const long long qp = blockIdx.x;
  BL s;
  {
    double *w = lds;
    s.Pv = w; w += p.nnzP; s.Av = w; w += p.nnzA;
    s.q = w; w += NP; s.x = w; w += NP; s.xt = w; w += NP; s.dx = w; w += NP; s.D = w; w += NP;
    s.tn = w; w += NP; s.b = w; w += NP;
    s.l = w; w += m; s.u = w; w += m; s.rho = w; w += m; s.rinv = w; w += m; s.z = w; w += m;
    s.y = w; w += m; s.zt = w; w += m; s.w = w; w += m; s.dy = w; w += m; s.E = w; w += m; s.tm = w; w += m;
    s.rowk = w; w += 2 * NP; s.colk = w; w += 2 * NP; s.red = w; w += 8;
    s.ctype = reinterpret_cast<int *>(w);
  }
  double a[T][T];

  // ---- load the problem -----------------------------------------------------
  const double *Pg = io.Px + qp * io.strideP, *Ag = io.Ax + qp * io.strideA;
  for (int k = tid; k < p.nnzP; k += BT) s.Pv[k] = Pg[k];
  for (int k = tid; k < p.nnzA; k += BT) s.Av[k] = Ag[k];
  for (int j = tid; j < NP; j += BT) {
    s.q[j] = j < n ? io.Q[qp * n + j] : 0.0;
    s.x[j] = 0.0; s.xt[j] = 0.0; s.dx[j] = 0.0; s.D[j] = 1.0; s.tn[j] = 0.0; s.b[j] = 0.0;
  }
  for (int i = tid; i < m; i += BT) {
    s.l[i] = io.L[qp * m + i]; s.u[i] = io.U[qp * m + i];
    s.z[i] = 0.0; s.y[i] = 0.0; s.E[i] = 1.0; s.dy[i] = 0.0; s.zt[i] = 0.0;
  }
  __syncthreads();

  // ---- Ruiz equilibration (scaling.c:44-156), per QP -------------------------
  double cs = 1.0;   // cost scaling c
  for (int pass = 0; pass < st.scaling; ++pass) {
    for (int j = tid; j < n; j += BT) {
      double v = 0.0;
      for (int k = p.Fp[j]; k < p.Fp[j + 1]; ++k) v = fmax(v, fabs(s.Pv[p.Fk[k]]));
      for (int k = p.Ap[j]; k < p.Ap[j + 1]; ++k) v = fmax(v, fabs(s.Av[k]));
      s.tn[j] = 1.0 / sqrt(clip_scale(v));
    }
    for (int i = tid; i < m; i += BT) {
      double v = 0.0;
      for (int k = p.Rp[i]; k < p.Rp[i + 1]; ++k) v = fmax(v, fabs(s.Av[p.Rk[k]]));
      s.tm[i] = 1.0 / sqrt(clip_scale(v));
    }
    __syncthreads();
    for (int k = tid; k < p.nnzP; k += BT) s.Pv[k] = (s.Pv[k] * s.tn[p.Pi[k]]) * s.tn[p.Pc[k]];
    for (int k = tid; k < p.nnzA; k += BT) s.Av[k] = (s.Av[k] * s.tm[p.Ai[k]]) * s.tn[p.Ac[k]];
    for (int j = tid; j < n; j += BT) { s.q[j] = s.q[j] * s.tn[j]; s.D[j] = s.tn[j] * s.D[j]; }
    for (int i = tid; i < m; i += BT) s.E[i] = s.tm[i] * s.E[i];
    __syncthreads();
    // cost normalisation: mean column norm of P (sequential sum, reference order) vs |q|_inf
    double cn = 0.0, qn = 0.0;
    for (int j = tid; j < n; j += BT) {
      double v = 0.0;
      for (int k = p.Fp[j]; k < p.Fp[j + 1]; ++k) v = fmax(v, fabs(s.Pv[p.Fk[k]]));
      s.tn[j] = v;
      qn = fmax(qn, fabs(s.q[j]));
    }
    qn = b_max(qn, s.red);
    if (tid == 0) { double acc = 0.0; for (int j = 0; j < n; ++j) acc += s.tn[j]; s.red[6] = acc / (double)n; }
    __syncthreads();
    cn = s.red[6];
    double ct = fmax(cn, clip_scale(qn));
    ct = 1.0 / clip_scale(ct);
    for (int k = tid; k < p.nnzP; k += BT) s.Pv[k] *= ct;
    for (int j = tid; j < n; j += BT) s.q[j] *= ct;
    cs *= ct;
    __syncthreads();
  }
  const double cinv = 1.0 / cs;
  const bool unscaled = st.scaling && !st.scaled_termination;
  for (int i = tid; i < m; i += BT) { s.l[i] = s.l[i] * s.E[i]; s.u[i] = s.u[i] * s.E[i]; }
  __syncthreads();

  // ---- rho vector (auxil.c:76-98) and warm start -----------------------------
  double rho = (first_solve || !io.rho_io) ? st.rho : io.rho_io[qp];
  rho = fmin(fmax(rho, 1e-6), 1e6);
  for (int i = tid; i < m; i += BT) {
    int t = 0;
    if (s.l[i] < -BINF && s.u[i] > BINF) t = -1;
    else if (s.u[i] - s.l[i] < st.rho_tol) t = 1;
    s.ctype[i] = t;
    const double r = t == -1 ? 1e-6 : (t == 1 ? 1e3 * rho : rho);
    s.rho[i] = r; s.rinv[i] = 1.0 / r;
  }
  if (st.warm_start && !first_solve) {
    for (int j = tid; j < n; j += BT) s.x[j] = io.Xs[qp * n + j];
    for (int i = tid; i < m; i += BT) { s.z[i] = io.Zs[qp * m + i]; s.y[i] = io.Ys[qp * m + i]; }
  }
  __syncthreads();
  form_K<T>(a, p, s, st.sigma);
  invert_tiles<T>(a, s);

  // ---- ADMM loop (osqp.c:354-532) ---------------------------------------------
  const double alpha = st.alpha, oma = 1.0 - st.alpha, sigma = st.sigma;
  int iter = 0, status = OSQP_UNSOLVED, rho_updates = 0;
  double pri_res = 0, dua_res = 0, obj = 0, rho_est = rho;
  // scaled norms of the last residual evaluation (for the rho estimate)
  double n_pri_s = 0, n_dua_s = 0, n_z_s = 0, n_ax_s = 0, n_q_s = 0, n_aty_s = 0, n_px_s = 0;
  // unscaled (or scaled, when no unscaling) norms for the tolerances
  double n_z = 0, n_ax = 0, n_q = 0, n_aty = 0, n_px = 0;

  auto evaluate = [&](bool approximate) -> bool {
    // ---- update_info: residuals and norms (auxil.c:227-318) ----
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
    for (int i = tid; i < m; i += BT) {
      const double ax = a_row_dot(p, s.Av, s.x, i);
      const double pr = ax + (-1.0) * s.z[i];
      const double ei = unscaled ? 1.0 / s.E[i] : 1.0;
      s.tm[i] = ax;
      m0 = fmax(m0, fabs(ei * pr)); m1 = fmax(m1, fabs(pr));
      m2 = fmax(m2, fabs(ei * s.z[i])); m3 = fmax(m3, fabs(s.z[i]));
      m4 = fmax(m4, fabs(ei * ax)); m5 = fmax(m5, fabs(ax));
    }
    pri_res = m == 0 ? 0.0 : b_max(m0, s.red); n_pri_s = b_max(m1, s.red);
    n_z = b_max(m2, s.red); n_z_s = b_max(m3, s.red); n_ax = b_max(m4, s.red); n_ax_s = b_max(m5, s.red);
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0, ob = 0;
    for (int j = tid; j < n; j += BT) {
      const double px = p_row_dot(p, s.Pv, s.x, j);
      const double aty = a_col_dot(p, s.Av, s.y, j);
      double dr = s.q[j] + px;
      if (m > 0) dr = dr + aty;
      const double di = unscaled ? 1.0 / s.D[j] : 1.0;
      d0 = fmax(d0, fabs(di * dr)); d1 = fmax(d1, fabs(dr));
      d2 = fmax(d2, fabs(di * s.q[j])); d3 = fmax(d3, fabs(s.q[j]));
      d4 = fmax(d4, fabs(di * aty)); d5 = fmax(d5, fabs(aty));
      d6 = fmax(d6, fabs(di * px)); d7 = fmax(d7, fabs(px));
      ob += s.x[j] * (0.5 * px + s.q[j]);
    }
    dua_res = b_max(d0, s.red); n_dua_s = b_max(d1, s.red);
    n_q = b_max(d2, s.red); n_q_s = b_max(d3, s.red); n_aty = b_max(d4, s.red); n_aty_s = b_max(d5, s.red);
    n_px = b_max(d6, s.red); n_px_s = b_max(d7, s.red);
    obj = b_sum(ob, s.red) * (st.scaling ? cinv : 1.0);
    if (unscaled) { dua_res *= cinv; n_q *= cinv; n_aty *= cinv; n_px *= cinv; }
    else { pri_res = m == 0 ? 0.0 : n_pri_s; }

    // ---- check_termination (auxil.c:681-786) ----
    if (pri_res > 1e30 || dua_res > 1e30) { status = OSQP_NON_CVX; obj = OSQP_NAN; return true; }
    double ea = st.eps_abs, er = st.eps_rel, epi = st.eps_pinf, edi = st.eps_dinf;
    if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
    bool prim_ok = false, dual_ok = false, pinf = false, dinf = false;
    if (m == 0) prim_ok = true;
    else if (pri_res < ea + er * fmax(n_z, n_ax)) prim_ok = true;
    else {
      // is_primal_infeasible (auxil.c:361-424); dy projected into tm
      double nd = 0, lhs = 0;
      for (int i = tid; i < m; i += BT) {
        double dy = s.dy[i];
        if (s.u[i] > BINF) { if (s.l[i] < -BINF) dy = 0.0; else dy = fmin(dy, 0.0); }
        else if (s.l[i] < -BINF) dy = fmax(dy, 0.0);
        s.w[i] = dy;
        nd = fmax(nd, fabs(unscaled ? s.E[i] * dy : dy));
        lhs += s.u[i] * fmax(dy, 0.0) + s.l[i] * fmin(dy, 0.0);
      }
      nd = b_max(nd, s.red); lhs = b_sum(lhs, s.red);
      if (nd > 1e-30 && lhs < epi * nd) {
        double mx = 0;
        for (int j = tid; j < n; j += BT) {
          double v = a_col_dot(p, s.Av, s.w, j);
          if (unscaled) v = v / s.D[j];
          mx = fmax(mx, fabs(v));
        }
        mx = b_max(mx, s.red);
        pinf = mx < epi * nd;
      }
    }
    if (dua_res < ea + er * fmax(fmax(n_q, n_aty), n_px)) dual_ok = true;
    else {
      // is_dual_infeasible (auxil.c:426-512)
      double ndx = 0, qdx = 0;
      for (int j = tid; j < n; j += BT) {
        ndx = fmax(ndx, fabs(unscaled ? s.D[j] * s.dx[j] : s.dx[j]));
        qdx += s.q[j] * s.dx[j];
      }
      ndx = b_max(ndx, s.red); qdx = b_sum(qdx, s.red);
      const double csc_ = unscaled ? cs : 1.0;
      if (ndx > 1e-30 && qdx < csc_ * edi * ndx) {
        double mx = 0;
        for (int j = tid; j < n; j += BT) {
          double v = p_row_dot(p, s.Pv, s.dx, j);
          if (unscaled) v = v / s.D[j];
          mx = fmax(mx, fabs(v));
        }
        mx = b_max(mx, s.red);
        if (mx < csc_ * edi * ndx) {
          double viol = 0;
          for (int i = tid; i < m; i += BT) {
            double v = a_row_dot(p, s.Av, s.dx, i);
            if (unscaled) v = v / s.E[i];
            if ((s.u[i] < BINF && v > edi * ndx) || (s.l[i] > -BINF && v < -edi * ndx)) viol += 1.0;
          }
          viol = b_sum(viol, s.red);
          dinf = viol == 0.0;
        }
      }
    }
    if (prim_ok && dual_ok) { status = approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED; return true; }
    if (pinf) { status = approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE; obj = OSQP_INFTY; return true; }
    if (dinf) { status = approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE; obj = -OSQP_INFTY; return true; }
    return false;
  };

  auto rho_estimate = [&]() -> double {     // auxil.c:13-52
    const double pr = (m ? n_pri_s : 0.0) / (fmax(n_z_s, n_ax_s) + 1e-30);
    const double du = n_dua_s / (fmax(fmax(n_q_s, n_aty_s), n_px_s) + 1e-30);
    return fmin(fmax(rho * sqrt(pr / du), 1e-6), 1e6);
  };

  bool checked = false;
  for (iter = 1; iter <= st.max_iter; ++iter) {
    // rhs of the reduced system: b = sigma x - q + A'(rho z - y)
    for (int i = tid; i < m; i += BT) s.w[i] = s.rho[i] * s.z[i] - s.y[i];
    __syncthreads();
    for (int j = tid; j < NP; j += BT)
      s.b[j] = j < n ? (sigma * s.x[j] - s.q[j]) + a_col_dot(p, s.Av, s.w, j) : 0.0;
    __syncthreads();
    tile_gemv<T>(a, s.b, s.xt);
    for (int r = 0; r < st.refine; ++r) {   // xt += Kinv (b - K xt)
      for (int i = tid; i < m; i += BT) s.w[i] = s.rho[i] * a_row_dot(p, s.Av, s.xt, i);
      __syncthreads();
      for (int j = tid; j < NP; j += BT)
        s.tn[j] = j < n ? s.b[j] - (p_row_dot(p, s.Pv, s.xt, j) + sigma * s.xt[j] + a_col_dot(p, s.Av, s.w, j)) : 0.0;
      __syncthreads();
      tile_gemv<T>(a, s.tn, s.dx);          // dx is free until the x update below
      for (int j = tid; j < n; j += BT) s.xt[j] += s.dx[j];
      __syncthreads();
    }
    // z~ = A x~ ; x, z, y updates (auxil.c:185-225, proj.c:4-14)
    for (int i = tid; i < m; i += BT) {
      const double zt = a_row_dot(p, s.Av, s.xt, i);
      const double zo = s.z[i], yo = s.y[i];
      double v = alpha * zt + oma * zo + s.rinv[i] * yo;
      v = fmax(v, s.l[i]);
      const double zn = fmin(v, s.u[i]);
      const double dy = s.rho[i] * (alpha * zt + oma * zo - zn);
      s.z[i] = zn; s.dy[i] = dy; s.y[i] = yo + dy;
    }
    for (int j = tid; j < n; j += BT) {
      const double xo = s.x[j];
      const double xn = alpha * s.xt[j] + oma * xo;
      s.dx[j] = xn - xo; s.x[j] = xn;
    }
    __syncthreads();

    checked = st.check_termination && (iter % st.check_termination == 0);
    bool fresh = false;
    if (checked) { fresh = true; if (evaluate(false)) break; }
    if (st.adaptive_rho && st.rho_interval && (iter % st.rho_interval == 0)) {
      if (!fresh) { const int keep = status; evaluate(false); status = keep; }
      const double rn = rho_estimate();
      rho_est = rn;
      if (rn > rho * st.adapt_tol || rn < rho / st.adapt_tol) {
        rho = rn; rho_updates++;
        __syncthreads();
        for (int i = tid; i < m; i += BT) {
          const int t = s.ctype[i];
          if (t == 0) { s.rho[i] = rho; s.rinv[i] = 1.0 / rho; }
          else if (t == 1) { s.rho[i] = 1e3 * rho; s.rinv[i] = 1.0 / s.rho[i]; }
        }
        __syncthreads();
        form_K<T>(a, p, s, sigma);
        invert_tiles<T>(a, s);
      }
    }
  }
  if (iter > st.max_iter) iter = st.max_iter;
  if (!checked) evaluate(false);
  if (status == OSQP_UNSOLVED) { if (!evaluate(true)) status = OSQP_MAX_ITER_REACHED; }
  rho_est = rho_estimate();

  // ---- store_solution (auxil.c:524-562) ---------------------------------------
  const bool has_sol = !(status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE ||
                         status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE ||
                         status == OSQP_NON_CVX);
  __syncthreads();
  if (has_sol) {
    for (int j = tid; j < n; j += BT) {
      io.Xo[qp * n + j] = st.scaling ? s.x[j] * s.D[j] : s.x[j];
      io.Xs[qp * n + j] = s.x[j];
    }
    for (int i = tid; i < m; i += BT) {
      io.Yo[qp * m + i] = st.scaling ? (s.y[i] * s.E[i]) * cinv : s.y[i];
      io.Ys[qp * m + i] = s.y[i]; io.Zs[qp * m + i] = s.z[i];
    }
  } else {
    for (int j = tid; j < n; j += BT) { io.Xo[qp * n + j] = OSQP_NAN; io.Xs[qp * n + j] = 0.0; }
    for (int i = tid; i < m; i += BT) { io.Yo[qp * m + i] = OSQP_NAN; io.Ys[qp * m + i] = 0.0; io.Zs[qp * m + i] = 0.0; }
    if (status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE) {
      double mx = 0;
      for (int i = tid; i < m; i += BT) { s.w[i] = unscaled ? s.w[i] * s.E[i] : s.w[i]; mx = fmax(mx, fabs(s.w[i])); }
      mx = b_max(mx, s.red);
      for (int i = tid; i < m; i += BT) io.DYo[qp * m + i] = s.w[i] * (1.0 / mx);
    }
    if (status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE) {
      double mx = 0;
      for (int j = tid; j < n; j += BT) { s.tn[j] = unscaled ? s.dx[j] * s.D[j] : s.dx[j]; mx = fmax(mx, fabs(s.tn[j])); }
      mx = b_max(mx, s.red);
      for (int j = tid; j < n; j += BT) io.DXo[qp * n + j] = s.tn[j] * (1.0 / mx);
    }
  }
  if (tid == 0) {
    double *inf = io.info + qp * 8;
    inf[0] = iter; inf[1] = status; inf[2] = obj; inf[3] = pri_res; inf[4] = dua_res;
    inf[5] = rho_updates; inf[6] = rho_est; inf[7] = rho;
    if (io.rho_io) io.rho_io[qp] = rho;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct osqp_amd_batch {
  int device = 0, tile = 8;
  long long B = 0;
  int n = 0, m = 0, nnzP = 0, nnzA = 0;
  hipStream_t stream = nullptr;
  BPattern pat{};
  BSettings st{};
  BIO io{};
  std::vector<void *> allocs;
  size_t lds_bytes = 0;
  int solves = 0;
  std::vector<double> h_info;
};

template <typename Tp>
static int balloc(osqp_amd_batch *b, Tp **p, size_t cnt) {
  void *q = nullptr;
  if (!cnt) cnt = 1;
  BCHK(hipMalloc(&q, cnt * sizeof(Tp)));
  BCHK(hipMemsetAsync(q, 0, cnt * sizeof(Tp), b->stream));
  b->allocs.push_back(q);
  *p = static_cast<Tp *>(q);
  return 0;
}
template <typename Tp>
static int bupload(osqp_amd_batch *b, const Tp **dst, const std::vector<Tp> &src) {
  Tp *d = nullptr;
  if (balloc(b, &d, src.size())) return -102;
  if (!src.empty()) BCHK(hipMemcpyAsync(d, src.data(), src.size() * sizeof(Tp), hipMemcpyHostToDevice, b->stream));
  *dst = d;
  return 0;
}

static void fill_settings(osqp_amd_batch *b, const OSQPSettings *s) {
  BSettings &t = b->st;
  t.rho = s->rho; t.sigma = s->sigma; t.alpha = s->alpha; t.eps_abs = s->eps_abs; t.eps_rel = s->eps_rel;
  t.eps_pinf = s->eps_prim_inf; t.eps_dinf = s->eps_dual_inf; t.rho_tol = RHO_TOL;
  t.adapt_tol = s->adaptive_rho_tolerance;
  t.scaling = (int)s->scaling; t.adaptive_rho = (int)s->adaptive_rho;
  t.rho_interval = (int)s->adaptive_rho_interval;
  if (t.adaptive_rho && !t.rho_interval)   // deterministic stand-in for the timing rule (osqp.c:267-279)
    t.rho_interval = s->check_termination ? 4 * (int)s->check_termination : 100;
  t.max_iter = (int)s->max_iter; t.check_termination = (int)s->check_termination;
  t.scaled_termination = (int)s->scaled_termination; t.warm_start = (int)s->warm_start;
  const char *e = getenv("OSQP_AMD_BATCH_REFINE");
  t.refine = e ? atoi(e) : 1;
}

extern "C" c_int osqp_amd_batch_setup(osqp_amd_batch **out, c_int batch, const csc *P, const csc *A,
                                     const c_float *Px_all, const c_float *Ax_all,
                                     const c_float *Q, const c_float *L, const c_float *U,
                                     const OSQPSettings *settings, c_int device) {
  if (!out || !P || !A || !Q || !settings || batch <= 0) return OSQP_DATA_VALIDATION_ERROR;
  *out = nullptr;
  const int n = (int)P->n, m = (int)A->m;
  if (P->m != P->n || A->n != P->n || n <= 0 || (m > 0 && (!L || !U))) return OSQP_DATA_VALIDATION_ERROR;
  for (c_int j = 0; j < n; j++)
    for (c_int k = P->p[j]; k < P->p[j + 1]; k++) if (P->i[k] > j) return OSQP_DATA_VALIDATION_ERROR;
  if (settings->adaptive_rho_tolerance < 1.0) return OSQP_SETTINGS_VALIDATION_ERROR;
  if (settings->rho <= 0 || settings->sigma <= 0 || settings->alpha <= 0 || settings->alpha >= 2 ||
      settings->max_iter <= 0 || settings->scaling < 0 || settings->check_termination < 0)
    return OSQP_SETTINGS_VALIDATION_ERROR;
  if (n > 128) {
    fprintf(stderr, "osqp_amd batch: n = %d > 128 is not supported by the register-tiled engine; "
                    "use one osqp_setup workspace per QP (one-QP-per-stream)\n", n);
    return OSQP_LINSYS_SOLVER_INIT_ERROR;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "osqp_amd batch: no HIP device available -- there is no CPU fallback\n");
    return OSQP_LINSYS_SOLVER_LOAD_ERROR;
  }
  osqp_amd_batch *b = new (std::nothrow) osqp_amd_batch();
  if (!b) return OSQP_MEM_ALLOC_ERROR;
  b->device = (int)device; b->B = batch; b->n = n; b->m = m;
  b->nnzP = (int)P->p[n]; b->nnzA = (int)A->p[n];
  b->tile = n <= 64 ? 4 : 8;
  if (hipSetDevice(b->device) != hipSuccess || hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
    delete b; return OSQP_LINSYS_SOLVER_LOAD_ERROR;
  }
  fill_settings(b, settings);

  // ---- shared patterns -------------------------------------------------------
  std::vector<int> Pp(n + 1), Pi(b->nnzP), Pc(b->nnzP), Ap(n + 1), Ai(b->nnzA), Ac(b->nnzA);
  for (int j = 0; j <= n; j++) { Pp[j] = (int)P->p[j]; Ap[j] = (int)A->p[j]; }
  for (int j = 0; j < n; j++) {
    for (int k = Pp[j]; k < Pp[j + 1]; k++) { Pi[k] = (int)P->i[k]; Pc[k] = j; }
    for (int k = Ap[j]; k < Ap[j + 1]; k++) { Ai[k] = (int)A->i[k]; Ac[k] = j; }
  }
  // full symmetric pattern of P by columns, each column in the reference's
  // summation order: rows >= ... (upper part: entries (j, c>=j) come from row j
  // of triu) then the column's own strictly-upper entries
  std::vector<int> cnt(n, 0);
  for (int k = 0; k < b->nnzP; k++) { cnt[Pi[k]]++; if (Pi[k] != Pc[k]) cnt[Pc[k]]++; }
  std::vector<int> Fp(n + 1, 0);
  for (int j = 0; j < n; j++) Fp[j + 1] = Fp[j] + cnt[j];
  std::vector<int> Fi(Fp[n]), Fk(Fp[n]), nx(Fp.begin(), Fp.end() - 1);
  for (int j = 0; j < n; j++)           // upper part of row i: ascending column
    for (int k = Pp[j]; k < Pp[j + 1]; k++) { int i = Pi[k]; Fi[nx[i]] = j; Fk[nx[i]] = k; nx[i]++; }
  for (int j = 0; j < n; j++)           // lower part of row j: the column's own entries
    for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] != j) { Fi[nx[j]] = Pi[k]; Fk[nx[j]] = k; nx[j]++; }
  // CSR view of A
  std::vector<int> Rp(m + 1, 0);
  for (int k = 0; k < b->nnzA; k++) Rp[Ai[k] + 1]++;
  for (int i = 0; i < m; i++) Rp[i + 1] += Rp[i];
  std::vector<int> Rj(b->nnzA), Rk(b->nnzA), rn(Rp.begin(), Rp.end() - 1);
  for (int j = 0; j < n; j++)
    for (int k = Ap[j]; k < Ap[j + 1]; k++) { int d = rn[Ai[k]]++; Rj[d] = j; Rk[d] = k; }

  BPattern &pt = b->pat;
  pt.n = n; pt.m = m; pt.nnzP = b->nnzP; pt.nnzA = b->nnzA; pt.nnzPf = Fp[n];
  int rc = 0;
  rc |= bupload(b, &pt.Pp, Pp); rc |= bupload(b, &pt.Pi, Pi); rc |= bupload(b, &pt.Pc, Pc);
  rc |= bupload(b, &pt.Fp, Fp); rc |= bupload(b, &pt.Fi, Fi); rc |= bupload(b, &pt.Fk, Fk);
  rc |= bupload(b, &pt.Ap, Ap); rc |= bupload(b, &pt.Ai, Ai); rc |= bupload(b, &pt.Ac, Ac);
  rc |= bupload(b, &pt.Rp, Rp); rc |= bupload(b, &pt.Rj, Rj); rc |= bupload(b, &pt.Rk, Rk);

  // ---- values and per-QP arrays ----------------------------------------------
  BIO &io = b->io;
  const size_t B = (size_t)batch;
  double *dPx = nullptr, *dAx = nullptr, *dQ = nullptr, *dL = nullptr, *dU = nullptr;
  io.strideP = Px_all ? b->nnzP : 0; io.strideA = Ax_all ? b->nnzA : 0;
  rc |= balloc(b, &dPx, Px_all ? B * b->nnzP : (size_t)b->nnzP);
  rc |= balloc(b, &dAx, Ax_all ? B * b->nnzA : (size_t)b->nnzA);
  rc |= balloc(b, &dQ, B * n); rc |= balloc(b, &dL, B * m); rc |= balloc(b, &dU, B * m);
  rc |= balloc(b, &io.Xs, B * n); rc |= balloc(b, &io.Zs, B * m); rc |= balloc(b, &io.Ys, B * m);
  rc |= balloc(b, &io.Xo, B * n); rc |= balloc(b, &io.Yo, B * m);
  rc |= balloc(b, &io.DXo, B * n); rc |= balloc(b, &io.DYo, B * m);
  rc |= balloc(b, &io.rho_io, B); rc |= balloc(b, &io.info, B * 8);
  if (rc) { osqp_amd_batch_cleanup(b); return OSQP_MEM_ALLOC_ERROR; }
  io.Px = dPx; io.Ax = dAx; io.Q = dQ; io.L = dL; io.U = dU;
  auto up = [&](double *d, const c_float *s, size_t cnt) -> int {
    if (cnt && s && hipMemcpyAsync(d, s, cnt * sizeof(double), hipMemcpyHostToDevice, b->stream) != hipSuccess) return 1;
    return 0;
  };
  rc |= up(dPx, Px_all ? Px_all : P->x, Px_all ? B * b->nnzP : (size_t)b->nnzP);
  rc |= up(dAx, Ax_all ? Ax_all : A->x, Ax_all ? B * b->nnzA : (size_t)b->nnzA);
  rc |= up(dQ, Q, B * n); rc |= up(dL, L, B * m); rc |= up(dU, U, B * m);
  if (rc || hipStreamSynchronize(b->stream) != hipSuccess) { osqp_amd_batch_cleanup(b); return OSQP_LINSYS_SOLVER_INIT_ERROR; }

  const int NP = 16 * b->tile;
  b->lds_bytes = sizeof(double) * ((size_t)b->nnzP + b->nnzA + 7 * NP + 11 * (size_t)m + 4 * NP + 8) +
                 sizeof(int) * (size_t)(m + 4);
  b->lds_bytes = (b->lds_bytes + 15) & ~(size_t)15;
  if (b->lds_bytes > 160 * 1024) {
    fprintf(stderr, "osqp_amd batch: problem needs %zu B of LDS per QP (> 160 KiB)\n", b->lds_bytes);
    osqp_amd_batch_cleanup(b);
    return OSQP_LINSYS_SOLVER_INIT_ERROR;
  }
  if (b->lds_bytes > 64 * 1024) {
    if (b->tile == 8) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_solve<8>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_solve<4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes);
  }
  b->h_info.assign(B * 8, 0.0);
  *out = b;
  return 0;
}

extern "C" void osqp_amd_batch_cleanup(osqp_amd_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  for (void *p : b->allocs) (void)hipFree(p);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

extern "C" c_int osqp_amd_batch_update(osqp_amd_batch *b, const c_float *Q, const c_float *L, const c_float *U) {
  if (!b) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  BCHK(hipSetDevice(b->device));
  const size_t B = (size_t)b->B;
  if (Q) BCHK(hipMemcpyAsync(const_cast<double *>(b->io.Q), Q, B * b->n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  if (L) BCHK(hipMemcpyAsync(const_cast<double *>(b->io.L), L, B * b->m * sizeof(double), hipMemcpyHostToDevice, b->stream));
  if (U) BCHK(hipMemcpyAsync(const_cast<double *>(b->io.U), U, B * b->m * sizeof(double), hipMemcpyHostToDevice, b->stream));
  BCHK(hipStreamSynchronize(b->stream));
  return 0;
}

extern "C" c_int osqp_amd_batch_solve(osqp_amd_batch *b) {
  if (!b) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  BCHK(hipSetDevice(b->device));
  const int first = b->solves == 0;
  if (b->tile == 8)
    hipLaunchKernelGGL(k_batch_solve<8>, dim3((unsigned)b->B), dim3(BT), b->lds_bytes, b->stream, b->pat, b->st, b->io, first);
  else
    hipLaunchKernelGGL(k_batch_solve<4>, dim3((unsigned)b->B), dim3(BT), b->lds_bytes, b->stream, b->pat, b->st, b->io, first);
  BCHK(hipGetLastError());
  BCHK(hipStreamSynchronize(b->stream));
  b->solves++;
  return 0;
}

extern "C" c_int osqp_amd_batch_get(osqp_amd_batch *b, c_float *X, c_float *Y, c_float *info8,
                                   c_float *DX, c_float *DY) {
  if (!b) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  BCHK(hipSetDevice(b->device));
  const size_t B = (size_t)b->B;
  if (X) BCHK(hipMemcpyAsync(X, b->io.Xo, B * b->n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (Y && b->m) BCHK(hipMemcpyAsync(Y, b->io.Yo, B * b->m * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (info8) BCHK(hipMemcpyAsync(info8, b->io.info, B * 8 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (DX) BCHK(hipMemcpyAsync(DX, b->io.DXo, B * b->n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (DY && b->m) BCHK(hipMemcpyAsync(DY, b->io.DYo, B * b->m * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  BCHK(hipStreamSynchronize(b->stream));
  return 0;
}

// device pointers of the result arrays (for device-side gathers: RCCL all_gather)
extern "C" c_int osqp_amd_batch_device_ptrs(osqp_amd_batch *b, void **X, void **Y, void **info8) {
  if (!b) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (X) *X = b->io.Xo;
  if (Y) *Y = b->io.Yo;
  if (info8) *info8 = b->io.info;
  return 0;
}
