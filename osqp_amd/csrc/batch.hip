// batch.hip -- batched OSQP engine for many small QPs with one sparsity pattern
// (MPC-style: BASELINE config 4, 1024 x (n=120, m=240)), gfx950 / MI355X.
//
// One 512-thread workgroup solves one QP from raw data to unscaled solution in
// a single kernel launch; the grid is the batch.  Nothing leaves the CU during
// the ADMM loop:
//   * problem vectors and the (per-QP) scaled matrix values live in LDS;
//   * the KKT solve of update_xz_tilde (reference src/auxil.c:177-183) is a
//     direct one: K = P + sigma I + A' diag(rho) A (n x n, SPD) is formed and
//     inverted in place by Gauss-Jordan with the matrix held in REGISTERS,
//     tiled TR x TC over a 16 x 32 thread grid (n <= 16*TR = 32*TC); every ADMM
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

#define BT 512            // 8 wavefronts; thread grid 16 (row blocks) x 32 (column blocks)
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
  const int *packed;             // all twelve arrays back to back (copied to LDS by the kernel)
};

struct BSettings {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_pinf, eps_dinf, rho_tol, adapt_tol, refine_tol;
  int scaling, adaptive_rho, rho_interval, max_iter, check_termination, scaled_termination,
      warm_start, refine, profile, ablate;
};

struct BIO {               // per-batch arrays (device)
  const double *Px, *Ax;   // shared values, or per-QP values when strideP/strideA != 0
  long long strideP, strideA;
  const double *Q, *L, *U; // [B][n], [B][m], [B][m]
  double *Xs, *Zs, *Ys;    // scaled iterates kept between solves (warm start) [B][..]
  double *Xo, *Yo;         // unscaled solution out
  double *DXo, *DYo;       // certificates out
  double *rho_io;          // current rho per QP (persists between solves)
  // per-QP workspace written by the setup phase (the analogue of the reference's
  // scaled OSQPData + factorisation, kept across solves like its workspace)
  double *Wv;              // [B][nnzP + nnzA] scaled matrix values
  double *Wq, *Wl, *Wu;    // scaled q, l, u
  double *Wd, *We, *Wc;    // D [B][n], E [B][m], c [B]
  double *Wk;              // [B][NP*NP] K^-1 in per-thread tile order
  int    *Wt;              // [B][m] constraint class
  int    *flag;            // [B] 1 = K^-1 must be rebuilt (constraint class changed)
  double *info;            // [B][8]: iter, status, obj, pri, dua, rho_updates, rho_estimate, rho
  const int *order;        // [B] solve phase: workgroup k works on QP order[k] (longest expected first)
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
template <int NW>
__device__ __forceinline__ double b_sum(double v, double *red) {
  v = b_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = (red[0] + red[1]) + (red[2] + red[3]);
  if (NW == 8) t += (red[4] + red[5]) + (red[6] + red[7]);
  return t;
}
template <int NW>
__device__ __forceinline__ double b_max(double v, double *red) {
  v = b_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  if (NW == 8) t = fmax(t, fmax(fmax(red[4], red[5]), fmax(red[6], red[7])));
  return t;
}
__device__ __forceinline__ double clip_scale(double v) {
  if (v < 1e-4) v = 1.0;
  if (v > 1e4) v = 1e4;
  return v;
}

// LDS working set of one QP.  n-vectors are NP apart, m-vectors m apart, index
// arrays are one int block: a handful of base pointers instead of ~40.
struct BL {
  double *Pv, *Av;          // scaled matrix values (triu P, CSC A)
  double *nv, *mv;          // n-vector block (stride NP), m-vector block (stride m)
  double *rowk, *colk;      // Gauss-Jordan exchange (2 x 2 x NP)
  double *gp;               // scratch of the fused reductions (256 doubles)
  double *red;
  int *ctype;
  int NP, m;
  // shared sparsity pattern, copied into LDS once per workgroup
  const int *Pp, *Pi, *Pc, *Fp, *Fi, *Fk, *Ap, *Ai, *Ac, *Rp, *Rj, *Rk;
};
#define NV(k) (s.nv + (k) * s.NP)
#define MV(k) (s.mv + (k) * s.m)
#define s_q NV(0)
#define s_x NV(1)
#define s_xt NV(2)
#define s_dx NV(3)
#define s_D NV(4)
#define s_tn NV(5)
#define s_b NV(6)
#define s_l MV(0)
#define s_u MV(1)
#define s_rho MV(2)
#define s_rinv MV(3)
#define s_z MV(4)
#define s_y MV(5)
#define s_ws MV(6)   /* m-scratch: refinement, certificates */
#define s_w MV(7)    /* rho z - y, kept current by the z/y update */
#define s_dy MV(8)
#define s_E MV(9)
#define s_tm MV(10)

// y_i = sum_j A_ij v_j  (row gather through the CSR view of the CSC values)
__device__ __forceinline__ double a_row_dot(const BL &s, const double *v, int i) {
  double acc = 0.0;
  for (int k = s.Rp[i]; k < s.Rp[i + 1]; ++k) acc += s.Av[s.Rk[k]] * v[s.Rj[k]];
  return acc;
}
// (A' v)_j  (column gather)
__device__ __forceinline__ double a_col_dot(const BL &s, const double *v, int j) {
  double acc = 0.0;
  for (int k = s.Ap[j]; k < s.Ap[j + 1]; ++k) acc += s.Av[k] * v[s.Ai[k]];
  return acc;
}
// (P v)_j from the full symmetric pattern
__device__ __forceinline__ double p_row_dot(const BL &s, const double *v, int j) {
  double acc = 0.0;
  for (int k = s.Fp[j]; k < s.Fp[j + 1]; ++k) acc += s.Pv[s.Fk[k]] * v[s.Fi[k]];
  return acc;
}

// ---------------------------------------------------------------------------
// register-tiled K^-1: thread (tr, tc) of a 16 x 16 grid owns rows tr*T.. and
// columns tc*T.. of the (padded) NP x NP matrix, NP = 16*T.
// Sparse dots split over adjacent lanes (the hot loop's rows and columns hold
// 1..6 entries: the serial chain per row, not the flop count, sets the latency).
// Lanes of one quad exchange through DPP quad_perm (no LDS traffic).
__device__ __forceinline__ double quad_xor1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_xor2(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// y_i = sum_j A_ij v_j by the two lanes (l = 0, 1) of a pair; both return the sum
__device__ __forceinline__ double a_row_dot2(const BL &s, const double *v, int i, int l) {
  double acc = 0.0;
  for (int k = s.Rp[i] + l; k < s.Rp[i + 1]; k += 2) acc += s.Av[s.Rk[k]] * v[s.Rj[k]];
  return acc + quad_xor1(acc);
}
// (A' v)_j by the four lanes (l = 0..3) of a quad; all return the sum
__device__ __forceinline__ double a_col_dot4(const BL &s, const double *v, int j, int l) {
  double acc = 0.0;
  for (int k = s.Ap[j] + l; k < s.Ap[j + 1]; k += 4) acc += s.Av[k] * v[s.Ai[k]];
  acc += quad_xor1(acc);
  return acc + quad_xor2(acc);
}

// (P v)_j from the full symmetric pattern by the four lanes of a quad; all return the sum
__device__ __forceinline__ double p_row_dot4(const BL &s, const double *v, int j, int l) {
  double acc = 0.0;
  for (int k = s.Fp[j] + l; k < s.Fp[j + 1]; k += 4) acc += s.Pv[s.Fk[k]] * v[s.Fi[k]];
  acc += quad_xor1(acc);
  return acc + quad_xor2(acc);
}
// Wavefront reductions whose result is uniform: two quad exchanges, two mirror steps inside
// the 16-lane row (DPP, no LDS), then the four row totals are read out by lane.
__device__ __forceinline__ double dpp_mirror(double v, bool half) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (half) { lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true); }
  else      { lo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true); }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_all_max(double v) {
  v = fmax(v, quad_xor1(v)); v = fmax(v, quad_xor2(v));
  v = fmax(v, dpp_mirror(v, true)); v = fmax(v, dpp_mirror(v, false));
  return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ double wave_all_sum(double v) {
  v += quad_xor1(v); v += quad_xor2(v);
  v += dpp_mirror(v, true); v += dpp_mirror(v, false);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
// NMAX maxima and NSUM sums over the workgroup with two barriers; the combined values are
// left in buf[NW*(NMAX+NSUM) ...] (maxima first).  buf: >= (NW+1)*(NMAX+NSUM) doubles of LDS.
template <int NW, int NMAX, int NSUM>
__device__ __forceinline__ void b_reduce_many(double (&mx)[NMAX], double (&sm)[NSUM], double *buf) {
  constexpr int K = NMAX + NSUM;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) { const double r = wave_all_max(mx[k]); if (lane == 0) buf[w * K + k] = r; }
#pragma unroll
  for (int k = 0; k < NSUM; ++k) { const double r = wave_all_sum(sm[k]); if (lane == 0) buf[w * K + NMAX + k] = r; }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    const int k = threadIdx.x;
    double r;
    if (k < NMAX) { r = buf[k]; for (int q = 1; q < NW; ++q) r = fmax(r, buf[q * K + k]); }
    else {
      r = (buf[k] + buf[K + k]) + (buf[2 * K + k] + buf[3 * K + k]);
      if (NW == 8) r += (buf[4 * K + k] + buf[5 * K + k]) + (buf[6 * K + k] + buf[7 * K + k]);
    }
    buf[NW * K + k] = r;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
template <int TR, int TC, int GC>
__device__ __forceinline__ void form_K(double (&a)[TR][TC], int n, const BL &s, double sigma) {
  const int tr = threadIdx.x / GC, tc = threadIdx.x % GC;
  const double *rho = s_rho;
#pragma unroll 1
  for (int r = 0; r < TR; ++r) {
    double v[TC];
#pragma unroll 1
    for (int c = 0; c < TC; ++c) {
      const int i = tr * TR + r, j = tc * TC + c;
      double acc = 0.0;
      if (i < n && j < n) {
        // P_ij from the upper triangle: column max(i,j), row min(i,j)
        const int cj = i > j ? i : j, ri = i > j ? j : i;
        for (int k = s.Pp[cj]; k < s.Pp[cj + 1]; ++k) if (s.Pi[k] == ri) acc += s.Pv[k];
        if (i == j) acc += sigma;
        // sum_t rho_t A_ti A_tj : merge of the two sorted columns
        int ka = s.Ap[i], kb = s.Ap[j];
        const int ea = s.Ap[i + 1], eb = s.Ap[j + 1];
        while (ka < ea && kb < eb) {
          const int ra = s.Ai[ka], rb = s.Ai[kb];
          if (ra == rb) { acc += rho[ra] * s.Av[ka] * s.Av[kb]; ++ka; ++kb; }
          else if (ra < rb) ++ka; else ++kb;
        }
      } else if (i == j) acc = 1.0;    // identity padding keeps the inverse well defined
#pragma unroll
      for (int cc = 0; cc < TC; ++cc) if (cc == c) v[cc] = acc;
    }
#pragma unroll
    for (int rr = 0; rr < TR; ++rr)
#pragma unroll
      for (int cc = 0; cc < TC; ++cc) if (rr == r) a[rr][cc] = v[cc];
  }
}

// In-place Gauss-Jordan inversion without pivoting (K is SPD).  One barrier per
// pivot: the pivot row / column are exchanged through double-buffered LDS.
template <int TR, int TC, int GC>
__device__ __forceinline__ void invert_tiles(double (&a)[TR][TC], const BL &s, int n) {
  // The pivot loop is unrolled by TR (a multiple of TC) so that the pivot's
  // position inside a tile (ko, kco) is a compile-time constant: the register
  // tile is only ever indexed statically.  Per pivot: owners publish row k and
  // column k to LDS, one barrier, one FMA per tile element, then the owners of
  // row k / column k overwrite their strip with the Gauss-Jordan special cases.
  static_assert(TR % TC == 0, "tile shape");
  constexpr int NP = 16 * TR;
  const int tr = threadIdx.x / GC, tc = threadIdx.x % GC;
  const int nkb = (n + TR - 1) / TR;    // padded rows/columns are identity: nothing to eliminate
#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
    for (int ko = 0; ko < TR; ++ko) {
      const int k = kb * TR + ko;
      const int kco = ko % TC;               // static after unrolling
      const int kc = k / TC;                 // column block that owns column k
      double *rowk = s.rowk + (k & 1) * NP, *colk = s.colk + (k & 1) * NP;
      if (tr == kb) {
#pragma unroll
        for (int c = 0; c < TC; ++c) rowk[c * GC + tc] = a[ko][c];   // [c][tc]: lane stride 8 B
      }
      if (tc == kc) {
#pragma unroll
        for (int r = 0; r < TR; ++r) colk[tr * TR + r] = a[r][kco];
      }
      __syncthreads();
      const double piv = 1.0 / rowk[kco * GC + kc];
      double rk[TC];
#pragma unroll
      for (int c = 0; c < TC; ++c) rk[c] = rowk[c * GC + tc] * piv;
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const double ci = colk[tr * TR + r];
#pragma unroll
        for (int c = 0; c < TC; ++c) a[r][c] = __builtin_fma(-ci, rk[c], a[r][c]);
      }
      if (tc == kc) {          // column k: a_ik <- -a_ik / a_kk
#pragma unroll
        for (int r = 0; r < TR; ++r) a[r][kco] = 0.0 - colk[tr * TR + r] * piv;
      }
      if (tr == kb) {          // row k: a_kj <- a_kj / a_kk, a_kk <- 1 / a_kk
#pragma unroll
        for (int c = 0; c < TC; ++c) a[ko][c] = (tc * TC + c == k) ? piv : rk[c];
      }
    }
  }
  __syncthreads();
}

// K^-1 for the per-iteration GEMV lives in a second register layout ("G"): eight adjacent
// lanes share a group of RG = NP/64 rows, lane q of the eight holds the columns
// {16k + 2q, 16k + 2q + 1}.  The eight partial sums of a row meet through three DPP exchanges
// (no LDS round trip, one barrier per GEMV), and for a fixed k the eight lanes read 128
// contiguous bytes of the input vector.  The Gauss-Jordan tiles (layout "I") are converted
// once per inversion through the per-QP K^-1 array in HBM, which is stored in G order.
template <int NP> struct GL { static constexpr int RG = NP / 64, CG = NP / 8; };

// out_i = sum_j Kinv_ij in_j ; in / out are LDS vectors of length >= NP
template <int NP>
__device__ __forceinline__ void tile_gemv(const double (&ag)[GL<NP>::RG][GL<NP>::CG], const double *in, double *out) {
  constexpr int RG = GL<NP>::RG, CG = GL<NP>::CG;
  const int g = threadIdx.x >> 3, q = threadIdx.x & 7;
  double acc[RG];
#pragma unroll
  for (int r = 0; r < RG; ++r) acc[r] = 0.0;
#pragma unroll
  for (int k = 0; k < CG / 2; ++k) {
    const double2 bv = *reinterpret_cast<const double2 *>(in + 16 * k + 2 * q);
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      acc[r] = __builtin_fma(ag[r][2 * k], bv.x, acc[r]);
      acc[r] = __builtin_fma(ag[r][2 * k + 1], bv.y, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    double v = acc[r];
    v += quad_xor1(v); v += quad_xor2(v); v += dpp_mirror(v, true);
    if (q == 0) out[g * RG + r] = v;
  }
  __syncthreads();
}
// element (i, j) of K^-1 -> position in the per-QP array (G order, slot-major so that the
// loads of the solve phase are coalesced)
template <int NP, int NT>
__device__ __forceinline__ int g_index(int i, int j) {
  constexpr int RG = GL<NP>::RG, CG = GL<NP>::CG;
  const int tg = (i / RG) * 8 + ((j & 15) >> 1);
  const int sg = (i % RG) * CG + ((j >> 4) << 1) + (j & 1);
  return sg * NT + tg;
}
template <int TR, int TC, int GC>
__device__ __forceinline__ void store_kinv(const double (&a)[TR][TC], double *Wk) {
  constexpr int NP = 16 * TR, NT = 16 * GC;
  const int tr = threadIdx.x / GC, tc = threadIdx.x % GC;
#pragma unroll
  for (int r = 0; r < TR; ++r)
#pragma unroll
    for (int c = 0; c < TC; ++c) Wk[g_index<NP, NT>(tr * TR + r, tc * TC + c)] = a[r][c];
}
// (agent-scope loads: inside the loop the array was just rewritten by other lanes of this workgroup)
template <int NP, int NT>
__device__ __forceinline__ void load_kinv(double (&ag)[GL<NP>::RG][GL<NP>::CG], double *Wk) {
#pragma unroll
  for (int r = 0; r < GL<NP>::RG; ++r)
#pragma unroll
    for (int c = 0; c < GL<NP>::CG; ++c)
      ag[r][c] = __hip_atomic_load(Wk + (r * GL<NP>::CG + c) * NT + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// form K, invert it in the Gauss-Jordan tile layout and leave it in HBM in GEMV order
template <int TR, int TC, int GC>
__device__ __forceinline__ void rebuild_kinv(int n, const BL &s, double sigma, double *Wk) {
  double a[TR][TC];
  form_K<TR, TC, GC>(a, n, s, sigma);
  invert_tiles<TR, TC, GC>(a, s, n);
  store_kinv<TR, TC, GC>(a, Wk);
  __threadfence();               // the stores are re-read by other lanes of this workgroup
  __syncthreads();
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
// PH = 0: setup phase (scale, classify, build K^-1, store the workspace);
// PH = 1: solve phase (load the workspace, ADMM loop, store the solution).
// Phase time stamps / per-phase accumulators exist only in -DOSQP_AMD_BATCH_DEBUG builds
// (make BATCH_DEBUG=1): they cost ~26 registers and a 2 us s_memrealtime each.
#ifdef OSQP_AMD_BATCH_DEBUG
#define DBG(...) __VA_ARGS__
#else
#define DBG(...)
#endif
#ifndef BATCH_WAVES_PER_SIMD
#define BATCH_WAVES_PER_SIMD 2   // 4 (two workgroups per CU, <= 128 registers) was measured slower: spills lengthen the slowest QP
#endif
template <int TR, int TC, int GC, int PH>
__global__ void __launch_bounds__(16 * GC, PH == 1 ? BATCH_WAVES_PER_SIMD : 2) k_batch_solve(BPattern p, BSettings st, BIO io) {
  constexpr int NP = 16 * TR, NT = 16 * GC, NW = NT / 64, phase = PH;
  static_assert(GC * TC == NP, "tile shape");
  static_assert(4 * NP <= NT, "the GEMV reduction and the column dots use four lanes per row/column");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int n = p.n, m = p.m, tid = threadIdx.x;
  const long long qp = (phase == 1 && io.order) ? io.order[blockIdx.x] : (int)blockIdx.x;
  BL s;
  {
    double *w = lds;
    s.NP = NP; s.m = m;
    s.Pv = w; w += p.nnzP; s.Av = w; w += p.nnzA;
    w += (p.nnzP + p.nnzA) & 1;          // n-vectors are read 16 bytes at a time by the GEMV
    s.nv = w; w += 7 * NP; s.mv = w; w += 11 * m;
    s.rowk = w; w += 2 * NP; s.colk = w; w += 2 * NP; s.red = w; w += 64; s.gp = w; w += 256;
    int *iw = reinterpret_cast<int *>(w);
    s.ctype = iw; iw += m;
    int *ib = iw;
    s.Pp = iw; iw += n + 1; s.Pi = iw; iw += p.nnzP; s.Pc = iw; iw += p.nnzP;
    s.Fp = iw; iw += n + 1; s.Fi = iw; iw += p.nnzPf; s.Fk = iw; iw += p.nnzPf;
    s.Ap = iw; iw += n + 1; s.Ai = iw; iw += p.nnzA; s.Ac = iw; iw += p.nnzA;
    s.Rp = iw; iw += m + 1; s.Rj = iw; iw += p.nnzA; s.Rk = iw; iw += p.nnzA;
    // the host packs the twelve index arrays back to back in this order
    const int tot = (int)(iw - ib);
    for (int k = tid; k < tot; k += NT) ib[k] = p.packed[k];
  }
  double ag[GL<NP>::RG][GL<NP>::CG];     // K^-1 in the GEMV layout
  DBG(unsigned long long tstamp[8]; tstamp[0] = wall_clock64(); const unsigned long long cyc0 = clock64();)

  // ---- load: raw problem (setup phase) or the per-QP workspace (solve phase) ---
  double cs = 1.0;   // cost scaling c
  for (int j = tid; j < NP; j += NT) {
    s_q[j] = 0.0; s_x[j] = 0.0; s_xt[j] = 0.0; s_dx[j] = 0.0; s_D[j] = 1.0; s_tn[j] = 0.0; s_b[j] = 0.0;
  }
  for (int i = tid; i < m; i += NT) { s_z[i] = 0.0; s_y[i] = 0.0; s_E[i] = 1.0; s_dy[i] = 0.0; s_ws[i] = 0.0; }
  __syncthreads();
  const long long nv_ = (long long)p.nnzP + p.nnzA;
  if (phase == 0) {
    const double *Pg = io.Px + qp * io.strideP, *Ag = io.Ax + qp * io.strideA;
    for (int k = tid; k < p.nnzP; k += NT) s.Pv[k] = Pg[k];
    for (int k = tid; k < p.nnzA; k += NT) s.Av[k] = Ag[k];
    for (int j = tid; j < n; j += NT) s_q[j] = io.Q[qp * n + j];
    for (int i = tid; i < m; i += NT) { s_l[i] = io.L[qp * m + i]; s_u[i] = io.U[qp * m + i]; }
  } else {
    const double *Wv = io.Wv + qp * nv_;
    for (int k = tid; k < p.nnzP; k += NT) s.Pv[k] = Wv[k];
    for (int k = tid; k < p.nnzA; k += NT) s.Av[k] = Wv[p.nnzP + k];
    for (int j = tid; j < n; j += NT) { s_q[j] = io.Wq[qp * n + j]; s_D[j] = io.Wd[qp * n + j]; }
    for (int i = tid; i < m; i += NT) {
      s_l[i] = io.Wl[qp * m + i]; s_u[i] = io.Wu[qp * m + i]; s_E[i] = io.We[qp * m + i];
      s.ctype[i] = io.Wt[qp * m + i];
    }
    cs = io.Wc[qp];
  }
  __syncthreads();

  DBG(tstamp[1] = wall_clock64();)
  // ---- Ruiz equilibration (scaling.c:44-156), per QP -------------------------
  for (int pass = 0; phase == 0 && pass < st.scaling; ++pass) {
    for (int j = tid; j < n; j += NT) {
      double v = 0.0;
      for (int k = s.Fp[j]; k < s.Fp[j + 1]; ++k) v = fmax(v, fabs(s.Pv[s.Fk[k]]));
      for (int k = s.Ap[j]; k < s.Ap[j + 1]; ++k) v = fmax(v, fabs(s.Av[k]));
      s_tn[j] = 1.0 / sqrt(clip_scale(v));
    }
    for (int i = tid; i < m; i += NT) {
      double v = 0.0;
      for (int k = s.Rp[i]; k < s.Rp[i + 1]; ++k) v = fmax(v, fabs(s.Av[s.Rk[k]]));
      s_tm[i] = 1.0 / sqrt(clip_scale(v));
    }
    __syncthreads();
    for (int k = tid; k < p.nnzP; k += NT) s.Pv[k] = (s.Pv[k] * s_tn[s.Pi[k]]) * s_tn[s.Pc[k]];
    for (int k = tid; k < p.nnzA; k += NT) s.Av[k] = (s.Av[k] * s_tm[s.Ai[k]]) * s_tn[s.Ac[k]];
    for (int j = tid; j < n; j += NT) { s_q[j] = s_q[j] * s_tn[j]; s_D[j] = s_tn[j] * s_D[j]; }
    for (int i = tid; i < m; i += NT) s_E[i] = s_tm[i] * s_E[i];
    __syncthreads();
    // cost normalisation: mean column norm of P (sequential sum, reference order) vs |q|_inf
    double cn = 0.0, qn = 0.0;
    for (int j = tid; j < n; j += NT) {
      double v = 0.0;
      for (int k = s.Fp[j]; k < s.Fp[j + 1]; ++k) v = fmax(v, fabs(s.Pv[s.Fk[k]]));
      s_tn[j] = v;
      qn = fmax(qn, fabs(s_q[j]));
    }
    qn = b_max<NW>(qn, s.red);
    if (tid == 0) { double acc = 0.0; for (int j = 0; j < n; ++j) acc += s_tn[j]; s.red[12] = acc / (double)n; }
    __syncthreads();
    cn = s.red[12];
    double ct = fmax(cn, clip_scale(qn));
    ct = 1.0 / clip_scale(ct);
    for (int k = tid; k < p.nnzP; k += NT) s.Pv[k] *= ct;
    for (int j = tid; j < n; j += NT) s_q[j] *= ct;
    cs *= ct;
    __syncthreads();
  }
  const double cinv = 1.0 / cs;
  const bool unscaled = st.scaling && !st.scaled_termination;
  if (phase == 0) { for (int i = tid; i < m; i += NT) { s_l[i] = s_l[i] * s_E[i]; s_u[i] = s_u[i] * s_E[i]; } }
  __syncthreads();

  DBG(tstamp[2] = wall_clock64();)
  // ---- rho vector (auxil.c:76-98) and warm start -----------------------------
  double rho = phase == 0 ? st.rho : io.rho_io[qp];
  rho = fmin(fmax(rho, 1e-6), 1e6);
  for (int i = tid; i < m; i += NT) {
    int t = 0;
    if (phase == 0) {
      if (s_l[i] < -BINF && s_u[i] > BINF) t = -1;
      else if (s_u[i] - s_l[i] < st.rho_tol) t = 1;
      s.ctype[i] = t;
    } else t = s.ctype[i];
    const double r = t == -1 ? 1e-6 : (t == 1 ? 1e3 * rho : rho);
    s_rho[i] = r; s_rinv[i] = 1.0 / r;
  }
  if (st.warm_start && phase != 0) {
    for (int j = tid; j < n; j += NT) s_x[j] = io.Xs[qp * n + j];
    for (int i = tid; i < m; i += NT) { s_z[i] = io.Zs[qp * m + i]; s_y[i] = io.Ys[qp * m + i]; }
  }
  __syncthreads();
  DBG(tstamp[3] = wall_clock64();)
  // refinement is applied only to QPs whose K^-1 left a relative residual above 1e-10 in the
  // first solve after it was (re)built; the verdict is kept in bit 1 of flag[]
  const int qflag = phase == 0 ? 1 : io.flag[qp];
  bool need_refine = (qflag & 2) != 0, check_pending = false;
  double *Wk = io.Wk + qp * (long long)(NP * NP);
  if (phase == 0 || (qflag & 1)) {
    rebuild_kinv<TR, TC, GC>(n, s, st.sigma, Wk);
    check_pending = true;
  } else if (qflag & 4) check_pending = true;
  DBG(tstamp[4] = wall_clock64();)
  if (phase != 0) load_kinv<NP, NT>(ag, Wk);
  DBG(tstamp[5] = wall_clock64();)
  if (phase == 0) {
    // ---- store the workspace and stop: the solve phase starts from here -------
    double *Wv = io.Wv + qp * nv_;
    for (int k = tid; k < p.nnzP; k += NT) Wv[k] = s.Pv[k];
    for (int k = tid; k < p.nnzA; k += NT) Wv[p.nnzP + k] = s.Av[k];
    for (int j = tid; j < n; j += NT) { io.Wq[qp * n + j] = s_q[j]; io.Wd[qp * n + j] = s_D[j]; io.Xs[qp * n + j] = 0.0; }
    for (int i = tid; i < m; i += NT) {
      io.Wl[qp * m + i] = s_l[i]; io.Wu[qp * m + i] = s_u[i]; io.We[qp * m + i] = s_E[i];
      io.Wt[qp * m + i] = s.ctype[i]; io.Zs[qp * m + i] = 0.0; io.Ys[qp * m + i] = 0.0;
    }
    if (tid == 0) { io.Wc[qp] = cs; io.rho_io[qp] = rho; io.flag[qp] = 4; }   // 4: verdict on refinement still open
    return;
  }

  // ---- ADMM loop (osqp.c:354-532) ---------------------------------------------
  // Uniform scalars (norms, residuals, status) live in LDS (`sc`), not in
  // registers, and the residual/termination code has ONE call site: a small
  // stage machine replaces the reference's in-loop / post-loop / approximate
  // calls of update_info + check_termination (osqp.c:411-437, 537-581).
  const double alpha = st.alpha, oma = 1.0 - st.alpha, sigma = st.sigma;
  double *sc = s.red + 13;
  enum { S_PRI, S_DUA, S_OBJ, S_NPRI_S, S_NDUA_S, S_NZ_S, S_NAX_S, S_NQ_S, S_NATY_S, S_NPX_S,
         S_NZ, S_NAX, S_NQ, S_NATY, S_NPX, S_STATUS, S_RHO, S_ND, S_LHS, S_NDX, S_QDX, S_COUNT_ };
  enum { F_NORMS = 1, F_STATUS = 2, F_APPROX = 4 };
  if (tid == 0) { for (int k = 0; k < S_COUNT_; ++k) sc[k] = 0.0; sc[S_STATUS] = OSQP_UNSOLVED; sc[S_RHO] = rho; }
  for (int i = tid; i < m; i += NT) s_w[i] = s_rho[i] * s_z[i] - s_y[i];
  __syncthreads();
  int iter = 0, rho_updates = 0, stage = 0, probe_until = 0;
#ifdef OSQP_AMD_BATCH_DEBUG
  unsigned long long pacc[5] = {0, 0, 0, 0, 0}, pt0 = 0, pt1 = 0;
#define PSTAMP(slot) do { if (st.profile) { pt1 = wall_clock64(); pacc[slot] += pt1 - pt0; pt0 = pt1; } } while (0)
#define ABL(bit) (st.ablate & (bit))
#else
#define PSTAMP(slot) do { } while (0)
#define ABL(bit) 0
#endif
  bool norms_fresh = false;

  while (stage != 3) {
    int flags = 0;
    bool checked = false, adapt_due = false;
    if (stage == 0) {
      ++iter;
      DBG(if (st.profile) pt0 = wall_clock64();)
      // rhs of the reduced system: b = sigma x - q + A'(rho z - y); w = rho z - y is kept
      // up to date by the z/y update below.  Four lanes per column.
      if (!ABL(1)) {
        const int j = tid >> 2, l = tid & 3;
        if (j < NP) {
          const double acc = j < n ? a_col_dot4(s, s_w, j, l) : 0.0;
          if (l == 0) s_b[j] = j < n ? (sigma * s_x[j] - s_q[j]) + acc : 0.0;
        }
        __syncthreads();
      }
      PSTAMP(0);
      if (!ABL(2)) tile_gemv<NP>(ag, s_b, s_xt);
      PSTAMP(1);
      // One step of iterative refinement, xt += Kinv (b - K xt), for QPs whose K^-1 needs it.
      // Whether it does is probed (relative residual of the solve above refine_tol) in the first
      // four iterations after K^-1 was built; one hit turns refinement on for good (kept per QP
      // across solves).  refine = 2: always on.
      const bool probe = !need_refine && (check_pending || iter <= probe_until);
      if (st.refine && (st.refine == 2 || need_refine || probe)) {
        for (int i = tid; i < m; i += NT) s_ws[i] = s_rho[i] * a_row_dot(s, s_xt, i);
        __syncthreads();
        double rmax = 0.0, bmax = 0.0;
        for (int j = tid; j < NP; j += NT) {
          s_tn[j] = j < n ? s_b[j] - (p_row_dot(s, s_xt, j) + sigma * s_xt[j] + a_col_dot(s, s_ws, j)) : 0.0;
          rmax = fmax(rmax, fabs(s_tn[j])); bmax = fmax(bmax, fabs(s_b[j]));
        }
        if (probe) {
          rmax = b_max<NW>(rmax, s.red); bmax = b_max<NW>(bmax, s.red);
          need_refine = rmax > st.refine_tol * bmax;
          if (check_pending) { check_pending = false; probe_until = iter + 3; }
        }
        __syncthreads();
        tile_gemv<NP>(ag, s_tn, s_dx);   // dx is free until the x update below
        for (int j = tid; j < n; j += NT) s_xt[j] += s_dx[j];
        __syncthreads();
      }
      PSTAMP(2);
      // z~ = A x~ ; x, z, y updates (auxil.c:185-225, proj.c:4-14); two lanes per row
      if (!ABL(4))
      for (int i = tid >> 1; i < m; i += NT / 2) {
        const double zt = a_row_dot2(s, s_xt, i, tid & 1);
        if ((tid & 1) == 0) {
          const double zo = s_z[i], yo = s_y[i], ri = s_rho[i];
          double v = alpha * zt + oma * zo + s_rinv[i] * yo;
          v = fmax(v, s_l[i]);
          const double zn = fmin(v, s_u[i]);
          const double dy = ri * (alpha * zt + oma * zo - zn);
          const double yn = yo + dy;
          s_z[i] = zn; s_dy[i] = dy; s_y[i] = yn;
          s_w[i] = ri * zn - yn;
        }
      }
      if (!ABL(8))
      for (int j = tid; j < n; j += NT) {
        const double xo = s_x[j];
        const double xn = alpha * s_xt[j] + oma * xo;
        s_dx[j] = xn - xo; s_x[j] = xn;
      }
      __syncthreads();
      PSTAMP(3);
      norms_fresh = false;
      checked = st.check_termination && (iter % st.check_termination == 0);
      adapt_due = st.adaptive_rho && st.rho_interval && (iter % st.rho_interval == 0);
      if (checked) flags = F_NORMS | F_STATUS;
      else if (adapt_due) flags = F_NORMS;
    } else if (stage == 1) flags = F_STATUS | (norms_fresh ? 0 : F_NORMS);
    else flags = F_STATUS | F_APPROX;

    bool term = false;
    if (flags & F_NORMS) {
      // ---- update_info: residuals and norms (auxil.c:227-318), plus the cheap halves of both
      // infeasibility tests (auxil.c:361-512), in two passes and ONE workgroup reduction ----
      double mx[16], sm[3];
#pragma unroll
      for (int k = 0; k < 16; ++k) mx[k] = 0.0;
      sm[0] = sm[1] = sm[2] = 0.0;
      for (int i = tid >> 1; i < m; i += NT / 2) {          // rows: two lanes each
        const double ax = a_row_dot2(s, s_x, i, tid & 1);
        if ((tid & 1) == 0) {
          const double zi = s_z[i], pr = ax + (-1.0) * zi;
          const double ei = unscaled ? 1.0 / s_E[i] : 1.0;
          mx[0] = fmax(mx[0], fabs(ei * pr)); mx[1] = fmax(mx[1], fabs(pr));
          mx[2] = fmax(mx[2], fabs(ei * zi)); mx[3] = fmax(mx[3], fabs(zi));
          mx[4] = fmax(mx[4], fabs(ei * ax)); mx[5] = fmax(mx[5], fabs(ax));
          // delta_y projected on the polar of the recession cone (is_primal_infeasible)
          double dy = s_dy[i];
          const double li = s_l[i], ui = s_u[i];
          if (ui > BINF) { if (li < -BINF) dy = 0.0; else dy = fmin(dy, 0.0); }
          else if (li < -BINF) dy = fmax(dy, 0.0);
          s_ws[i] = dy;
          mx[14] = fmax(mx[14], fabs(unscaled ? s_E[i] * dy : dy));
          sm[1] += ui * fmax(dy, 0.0) + li * fmin(dy, 0.0);
        }
      }
      {                                                     // columns: four lanes each
        const int j = tid >> 2, l = tid & 3;
        if (j < n) {
          const double px = p_row_dot4(s, s_x, j, l);
          const double aty = a_col_dot4(s, s_y, j, l);
          if (l == 0) {
            const double qj = s_q[j], xj = s_x[j], dxj = s_dx[j];
            double dr = qj + px;
            if (m > 0) dr = dr + aty;
            const double di = unscaled ? 1.0 / s_D[j] : 1.0;
            mx[6] = fabs(di * dr); mx[7] = fabs(dr);
            mx[8] = fabs(di * qj); mx[9] = fabs(qj);
            mx[10] = fabs(di * aty); mx[11] = fabs(aty);
            mx[12] = fabs(di * px); mx[13] = fabs(px);
            sm[0] = xj * (0.5 * px + qj);
            mx[15] = fabs(unscaled ? s_D[j] * dxj : dxj);      // is_dual_infeasible: |delta_x|, q'delta_x
            sm[2] = qj * dxj;
          }
        }
      }
      b_reduce_many<NW, 16, 3>(mx, sm, s.gp);
      if (tid == 0) {
        const double *g = s.gp + NW * 19;                 // combined values
        sc[S_PRI] = m == 0 ? 0.0 : (unscaled ? g[0] : g[1]);
        sc[S_NPRI_S] = g[1]; sc[S_NZ] = unscaled ? g[2] : g[3]; sc[S_NZ_S] = g[3];
        sc[S_NAX] = unscaled ? g[4] : g[5]; sc[S_NAX_S] = g[5];
        const double f = unscaled ? cinv : 1.0;
        sc[S_DUA] = unscaled ? g[6] * cinv : g[7]; sc[S_NDUA_S] = g[7];
        sc[S_NQ] = (unscaled ? g[8] : g[9]) * f; sc[S_NQ_S] = g[9];
        sc[S_NATY] = (unscaled ? g[10] : g[11]) * f; sc[S_NATY_S] = g[11];
        sc[S_NPX] = (unscaled ? g[12] : g[13]) * f; sc[S_NPX_S] = g[13];
        sc[S_OBJ] = g[16] * (st.scaling ? cinv : 1.0);
        sc[S_ND] = g[14]; sc[S_LHS] = g[17]; sc[S_NDX] = g[15]; sc[S_QDX] = g[18];
      }
      __syncthreads();
      norms_fresh = true;
    }
    if (flags & F_STATUS) {
      // ---- check_termination (auxil.c:681-786) ----
      const bool approximate = flags & F_APPROX;
      const double pri_res = sc[S_PRI], dua_res = sc[S_DUA];
      int newstatus = 0;      // 0 = keep going
      double newobj = 0.0;
      if (pri_res > 1e30 || dua_res > 1e30) { newstatus = OSQP_NON_CVX; newobj = OSQP_NAN; }
      else {
        double ea = st.eps_abs, er = st.eps_rel, epi = st.eps_pinf, edi = st.eps_dinf;
        if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
        bool prim_ok = false, dual_ok = false, pinf = false, dinf = false;
        if (m == 0) prim_ok = true;
        else if (pri_res < ea + er * fmax(sc[S_NZ], sc[S_NAX])) prim_ok = true;
        else {
          // is_primal_infeasible (auxil.c:361-424); the projected delta_y is in ws
          const double nd = sc[S_ND], lhs = sc[S_LHS];
          if (nd > 1e-30 && lhs < epi * nd) {
            double mxv = 0;
            for (int j = tid; j < n; j += NT) {
              double v = a_col_dot(s, s_ws, j);
              if (unscaled) v = v / s_D[j];
              mxv = fmax(mxv, fabs(v));
            }
            mxv = b_max<NW>(mxv, s.red);
            pinf = mxv < epi * nd;
          }
        }
        if (dua_res < ea + er * fmax(fmax(sc[S_NQ], sc[S_NATY]), sc[S_NPX])) dual_ok = true;
        else {
          // is_dual_infeasible (auxil.c:426-512)
          const double ndx = sc[S_NDX], qdx = sc[S_QDX];
          const double csc_ = unscaled ? cs : 1.0;
          if (ndx > 1e-30 && qdx < csc_ * edi * ndx) {
            double mxv = 0;
            for (int j = tid; j < n; j += NT) {
              double v = p_row_dot(s, s_dx, j);
              if (unscaled) v = v / s_D[j];
              mxv = fmax(mxv, fabs(v));
            }
            mxv = b_max<NW>(mxv, s.red);
            if (mxv < csc_ * edi * ndx) {
              double viol = 0;
              for (int i = tid; i < m; i += NT) {
                double v = a_row_dot(s, s_dx, i);
                if (unscaled) v = v / s_E[i];
                if ((s_u[i] < BINF && v > edi * ndx) || (s_l[i] > -BINF && v < -edi * ndx)) viol += 1.0;
              }
              viol = b_sum<NW>(viol, s.red);
              dinf = viol == 0.0;
            }
          }
        }
        if (prim_ok && dual_ok) newstatus = approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED;
        else if (pinf) { newstatus = approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE; newobj = OSQP_INFTY; }
        else if (dinf) { newstatus = approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE; newobj = -OSQP_INFTY; }
      }
      __syncthreads();
      if (newstatus != 0) {
        term = true;
        if (tid == 0) { sc[S_STATUS] = newstatus; if (newstatus != OSQP_SOLVED && newstatus != OSQP_SOLVED_INACCURATE) sc[S_OBJ] = newobj; }
      }
      __syncthreads();
    }
    if (stage == 0) {
      if (checked && term) { stage = 3; continue; }
      if (adapt_due) {     // adapt_rho (auxil.c:13-74)
        const double pr = (m ? sc[S_NPRI_S] : 0.0) / (fmax(sc[S_NZ_S], sc[S_NAX_S]) + 1e-30);
        const double du = sc[S_NDUA_S] / (fmax(fmax(sc[S_NQ_S], sc[S_NATY_S]), sc[S_NPX_S]) + 1e-30);
        const double rn = fmin(fmax(rho * sqrt(pr / du), 1e-6), 1e6);
        if (rn > rho * st.adapt_tol || rn < rho / st.adapt_tol) {
          rho = rn; rho_updates++;
          for (int i = tid; i < m; i += NT) {
            const int t = s.ctype[i];
            if (t == 0) { s_rho[i] = rho; s_rinv[i] = 1.0 / rho; }
            else if (t == 1) { s_rho[i] = 1e3 * rho; s_rinv[i] = 1.0 / s_rho[i]; }
            s_w[i] = s_rho[i] * s_z[i] - s_y[i];
          }
          __syncthreads();
          rebuild_kinv<TR, TC, GC>(n, s, sigma, Wk);
          load_kinv<NP, NT>(ag, Wk);
          check_pending = true;
        }
      }
      if (iter >= st.max_iter) stage = checked ? 2 : 1;
    } else if (stage == 1) stage = term ? 3 : 2;
    else {
      if (!term && tid == 0) sc[S_STATUS] = OSQP_MAX_ITER_REACHED;
      __syncthreads();
      stage = 3;
    }
  }
  DBG(tstamp[6] = wall_clock64();)
  const int status = (int)sc[S_STATUS];
  const double pri_res = sc[S_PRI], dua_res = sc[S_DUA], obj = sc[S_OBJ];
  double rho_est;
  {
    const double pr = (m ? sc[S_NPRI_S] : 0.0) / (fmax(sc[S_NZ_S], sc[S_NAX_S]) + 1e-30);
    const double du = sc[S_NDUA_S] / (fmax(fmax(sc[S_NQ_S], sc[S_NATY_S]), sc[S_NPX_S]) + 1e-30);
    rho_est = fmin(fmax(rho * sqrt(pr / du), 1e-6), 1e6);
  }

  // ---- store_solution (auxil.c:524-562) ---------------------------------------
  const bool has_sol = !(status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE ||
                         status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE ||
                         status == OSQP_NON_CVX);
  __syncthreads();
  if (has_sol) {
    for (int j = tid; j < n; j += NT) {
      io.Xo[qp * n + j] = st.scaling ? s_x[j] * s_D[j] : s_x[j];
      io.Xs[qp * n + j] = s_x[j];
    }
    for (int i = tid; i < m; i += NT) {
      io.Yo[qp * m + i] = st.scaling ? (s_y[i] * s_E[i]) * cinv : s_y[i];
      io.Ys[qp * m + i] = s_y[i]; io.Zs[qp * m + i] = s_z[i];
    }
  } else {
    for (int j = tid; j < n; j += NT) { io.Xo[qp * n + j] = OSQP_NAN; io.Xs[qp * n + j] = 0.0; }
    for (int i = tid; i < m; i += NT) { io.Yo[qp * m + i] = OSQP_NAN; io.Ys[qp * m + i] = 0.0; io.Zs[qp * m + i] = 0.0; }
    if (status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE) {
      double mx = 0;
      for (int i = tid; i < m; i += NT) { s_ws[i] = unscaled ? s_ws[i] * s_E[i] : s_ws[i]; mx = fmax(mx, fabs(s_ws[i])); }
      mx = b_max<NW>(mx, s.red);
      for (int i = tid; i < m; i += NT) io.DYo[qp * m + i] = s_ws[i] * (1.0 / mx);
    }
    if (status == OSQP_DUAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE_INACCURATE) {
      double mx = 0;
      for (int j = tid; j < n; j += NT) { s_tn[j] = unscaled ? s_dx[j] * s_D[j] : s_dx[j]; mx = fmax(mx, fabs(s_tn[j])); }
      mx = b_max<NW>(mx, s.red);
      for (int j = tid; j < n; j += NT) io.DXo[qp * n + j] = s_tn[j] * (1.0 / mx);
    }
  }
  DBG(if (st.profile && tid == 0) {
    tstamp[7] = wall_clock64();
    for (int k = 0; k < 8; ++k) io.DXo[qp * n + k] = (double)(tstamp[k] - tstamp[0]);
    for (int k = 0; k < 4; ++k) io.DXo[qp * n + 8 + k] = (double)pacc[k];
    io.DXo[qp * n + 12] = (double)(clock64() - cyc0);
    io.DXo[qp * n + 13] = (double)tstamp[0]; io.DXo[qp * n + 14] = (double)tstamp[7];
  })
  if (tid == 0) {
    io.flag[qp] = check_pending ? 4 : (need_refine ? 2 : 0);
    double *inf = io.info + qp * 8;
    inf[0] = iter; inf[1] = status; inf[2] = obj; inf[3] = pri_res; inf[4] = dua_res;
    inf[5] = rho_updates; inf[6] = rho_est; inf[7] = rho;
    if (io.rho_io) io.rho_io[qp] = rho;
  }
}

// osqp_update_lin_cost / osqp_update_bounds for every QP of the batch
// (src/osqp.c:765-846): new raw vectors are scaled with the stored D, E, c; rows
// are re-classified and a changed class requests a rebuild of K^-1
// (update_rho_vec, src/auxil.c:100-142).
__global__ void __launch_bounds__(256) k_batch_update(int n, int m, BIO io, const double *Q, const double *L,
                                                      const double *U, double rho_tol) {
  const long long qp = blockIdx.x;
  __shared__ int changed;
  if (threadIdx.x == 0) changed = 0;
  __syncthreads();
  if (Q) {
    const double c = io.Wc[qp];
    for (int j = threadIdx.x; j < n; j += blockDim.x) io.Wq[qp * n + j] = (Q[qp * n + j] * io.Wd[qp * n + j]) * c;
  }
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    const double e = io.We[qp * m + i];
    if (L) io.Wl[qp * m + i] = L[qp * m + i] * e;
    if (U) io.Wu[qp * m + i] = U[qp * m + i] * e;
    if (L || U) {
      const double l = io.Wl[qp * m + i], u = io.Wu[qp * m + i];
      int t = 0;
      if (l < -BINF && u > BINF) t = -1;
      else if (u - l < rho_tol) t = 1;
      if (t != io.Wt[qp * m + i]) { io.Wt[qp * m + i] = t; changed = 1; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && changed) io.flag[qp] = 1;   // rebuild; the refinement verdict is re-taken after it
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Dispatch order of the next solve: QPs sorted by the iteration count of the solve
// that just finished, longest first.  Workgroups are handed to CUs in blockIdx
// order as CUs free up, so this is longest-processing-time-first list scheduling
// and the batch no longer ends on a late-started slow QP.  (The order only moves
// work between CUs; every QP's arithmetic is untouched, so the order inside a bucket
// may vary from run to run.)  Counting sort in one workgroup: 1024 buckets over
// [0, max iterations], histogram, exclusive scan, scatter through bucket cursors.
__global__ void __launch_bounds__(1024) k_batch_order(long long B, const double *info, int *order) {
  __shared__ int hist[1024], wmax[16];
  const int tid = threadIdx.x;
  int mx = 1;
  for (long long b = tid; b < B; b += 1024) mx = max(mx, (int)info[b * 8]);
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_down(mx, o, 64));
  hist[tid] = 0;
  if ((tid & 63) == 0) wmax[tid >> 6] = mx;
  __syncthreads();
  mx = wmax[0];
  for (int k = 1; k < 16; ++k) mx = max(mx, wmax[k]);
  const double sc = 1023.0 / (double)mx;
  for (long long b = tid; b < B; b += 1024) atomicAdd(&hist[1023 - (int)(info[b * 8] * sc)], 1);
  __syncthreads();
  // exclusive scan of the 1024 bucket counts (bucket 0 = most iterations): wavefront scans + 16 totals
  const int cnt = hist[tid];
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
  if ((tid & 63) == 63) wmax[tid >> 6] = incl;
  __syncthreads();
  int base = incl - cnt;
  for (int k = 0; k < (tid >> 6); ++k) base += wmax[k];
  __syncthreads();
  hist[tid] = base;                         // now the bucket's write cursor
  __syncthreads();
  for (long long b = tid; b < B; b += 1024) order[atomicAdd(&hist[1023 - (int)(info[b * 8] * sc)], 1)] = (int)b;
}

struct osqp_amd_batch {
  int device = 0, tile = 8, threads = 512;
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
  double *dQ = nullptr, *dL = nullptr, *dU = nullptr;   // staging for updates
  int *d_order = nullptr;   // dispatch order for the next solve (k_batch_order)
  int lpt = 1;
};

static void batch_launch(osqp_amd_batch *b, int phase);
static void batch_set_lds(osqp_amd_batch *b);

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
  e = getenv("OSQP_AMD_BATCH_REFINE_TOL");
  t.refine_tol = e ? atof(e) : 1e-12;
  e = getenv("OSQP_AMD_BATCH_PROFILE");
  t.profile = e ? atoi(e) : 0;
  e = getenv("OSQP_AMD_BATCH_ABLATE");   // timing experiments only: skip phases of the loop (results are garbage)
  t.ablate = e ? atoi(e) : 0;
  e = getenv("OSQP_AMD_BATCH_LPT");
  b->lpt = e ? atoi(e) : 1;   // phase time stamps (wall_clock64 ticks) written into DX[0..7]
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
  b->threads = 512;   // 8x4 (n <= 128) or 4x2 (n <= 64) register tiles; 256-thread variants (8x8 tiles) spill
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
  {
    std::vector<int> pk;
    for (const std::vector<int> *v : {&Pp, &Pi, &Pc, &Fp, &Fi, &Fk, &Ap, &Ai, &Ac, &Rp, &Rj, &Rk})
      pk.insert(pk.end(), v->begin(), v->end());
    rc |= bupload(b, &pt.packed, pk);
  }

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
  { int *ord = nullptr; rc |= balloc(b, &ord, B); b->d_order = ord; }
  {
    const size_t NPs = (size_t)16 * b->tile;
    rc |= balloc(b, &io.Wv, B * ((size_t)b->nnzP + b->nnzA));
    rc |= balloc(b, &io.Wq, B * n); rc |= balloc(b, &io.Wl, B * m); rc |= balloc(b, &io.Wu, B * m);
    rc |= balloc(b, &io.Wd, B * n); rc |= balloc(b, &io.We, B * m); rc |= balloc(b, &io.Wc, B);
    rc |= balloc(b, &io.Wk, B * NPs * NPs); rc |= balloc(b, &io.Wt, B * m); rc |= balloc(b, &io.flag, B);
    rc |= balloc(b, &b->dQ, B * n); rc |= balloc(b, &b->dL, B * m); rc |= balloc(b, &b->dU, B * m);
  }
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
  b->lds_bytes = sizeof(double) * ((size_t)b->nnzP + b->nnzA + 1 + 7 * NP + 11 * (size_t)m + 4 * NP + 64 + 256) +
                 sizeof(int) * ((size_t)m + 4 + 3 * ((size_t)n + 1) + 2 * (size_t)b->nnzP + 2 * (size_t)Fp[n] +
                                4 * (size_t)b->nnzA + (size_t)m + 1);
  b->lds_bytes = (b->lds_bytes + 15) & ~(size_t)15;
  if (b->lds_bytes > 160 * 1024) {
    fprintf(stderr, "osqp_amd batch: problem needs %zu B of LDS per QP (> 160 KiB)\n", b->lds_bytes);
    osqp_amd_batch_cleanup(b);
    return OSQP_LINSYS_SOLVER_INIT_ERROR;
  }
  if (b->lds_bytes > 64 * 1024) batch_set_lds(b);
  b->h_info.assign(B * 8, 0.0);
  // setup phase on the device: Ruiz scaling, rho classes, K^-1 (one workgroup per QP)
  batch_launch(b, 0);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) {
    osqp_amd_batch_cleanup(b);
    return OSQP_LINSYS_SOLVER_INIT_ERROR;
  }
  *out = b;
  return 0;
}

static void batch_launch(osqp_amd_batch *b, int phase) {
  const dim3 g((unsigned)b->B);
#define BL_(TR, TC, GC, PH) hipLaunchKernelGGL((k_batch_solve<TR, TC, GC, PH>), g, dim3(16 * GC), b->lds_bytes, b->stream, b->pat, b->st, b->io)
  if (b->tile == 8) { if (phase == 0) BL_(8, 4, 32, 0); else BL_(8, 4, 32, 1); }
  else              { if (phase == 0) BL_(4, 2, 32, 0); else BL_(4, 2, 32, 1); }
#undef BL_
}
static void batch_set_lds(osqp_amd_batch *b) {
#define SA_(TR, TC, GC, PH) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch_solve<TR, TC, GC, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes)
  if (b->tile == 8) { SA_(8, 4, 32, 0); SA_(8, 4, 32, 1); }
  else              { SA_(4, 2, 32, 0); SA_(4, 2, 32, 1); }
#undef SA_
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
  if (L && U)
    for (size_t k = 0; k < B * (size_t)b->m; k++) if (L[k] > U[k]) return 1;   // osqp.c:815-822
  if (Q) BCHK(hipMemcpyAsync(b->dQ, Q, B * b->n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  if (L) BCHK(hipMemcpyAsync(b->dL, L, B * b->m * sizeof(double), hipMemcpyHostToDevice, b->stream));
  if (U) BCHK(hipMemcpyAsync(b->dU, U, B * b->m * sizeof(double), hipMemcpyHostToDevice, b->stream));
  hipLaunchKernelGGL(k_batch_update, dim3((unsigned)B), dim3(256), 0, b->stream, b->n, b->m, b->io,
                     Q ? b->dQ : nullptr, L ? b->dL : nullptr, U ? b->dU : nullptr, (double)RHO_TOL);
  BCHK(hipGetLastError());
  BCHK(hipStreamSynchronize(b->stream));
  return 0;
}

extern "C" c_int osqp_amd_batch_solve(osqp_amd_batch *b) {
  if (!b) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  BCHK(hipSetDevice(b->device));
  b->io.order = (b->lpt && b->solves > 0) ? b->d_order : nullptr;   // first solve: no history, index order
  batch_launch(b, 1);
  if (b->lpt) hipLaunchKernelGGL(k_batch_order, dim3(1), dim3(1024), 0, b->stream, b->B, b->io.info, b->d_order);
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
