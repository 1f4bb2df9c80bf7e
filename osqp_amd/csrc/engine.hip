// engine.hip -- device-resident OSQP ADMM engine for gfx950 (MI355X, CDNA4).
//
// What runs here (all fp64, int32 indices, one HIP stream per engine):
//   * the ADMM iteration of osqp_solve (reference src/osqp.c:356-370,
//     src/auxil.c:161-225, src/proj.c:4-14) as hipGraph replays of
//       k_pcg_init -> k_cg_A, k_cg_B (operator on u0) -> K x { k_cg_A, k_cg_B } -> k_admm_finalize
//   * the linear solve of update_xz_tilde as an indirect method: Jacobi-PCG on
//       (P + sigma I + A' diag(rho) A) x~ = sigma x - q + A'(rho.z - y)
//     (docs/solver/index.rst:50-55 in the reference), then z~ = A x~
//   * residual / tolerance / rho-estimate / infeasibility reductions of
//     update_info + check_termination (src/auxil.c:13-52, 227-512)
//
// Data layout in HBM:
//   A      : CSR (row gather for A x), built from the CSC the API hands over
//   M      : fused row matrix [P_full | A'] (n rows, n+m columns).  Row j holds
//            P(j, j..n) ascending, then P(j, 0..j) ascending, then column j of
//            A with column ids n+i -- i.e. exactly the summation order of the
//            reference's mat_vec + mat_tpose_vec(skip_diag) + mat_tpose_vec
//            (lin_alg.c:241-322), so P x and A' y are reproduced bit for bit.
//   vectors: [x | y], [x~ | rho z~], [0 | rho z - y], [p | t] (x2) are stored
//            contiguously so one fused SpMV over M consumes "n-part | m-part".
//
// SpMV scheme (all matrices): row blocks of <= chunk non-zeros are streamed
// with fully coalesced index/value loads, the products val*x[col] are staged in
// LDS, then each row segment is summed sequentially from LDS by one lane
// (CSR-stream).  Rows longer than a chunk are reduced by the whole workgroup
// with 64-lane wavefront shuffles.  Dot products are written as one partial
// per workgroup and re-reduced in a fixed order by every workgroup of the next
// kernel, so results are bitwise reproducible and PCG scalars never visit the
// host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <new>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <map>
#include <algorithm>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <chrono>
#include "../../include/osqp_amd_engine.h"

#define TB 256            // threads per workgroup (4 wavefronts of 64)
#define MAX_CHUNK 2048    // products staged in LDS per workgroup (x2 for dual stream)
#define LONG_ROW 512      // a row with at least this many entries gets a workgroup of its own (no LDS staging)
#define HUGE_ROW 8192     // A rows this long are sliced over the whole grid in k_cg_A (partials finished by k_huge_reduce)
#define MAX_PARTS 1024    // upper bound on workgroups that emit dot partials
#define MAX_HUGE_FOLD 4   // up to this many huge rows are folded into k_cg_B (more: k_huge_reduce finishes them)
#define INF_BOUND 1e26    // OSQP_INFTY * MIN_SCALING

#define HIPCHK(call)                                                            \
  do {                                                                          \
    hipError_t _e = (call);                                                     \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "osqp_amd: HIP error %s at %s:%d (%s)\n",                 \
              hipGetErrorString(_e), __FILE__, __LINE__, #call);                \
      return HIPENG_ERR_HIP;                                                    \
    }                                                                           \
  } while (0)

// ---------------------------------------------------------------------------
// device-side descriptors
// ---------------------------------------------------------------------------
struct RowBlk { int r0, r1, k0, k1; };
#define IS_LONG(b) ((b).r1 - (b).r0 == 1 && (b).k1 - (b).k0 >= LONG_ROW)
// per-column record gathered by k_cg_A: one 32-byte access instead of four 8-byte
// gathers from four arrays (residual, w = K u, s = K p, Jacobi inverse)
struct __attribute__((aligned(32))) G4 { double r, w, s, m; };

struct DevMat {
  int nrows, nblk;
  int nstream;           // blocks [0,nstream) are multi-row stream blocks, [nstream,nblk) single long rows
  int nwave;             // PCG kernels: long rows [nstream,nwave) take one wavefront each, [nwave,nblk) are huge
  const int    *rowptr;
  const int    *col;
  const unsigned short *col16;   // the same column ids as 16-bit words where every id fits (null otherwise): the long-row passes
                                 // stream 10 instead of 12 bytes per entry (config 3: 26 -> 22 us per pass over A)
  const double *val;
  const int    *split;   // M only: first entry of the A' part of each row
  const RowBlk *blk;
};

// Dense diagonal blocks of P (portfolio-style block-diagonal covariance): such a block is
// kept as a plain dense b x b array (8 B per entry, no index) for the PCG operator kernel;
// everything else of its rows (the A' part) stays in the sparse remainder matrix.
#define DENSE_MAX 128          // lanes cover the block's columns in two 64-lane halves
struct DenseBlk { int c0, b, off, pitch; };   // first row/column, size, offset into val, row pitch in doubles (even, zero padded)
struct DenseP {
  int nblk;                  // dense blocks (0: feature off)
  const DenseBlk *blk;
  const double *val;         // blocks back to back, each b rows of `pitch` doubles (zero padded; row-major = column-major: symmetric)
};

// Written by kernels only; read by the host between windows.
struct State {
  int    run;          // this ADMM iteration is active (written by the first kernel)
  int    done;         // PCG converged (written by k_cg_A)
  int    stalled;      // PCG ran out of unrolled iterations (written by finalize)
  int    neg_curv;     // p'Kp <= 0 seen
  int    iters[2];     // PCG iterations of the current solve (ping-pong on parity)
  int    neg_curv_seen;
  int    iters_last;
  int    iters_max;
  int    iters_min;        // smallest PCG iteration count of a solve since the host last reset it
  int    forced;
  long long iters_total;
  long long admm_done;
  double rz[2];
  double tol2;
  double gam[2], alp[2];   // Chronopoulos-Gear scalars (ping-pong on parity)
  int    hist_r;           // ADMM iterations since the PCG start vector history was reset (k_pcg_init counts)
  long long admm_target;   // k_pcg_init starts no new ADMM iteration once admm_done has reached this
  unsigned res_epoch;      // resident PCG: tag of the last exchange of the previous launch
  int    res_fail;         // resident PCG: a wait timed out (workgroups not co-resident); the host falls back to the launch-per-step path
  int    res_pipe_off;     // resident PCG: the pipelined recurrences failed a true-residual check for this K (k_form_K clears it)
  int    res_chk_fail;     // resident PCG: true-residual checks that failed since create (each continued the solve from the true residual)
  int    res_dbg[4];       // resident PCG, first timed-out wait: 1 = flags / 2 = granules, exchange number, waiting workgroup, first missing workgroup
  int    res_slow;         // resident PCG: waits that ended well but took more than 50 us (the host adds them up: hipeng_resident_info)
  int    res_slow_max;     // ... the longest of them, in ticks of the 100 MHz clock
  int    res_repub;        // ... flags / granules that a workgroup stored again because its own wait went on (every 64th round)
  int    res_slow_hist[4]; // ... slow waits by length: 50-200 us, 200-500 us, 0.5-2 ms, longer
  unsigned res_slow_xcc;   // ... bit k: a wavefront on XCC k had one
  int    res_slow_last[4]; // ... the latest: exchange number, workgroup, first workgroup it had missed at its 16th round, wait kind (1 flags, 2 granules)
};

struct Params {          // mutable scalars (host writes, kernels read)
  double sigma, alpha, eps_rel, eps_abs, cinv;
  int    pcg_max_iter, use_cvec, has_scaling, pad0;
  double ex_theta;         // extrapolation of the PCG start vector (0 = plain warm start)
  int    ex_h0, ex_h1;     // no extrapolation before iteration ex_h0 after a reset, half a step before ex_h1
  int    no_restart;       // setup-time convexity probe: any loss of positivity counts as negative curvature
};

struct Ctx {             // static pointers / sizes, passed by value
  int n, m;
  DevMat A, M;
  DevMat Mk;             // what k_cg_B streams: M itself, or its remainder when P has dense blocks
  DenseP dP;
  int nh, hrow[MAX_HUGE_FOLD];     // huge rows of A folded into k_cg_B, and hcol: nh dense vectors hcol_k[j] = A(h_k, j)
  const double *hcol;
  int gridA, gridM;      // launch grids (>=1) of row kernels over A / over M
  double *xy, *z, *zt, *va, *vb, *q, *l, *u, *rho, *rhoinv, *minv, *pdiag;
  double *r, *zz, *kp, *pt0, *pt1, *dxy, *dy, *cvec;
  double *init_r, *init_z;        // where k_pcg_init puts r0 and Minv r0
  double *vx, *vold;               // PCG start vector [x~0 | rho z~0] (extrapolated) and the previous [x~ | rho z~]
  double *pdir, *ut;               // Chronopoulos-Gear PCG: p, [u|t]
  G4     *g4;                      // ... and {r, w, s, Minv} records, ping-ponged on parity (2n)
  int     init_stride;             // element stride of init_r (4 when it points into g4)
  int     fin_wave_rows;           // k_admm_finalize: one wavefront per long row of A (dense-direct engines)
  double *vd;                      // dense-direct engines: vb - vx over all n + m entries (k_vd, before k_pcg_init): the long rows of k_pcg_init then
                                   // gather ONE vector per entry -- they need the residual b - K x~0 only, not b itself
  int     plain_rhs;               // block-direct engines: k_pcg_init leaves b0 = b - S' beta (the right-hand side without the low-rank rows' terms) instead of
  const int *plain_skip;           // the residual b - K x~0: no pass over P.  plain_skip[i] >= 0: row i of A is a coupling row (null: only the folded huge rows are)
  // resident PCG: k_pcg_init also leaves u0 = Minv r0 in the layout of the exchanged vector (position u0map[j] of u0pos), so that
  // the resident launch takes it in with one coalesced sweep instead of a 2-byte-indexed gather (null: not a k_pcg_resident engine)
  const unsigned short *u0map;
  double *u0pos;
  // Slack-like variables eliminated from the linear system (k_elim_refresh): nelim = 0 switches all of it off (rhoe == rho then)
  int     nelim;
  const int *erow, *ecol, *epos;   // [n] row of an eliminated variable (-1: not eliminated); [m] the row's eliminated variable (-1: none), slot of that entry in A.val
  double *rhoe, *ecoef, *edinv, *xte;   // [m] effective rho of the row, A(i, ecol[i]), 1 / D_y, x~ of the row's eliminated variable
  double *D, *Dinv, *E, *Einv;
  double *part_rz, *part_rr, *part_bb, *part_pkp, *part_s0, *part_s1, *part_s2, *part_gam, *part_del;
  const double *fin_rr;  // where k_admm_finalize finds the partials of the last ||r||^2, and how many
  int fin_cnt;
  double *scal;          // reduction outputs (see SC_* below)
  double *part_h;        // huge A rows: [row][workgroup of k_cg_A] partial dots
  State  *st;
  const Params *prm;
#ifdef OSQP_AMD_TIMELINE
  unsigned long long *tl;  // make TIMELINE=1: [0] = count, then one word per kernel start: id << 56 | wall_clock64 (100 MHz)
#endif
};

#ifdef OSQP_AMD_TIMELINE
#define TL_CAP (1 << 20)
#define TL_MARK(c, id) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long k_ = atomicAdd((c).tl, 1ull); \
    if (k_ < TL_CAP - 2) (c).tl[1 + k_] = ((unsigned long long)(id) << 56) | (wall_clock64() & 0x00FFFFFFFFFFFFFFull); } } while (0)
#else
#define TL_MARK(c, id) do { } while (0)
#endif

enum {  // slots of Ctx::scal (max slots are bit patterns of non-negative doubles)
  SC_PRI_U, SC_PRI_S, SC_Z_U, SC_Z_S, SC_AX_U, SC_AX_S,
  SC_DUA_U, SC_DUA_S, SC_Q_U, SC_Q_S, SC_ATY_U, SC_ATY_S, SC_PX_U, SC_PX_S,
  SC_DYN_U, SC_DYN_S, SC_DXN_U, SC_DXN_S,
  SC_ATDY_U, SC_ATDY_S, SC_PDX_U, SC_PDX_S, SC_ADX_VIOL,
  SC_OBJ, SC_DYLHS, SC_QDX,
  SC_COUNT
};
#define SCI(x) ((x) * 16)   // one 128-byte line per slot: atomics on different slots do not serialise

// ---------------------------------------------------------------------------
// wavefront / workgroup reductions (wave = 64 lanes on gfx950)
// ---------------------------------------------------------------------------
// Sum over the 64 lanes, result in every lane.  Four DPP exchanges inside the 16-lane rows
// (quad xor 1, quad xor 2, half-row mirror, row mirror: no LDS traffic, unlike ds_bpermute
// shuffles), then the four row totals are read out by lane and added in a fixed order.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// Sum over the workgroup, result broadcast to every thread.  `red` holds >= 5
// doubles of LDS.  Fixed tree => deterministic.
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) red[4] = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return red[4];
}

// Three sums in one pass (one barrier pair instead of three); results broadcast.
// `red` holds >= 16 doubles.
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *red) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { red[w] = a; red[4 + w] = b; red[8 + w] = c; }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
  c = (red[8] + red[9]) + (red[10] + red[11]);
}

// Re-reduce up to three arrays of per-workgroup partials (same length).
template <int NA>
__device__ __forceinline__ void reduce_parts(const double *a0, const double *a1,
                                             const double *a2, int count,
                                             double *red, double out[NA]) {
  double s0 = 0, s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < count; i += TB) {
    s0 += a0[i];
    if (NA > 1) s1 += a1[i];
    if (NA > 2) s2 += a2[i];
  }
  if (NA == 1) { out[0] = block_sum(s0, red); return; }
  block_sum3(s0, s1, s2, red);
  out[0] = s0; out[1] = s1;
  if (NA > 2) out[2] = s2;
}

__device__ __forceinline__ void atomic_max_pos(double *slot, double v) {
  // v >= 0: IEEE ordering of non-negative doubles equals unsigned ordering
  atomicMax(reinterpret_cast<unsigned long long *>(slot),
            static_cast<unsigned long long>(__double_as_longlong(v)));
}

__device__ __forceinline__ void block_max_to(double v, double *slot, double *red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomic_max_pos(slot, fmax(fmax(red[0], red[1]), fmax(red[2], red[3])));
}

// ---------------------------------------------------------------------------
// CSR-stream building blocks
// ---------------------------------------------------------------------------
// Stage products of the block's entries with one or two input vectors.
// GATHER_P: input 0 is formed on the fly as zz[c] + beta*pold[c] for c < n
// (the PCG direction update fused into the gather).
template <int NV>
__device__ __forceinline__ void stage_products(const DevMat &Mx, const RowBlk b,
                                               const double *in0, const double *in1,
                                               double *l0, double *l1) {
  // all index/value loads of the chunk are issued first, then all gathers, then
  // the LDS writes: up to 2*MAX_CHUNK/TB independent loads in flight per lane
  constexpr int E = MAX_CHUNK / TB;
  const int cnt = b.k1 - b.k0;
  int cc[E]; double vv[E], x0[E], x1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = threadIdx.x + e * TB;
    if (k < cnt) { cc[e] = Mx.col[b.k0 + k]; vv[e] = Mx.val[b.k0 + k]; }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = threadIdx.x + e * TB;
    if (k < cnt) { x0[e] = in0[cc[e]]; if (NV == 2) x1[e] = in1[cc[e]]; }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = threadIdx.x + e * TB;
    if (k < cnt) { l0[k] = vv[e] * x0[e]; if (NV == 2) l1[k] = vv[e] * x1[e]; }
  }
}

// Sequential (reference-order) sum of one row segment held in LDS.
__device__ __forceinline__ double row_sum(const double *l, int a, int b) {
  double s = 0.0;
  for (int k = a; k < b; ++k) s += l[k];
  return s;
}

// Row segment summed by L adjacent lanes (strided) + xor tree: used by the PCG
// kernels, where the summation order is free; every lane returns the sum.
__device__ __forceinline__ double row_sum_par(const double *l, int a, int b, int lane, int L) {
  double s = 0.0;
  for (int k = a + lane; k < b; k += L) s += l[k];
  // L is a power of two <= 64; the first four butterfly steps stay inside a 16-lane row (DPP)
  if (L >= 2) s += dpp_move<0xB1>(s);
  if (L >= 4) s += dpp_move<0x4E>(s);
  if (L >= 8) s += dpp_move<0x141>(s);
  if (L >= 16) s += dpp_move<0x140>(s);
  if (L >= 32) s += __shfl_xor(s, 16, 64);
  if (L == 64) s += __shfl_xor(s, 32, 64);
  return s;
}
// lanes per row for a block: as many as fit one pass over its rows (1..64)
__device__ __forceinline__ int lanes_for(int nrows) {
  int L = 64;
  while (L > 1 && L * nrows > TB) L >>= 1;
  return L;
}

// A row that does not fit one chunk: whole-workgroup strided reduction of
// sum_k val[k]*in[col[k]] over [ka, kb).  (Summation order differs from the
// sequential reference order; only rows longer than MAX_CHUNK take this path.)
__device__ __forceinline__ double long_row_dot(const DevMat &Mx, int ka, int kb,
                                               const double *in, double *red) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = ka + threadIdx.x;
  for (; k + 3 * TB < kb; k += 4 * TB) {          // four independent index/value/gather chains
    const int c0 = Mx.col[k], c1 = Mx.col[k + TB], c2 = Mx.col[k + 2 * TB], c3 = Mx.col[k + 3 * TB];
    const double v0 = Mx.val[k], v1 = Mx.val[k + TB], v2 = Mx.val[k + 2 * TB], v3 = Mx.val[k + 3 * TB];
    s0 += v0 * in[c0]; s1 += v1 * in[c1]; s2 += v2 * in[c2]; s3 += v3 * in[c3];
  }
  for (; k < kb; k += TB) s0 += Mx.val[k] * in[Mx.col[k]];
  return block_sum((s0 + s1) + (s2 + s3), red);
}

// One wavefront per long row; no LDS, no barrier.  The row is taken in chunks of 64 x ROW_U entries: every index and value
// load of a chunk is issued before the first gather, every gather before the first product, lanes beyond the row's end
// are masked (clamped address, zero value).  Measured on the Lasso's pass over A (75 MB with 16-bit column ids): 24.5 us, of which
// 7.7 us are the gathers (16.8 us with the gathered value replaced by a constant: 4.5 TB/s for the stream alone) -- 64 scattered
// 8-byte reads per wave instruction keep the vector L1 busy far longer than their bytes; round 2's form (four chains per lane and
// a one-load-at-a-time remainder loop) took the same time, so neither loop shape is the bound.
#define ROW_U 12
template <bool NARROW, class F>
__device__ __forceinline__ double wave_row_dot_t(const DevMat &Mx, int ka, int kb, F xval) {
  const int lane = threadIdx.x & 63;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int k0 = ka + lane; k0 - lane < kb; k0 += 64 * ROW_U) {
    int cc[ROW_U]; double vv[ROW_U], xx[ROW_U];
#pragma unroll
    for (int q = 0; q < ROW_U; ++q) {
      const int kc = min(k0 + 64 * q, kb - 1);           // (unconditional loads at a clamped address: no branch per entry)
      cc[q] = NARROW ? (int)Mx.col16[kc] : Mx.col[kc];
      vv[q] = Mx.val[kc];
    }
#pragma unroll
    for (int q = 0; q < ROW_U; ++q) xx[q] = xval(cc[q]);
#pragma unroll
    for (int q = 0; q < ROW_U; ++q) if (k0 + 64 * q >= kb) vv[q] = 0.0;
#pragma unroll
    for (int q = 0; q < ROW_U; q += 4) { s0 += vv[q] * xx[q]; s1 += vv[q + 1] * xx[q + 1]; s2 += vv[q + 2] * xx[q + 2]; s3 += vv[q + 3] * xx[q + 3]; }
  }
  return wave_sum((s0 + s1) + (s2 + s3));      // valid in lane 0
}
template <class F>
__device__ __forceinline__ double wave_row_dot(const DevMat &Mx, int ka, int kb, F xval) {
  return Mx.col16 ? wave_row_dot_t<true>(Mx, ka, kb, xval) : wave_row_dot_t<false>(Mx, ka, kb, xval);
}

// y = P_b x_b for one dense diagonal block by the whole workgroup.  P_b is symmetric, so
// y_r = sum_j P_b[j][r] x_j: a lane walks DOWN its columns while the wavefront reads row j contiguously.
// Rows are stored with an even pitch (b rounded up; never a multiple of 32 doubles -- rows that all start on the
// same 256-byte phase camp on a few memory channels: 19.5 instead of 12 us at pitch 128), so one wave instruction
// fetches a whole row with 16-byte loads: lane l owns columns 2l and 2l+1 (measured against 8-byte loads at pitch b and
// against v_mfma_f64_16x16x4_f64 in tools/mfma_dense_probe.hip: 7.65 vs 9.21 vs 9.83 us for the 400 blocks of
// config 5, profiles/r02_mfma_dense_probe.json -- one right-hand side leaves the matrix cores nothing to do).
// Wavefront w takes the rows j = w, w+4, ...; every load of the block is issued before anything is consumed
// (<= 32 rows per wavefront); the four partial vectors meet in LDS.  No index loads, no gathers.
// Returns y_r for r = threadIdx.x < b (0 otherwise); scratch: 5*DENSE_MAX doubles, x_b is left
// in scratch[4*DENSE_MAX ...].  Ends with a barrier; the caller adds one before reusing scratch.
// EARLY: the input element comes out of a chain of dependent loads (k_blk_apply_back).  vmcnt completes in order, so behind the
// 32 block loads every link of the chain would wait for all of them; the chain runs first, alone, and is waited for.
template <bool EARLY = false, class XL>
__device__ __forceinline__ double dense_block_mv_x(const DenseP &dP, const DenseBlk d, XL xload, double *scratch) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double2 *dv = reinterpret_cast<const double2 *>(dP.val + d.off);
  const int hp = d.pitch >> 1;
  double xe = 0.0;
  if (EARLY) {
    if ((int)threadIdx.x < d.b) xe = xload(d.c0 + (int)threadIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  double2 v[DENSE_MAX / 4];
#pragma unroll
  for (int q = 0; q < DENSE_MAX / 4; ++q) {
    const int j = w + 4 * q;
    v[q] = (j < d.b && lane < hp) ? dv[(size_t)j * hp + lane] : double2{0.0, 0.0};
  }
  double *xl = scratch + 4 * DENSE_MAX;
  if ((int)threadIdx.x < DENSE_MAX) xl[threadIdx.x] = EARLY ? xe : ((int)threadIdx.x < d.b ? xload(d.c0 + (int)threadIdx.x) : 0.0);
  __syncthreads();
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int q = 0; q < DENSE_MAX / 4; ++q) {
    const double u = xl[min(w + 4 * q, DENSE_MAX - 1)];
    a0 += v[q].x * u; a1 += v[q].y * u;
  }
  scratch[w * DENSE_MAX + 2 * lane] = a0; scratch[w * DENSE_MAX + 2 * lane + 1] = a1;
  __syncthreads();
  const int r = threadIdx.x;
  return r < d.b ? (scratch[r] + scratch[DENSE_MAX + r]) + (scratch[2 * DENSE_MAX + r] + scratch[3 * DENSE_MAX + r]) : 0.0;
}
__device__ __forceinline__ double dense_block_mv(const DenseP &dP, const DenseBlk d, const double *x, double *scratch) {
  return dense_block_mv_x<false>(dP, d, [x](int j) { return x[j]; }, scratch);
}

#define LDS_DECL(NV)                                   \
  __shared__ double lprod[(NV) * MAX_CHUNK];           \
  __shared__ double red[16]

// ---------------------------------------------------------------------------
// PCG kernels
// ---------------------------------------------------------------------------
// Every row kernel walks its row blocks with a workgroup-stride loop, so the
// number of dot-product partials equals the launch grid (<= MAX_PARTS).

// First kernel of an ADMM iteration: right-hand side b = sigma x - q + A'(rho z - y)
// (compute_rhs folded into the reduced system), initial residual r = b - K x~0
// with the warm start x~0 (the previous x~, linearly extrapolated by k_admm_finalize once the
// iterates move smoothly: ADMM iterates follow a linear recurrence, so 2 x~_k - x~_{k-1} is 2-10x
// closer to x~_{k+1} than x~_k is), preconditioned residual and the three start-up dot
// products.  One dual-stream pass over M.
__global__ void __launch_bounds__(TB) k_vd(Ctx c) {
  for (int k = blockIdx.x * TB + threadIdx.x; k < c.n + c.m; k += gridDim.x * TB) c.vd[k] = c.vb[k] - c.vx[k];
}
template <bool DENSE>
__global__ void __launch_bounds__(TB) k_pcg_init(Ctx c, int bench) {
  State *st = c.st;
  TL_MARK(c, 1);
  // A solve that ran out of unrolled PCG iterations (`stalled`, raised by k_admm_finalize) is continued by
  // this graph launch: no new right-hand side, the iteration kernels below pick the recurrences up where they
  // stopped (K is even, so the parity of every ping-pong buffer is preserved).
  if (!bench) {
    if (st->stalled || st->res_fail) return;
    if (st->admm_done >= st->admm_target) { if (blockIdx.x == 0 && threadIdx.x == 0) st->run = 0; return; }
  }
  LDS_DECL(2);
  const Params prm = *c.prm;
  double prz = 0, prr = 0, pbb = 0;
  // rows inside dense diagonal blocks of P: P x~0 by the block product, A' part from the remainder
  // matrix; all other rows below (Mm = remainder matrix then, M itself otherwise)
  const DevMat &Mm = DENSE ? c.Mk : c.M;
  if (DENSE)
  for (int db = blockIdx.x; db < c.dP.nblk; db += gridDim.x) {
    const DenseBlk d = c.dP.blk[db];
    // plain_rhs (block-direct engines): no P x~0 -- the 50 MB pass over the blocks of P that the residual form needs -- and the
    // low-rank rows' terms stay out of the right-hand side (k_blk_finish / k_cpl_solve put them back in closed form)
    double sA = c.plain_rhs ? 0.0 : dense_block_mv(c.dP, d, c.vx, lprod);
    if ((int)threadIdx.x < d.b) {
      const int j = d.c0 + threadIdx.x;
      double sB = 0.0;
      if (c.plain_rhs) {
        for (int k = Mm.rowptr[j]; k < Mm.rowptr[j + 1]; ++k) { const int cc = Mm.col[k]; if (c.plain_skip && c.plain_skip[cc - c.n] >= 0) continue; sB += Mm.val[k] * c.vb[cc]; }
      } else {
        for (int k = Mm.rowptr[j]; k < Mm.rowptr[j + 1]; ++k) { const int cc = Mm.col[k]; const double v = Mm.val[k]; sA += v * c.vx[cc]; sB += v * c.vb[cc]; }
        for (int k = 0; k < c.nh; ++k) { const double hc = c.hcol[(size_t)k * c.n + j]; sA += hc * c.vx[c.n + c.hrow[k]]; sB += hc * c.vb[c.n + c.hrow[k]]; }   // huge rows of A live outside Mk
      }
      const double base = prm.use_cvec ? c.cvec[j] : (prm.sigma * c.xy[j] - c.q[j]);
      const double bj = base + sB;
      const double rj = c.plain_rhs ? bj : bj - prm.sigma * c.vx[j] - sA;
      const double zj = c.minv[j] * rj;
      c.init_r[(size_t)j * c.init_stride] = rj;
      c.init_z[j] = zj;
      if (c.u0pos) c.u0pos[c.u0map[j]] = zj;
      prz += rj * zj; prr += rj * rj; pbb += bj * bj;
    }
    __syncthreads();
  }
  // row j of the start-up given sA = (M [x~0 | rho z~0])_j and sB = (M [0 | rho z - y])_j, by one thread
  auto row_start = [&](int j, double sA, double sB) {
    if (DENSE) for (int k = 0; k < c.nh; ++k) { const double hc = c.hcol[(size_t)k * c.n + j]; sA += hc * c.vx[c.n + c.hrow[k]]; sB += hc * c.vb[c.n + c.hrow[k]]; }
    const double base = prm.use_cvec ? c.cvec[j] : (prm.sigma * c.xy[j] - c.q[j]);
    const bool gone = c.nelim && c.erow[j] >= 0;           // eliminated from the system: no residual, no direction
    const double bj = gone ? 0.0 : base + sB;
    const double rj = gone ? 0.0 : bj - prm.sigma * c.vx[j] - sA;
    const double zj = c.minv[j] * rj;
    c.init_r[(size_t)j * c.init_stride] = rj;
    c.init_z[j] = zj;
    if (c.u0pos) c.u0pos[c.u0map[j]] = zj;
    prz += rj * zj; prr += rj * rj; pbb += bj * bj;
  };
  for (int bi = blockIdx.x; bi < Mm.nstream; bi += gridDim.x) {
    const RowBlk b = Mm.blk[bi];
    stage_products<2>(Mm, b, c.vx, c.vb, lprod, lprod + MAX_CHUNK);
    __syncthreads();
    // the summation order is free here (PCG-internal): several lanes per row
    const int RL = lanes_for(b.r1 - b.r0), rg = threadIdx.x / RL, rlane = threadIdx.x % RL;
    for (int j = b.r0 + rg; j < b.r1; j += TB / RL) {
      const int a0 = Mm.rowptr[j] - b.k0, a1 = Mm.rowptr[j + 1] - b.k0;
      const double sA = row_sum_par(lprod, a0, a1, rlane, RL);
      const double sB = row_sum_par(lprod + MAX_CHUNK, a0, a1, rlane, RL);
      if (rlane == 0) row_start(j, sA, sB);
    }
    __syncthreads();
  }
  // long rows: one wavefront each, both products in one pass over the row (two whole-workgroup reductions per row made this
  // kernel 44 us on the Lasso)
  for (int bi = Mm.nstream + blockIdx.x * (TB / 64) + (threadIdx.x >> 6); bi < Mm.nblk; bi += gridDim.x * (TB / 64)) {
    const RowBlk lb = Mm.blk[bi];
    const int lane = threadIdx.x & 63;
    if (c.vd) {              // r_j = base - sigma x~0_j + (M (vb - vx))_j: one gather per entry
      const double acc = wave_row_dot(Mm, lb.k0, lb.k1, [&](int cc) { return c.vd[cc]; });
      if (lane == 0) row_start(lb.r0, -acc, 0.0);
      continue;
    }
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    const bool narrow = Mm.col16 != nullptr;
    constexpr int UI = 8;
    for (int k0 = lb.k0 + lane; k0 - lane < lb.k1; k0 += 64 * UI) {
      int cc[UI]; double vv[UI], xa[UI], xb[UI];
#pragma unroll
      for (int q = 0; q < UI; ++q) {
        const int kc = min(k0 + 64 * q, lb.k1 - 1);
        cc[q] = narrow ? (int)Mm.col16[kc] : Mm.col[kc];
        vv[q] = Mm.val[kc];
      }
#pragma unroll
      for (int q = 0; q < UI; ++q) { xa[q] = c.vx[cc[q]]; xb[q] = c.vb[cc[q]]; }
#pragma unroll
      for (int q = 0; q < UI; ++q) if (k0 + 64 * q >= lb.k1) vv[q] = 0.0;
#pragma unroll
      for (int q = 0; q < UI; q += 2) { a0 += vv[q] * xa[q]; b0 += vv[q] * xb[q]; a1 += vv[q + 1] * xa[q + 1]; b1 += vv[q + 1] * xb[q + 1]; }
    }
    const double sA = wave_sum(a0 + a1), sB = wave_sum(b0 + b1);
    if (lane == 0) row_start(lb.r0, sA, sB);
  }
  block_sum3(prz, prr, pbb, red);
  if (threadIdx.x == 0) {
    c.part_rz[blockIdx.x] = prz; c.part_rr[blockIdx.x] = prr; c.part_bb[blockIdx.x] = pbb;
    if (blockIdx.x == 0) {
      st->run = 1; st->done = 0; st->neg_curv = 0; st->iters[0] = 0; st->iters[1] = 0;
      if (st->hist_r < (1 << 20)) st->hist_r += 1;
    }
  }
}

// ---------------------------------------------------------------------------
// Chronopoulos-Gear PCG: two kernels per iteration instead of three.
//   u = Minv r, w = K u, gamma = (r,u), delta = (w,u)
//   beta = gamma/gamma_old, alpha = gamma / (delta - beta*gamma/alpha_old)
//   p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s
// k_cg_A does the vector update for its own slice AND t = rho.(A u_new), with
// u_new recomputed from (r_old, w, s_old, Minv) at every gathered column (same
// expression, hence the same bits, as the value the owner stores); r and s are
// ping-ponged on the parity of `it` so gathers never see half-updated data.
// k_cg_B applies w = [P|A'][u;t] + sigma u and emits the three dot partials.
// flags: bit2 = benchmark, bit3 = "pre" pass (pure operator apply on u0, first convergence test),
//        bit4 / bit5 = vector update only / operator apply only (split mode)
// ---------------------------------------------------------------------------
#define EPT (MAX_CHUNK / TB)

// Scalars of one Chronopoulos-Gear step from the three reduced dot products.
struct CgStep { double alpha, beta; bool first, stop, conv, bad; };
__device__ __forceinline__ CgStep cg_step(double rr, double gam, double del, double gam_old, double alp_old,
                                          double tol2, int iters_prev, int neg, const Params &prm, bool bench) {
  CgStep s;
  s.first = gam_old == 0.0;              // the pre pass zeroes gam[1]: first update of this solve
  s.bad = false;
  if (s.first) { s.beta = 0.0; s.alpha = gam / del; }
  else {
    s.beta = gam / gam_old;
    const double den = del - s.beta * gam / alp_old;     // = p'Kp in exact arithmetic
    if (!(den > 0.0) && del > 0.0) {
      // u'Ku > 0 but the two-term recurrence lost positivity.  Near the stop (||r|| within 1e4 of the
      // tolerance) this is the rounding floor of an ill-conditioned system: restart the directions
      // from u (p = u, s = w).  Anywhere else -- and always during the setup-time convexity probe --
      // it is what an indefinite K produces, and is reported as negative curvature.
      if (!prm.no_restart && rr <= 1e8 * tol2) { s.first = true; s.beta = 0.0; s.alpha = gam / del; }
      else { s.alpha = gam / den; s.bad = true; }
    } else s.alpha = gam / den;
  }
  if (bench) { s.beta = 0.5; s.alpha = 1e-3; s.bad = false; }
  s.conv = rr <= tol2;
  s.bad = s.bad || !(s.alpha > 0.0) || !(del > 0.0);      // breakdown / negative curvature
  s.stop = !bench && (s.conv || iters_prev >= prm.pcg_max_iter || neg || s.bad);
  return s;
}
__global__ void __launch_bounds__(TB) k_cg_A(Ctx c, int it, int flags) {
  State *st = c.st;
  TL_MARK(c, 2);
  const bool bench = flags & 4, upd_only = flags & 16, apply_only = flags & 32;
  const bool pre = (flags & 8) || apply_only;   // gather u directly (no recompute)
  // An unrolled slot whose turn never comes costs a whole kernel if it prefetches its operands before it
  // looks at the flags (the wave cannot retire before its loads land: 4-6 us instead of 1.4, measured with
  // `make TIMELINE=1`).  Slots at or beyond the PREVIOUS solve's iteration count (left in the state by
  // k_admm_finalize, same cache line as the flags) therefore look first; so does a graph that continues a
  // stalled solve.  The count moves by about one between consecutive ADMM iterations.  Looking first costs a
  // dependent load before anything else is issued, so slots below the smallest count of the last window
  // (baked into the graph node: flags >> 8) skip it.
  if (!bench && it >= (flags >> 8) && (it >= st->iters_last || st->stalled) && (!st->run || st->done)) return;
  // ---- issue every independent load before looking at the flags ----
  const int stalled = st->stalled, run = st->run, done = st->done, neg = st->neg_curv;
  const int iters_prev = st->iters[(it + 1) & 1];
  const double tol2_old = st->tol2, gam_old = st->gam[(it + 1) & 1], alp_old = st->alp[(it + 1) & 1];
  const Params prm = *c.prm;
  const bool has_blk = (int)blockIdx.x < c.A.nstream;
  RowBlk b = {0, 0, 0, 0};
  if (has_blk) b = c.A.blk[blockIdx.x];
  const int cnt = b.k1 - b.k0;
  const bool small = has_blk;
  const G4 *gold = c.g4 + (size_t)((it + 1) & 1) * c.n;
  G4 *gnew = c.g4 + (size_t)(it & 1) * c.n;
  int ecol[EPT]; double eval[EPT], g0[EPT], g1[EPT], g2[EPT], g3[EPT];
  int RL = lanes_for(b.r1 - b.r0);
  int rg = threadIdx.x / RL, rlane = threadIdx.x % RL;
  int rp0 = 0, rp1 = 0;
  double prho = 0.0;                   // rho of this lane group's first row
  if (small) {
    if (b.r0 + rg < b.r1) { rp0 = c.A.rowptr[b.r0 + rg]; rp1 = c.A.rowptr[b.r0 + rg + 1]; prho = c.rhoe[b.r0 + rg]; }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int k = threadIdx.x + e * TB;
      if (k < cnt) { ecol[e] = c.A.col[b.k0 + k]; eval[e] = c.A.val[b.k0 + k]; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int k = threadIdx.x + e * TB;
      if (k < cnt) {
        const int cc = ecol[e];
        if (pre) g0[e] = c.ut[cc];
        else { const G4 g = gold[cc]; g0[e] = g.r; g1[e] = g.w; g2[e] = g.s; g3[e] = g.m; }
      }
    }
  }
  // dot-product partials of the previous kernel (pre: rr, bb ; else: rr, gamma, delta)
  double q0 = 0, q1 = 0, q2 = 0;
  if (!apply_only) {
    const double *a1 = pre ? c.part_bb : c.part_gam;
    for (int i = threadIdx.x; i < c.gridM; i += TB) { q0 += c.part_rr[i]; q1 += a1[i]; if (!pre) q2 += c.part_del[i]; }
  }
  const int j0 = blockIdx.x * TB + threadIdx.x;
  double o_u = 0, o_w = 0, o_p = 0, o_s = 0, o_r = 0, o_x = 0, o_m = 0;
  if (!pre && j0 < c.n) {
    const G4 g = gold[j0];
    o_u = c.ut[j0]; o_w = g.w; o_p = c.pdir[j0]; o_s = g.s; o_r = g.r; o_x = gam_old == 0.0 ? c.vx[j0] : c.va[j0]; o_m = g.m;
  }
  if (!bench && (!run || done || ((flags & 8) && stalled))) return;   // pre pass: skipped when this graph continues a solve
  LDS_DECL(1);
  double sums[3];
  bool first = false;
  double alpha = 0.0, beta = 0.0, tol2 = tol2_old;
  sums[0] = q0; sums[1] = q1; sums[2] = q2;
  if (!apply_only) block_sum3(sums[0], sums[1], sums[2], red);
  if (apply_only) {
    // second half of a split iteration: the update kernel before this one did the scalars
  } else if (pre) {
    tol2 = fmax(prm.eps_rel * prm.eps_rel * sums[1], prm.eps_abs * prm.eps_abs);
    if (!bench && sums[0] <= tol2) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->tol2 = tol2; }
      return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->tol2 = tol2; st->gam[1] = 0.0; st->alp[1] = 0.0; }
  } else {
    const double gam = sums[1];
    const bool fresh = gam_old == 0.0;
    const CgStep cs = cg_step(sums[0], gam, sums[2], gam_old, alp_old, tol2, iters_prev, neg, prm, bench);
    first = cs.first; alpha = cs.alpha; beta = cs.beta;
    if (cs.stop) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = cs.conv ? 1 : 2; if (cs.bad && !cs.conv) st->neg_curv = 1; }
      return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->gam[it & 1] = gam; st->alp[it & 1] = alpha; st->iters[it & 1] = iters_prev + 1;
    }
    // own slice: p, s, x, r, u (first element per thread was prefetched above)
    for (int j = j0; j < c.n; j += gridDim.x * TB) {
      double uj, wj, po, so, ro, xo, mi;
      if (j == j0) { uj = o_u; wj = o_w; po = o_p; so = o_s; ro = o_r; xo = o_x; mi = o_m; }
      else { const G4 g = gold[j]; uj = c.ut[j]; wj = g.w; po = c.pdir[j]; so = g.s; ro = g.r; xo = fresh ? c.vx[j] : c.va[j]; mi = g.m; }
      const double pj = first ? uj : (uj + beta * po);
      const double sj = first ? wj : (wj + beta * so);
      const double rj = ro - alpha * sj;
      c.pdir[j] = pj; gnew[j].s = sj; gnew[j].r = rj;
      c.va[j] = xo + alpha * pj;
      c.ut[j] = mi * rj;
    }
  }
  if (upd_only) return;
  // ---- t = rho . (A u_new) ----
  double *t = c.ut + c.n;
  for (int bi = blockIdx.x; bi < c.A.nstream; bi += gridDim.x) {
    if (bi != (int)blockIdx.x) { b = c.A.blk[bi]; RL = lanes_for(b.r1 - b.r0); rg = threadIdx.x / RL; rlane = threadIdx.x % RL; }
    const int cn = b.k1 - b.k0;
    if (bi != (int)blockIdx.x) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int k = threadIdx.x + e * TB;
        if (k < cn) { ecol[e] = c.A.col[b.k0 + k]; eval[e] = c.A.val[b.k0 + k]; }
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int k = threadIdx.x + e * TB;
        if (k < cn) {
          const int cc = ecol[e];
          if (pre) g0[e] = c.ut[cc];
          else { const G4 g = gold[cc]; g0[e] = g.r; g1[e] = g.w; g2[e] = g.s; g3[e] = g.m; }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int k = threadIdx.x + e * TB;
      if (k < cn) {
        double uv;
        if (pre) uv = g0[e];
        else {
          const double sj = first ? g1[e] : (g1[e] + beta * g2[e]);
          uv = g3[e] * (g0[e] - alpha * sj);
        }
        lprod[k] = eval[e] * uv;
      }
    }
    __syncthreads();
    for (int i = b.r0 + rg; i < b.r1; i += TB / RL) {
      int a0, a1;
      if (bi == (int)blockIdx.x && i == b.r0 + rg) { a0 = rp0; a1 = rp1; }
      else { a0 = c.A.rowptr[i]; a1 = c.A.rowptr[i + 1]; }
      const double acc = row_sum_par(lprod, a0 - b.k0, a1 - b.k0, rlane, RL);
      if (rlane == 0) t[i] = ((bi == (int)blockIdx.x && i == b.r0 + rg) ? prho : c.rhoe[i]) * acc;
    }
    __syncthreads();
  }
  // long rows: one wavefront each
  for (int bi = c.A.nstream + blockIdx.x * (TB / 64) + (threadIdx.x >> 6); bi < c.A.nwave; bi += gridDim.x * (TB / 64)) {
    const RowBlk lb = c.A.blk[bi];
    double acc;
    if (pre) acc = wave_row_dot(c.A, lb.k0, lb.k1, [&](int cc) { return c.ut[cc]; });
    else acc = wave_row_dot(c.A, lb.k0, lb.k1, [&](int cc) {
      const G4 g = gold[cc];
      const double sj = first ? g.w : (g.w + beta * g.s);
      return g.m * (g.r - alpha * sj);
    });
    if ((threadIdx.x & 63) == 0) t[lb.r0] = c.rhoe[lb.r0] * acc;
  }
  // huge rows (a budget constraint over every variable, ...): every workgroup takes a
  // strided slice and leaves one partial; k_huge_reduce adds them in a fixed order
  for (int bi = c.A.nwave; bi < c.A.nblk; ++bi) {
    const RowBlk hb = c.A.blk[bi];
    double s0 = 0.0, s1 = 0.0;
    int k = hb.k0 + blockIdx.x * TB + threadIdx.x;
    const int stride = gridDim.x * TB;
    auto xv = [&](int cc) -> double {
      if (pre) return c.ut[cc];
      const G4 g = gold[cc];
      const double sj = first ? g.w : (g.w + beta * g.s);
      return g.m * (g.r - alpha * sj);
    };
    for (; k + stride < hb.k1; k += 2 * stride) {
      const int c0 = c.A.col[k], c1 = c.A.col[k + stride];
      const double v0 = c.A.val[k], v1 = c.A.val[k + stride];
      s0 += v0 * xv(c0); s1 += v1 * xv(c1);
    }
    if (k < hb.k1) s0 += c.A.val[k] * xv(c.A.col[k]);
    const double tot = block_sum(s0 + s1, red);
    if (threadIdx.x == 0) c.part_h[(size_t)(bi - c.A.nwave) * gridDim.x + blockIdx.x] = tot;
  }
}

// t_i = rho_i * (sum of the k_cg_A partials) for the huge rows of A; one workgroup per row.
__global__ void __launch_bounds__(TB) k_huge_reduce(Ctx c, int flags) {
  const State *st = c.st;
  if (!(flags & 4) && (!st->run || st->done)) return;
  __shared__ double red[16];
  const int row = c.A.blk[c.A.nwave + blockIdx.x].r0;
  const double *part = c.part_h + (size_t)blockIdx.x * c.gridA;
  double s = 0.0;
  for (int i = threadIdx.x; i < c.gridA; i += TB) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) c.ut[c.n + row] = c.rhoe[row] * s;
}

// DENSE: P has dense diagonal blocks (their column walk keeps 64 loads per lane in flight and
// needs ~190 registers; the plain instantiation stays lean for the sparse stream path).
template <bool DENSE, bool FOLD>
__global__ void __launch_bounds__(TB) k_cg_B(Ctx c, int it, int flags) {
  State *st = c.st;
  TL_MARK(c, 3);
  const bool bench = flags & 4;
  if (!bench && it >= (flags >> 8) && (it >= st->iters_last || st->stalled) && (!st->run || st->done)) return;     // see k_cg_A
  const int run = st->run, done = st->done, skip_pre = it < 0 && st->stalled;
  const double sigma = c.prm->sigma;
  const bool has_blk = (int)blockIdx.x < c.Mk.nstream;
  RowBlk b = {0, 0, 0, 0};
  if (has_blk) b = c.Mk.blk[blockIdx.x];
  const int cnt = b.k1 - b.k0;
  const bool small = has_blk;
  double ev[EPT];
  int RL = lanes_for(b.r1 - b.r0);
  int rg = threadIdx.x / RL, rlane = threadIdx.x % RL;
  int rp0 = 0, rp1 = 0;
  G4 *gc = c.g4 + (size_t)(it & 1) * c.n;
  double pu = 0.0, pr = 0.0;           // u_j, r_j of this lane group's first row (off the tail's critical path)
  // huge rows of A folded in: this thread's share of the k_cg_A partials of each (summed below), and the
  // entries hcol_k[j] of this lane group's first row
  double hs[MAX_HUGE_FOLD], ph[MAX_HUGE_FOLD], hrho[MAX_HUGE_FOLD];
#pragma unroll
  for (int k = 0; k < MAX_HUGE_FOLD; ++k) {
    hs[k] = 0.0; ph[k] = 0.0; hrho[k] = 0.0;
    if (FOLD && k < c.nh) {
      hrho[k] = c.rhoe[c.hrow[k]];
      for (int i = threadIdx.x; i < c.gridA; i += TB) hs[k] += c.part_h[(size_t)k * c.gridA + i];
    }
  }
  if (small) {
    if (b.r0 + rg < b.r1) {
      rp0 = c.Mk.rowptr[b.r0 + rg]; rp1 = c.Mk.rowptr[b.r0 + rg + 1];
      pu = c.ut[b.r0 + rg]; pr = gc[b.r0 + rg].r;
#pragma unroll
      for (int k = 0; k < MAX_HUGE_FOLD; ++k) if (FOLD && k < c.nh) ph[k] = c.hcol[(size_t)k * c.n + b.r0 + rg];
    }
    int ecol[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int k = threadIdx.x + e * TB;
      if (k < cnt) { ecol[e] = c.Mk.col[b.k0 + k]; ev[e] = c.Mk.val[b.k0 + k]; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int k = threadIdx.x + e * TB;
      if (k < cnt) ev[e] = ev[e] * c.ut[ecol[e]];
    }
  }
  if (!bench && (!run || done || skip_pre)) return;
  LDS_DECL(1);
  double th[MAX_HUGE_FOLD];            // t_h = rho_h * (A u)_h of the folded huge rows, formed by every workgroup itself
#pragma unroll
  for (int k = 0; k < MAX_HUGE_FOLD; ++k) th[k] = (FOLD && k < c.nh) ? hrho[k] * block_sum(hs[k], red) : 0.0;
  auto hterm = [&](int j, bool pref) {
    double s = 0.0;
    if (FOLD) {
#pragma unroll
      for (int k = 0; k < MAX_HUGE_FOLD; ++k) if (k < c.nh) s += (pref ? ph[k] : c.hcol[(size_t)k * c.n + j]) * th[k];
    }
    return s;
  };
  double pg = 0, pd = 0, prr = 0;
  for (int bi = blockIdx.x; bi < c.Mk.nstream; bi += gridDim.x) {
    if (bi != (int)blockIdx.x) { b = c.Mk.blk[bi]; RL = lanes_for(b.r1 - b.r0); rg = threadIdx.x / RL; rlane = threadIdx.x % RL; }
    const int cn = b.k1 - b.k0;
    if (bi == (int)blockIdx.x) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) { const int k = threadIdx.x + e * TB; if (k < cn) lprod[k] = ev[e]; }
    } else stage_products<1>(c.Mk, b, c.ut, nullptr, lprod, nullptr);
    __syncthreads();
    for (int j = b.r0 + rg; j < b.r1; j += TB / RL) {
      int a0, a1;
      if (bi == (int)blockIdx.x && j == b.r0 + rg) { a0 = rp0; a1 = rp1; }
      else { a0 = c.Mk.rowptr[j]; a1 = c.Mk.rowptr[j + 1]; }
      const double acc = row_sum_par(lprod, a0 - b.k0, a1 - b.k0, rlane, RL);
      if (rlane == 0) {
        const bool first = bi == (int)blockIdx.x && j == b.r0 + rg;
        const double uj = first ? pu : c.ut[j], rj = first ? pr : gc[j].r;
        const double wj = (c.nelim && c.erow[j] >= 0) ? 0.0 : (acc + sigma * uj) + hterm(j, first);
        gc[j].w = wj;
        pg += rj * uj; pd += wj * uj; prr += rj * rj;
      }
    }
    __syncthreads();
  }
  // long rows: one wavefront each
  for (int bi = c.Mk.nstream + blockIdx.x * (TB / 64) + (threadIdx.x >> 6); bi < c.Mk.nblk; bi += gridDim.x * (TB / 64)) {
    const RowBlk lb = c.Mk.blk[bi];
    const double acc = wave_row_dot(c.Mk, lb.k0, lb.k1, [&](int cc) { return c.ut[cc]; });
    if ((threadIdx.x & 63) == 0) {
      const int j = lb.r0;
      const double uj = c.ut[j], wj = (acc + sigma * uj) + hterm(j, false), rj = gc[j].r;
      gc[j].w = wj; pg += rj * uj; pd += wj * uj; prr += rj * rj;
    }
  }
  // dense diagonal blocks of P, one workgroup per block
  if (DENSE)
  for (int db = blockIdx.x; db < c.dP.nblk; db += gridDim.x) {
    const DenseBlk d = c.dP.blk[db];
    double acc = dense_block_mv(c.dP, d, c.ut, lprod);
    if ((int)threadIdx.x < d.b) {
      const int jrow = d.c0 + threadIdx.x;
      for (int k = c.Mk.rowptr[jrow]; k < c.Mk.rowptr[jrow + 1]; ++k) acc += c.Mk.val[k] * c.ut[c.Mk.col[k]];   // A' part
      const double uj = lprod[4 * DENSE_MAX + threadIdx.x], wj = (acc + sigma * uj) + hterm(jrow, false), rj = gc[jrow].r;
      gc[jrow].w = wj; pg += rj * uj; pd += wj * uj; prr += rj * rj;
    }
    __syncthreads();
  }
  block_sum3(pg, pd, prr, red);
  if (threadIdx.x == 0) { c.part_gam[blockIdx.x] = pg; c.part_del[blockIdx.x] = pd; c.part_rr[blockIdx.x] = prr; }
}


// Slices of A_i . x for the huge rows of A (every workgroup one strided slice, one partial
// each); the consumer adds the partials in a fixed order.  Launched only when A has huge rows.
__global__ void __launch_bounds__(TB) k_huge_dot(Ctx c, const double *x, int always) {
  if (!always && !c.st->run) return;
  if (!x) {   // x~ as k_admm_finalize will see it (the start vector when no PCG update ran)
    const State *st = c.st;
    x = (c.vx != c.va && st->iters[0] == 0 && st->iters[1] == 0) ? c.vx : c.va;
  }
  __shared__ double red[16];
  for (int bi = c.A.nwave; bi < c.A.nblk; ++bi) {
    const RowBlk hb = c.A.blk[bi];
    double s0 = 0.0, s1 = 0.0;
    int k = hb.k0 + blockIdx.x * TB + threadIdx.x;
    const int stride = gridDim.x * TB;
    for (; k + stride < hb.k1; k += 2 * stride) {
      const int c0 = c.A.col[k], c1 = c.A.col[k + stride];
      s0 += c.A.val[k] * x[c0]; s1 += c.A.val[k + stride] * x[c1];
    }
    if (k < hb.k1) s0 += c.A.val[k] * x[c.A.col[k]];
    const double tot = block_sum(s0 + s1, red);
    if (threadIdx.x == 0) c.part_h[(size_t)(bi - c.A.nwave) * gridDim.x + blockIdx.x] = tot;
  }
}
__device__ __forceinline__ double huge_row_sum(const Ctx &c, int bi, double *red) {
  const double *part = c.part_h + (size_t)(bi - c.A.nwave) * c.gridA;
  double s = 0.0;
  for (int i = threadIdx.x; i < c.gridA; i += TB) s += part[i];
  return block_sum(s, red);
}

// what the next right-hand side carries for row i: rho z - y, minus rho a b_y / D_y where the row has an eliminated variable y
__device__ __forceinline__ double elim_vb(const Ctx &c, const Params &prm, int i, double rho, double vbn, double x_y) {
  const int y = c.ecol[i];
  if (y < 0) return vbn;
  const double a = c.ecoef[i];
  const double by = (prm.use_cvec ? c.cvec[y] : (prm.sigma * x_y - c.q[y])) + a * vbn;
  return vbn - rho * a * by * c.edinv[i];
}

// Last kernel of an ADMM iteration: z~ = A x~, then update_x / update_z (+project)
// / update_y (auxil.c:185-225, proj.c:4-14) and the m-parts of the next
// right-hand side.  If the PCG has not converged within the unrolled
// iterations the iteration is left untouched and `stalled` is raised.
__global__ void __launch_bounds__(TB) k_admm_finalize(Ctx c) {
  State *st = c.st;
  TL_MARK(c, 4);
  if (!st->run || st->res_fail) return;
  LDS_DECL(1);
  const Params prm = *c.prm;
  double rr[1];
  reduce_parts<1>(c.fin_rr, nullptr, nullptr, c.fin_cnt, red, rr);
  const int iters = st->iters[0] > st->iters[1] ? st->iters[0] : st->iters[1];
  const bool conv = st->done == 1 || rr[0] <= st->tol2;
  const bool force = st->done == 2 || iters >= prm.pcg_max_iter || st->neg_curv;
  if (!conv && !force) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->stalled = 1;
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->stalled = 0;
    st->admm_done += 1;
    st->iters_last = iters;
    st->iters_total += iters;
    if (iters > st->iters_max) st->iters_max = iters;
    if (iters < st->iters_min) st->iters_min = iters;
    if (!conv) st->forced += 1;
    if (st->neg_curv) st->neg_curv_seen += 1;
  }
  const double alpha = prm.alpha, oma = 1.0 - prm.alpha;
  // start vector of the next PCG solve: [x~ | rho z~] + theta * (change since the previous ADMM
  // iteration).  No extrapolation in the first iterations after a reset (the iterates still jump),
  // half a step for a while, then the full linear step.  hist_r is advanced by k_pcg_init only.
  const int hist = st->hist_r;
  const double th = (hist < prm.ex_h0 ? 0.0 : (hist < prm.ex_h1 ? 0.5 : 1.0)) * prm.ex_theta;
  const bool ex = c.vx != c.va;
  // the start vector itself passed the stop test (no PCG update ran): x~ is the start vector
  const bool from_start = ex && iters == 0;
  const double *xts = from_start ? c.vx : c.va;
  double *x = c.xy, *y = c.xy + c.n;
  for (int j = blockIdx.x * TB + threadIdx.x; j < c.n; j += gridDim.x * TB) {
    if (c.nelim && c.erow[j] >= 0) continue;      // eliminated from the linear system: its row's thread back-substitutes and updates it below
    const double xo = x[j], xt = xts[j];
    const double xn = alpha * xt + oma * xo;
    c.dxy[j] = xn - xo;
    x[j] = xn;
    if (ex) { c.vx[j] = xt + th * (xt - c.vold[j]); c.vold[j] = xt; }
    if (from_start) c.va[j] = xt;
  }
  // update_z / update_y of row i given z~_i = (A x~)_i (without the eliminated variable's term), by one thread
  auto row_update = [&](int i, double zt) {
    const double rho = c.rho[i], rinv = c.rhoinv[i], yo = y[i], zo = c.z[i];
    const int ye = c.nelim ? c.ecol[i] : -1;
    double rz = rho * zt, xny = 0.0;
    if (ye >= 0) {
      // zt so far is (A_X x~_X)_i: the PCG vector is zero at the eliminated variable.  Back substitution
      // x~_y = (b_y - rho a (A_X x~_X)_i) / D_y with the b_y of the system just solved, then the variable's own update_x.
      const double a = c.ecoef[i], xoy = x[ye];
      const double by = (prm.use_cvec ? c.cvec[ye] : (prm.sigma * xoy - c.q[ye])) + a * (rho * zo - yo);
      const double xty = (by - rho * a * zt) * c.edinv[i];
      rz = c.rhoe[i] * zt;                     // what the operator of the reduced system applies to this row
      zt += a * xty;
      xny = alpha * xty + oma * xoy;
      c.dxy[ye] = xny - xoy; x[ye] = xny; c.xte[i] = xty;
    }
    double v = alpha * zt + oma * zo + rinv * yo;
    v = fmax(v, c.l[i]);
    const double zn = fmin(v, c.u[i]);
    const double dy = rho * (alpha * zt + oma * zo - zn);
    const double yn = yo + dy;
    c.z[i] = zn; y[i] = yn; c.dy[i] = dy; c.zt[i] = zt;
    c.va[c.n + i] = rz;
    if (ex) { c.vx[c.n + i] = rz + th * (rz - c.vold[c.n + i]); c.vold[c.n + i] = rz; }
    c.vb[c.n + i] = ye >= 0 ? elim_vb(c, prm, i, rho, rho * zn - yn, xny) : rho * zn - yn;
  };
  // stream blocks (whole rows staged through LDS)
  for (int bi = blockIdx.x; bi < c.A.nstream; bi += gridDim.x) {
    const RowBlk b = c.A.blk[bi];
    stage_products<1>(c.A, b, xts, nullptr, lprod, nullptr);
    __syncthreads();
    const int RL = lanes_for(b.r1 - b.r0), rg = threadIdx.x / RL, rlane = threadIdx.x % RL;
    for (int i = b.r0 + rg; i < b.r1; i += TB / RL) {
      const double zt = row_sum_par(lprod, c.A.rowptr[i] - b.k0, c.A.rowptr[i + 1] - b.k0, rlane, RL);
      if (rlane == 0) row_update(i, zt);
    }
    __syncthreads();
  }
  // long rows: the whole workgroup per row.  (One wavefront per row as in the PCG kernels is twice as fast on the Lasso's 10 000 rows
  // -- 2 % of its ADMM iteration -- but z~ then differs in its last bits, and a solve at the floor of attainable accuracy
  // (test_tight_tolerance_on_ill_conditioned_system, eps = 1e-9 on a cond 1e6 system) moved from 225 to 325 iterations: not worth it.)
  if (c.fin_wave_rows) {
    // dense-direct engines (dense_direct.h): one wavefront per long row with the PCG kernels' row dot -- there this pass is
    // 40 % of an ADMM iteration (72 -> 2x us at the Lasso's 10 000 rows), and the linear solves no longer stop on a residual
    // that the last bits of z~ could move
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int bi = c.A.nstream + blockIdx.x * (TB / 64) + wv; bi < c.A.nwave; bi += gridDim.x * (TB / 64)) {
      const RowBlk lb = c.A.blk[bi];
      const double zt = wave_row_dot(c.A, lb.k0, lb.k1, [&](int cc) { return xts[cc]; });
      if (lane == 0) row_update(lb.r0, zt);
    }
  } else
  for (int bi = c.A.nstream + blockIdx.x; bi < c.A.nwave; bi += gridDim.x) {
    const RowBlk lb = c.A.blk[bi];
    const double zt = long_row_dot(c.A, lb.k0, lb.k1, xts, red);
    if (threadIdx.x == 0) row_update(lb.r0, zt);
    __syncthreads();
  }
  // huge rows: partials left by k_huge_dot
  for (int bi = c.A.nwave + blockIdx.x; bi < c.A.nblk; bi += gridDim.x) {
    const double zt = huge_row_sum(c, bi, red);
    if (threadIdx.x == 0) row_update(c.A.blk[bi].r0, zt);
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------
// Resident PCG: the whole linear solve of one ADMM iteration in ONE launch.
//
// For problems whose reduced matrix K = P + sigma I + A' diag(rho) A fits the chip's register files
// (n <= 15616, nnz(K) <= 256 workgroups x 448 threads x 64 entries) every workgroup keeps a block of
// rows of K -- values and 16-bit column positions -- in registers for the whole solve, one workgroup per CU:
// wavefronts 1..7 hold the entries and multiply, wavefront 0 owns the rows' vector elements, decides and talks.
// The workgroups exchange vectors through memory INSIDE the launch instead of ending a kernel:
//   (1) an n-vector: every workgroup stores its rows write-through (sc1), drains, raises its flag; wavefront 0
//       polls the 256 flags; then all wavefronts read the vector into LDS with L1-bypassing (sc1) loads -- no
//       fence (MI355X_MICROARCH.md, visibility: the flag/sc1 row).  3.5 us alone (tools/exchange_probe.hip),
//       4.6 us in the loop, against 12.2 us for a k_cg_A/k_cg_B pair;
//   (2) three scalars per workgroup as 8-byte {32 bits, tag} granules, stored and polled with agent-scope atomics.
// Rules the protocol learned the hard way: no 128-byte line of an exchange buffer is written by two
// CUs; a tag never shares a 16-byte store with data it vouches for in the other half; polls are 4- or 8-byte atomic
// loads, never 16-byte buffer loads.
// Recurrences: pipelined CG (Ghysels-Vanroose, ONE exchange per iteration) as long as its true-residual checks
// pass, else Chronopoulos-Gear with a fresh product (the recurrences, scalars and stop rule of k_cg_A / k_cg_B:
// cg_step).  The operator is the explicit K (k_form_K re-forms its values whenever rho, sigma or the matrices change).
// Every wait is bounded in time: if the grid is not co-resident (another process or stream holds CUs) the
// launch gives up, sets State::res_fail, and the host continues with the launch-per-step kernels.
// ---------------------------------------------------------------------------
#define RES_TB 512
#define RES_PT (RES_TB - 64)   // threads that hold entries of K (wavefronts 1..7); wavefront 0 owns the rows and talks
#define RES_MAXROWS 61    // rows per workgroup: they and the three riding partials are stored by the 64 lanes of wavefront 0
#define RES_MAXN (RES_MAXROWS * 256)
#define RES_MAXPAD 16384  // longest exchanged vector (doubles, lines padded)
#define RES_MAXLD (RES_MAXPAD / 2 / RES_TB)   // 16-byte loads per thread that sweep the exchanged vector
#define RES_WAIT_TICKS 2000000LL      // 20 ms of the 100 MHz wall clock: the limit of every wait inside a resident launch (the grid may still be
                                      // arriving in the first one; a later one that long means a workgroup is gone).  Two readings 16 rounds apart
                                      // must both say so, and a reading that cannot be (negative, or hours) restarts the wait's clock.
#define RES_SLOW_TICKS 5000LL         // a wait that ended well after more than 50 us is counted (State::res_slow)
#define AUX_SC1 16
#define AUX_VOL ((int)0x80000000)     // volatile: a poll load the compiler must neither hoist out of its loop nor merge
#define RES_DBG 16                    // ints per workgroup in ResCtx::dbg
#define RES_FSTRIDE 32   // 4-byte words between two workgroups' flags: a 128-byte line each
#define RES_GSTRIDE 16   // doubles between two workgroups' granule slots: a 128-byte line each (no line is written by two CUs)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct ResWG { int r0, nr, cnt, pos; };   // first row, rows (<= RES_MAXROWS), entries of K, first position in the exchanged vector
struct ResCtx {
  int nwg, E, npad;                  // workgroups, entries per thread, length of the exchanged vector: every workgroup's rows + its three
                                     // dot partials, padded to whole 128-byte lines (a line written by two CUs can lose one of the writes)
  int npw;                           // wavefronts of each workgroup that poll: ceil(nwg / 64), one flag (or one workgroup's granules) per lane
  const unsigned short *rowpos;      // position of row j in that vector
  int pipe;                          // 1: pipelined recurrences where they pass their checks (OSQP_AMD_RESIDENT_PIPE=0: never)
  int u0_direct;                     // 1: the first product reads u0 from global memory instead of exchanging it
  int inject;                        // test hook (OSQP_AMD_RESIDENT_INJECT=k): in the launch of ADMM iteration k one workgroup walks away,
                                     // the others' waits time out -- the give-up path on demand
  const ResWG *wg;
  double *val;                       // [nwg][E][RES_PT]: entry t*E + k of the workgroup's row-major list at (k, t)
  const unsigned short *col;         // same layout: POSITION of the entry's column in the exchanged vector
  const unsigned char *rowl;         // same layout: local row of the entry
  const int *krp, *kcj, *kps, *kdst; // K row by row for k_form_K: row pointers (n + 1), column, slot in M of P(i,j) (-1: none), slot in val
  const unsigned long long *brk;     // [nwg][RES_PT]: bit k set = entry k ends a row segment of this thread
  const unsigned short *slot0;       // [nwg][RES_PT]: first segment slot of the thread
  const unsigned short *segrow;      // [nwg][RES_MAXROWS + 1]: first segment slot of each local row
  double *ubuf;                      // 2 x npad doubles (parity of the tag)
  unsigned *flags;                   // nwg x RES_FSTRIDE words (one 128-byte line each)
  int *dbg;                          // nwg x RES_DBG, written by a launch that gives up: see res_note()
  double *sbuf;                      // 2 x nwg x RES_GSTRIDE 8-byte words: six granules {32 bits of a double, tag} per workgroup, one line each
};

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t res_rsrc(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// ---- waits inside a resident launch (k_pcg_resident, k_pcg_blockres) ----
// Every wait is a bounded spin of ceil(nwg / 64) wavefronts, one polled workgroup per lane, re-loading only what is still
// missing.  Every 16th round the wave looks at the clock and at the give-up word; every 64th the workgroup stores its own
// flag / granules again (idempotent: should a store ever go missing, the others do not hang on it) -- counted in
// State::res_repub; a wait that ends well after more than 50 us is counted in State::res_slow.
// res_spin_check: 0 = keep polling, 1 = this wait has timed out, 2 = another workgroup gave up.
__device__ __forceinline__ int res_spin_check(State *st, long long &t0, int &late) {
  const long long now = wall_clock64();
  const long long el = now - t0;
  if (el < 0 || el > (1LL << 40)) { t0 = now; late = 0; return 0; }      // not a time: start this wait's clock again
  if (__hip_atomic_load(&st->res_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 2;
  if (el > RES_WAIT_TICKS) { if (late) return 1; late = 1; } else late = 0;
  return 0;
}
// What a workgroup leaves for the host when it gives up (ResCtx::dbg, RES_DBG ints per workgroup):
//   [0] exchange number  [1] mode  [2] PCG iterations          (written at the kernel's exit)
//   [3] 1 / 2: its flag / granule wait timed out, 3: it left because another workgroup had given up
//   [4] first workgroup it still missed (-1: none)   [5] the word it last read from that workgroup (flag, or a granule's tag)
//   [6] ticks it had waited (100 MHz)   [7] XCC id   [8] clock at the give-up (low 32 bits)   [9] rounds   [10] granule index
// Publishing time of a workgroup's last flag = [8] - [6]: its wait started right behind its own store.
__device__ __forceinline__ void res_note(State *st, int *dbg, int g, int verdict, int kind, int nx, int first, unsigned seen, long long t0, unsigned rounds, int gidx) {
  if (verdict == 1 && atomicCAS(&st->res_dbg[0], 0, kind) == 0) { st->res_dbg[1] = nx; st->res_dbg[2] = g; st->res_dbg[3] = first; }
  int *d = dbg + (size_t)RES_DBG * g;
  const long long now = wall_clock64();
  d[3] = verdict == 1 ? kind : 3; d[4] = first; d[5] = (int)seen; d[6] = (int)(now - t0);
  d[7] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);        // HW_REG_XCC_ID, bits 3:0
  d[8] = (int)(unsigned)now; d[9] = (int)rounds; d[10] = gidx;
}
__device__ __forceinline__ void res_slow_note(State *st, long long t0, int nx, int g, int missed, int kind) {
  const long long el = wall_clock64() - t0;
  if (el > RES_SLOW_TICKS && el < (1LL << 40)) {
    atomicAdd(&st->res_slow, 1); atomicMax(&st->res_slow_max, (int)(el > 0x7fffffffLL ? 0x7fffffffLL : el));
    atomicAdd(&st->res_slow_hist[el < 20000 ? 0 : (el < 50000 ? 1 : (el < 200000 ? 2 : 3))], 1);
    atomicOr(&st->res_slow_xcc, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15));
    st->res_slow_last[0] = nx; st->res_slow_last[1] = g; st->res_slow_last[2] = missed; st->res_slow_last[3] = kind;
  }
}

// K values in the resident layout.  K_ij = P_ij + sigma [i == j] + sum_k rho_k (A_ki A_kj): the sum runs over the rows k
// of A that column i reaches, in ascending k -- the same order and the same products for (i,j) and (j,i), so K is
// symmetric to the bit.  One wavefront per row i of K: each lane keeps up to four entries (i, j) of the row; the
// wavefront walks the rows k of A that column i touches (the A' part of row i of M) and broadcasts their entries
// (k, j') lane by lane; a lane whose j equals j' adds rho_k A_ki A_kj'.  All loads are wavefront-uniform or coalesced:
// 25 us for the 2.09 M entries of config 2 (a two-list merge per entry took 390 us on dependent scattered loads).
__device__ __forceinline__ int lane_int(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__global__ void __launch_bounds__(TB) k_form_K(Ctx c, ResCtx rc) {
  const double sigma = c.prm->sigma;
  if (blockIdx.x == 0 && threadIdx.x == 0) c.st->res_pipe_off = 0;     // a new K: the pipelined recurrences get another chance
  const int lane = threadIdx.x & 63, nwaves = gridDim.x * (TB / 64);
  for (int i = blockIdx.x * (TB / 64) + (threadIdx.x >> 6); i < c.n; i += nwaves) {
    const int e0 = rc.krp[i], e1 = rc.krp[i + 1];
    const int ka = c.M.split[i], kb = c.M.rowptr[i + 1];
    for (int eb = e0; eb < e1; eb += 256) {
      int j[4]; double v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = eb + q * 64 + lane;
        j[q] = -1; v[q] = 0.0;
        if (idx < e1) {
          j[q] = rc.kcj[idx];
          const int ps = rc.kps[idx];
          v[q] = (ps >= 0 ? c.M.val[ps] : 0.0) + (j[q] == i ? sigma : 0.0);
        }
      }
      for (int kc = ka; kc < kb; kc += 64) {
        const bool on = kc + lane < kb;
        const int kk = on ? c.M.col[kc + lane] - c.n : 0;
        const double ai = on ? c.M.val[kc + lane] : 0.0, rk = on ? c.rhoe[kk] : 0.0;     // (rhoe: the rows' effective weights when variables are eliminated, rho itself otherwise)
        const int p0 = on ? c.A.rowptr[kk] : 0, p1 = on ? c.A.rowptr[kk + 1] : 0;
        const int cnt = min(64, kb - kc);
        for (int q = 0; q < cnt; ++q) {
          const double a_i = lane_value(ai, q), r = lane_value(rk, q);
          const int s0 = lane_int(p0, q), s1 = lane_int(p1, q);
          for (int ec = s0; ec < s1; ec += 64) {
            const bool eon = ec + lane < s1;
            const int cj = eon ? c.A.col[ec + lane] : -1;
            const double av = eon ? c.A.val[ec + lane] : 0.0;
            const int len = min(64, s1 - ec);
            for (int e = 0; e < len; ++e) {
              const int cjb = lane_int(cj, e);
              const double t = r * (a_i * lane_value(av, e));
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) if (j[q4] == cjb) v[q4] += t;
            }
          }
        }
      }
      // eliminated variables (k_elim_refresh) are not part of the system the launch solves: their rows and columns of K hold
      // nothing but a diagonal entry (the PCG vectors are zero there, so its value never enters a product)
      const bool igone = c.nelim && c.erow[i] >= 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = eb + q * 64 + lane;
        if (idx < e1) {
          const bool gone = c.nelim && (igone || c.erow[j[q]] >= 0);
          rc.val[rc.kdst[idx]] = gone ? (j[q] == i ? 1.0 : 0.0) : v[q];
        }
      }
    }
  }
}

template <int E>
__global__ void __launch_bounds__(RES_TB) k_pcg_resident(Ctx c, ResCtx rc) {
  extern __shared__ __attribute__((aligned(16))) double rlds[];
  State *st = c.st;
  TL_MARK(c, 5);
  if (st->stalled || !st->run || st->res_fail) return;
  double *uv = rlds;                              // npad doubles: the exchanged vector (rows and riding partials of every workgroup)
  double *seg = uv + rc.npad;                     // RES_TB + RES_MAXROWS row-segment sums
  double *sc = seg + RES_TB + RES_MAXROWS;        // 8 doubles: gamma, delta, rr, fail word, verdict
  double *sval = sc + 8;                          // 3 x 256: the workgroups' dot partials during exchange (2)
#ifdef OSQP_AMD_TIMELINE
  long long *tls = reinterpret_cast<long long *>(sval + 768);   // phase stamps of the first 64 exchanges (workgroup 0)
#define RTL(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && nx >= 1 && nx <= 64) tls[(nx - 1) * 5 + (k)] = wall_clock64(); } while (0)
  long long *wtl = tls + 5 * 64;                             // [phase][wavefront] stamps of exchange 8 (workgroup 0), then the same of exchange 1
#define WTL(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (nx == 8 || nx == 1)) wtl[(nx == 1 ? 40 : 0) + (k) * 8 + (threadIdx.x >> 6)] = wall_clock64(); } while (0)
  if (blockIdx.x == 0 && threadIdx.x == 0) wtl[80] = wall_clock64();       // kernel entry
#else
#define RTL(k) do { } while (0)
#define WTL(k) do { } while (0)
#endif
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nwg = rc.nwg, npad = rc.npad;
  const ResWG w = rc.wg[g];
  const Params prm = *c.prm;
  const unsigned ep0 = st->res_epoch;
  const bool sabotage = rc.inject > 0 && g == rc.nwg - 1 && st->admm_done + 1 == rc.inject;
  // pipelined recurrences only where a drift would be caught: not in the convexity probe, not after a failed check
  bool pipe = rc.pipe && !st->res_pipe_off && !prm.no_restart;
  // ---- own slice of K into registers (issued first: in flight under everything below) ----
  // (wavefronts 1..7; wavefront 0 holds no entries: its loads are the small ones the start-up and the exchanges wait for)
  const bool prod = wv != 0;
  const int tt = t - 64;
  double kv[E]; unsigned short kc[E];
#pragma unroll
  for (int k = 0; k < E; ++k) { kv[k] = 0.0; kc[k] = 0; }
  unsigned long long brk = 0ull;
  int slot0 = 0;
  if (prod) {
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const size_t idx = ((size_t)g * E + k) * RES_PT + tt;
      kv[k] = rc.val[idx]; kc[k] = rc.col[idx];
    }
    brk = rc.brk[(size_t)g * RES_PT + tt];
    slot0 = rc.slot0[(size_t)g * RES_PT + tt];
  }
  int sr0 = 0, sr1 = 0;
  int ppos[4];                         // where the partials of workgroups lane, lane + 64, ... sit in the exchanged vector
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int o = lane + 64 * q; ppos[q] = 0; if (o < rc.nwg) { const ResWG wo = rc.wg[o]; ppos[q] = wo.pos + wo.nr; } }
  // ---- start-up scalars: ||r0||^2, ||b||^2 from k_pcg_init's partials (every wavefront, same order) ----
  double rr0 = 0.0, bb = 0.0;
  for (int i = lane; i < c.gridM; i += 64) { rr0 += c.part_rr[i]; bb += c.part_bb[i]; }
  rr0 = wave_sum(rr0); bb = wave_sum(bb);
  const double tol2 = fmax(prm.eps_rel * prm.eps_rel * bb, prm.eps_abs * prm.eps_abs);
  if (rr0 <= tol2) {
    if (g == 0 && t == 0) { st->done = 1; st->tol2 = tol2; }
    return;
  }
  // rows of this workgroup live in the lanes of wavefront 0
  const bool own = wv == 0 && lane < w.nr;
  const int j = w.r0 + lane;
  double r_ = 0, u_ = 0, w_ = 0, p_ = 0, s_ = 0, x_ = 0, mi = 0, x0_ = 0, r0_ = 0;
  double z_ = 0, q_ = 0;               // pipelined recurrences: z = K q, q = Minv s
  if (own) {
    r_ = c.init_r[(size_t)j * c.init_stride]; u_ = c.init_z[j]; x_ = c.vx[j]; mi = c.minv[j];
    x0_ = x_; r0_ = r_;
    sr0 = rc.segrow[(size_t)g * (RES_MAXROWS + 1) + lane]; sr1 = rc.segrow[(size_t)g * (RES_MAXROWS + 1) + lane + 1];
  }
  if (t == 0) sc[3] = 0.0;
  __syncthreads();
  int nx = 0;                          // vector exchanges so far; exchange k carries the tag ep0 + k
  unsigned tag = ep0;
  int par = 0;

  // Exchange (1): every workgroup's rows of one vector (+ 3 doubles riding along) to every workgroup's LDS.
  // Write-through stores, drain, flag; wavefront 0 polls the flags; sc1 sweep.  False when a wait timed out.
  auto vec_exchange = [&](double val, bool ride) __attribute__((always_inline)) -> bool {
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;     // riding partials of (r,u), (w,u), (r,r) when `ride`
    ++nx; tag = ep0 + (unsigned)nx; par = (int)(tag & 1u);
    if (sabotage && nx == 3) return false;     // (test hook: this workgroup walks away without a word)
    RTL(0);
    const __amdgpu_buffer_rsrc_t rs = res_rsrc(rc.ubuf + (size_t)par * npad, (size_t)npad * 8);
    if (wv == 0) {
      // rows at pos .. pos + nr - 1, the three riding partials right behind them: all inside this workgroup's own lines
      if (lane < w.nr) {
        u32x2 d; d.x = (unsigned)__double2loint(val); d.y = (unsigned)__double2hiint(val);
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, (w.pos + lane) * 8, 0, AUX_SC1);
      }
      // (the partials are reduced while the rows are on their way)
      if (ride) { e0 = wave_sum(own ? r_ * u_ : 0.0); e1 = wave_sum(own ? w_ * u_ : 0.0); e2 = wave_sum(own ? r_ * r_ : 0.0); }
      if (lane < 3) {
        const double v = lane == 0 ? e0 : (lane == 1 ? e1 : e2);
        u32x2 d; d.x = (unsigned)__double2loint(v); d.y = (unsigned)__double2hiint(v);
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, (w.pos + w.nr + lane) * 8, 0, AUX_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(rc.flags + (size_t)g * RES_FSTRIDE, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wv < rc.npw) {
      // ceil(nwg / 64) wavefronts poll, ONE flag per lane (a lane that polled four flags with atomic loads waited for each
      // before it issued the next), and only until that flag has been seen; the own flag is not polled
      const int o = t;                         // flag of workgroup t
      bool pend = o < nwg && o != g;
      int miss16 = -1;
      unsigned seen = 0, rounds = 0;
      long long t0 = wall_clock64();
      int late = 0;
      bool gave = false;
      while (true) {
        if (pend) {
          seen = __hip_atomic_load(rc.flags + (size_t)o * RES_FSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pend = (int)(seen - tag) < 0;
        }
        if (!__any(pend)) break;
        if (++rounds & 15u) { __builtin_amdgcn_s_sleep(1); continue; }     // clock and give-up word only every 16th round
        const int verdict = res_spin_check(st, t0, late);
        if (verdict) {
          const unsigned long long miss = __ballot(pend);
          const int fl = miss ? (int)__ffsll((long long)miss) - 1 : 0;
          const unsigned sv = (unsigned)__builtin_amdgcn_readlane((int)seen, fl);
          if (lane == 0) { res_note(st, rc.dbg, g, verdict, 1, nx, miss ? 64 * wv + fl : -1, sv, t0, rounds, 0); sc[3] = 1.0; }
          gave = true;
          break;
        }
        if (rounds == 16u) { const unsigned long long ms = __ballot(pend); miss16 = ms ? 64 * wv + (int)__ffsll((long long)ms) - 1 : -1; }
        if ((rounds & 63u) == 0 && wv == 0 && lane == 0) {       // a long wait: say it again
          __hip_atomic_store(rc.flags + (size_t)g * RES_FSTRIDE, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicAdd(&st->res_repub, 1);
        }
        if ((rounds & 255u) == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_sleep(1);
      }
      if (rounds >= 16u && lane == 0 && !gave) res_slow_note(st, t0, nx, g, miss16, 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    RTL(1);
    if (sc[3] != 0.0) return false;
    {
      const int half = npad >> 1;
      u32x4 v[RES_MAXLD];              // every load of the sweep in flight at once
#pragma unroll
      for (int q = 0; q < RES_MAXLD; ++q) { const int i2 = q * RES_TB + t; if (i2 < half) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, i2 * 16, 0, AUX_SC1); }
#pragma unroll
      for (int q = 0; q < RES_MAXLD; ++q) {
        const int i2 = q * RES_TB + t;
        if (i2 < half) { uv[2 * i2] = __hiloint2double((int)v[q].y, (int)v[q].x); uv[2 * i2 + 1] = __hiloint2double((int)v[q].w, (int)v[q].z); }
      }
    }
    __syncthreads();
    RTL(2);
    return true;
  };
  // rows of K times the vector in LDS: E products per thread, row segments through LDS; the own-row value
  // in the lanes of wavefront 0 (ends with the barrier that makes the segments visible)
  // u0 = Minv r0 was left in global memory by k_pcg_init (an earlier launch: plain loads see it): no exchange needed
  auto load_u0 = [&]() __attribute__((always_inline)) {
    // (k_pcg_init left it in the layout of the exchanged vector: one coalesced sweep, every load in flight at once)
    const double2 *src = reinterpret_cast<const double2 *>(c.u0pos);
    const int half = npad >> 1;
#pragma unroll
    for (int h8 = 0; h8 < RES_MAXLD; h8 += 4) {          // four loads in flight per thread and round (registers are the entries of K's)
      if (h8 * RES_TB >= half) break;
      double2 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i2 = (h8 + q) * RES_TB + t; if (i2 < half) v[q] = src[i2]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i2 = (h8 + q) * RES_TB + t; if (i2 < half) { uv[2 * i2] = v[q].x; uv[2 * i2 + 1] = v[q].y; } }
    }
    __syncthreads();
  };
  auto products_issue = [&]() __attribute__((always_inline)) {
    if (prod) {
      // gathers first (all in flight together; the segment writes below could alias them as far as the compiler knows),
      // half of the entries at a time to stay inside the register budget
      double s = 0.0; int slot = slot0;
      constexpr int H = (E + 1) / 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double xv[H];
#pragma unroll
        for (int k = 0; k < H; ++k) if (h * H + k < E) xv[k] = uv[kc[h * H + k]];
#pragma unroll
        for (int k = 0; k < H; ++k)
          if (h * H + k < E) {
            s += kv[h * H + k] * xv[k];
            if ((brk >> (h * H + k)) & 1ull) { seg[slot++] = s; s = 0.0; }
          }
      }
    }
  };
  auto products_finish = [&]() __attribute__((always_inline)) -> double {
    __syncthreads();
    RTL(3);
    double a = 0.0;
    if (own)
      for (int q0 = sr0; q0 < sr1; q0 += 16) {       // 16 independent LDS reads in flight, summed in slot order
        double sv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) sv[q] = seg[min(q0 + q, sr1 - 1)];     // unconditional: no branch per read
#pragma unroll
        for (int q = 0; q < 16; ++q) a += q0 + q < sr1 ? sv[q] : 0.0;
      }
    return a;
  };
  // Exchange (2): three dot partials per workgroup as tagged granules (the data is the flag); totals in sc[0..2].
  auto scal_exchange = [&](double pg, double pd, double prr) __attribute__((always_inline)) -> bool {
    // granules of 8 bytes {32 bits of data, tag}: six per workgroup, in a line of the workgroup's own, stored with
    // agent-scope atomic stores
    unsigned long long *gb = reinterpret_cast<unsigned long long *>(rc.sbuf) + (size_t)par * nwg * RES_GSTRIDE;
    if (wv == 0) {
      pg = wave_sum(pg); pd = wave_sum(pd); prr = wave_sum(prr);
      if (lane < 6) {
        const double v = lane < 2 ? pg : (lane < 4 ? pd : prr);
        const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
        __hip_atomic_store(gb + (size_t)g * RES_GSTRIDE + lane, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (wv < rc.npw) {
      // ceil(nwg / 64) wavefronts poll, ONE workgroup's six granules per lane, all six loads in flight together
      // (buffer_load_dwordx2 sc1, volatile; six atomic loads went one after the other: 1.9-2.2 us per exchange against
      // 1.5-1.75 at up to 64 workgroups, tools/exchange_probe2.hip), and only the granules still missing.  Each granule vouches
      // for itself; a half arrives in LDS as soon as its tag matches.
      const int o = t;                         // the workgroup whose granules this thread fetches
      const __amdgpu_buffer_rsrc_t rg = res_rsrc(gb, (size_t)nwg * RES_GSTRIDE * 8);
      unsigned pend = o < nwg ? 63u : 0u;
      int miss16 = -1;
      unsigned *svw = reinterpret_cast<unsigned *>(sval) + 6 * o;
      unsigned seen = 0, rounds = 0;
      long long t0 = wall_clock64();
      int late = 0;
      bool gave = false;
      while (true) {
        u32x2 gv[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) if (pend & (1u << q)) gv[q] = __builtin_amdgcn_raw_buffer_load_b64(rg, (o * RES_GSTRIDE + q) * 8, 0, AUX_SC1 | AUX_VOL);
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (pend & (1u << q)) { if (gv[q].y == tag) { svw[q] = gv[q].x; pend &= ~(1u << q); } else seen = gv[q].y; }
        if (!__any(pend != 0)) break;
        asm volatile("" ::: "memory");
        if (++rounds & 15u) { __builtin_amdgcn_s_sleep(1); continue; }
        const int verdict = res_spin_check(st, t0, late);
        if (verdict) {
          const unsigned long long miss = __ballot(pend != 0);
          const int fl = miss ? (int)__ffsll((long long)miss) - 1 : 0;
          const unsigned sv = (unsigned)__builtin_amdgcn_readlane((int)seen, fl);
          const int gi = __builtin_amdgcn_readlane((int)pend, fl);
          if (lane == 0) { res_note(st, rc.dbg, g, verdict, 2, nx, miss ? 64 * wv + fl : -1, sv, t0, rounds, gi); sc[3] = 1.0; }
          gave = true;
          break;
        }
        if (rounds == 16u) { const unsigned long long ms = __ballot(pend != 0); miss16 = ms ? 64 * wv + (int)__ffsll((long long)ms) - 1 : -1; }
        if ((rounds & 63u) == 0 && wv == 0) {          // a long wait: say it again
          if (lane < 6) {
            const double v = lane < 2 ? pg : (lane < 4 ? pd : prr);
            const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
            __hip_atomic_store(gb + (size_t)g * RES_GSTRIDE + lane, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (lane == 0) atomicAdd(&st->res_repub, 1);
        }
        if ((rounds & 255u) == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // a poll that stays unanswered this long: drop whatever this CU still caches
        __builtin_amdgcn_s_sleep(1);
      }
      if (rounds >= 16u && lane == 0 && !gave) res_slow_note(st, t0, nx, g, miss16, 2);
    }
    __syncthreads();
    if (wv == 0) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lane + 64 * q < nwg) { a0 += sval[3 * (lane + 64 * q)]; a1 += sval[3 * (lane + 64 * q) + 1]; a2 += sval[3 * (lane + 64 * q) + 2]; }
      a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
      if (lane == 0) { sc[0] = a0; sc[1] = a1; sc[2] = a2; }
    }
    __syncthreads();
    RTL(4);
    return sc[3] == 0.0;
  };

  double gam_old = 0.0, alp_old = 0.0;
  int iters = 0;
  bool conv = false, bad = false, failed = false;
  // One loop, one copy of each exchange (four inlined copies cost 88 spilled registers).  Modes:
  //   0  w = K u0 for the pipelined phase           1  pipelined iteration (Ghysels-Vanroose): ONE exchange -- m = Minv w
  //   2  true-residual check r0 - K (x - x0) of        travels with the partials of (r,u), (w,u), (r,r); wavefront 0 forms
  //      EVERY solve that claims convergence           the scalars while the others multiply; w and u follow recurrences
  //   3  Chronopoulos-Gear iteration with a fresh
  //      product (the recurrences of k_cg_A/k_cg_B)
  // The pipelined recurrences drift on ill-conditioned systems, so: out after 64 iterations, out on any breakdown, and
  // every solve that claims convergence is checked against the true residual.  A failed check hands the solve to
  // mode 3 and switches the pipelined phase off until K changes.
  int mode = pipe ? 0 : 3;
  int check_why = 0;                   // 0: the pipelined recurrence says converged, 1: its breakdown / iteration cap, 2: long solve, 3: mode 3 says converged
  int checks3 = 0;                     // failed checks of mode-3 verdicts in this solve
  bool have_u0 = rc.u0_direct;         // the vector k_pcg_init left in global memory is still the u of the recurrences
  while (true) {
    double val = u_, m_ = 0.0;
    if (mode == 1) { m_ = mi * w_; val = m_; }
    else if (mode == 2) val = x_ - x0_;
    if (have_u0) { ++nx; tag = ep0 + (unsigned)nx; par = (int)(tag & 1u); load_u0(); have_u0 = false; }   // (a fresh tag for the granules of exchange (2))
    else if (!vec_exchange(val, mode == 1)) { failed = true; break; }
    WTL(0);
    products_issue();
    WTL(1);
    CgStep cs{};
    if (mode == 1 && wv == 0) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (lane + 64 * q < nwg) { a0 += uv[ppos[q]]; a1 += uv[ppos[q] + 1]; a2 += uv[ppos[q] + 2]; }
      const double gam = wave_sum(a0), del = wave_sum(a1), rr = wave_sum(a2);
      cs = cg_step(rr, gam, del, gam_old, alp_old, tol2, iters, 0, prm, false);
      gam_old = gam; alp_old = cs.alpha;
      if (lane == 0) sc[4] = cs.stop ? (cs.conv ? 1.0 : 2.0) : 0.0;
    }
    WTL(2);
    const double kx = products_finish();       // own row of K times the exchanged vector
    WTL(3);
    if (mode == 0) { w_ = kx; mode = 1; continue; }
    if (mode == 1) {
      const double verdict = sc[4];
      if (verdict != 0.0) { check_why = verdict == 1.0 ? 0 : 1; mode = 2; continue; }   // (the products of this trip are wasted)
      ++iters;
      if (own) {
        z_ = cs.first ? kx : (kx + cs.beta * z_);
        q_ = cs.first ? m_ : (m_ + cs.beta * q_);
        s_ = cs.first ? w_ : (w_ + cs.beta * s_);
        p_ = cs.first ? u_ : (u_ + cs.beta * p_);
        x_ += cs.alpha * p_;
        r_ -= cs.alpha * s_;
        u_ -= cs.alpha * q_;
        w_ -= cs.alpha * z_;
      }
      WTL(4);
      if (iters >= 64) { check_why = 2; mode = 2; }
      continue;
    }
    if (mode == 2) {
      const double rt = r0_ - kx;
      if (!scal_exchange(0.0, 0.0, own ? rt * rt : 0.0)) { failed = true; break; }
      if (sc[2] <= 2.0 * tol2) { conv = true; break; }
      // continue from the true residual with fresh directions; the recurrences are switched off for this K when they
      // had claimed convergence or broken down (not when the solve was merely long)
      if (own) { r_ = rt; u_ = mi * rt; }
      gam_old = 0.0; alp_old = 0.0;
      if (check_why == 3) ++checks3;
      if (g == 0 && t == 0) { if (check_why != 2) st->res_chk_fail += 1; if (check_why < 2) st->res_pipe_off = 1; }   // (a long solve cut at 64 iterations has not failed anything)
      __syncthreads();          // sc[] is rewritten by the next exchange
      mode = 3;
      continue;
    }
    w_ = kx;
    if (!scal_exchange(own ? r_ * u_ : 0.0, own ? w_ * u_ : 0.0, own ? r_ * r_ : 0.0)) { failed = true; break; }
    const double gam = sc[0], del = sc[1], rr = sc[2];
    cs = cg_step(rr, gam, del, gam_old, alp_old, tol2, iters, 0, prm, false);
    if (cs.stop) {
      // a converged solve is checked against the true residual here too (twice at most: on a system whose attainable
      // accuracy sits above the stop the recursive residual's verdict stands, as on the launch-per-step path)
      if (cs.conv && checks3 < 2) { check_why = 3; mode = 2; continue; }
      conv = cs.conv; bad = cs.bad && !cs.conv; break;
    }
    ++iters; gam_old = gam; alp_old = cs.alpha;
    if (own) {
      p_ = cs.first ? u_ : (u_ + cs.beta * p_);
      s_ = cs.first ? w_ : (w_ + cs.beta * s_);
      x_ += cs.alpha * p_;
      r_ -= cs.alpha * s_;
      u_ = mi * r_;
    }
  }
  if (failed) {
    if (t == 0) {
      __hip_atomic_store(&st->res_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int *d = rc.dbg + (size_t)RES_DBG * g;
      d[0] = nx; d[1] = mode; d[2] = iters;
      if (sabotage) { d[3] = 4; d[7] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20); d[8] = (int)(unsigned)wall_clock64(); }
    }
    return;
  }
  if (own && iters > 0) c.va[j] = x_;
#ifdef OSQP_AMD_TIMELINE
  if (g == 0 && t == 0) {
    const int cnt = (nx < 64 ? nx : 64) * 5;
    const unsigned long long k0 = atomicAdd(c.tl, (unsigned long long)cnt + 1);
    if (k0 + cnt + 1 < TL_CAP - 2) {
      for (int q = 0; q < cnt; ++q) c.tl[1 + k0 + q] = ((unsigned long long)(10 + q % 5) << 56) | ((unsigned long long)tls[q] & 0x00FFFFFFFFFFFFFFull);
      c.tl[1 + k0 + cnt] = (15ull << 56) | (wall_clock64() & 0x00FFFFFFFFFFFFFFull);
    }
    if (nx >= 8) {
      const unsigned long long k1 = atomicAdd(c.tl, 80ull);
      if (k1 + 80 < TL_CAP - 2) {
        for (int q = 0; q < 40; ++q) c.tl[1 + k1 + q] = ((unsigned long long)(20 + q) << 56) | ((unsigned long long)(wtl[q] - wtl[0]) & 0x00FFFFFFFFFFFFFFull);
        for (int q = 0; q < 40; ++q) c.tl[1 + k1 + 40 + q] = ((unsigned long long)(60 + q) << 56) | ((unsigned long long)(wtl[40 + q] - wtl[80]) & 0x00FFFFFFFFFFFFFFull);   // first trip, since kernel entry
      }
    }
  }
#endif
  if (g == 0 && t == 0) {
    st->iters[0] = iters; st->iters[1] = 0;
    st->done = conv ? 1 : 2;
    if (bad) st->neg_curv = 1;
    st->tol2 = tol2;
    st->res_epoch = ep0 + (unsigned)nx;
  }
}


// ---------------------------------------------------------------------------
// Block-resident PCG: the resident idea for the portfolio family (BASELINE config 5), whose n = 50 000 is far beyond
// the vector exchange of k_pcg_resident -- and which needs none.  If P consists of dense diagonal blocks only and every
// row of A either has a single entry (bounds) or is one of at most four huge rows (a budget constraint), then
//     K = blockdiag(P_b + sigma I + diag(sum_i rho_i a_ij^2))  +  sum_h rho_h a_h a_h'
// and a workgroup that owns whole blocks needs nobody else's vector elements: K_b u_b is a dense product from registers
// (each row of a block split over two threads, 64 entries each), the low-rank part needs the scalars s_h = a_h'u, and
// delta = (Ku, u) = sum_g (K_b u_b, u_b) + sum_h rho_h s_h^2.  So ONE exchange of 3 + nh scalars per workgroup and
// iteration (tagged 8-byte granules, the exchange (2) of k_pcg_resident) carries gamma, delta', ||r||^2 and the s_h;
// everything else is local.  Chronopoulos-Gear recurrences with a fresh product (cg_step), 3 barriers per iteration.
// The 50 MB of blocks are read once per launch instead of once per PCG iteration.
// ---------------------------------------------------------------------------
struct BrCtx { int nwg; const int *blk0; double *sbuf; int *dbg; };

template <int NH>
__global__ void __launch_bounds__(RES_TB) k_pcg_blockres(Ctx c, BrCtx bc) {
  constexpr int NV = 3 + NH;                 // scalars per workgroup and iteration
  __shared__ double ul[256];                 // u of the own rows
  __shared__ double red[8 * 8];              // per-wavefront partials
  __shared__ double tot[8];                  // totals; [7]: a wait timed out
  __shared__ double sval[256 * NV];          // every workgroup's scalars during the exchange
  State *st = c.st;
  TL_MARK(c, 6);
  if (st->stalled || !st->run || st->res_fail) return;
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, nwg = bc.nwg;
#ifdef OSQP_AMD_TIMELINE
  __shared__ long long btl[6 * 64];          // phase stamps of the first 64 iterations (workgroup 0)
#define BTL(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && nx >= 1 && nx <= 64) btl[(nx - 1) * 6 + (k)] = wall_clock64(); } while (0)
#else
#define BTL(k) do { } while (0)
#endif
  const Params prm = *c.prm;
  const unsigned ep0 = st->res_epoch;
  // the row of this thread pair: row rl of the workgroup's (at most two) blocks, columns [64 h, 64 h + 64) of its block
  const int rl = t >> 1, h = t & 1;
  int j = -1, cb = 0, bw = 0, c0 = 0, pitch = 0; size_t off = 0;
  {
    int base = 0;
    for (int q = bc.blk0[g]; q < bc.blk0[g + 1]; ++q) {
      const DenseBlk d = c.dP.blk[q];
      if (rl >= base && rl < base + d.b) { j = d.c0 + (rl - base); cb = base; bw = d.b; c0 = d.c0; off = (size_t)d.off; pitch = d.pitch; }
      base += d.b;
    }
  }
  const bool row = j >= 0, own = row && h == 0;
  double dadd = 0.0, hc[NH > 0 ? NH : 1], hrho[NH > 0 ? NH : 1];
#pragma unroll
  for (int q = 0; q < NH; ++q) { hc[q] = 0.0; hrho[q] = c.rho[c.hrow[q]]; }
  if (row) {
    dadd = prm.sigma;
    for (int k = c.M.split[j]; k < c.M.rowptr[j + 1]; ++k) {          // single-entry rows of A at this column
      const int i = c.M.col[k] - c.n;
      bool huge = false;
#pragma unroll
      for (int q = 0; q < NH; ++q) huge |= i == c.hrow[q];
      if (!huge) { const double a = c.M.val[k]; dadd += c.rho[i] * a * a; }
    }
#pragma unroll
    for (int q = 0; q < NH; ++q) hc[q] = c.hcol[(size_t)q * c.n + j];
  }
  double kv[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const int cc = h * 64 + k;
    // K_b is symmetric: entry (row, cc) is read as (cc, row), so that the lanes of a wave instruction -- consecutive rows --
    // read consecutive addresses (row-wise it was one 64-byte segment per lane: 100 us per launch)
    kv[k] = (row && cc < bw) ? c.dP.val[off + (size_t)cc * pitch + (j - c0)] + (cc == j - c0 ? dadd : 0.0) : 0.0;
  }
  double rr0 = 0.0, bb = 0.0;
  for (int i = lane; i < c.gridM; i += 64) { rr0 += c.part_rr[i]; bb += c.part_bb[i]; }
  rr0 = wave_sum(rr0); bb = wave_sum(bb);
  const double tol2 = fmax(prm.eps_rel * prm.eps_rel * bb, prm.eps_abs * prm.eps_abs);
  if (rr0 <= tol2) {
    if (g == 0 && t == 0) { st->done = 1; st->tol2 = tol2; }
    return;
  }
  double r_ = 0, u_ = 0, p_ = 0, s_ = 0, x_ = 0, mi = 0;
  if (own) { r_ = c.init_r[(size_t)j * c.init_stride]; u_ = c.init_z[j]; x_ = c.vx[j]; mi = c.minv[j]; }
  if (t < 256) ul[t] = 0.0;
  if (t == 0) tot[7] = 0.0;
  __syncthreads();
  double gam_old = 0.0, alp_old = 0.0;
  int iters = 0, nx = 0;
  bool conv = false, bad = false, failed = false;
  while (true) {
    ++nx;
    const unsigned tag = ep0 + (unsigned)nx;
    const int par = (int)(tag & 1u);
    BTL(0);
    if (own) ul[rl] = u_;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 64; ++k) acc += kv[k] * ul[cb + h * 64 + k];
    acc += dpp_move<0xB1>(acc);                      // the two halves of the row
    BTL(1);
    double v[NV];
    v[0] = own ? r_ * u_ : 0.0; v[1] = own ? acc * u_ : 0.0; v[2] = own ? r_ * r_ : 0.0;
#pragma unroll
    for (int q = 0; q < NH; ++q) v[3 + q] = own ? hc[q] * u_ : 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const double w_ = wave_sum(v[i]); if (lane == 0) red[wv * 8 + i] = w_; }
    __syncthreads();
    BTL(2);
    unsigned long long *gb = reinterpret_cast<unsigned long long *>(bc.sbuf) + (size_t)par * nwg * RES_GSTRIDE;
    if (wv == 0 && lane < 2 * NV) {
      const int i = lane >> 1;
      double sv = 0.0;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sv += red[w8 * 8 + i];
      const unsigned half = (lane & 1) ? (unsigned)__double2hiint(sv) : (unsigned)__double2loint(sv);
      __hip_atomic_store(gb + (size_t)g * RES_GSTRIDE + lane, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wv < ((nwg + 63) >> 6)) {
      // one workgroup's 2 NV granules per lane, all loads in flight together, only what is still missing (see k_pcg_resident)
      const int o = t;
      const __amdgpu_buffer_rsrc_t rg = res_rsrc(gb, (size_t)nwg * RES_GSTRIDE * 8);
      unsigned pend = o < nwg ? (1u << (2 * NV)) - 1u : 0u;
      int miss16 = -1;
      unsigned *svw = reinterpret_cast<unsigned *>(sval) + 2 * NV * o;
      unsigned seen = 0, rounds = 0;
      long long t0 = wall_clock64();
      int late = 0;
      bool gave = false;
      while (true) {
        u32x2 gv[2 * NV];
#pragma unroll
        for (int q = 0; q < 2 * NV; ++q) if (pend & (1u << q)) gv[q] = __builtin_amdgcn_raw_buffer_load_b64(rg, (o * RES_GSTRIDE + q) * 8, 0, AUX_SC1 | AUX_VOL);
#pragma unroll
        for (int q = 0; q < 2 * NV; ++q)
          if (pend & (1u << q)) { if (gv[q].y == tag) { svw[q] = gv[q].x; pend &= ~(1u << q); } else seen = gv[q].y; }
        if (!__any(pend != 0)) break;
        asm volatile("" ::: "memory");
        if (++rounds & 15u) { __builtin_amdgcn_s_sleep(1); continue; }
        const int verdict = res_spin_check(st, t0, late);
        if (verdict) {
          const unsigned long long miss = __ballot(pend != 0);
          const int fl = miss ? (int)__ffsll((long long)miss) - 1 : 0;
          const unsigned sv = (unsigned)__builtin_amdgcn_readlane((int)seen, fl);
          const int gi = __builtin_amdgcn_readlane((int)pend, fl);
          if (lane == 0) { res_note(st, bc.dbg, g, verdict, 2, nx, miss ? 64 * wv + fl : -1, sv, t0, rounds, gi); tot[7] = 1.0; }
          gave = true;
          break;
        }
        if (rounds == 16u) { const unsigned long long ms = __ballot(pend != 0); miss16 = ms ? 64 * wv + (int)__ffsll((long long)ms) - 1 : -1; }
        if ((rounds & 63u) == 0 && wv == 0) {          // a long wait: say it again
          if (lane < 2 * NV) {
            const int i = lane >> 1;
            double sv2 = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) sv2 += red[w8 * 8 + i];
            const unsigned half = (lane & 1) ? (unsigned)__double2hiint(sv2) : (unsigned)__double2loint(sv2);
            __hip_atomic_store(gb + (size_t)g * RES_GSTRIDE + lane, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (lane == 0) atomicAdd(&st->res_repub, 1);
        }
        if ((rounds & 255u) == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_sleep(1);
      }
      if (rounds >= 16u && lane == 0 && !gave) res_slow_note(st, t0, nx, g, miss16, 2);
    }
    BTL(3);
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (lane + 64 * q < nwg) a += sval[NV * (lane + 64 * q) + i];
        a = wave_sum(a);
        if (lane == 0) tot[i] = a;
      }
    }
    __syncthreads();
    BTL(4);
    if (tot[7] != 0.0) { failed = true; break; }
    const double gam = tot[0], rr = tot[2];
    double del = tot[1];
#pragma unroll
    for (int q = 0; q < NH; ++q) del += hrho[q] * (tot[3 + q] * tot[3 + q]);
    const CgStep cs = cg_step(rr, gam, del, gam_old, alp_old, tol2, iters, 0, prm, false);
    if (cs.stop) { conv = cs.conv; bad = cs.bad && !cs.conv; break; }
    ++iters; gam_old = gam; alp_old = cs.alpha;
    if (own) {
      double w_ = acc;
#pragma unroll
      for (int q = 0; q < NH; ++q) w_ += (hrho[q] * tot[3 + q]) * hc[q];
      p_ = cs.first ? u_ : (u_ + cs.beta * p_);
      s_ = cs.first ? w_ : (w_ + cs.beta * s_);
      x_ += cs.alpha * p_;
      r_ -= cs.alpha * s_;
      u_ = mi * r_;
    }
    BTL(5);
  }
#ifdef OSQP_AMD_TIMELINE
  if (g == 0 && t == 0 && !failed) {
    const int cnt = (nx - 1 < 64 ? nx - 1 : 64) * 6;          // (the last trip broke out before its stamp 5)
    const unsigned long long k0 = atomicAdd(c.tl, (unsigned long long)cnt + 1);
    if (k0 + cnt + 1 < TL_CAP - 2) {
      for (int q = 0; q < cnt; ++q) c.tl[1 + k0 + q] = ((unsigned long long)(40 + q % 6) << 56) | ((unsigned long long)btl[q] & 0x00FFFFFFFFFFFFFFull);
      c.tl[1 + k0 + cnt] = (46ull << 56) | (wall_clock64() & 0x00FFFFFFFFFFFFFFull);
    }
  }
#endif
  if (failed) {
    if (t == 0) {
      __hip_atomic_store(&st->res_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int *d = bc.dbg + (size_t)RES_DBG * g;
      d[0] = nx; d[1] = 3; d[2] = iters;
    }
    return;
  }
  if (own && iters > 0) c.va[j] = x_;
  if (g == 0 && t == 0) {
    st->iters[0] = iters; st->iters[1] = 0;
    st->done = conv ? 1 : 2;
    if (bad) st->neg_curv = 1;
    st->tol2 = tol2;
    st->res_epoch = ep0 + (unsigned)nx;
  }
}

// ---------------------------------------------------------------------------
// Block-direct solve: the portfolio family (BASELINE config 5) without any iteration at all.
//
// Same structure as k_pcg_blockres accepts: P = dense diagonal blocks only, every row of A single-entry except nh <= 4
// huge rows.  Then  K = B + A_h' R_h A_h  with B = blockdiag(P_b + sigma I + diag(sum_i rho_i a_ij^2)) and the nh huge
// rows as a low-rank term, and by the Woodbury identity
//     K^-1 r = t - W c,     t = B^-1 r,   W = B^-1 A_h' (n x nh),   c = (R_h^-1 + A_h W)^-1 (A_h t).
// The blocks are inverted explicitly (k_blk_invert: in-place Gauss-Jordan in LDS, no pivoting -- B_b is positive
// definite or the problem is not convex: a non-positive pivot is reported as negative curvature) whenever rho, sigma or
// the matrices change, and kept in HBM in the layout of the dense blocks of P.  One linear solve is then
//     k_blk_apply   t = B^-1 r0 : ONE pass over the 50 MB of inverse blocks (the dense_block_mv of k_cg_B, which the PCG
//                   ran 26 times per ADMM iteration) + the partials of A_h t
//     k_blk_finish  c from the partials (nh x nh system, inverted on the host at refresh), x~ = x~0 + t - W c
// applied to the residual r0 = b - K x~0 of the warm start k_pcg_init forms anyway (one step of iterative refinement
// on top of the previous x~: the error of the explicit inverse, ~cond(B) eps, multiplies the CORRECTION, not x~).
// Config 5: 258 -> 4x us per ADMM iteration.  OSQP_AMD_BLOCK_DIRECT=0 keeps the block-resident PCG.
// ---------------------------------------------------------------------------
struct BdCtx {
  double *binv;            // inverse blocks, layout of Ctx::dP.val (DenseBlk::off / pitch)
  double *t;               // [n] B^-1 r
  double *wh;              // [nh][n] W = B^-1 A_h'
  double *cinv;            // [nh][nh] (R_h^-1 + A_h W)^-1
  double *part;            // [MAX_HUGE_FOLD][nblk] partials of A_h t per block
  int    *flag;            // [0] a pivot was not positive
  double *chk;             // [2] checks of a fresh inverse (bit patterns of non-negative doubles, atomicMax): blocks, capacitance matrix
  double *cap0;            // [kc][kc] the capacitance matrix as formed (coupled form): its inverse is checked against it
  int     fin_dots;        // the kernel that writes x~ also leaves the folded huge rows' partials of A x~ for k_admm_finalize (no k_huge_dot launch)
  // coupled form (kc > 0): EVERY row of A with two or more entries -- sector rows, the budget row, whatever their length -- is a
  // term of the low-rank part; nothing of W is stored (see k_cpl_dot)
  int     kc;
  const int *crow;         // [kc] the coupling rows of A
  const int *cidx;         // [m] position of a row in crow, -1 for the single-entry rows
  const int *chuge;        // [kc] which folded huge row (Ctx::hrow) a coupling row is, -1: none
  const int *cbptr;        // [n + 1] / cbent: per column j the entries of the coupling rows, {position in crow, slot of the value in M.val}
  const int2 *cbent;
  double *cs, *cc;         // [kc] S t and Cinv S t
  double *cap;             // [kc][kc] capacitance matrix R^-1 + S B^-1 S', inverted in place (k_cap_invert)
  double *cap2;            // [kc][kc] the other copy of the pivot steps
  double *wm;              // [16][n] scratch of the refresh: B^-1 S' for 16 coupling rows
};

#define INV_TB 1024
__global__ void __launch_bounds__(INV_TB) k_blk_invert(Ctx c, BdCtx bd) {
  extern __shared__ __attribute__((aligned(16))) double bl[];       // b x b, row pitch b
  __shared__ double colp[DENSE_MAX], rowp[DENSE_MAX];
  const double sigma = c.prm->sigma;
  // thread -> column j and the rows i = i0, i0 + 8, ... (no index arithmetic in the sweep: a division per element made this
  // kernel 2.85 ms for the 400 blocks of config 5; with 256 threads -- one wavefront per SIMD, nothing to hide the LDS round trips
  // of the sweep behind -- it was 1.9 ms)
  constexpr int RS = INV_TB / DENSE_MAX, RU = DENSE_MAX / RS;        // row step 8, up to 16 rows per thread
  const int t = threadIdx.x, j = t & (DENSE_MAX - 1), i0 = t / DENSE_MAX;
  for (int db = blockIdx.x; db < c.dP.nblk; db += gridDim.x) {
    const DenseBlk d = c.dP.blk[db];
    const int b = d.b;
    if (j < b) for (int i = i0; i < b; i += RS) bl[i * b + j] = c.dP.val[d.off + (size_t)i * d.pitch + j];
    __syncthreads();
    if (t < b) {                       // diagonal: sigma + the single-entry rows of A at this column (the huge rows are the low-rank part)
      const int jj = d.c0 + t;
      double dadd = sigma;
      for (int k = c.Mk.rowptr[jj]; k < c.Mk.rowptr[jj + 1]; ++k) {
        const int row = c.Mk.col[k] - c.n;
        if (bd.kc && bd.cidx[row] >= 0) continue;           // (coupled form: such a row is part of S)
        const double a = c.Mk.val[k]; dadd += c.rho[row] * a * a;
      }
      bl[t * b + t] += dadd;
    }
    __syncthreads();
    for (int p = 0; p < b; ++p) {
      if (t < b) { colp[t] = bl[t * b + p]; rowp[t] = bl[p * b + t]; }
      __syncthreads();
      const double piv = rowp[p];
      if (!(piv > 0.0) && t == 0) atomicOr(bd.flag, 1);
      const double inv = 1.0 / piv;
      if (j < b) {
        const double rj = rowp[j] * inv;              // row p of the result (j != p)
        double cur[RU], cp[RU];                       // all rows of the thread: their LDS reads are in flight together
#pragma unroll
        for (int u = 0; u < RU; ++u) { const int i = i0 + RS * u; if (i < b) { cur[u] = bl[i * b + j]; cp[u] = colp[i]; } else { cur[u] = 0.0; cp[u] = 0.0; } }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = i0 + RS * u;
          if (i >= b) continue;
          double v;
          if (i == p) v = (j == p) ? inv : rj;
          else if (j == p) v = -cp[u] * inv;
          else v = cur[u] - cp[u] * rj;
          bl[i * b + j] = v;
        }
      }
      __syncthreads();
    }
    if (j < d.pitch) for (int i = i0; i < b; i += RS) bd.binv[d.off + (size_t)i * d.pitch + j] = j < b ? bl[i * b + j] : 0.0;
    __syncthreads();
  }
}

// out_b = Binv_b in_b for every block (in, out contiguous n-vectors), and the partials of A_h out for the huge rows
__global__ void __launch_bounds__(TB) k_blk_apply(Ctx c, BdCtx bd, const double *in, double *out, int gated) {
  if (gated) { const State *st = c.st; if (st->stalled || !st->run) return; }
  __shared__ double scratch[5 * DENSE_MAX];
  __shared__ double red[16];
  const DenseP inv{c.dP.nblk, c.dP.blk, bd.binv};
  for (int db = blockIdx.x; db < c.dP.nblk; db += gridDim.x) {
    const DenseBlk d = c.dP.blk[db];
    const double y = dense_block_mv(inv, d, in, scratch);
    const int j = d.c0 + threadIdx.x;
    const bool on = (int)threadIdx.x < d.b;
    if (on) out[j] = y;
    for (int h = 0; h < c.nh; ++h) {
      const double s = block_sum(on ? c.hcol[(size_t)h * c.n + j] * y : 0.0, red);
      if (threadIdx.x == 0) bd.part[(size_t)h * c.dP.nblk + db] = s;
    }
    __syncthreads();
  }
}

// (A x~)_h of the folded huge rows for k_admm_finalize, by the kernel that has just written x~: this workgroup's partial into its
// slot of Ctx::part_h, zeros into the slots no workgroup of this launch owns (k_huge_dot, which other paths run, fills all gridA)
__device__ __forceinline__ void fin_huge_partials(const Ctx &c, const double *hd, double *red) {
  for (int h = 0; h < c.nh; ++h) {
    const double s = block_sum(hd[h], red);
    if (threadIdx.x == 0)
      for (int slot = blockIdx.x; slot < c.gridA; slot += gridDim.x) c.part_h[(size_t)h * c.gridA + slot] = slot == (int)blockIdx.x ? s : 0.0;
  }
}

// x~ = x~0 + t - W c with c = Cinv (A_h t); the linear solve is complete (one "PCG iteration" in the statistics)
__global__ void __launch_bounds__(TB) k_blk_finish(Ctx c, BdCtx bd) {
  State *st = c.st;
  if (st->stalled || !st->run) return;
  __shared__ double red[16];
  __shared__ double cs[MAX_HUGE_FOLD];
  double sh[MAX_HUGE_FOLD];
#pragma unroll
  for (int h = 0; h < MAX_HUGE_FOLD; ++h) {
    double s = 0.0;
    if (h < c.nh) for (int i = threadIdx.x; i < c.dP.nblk; i += TB) s += bd.part[(size_t)h * c.dP.nblk + i];
    sh[h] = h < c.nh ? block_sum(s, red) : 0.0;
    // plain form: K^-1 b = t0 - W Cinv (A_h t0 - R^-1 beta), t0 = B^-1 b0, b = b0 + A_h' beta: beta_h = rho_h z_h - y_h is what
    // made `t - W c` cancel to eight digits when it sat inside b (rho_eq = 4 050 on the budget row); here it never meets B^-1
    if (c.plain_rhs && h < c.nh) sh[h] -= c.vb[c.n + c.hrow[h]] / c.rho[c.hrow[h]];
  }
  if (threadIdx.x < MAX_HUGE_FOLD) {
    double v = 0.0;
    for (int h = 0; h < c.nh; ++h) v += bd.cinv[threadIdx.x * MAX_HUGE_FOLD + h] * sh[h];
    cs[threadIdx.x] = (int)threadIdx.x < c.nh ? v : 0.0;
  }
  __syncthreads();
  static_assert(MAX_HUGE_FOLD == 4, "initialisers below");
  double hd[MAX_HUGE_FOLD] = {0.0, 0.0, 0.0, 0.0};
  for (int j = blockIdx.x * TB + threadIdx.x; j < c.n; j += gridDim.x * TB) {
    double v = bd.t[j];
    for (int h = 0; h < c.nh; ++h) v -= bd.wh[(size_t)h * c.n + j] * cs[h];
    const double xt = c.plain_rhs ? v : c.vx[j] + v;
    c.va[j] = xt;
    if (bd.fin_dots) for (int h = 0; h < c.nh; ++h) hd[h] += c.hcol[(size_t)h * c.n + j] * xt;
  }
  if (bd.fin_dots) fin_huge_partials(c, hd, red);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->iters[0] = 1; st->iters[1] = 0; st->done = 1;
    if (*bd.flag) st->neg_curv = 1;
  }
}

// ---- coupled form: P = dense diagonal blocks, A = single-entry rows + kc <= CPL_MAX rows of any shape ---------------------------
// (SURVEY C5 "+ 500 sparse sector rows": rows that tie variables of different blocks together.)  K = B + S' R S with S the kc
// coupling rows, and   K^-1 r = t - B^-1 S' c,   t = B^-1 r,   c = (R^-1 + S B^-1 S')^-1 S t.   W = B^-1 S' would be n x kc dense
// (200 MB at 500 rows), so it is never formed: the second term is a second pass over the inverse blocks (50 MB).  Per solve:
//     k_blk_apply (t)  ->  k_cpl_dot (S t, one workgroup per coupling row)  ->  k_cpl_solve (c: kc x kc dense product, one wavefront
//     per row)  ->  k_blk_apply_back (S' c gathered on the fly as the input of the block product; x~ = x~0 + t - result).
// The capacitance matrix is built at refresh from kc block passes (column r = S B^-1 S' e_r) and inverted by one workgroup
// in place (k_cap_invert: Gauss-Jordan without pivoting, the matrix is positive definite when K is).
#define CPL_MAX 512
// out[blockIdx.y * ostride + r] = (row crow[r] of A) . x[blockIdx.y * xstride ...]; use_part: a folded huge row takes the per-block
// partials the block pass left instead of walking its 50 000 entries (single right-hand side only)
__global__ void __launch_bounds__(TB) k_cpl_dot(Ctx c, BdCtx bd, const double *x, double *out, int gated, int use_part, long long xstride, long long ostride, int nrhs) {
  if (gated) { const State *st = c.st; if (st->stalled || !st->run) return; }
  if ((int)blockIdx.y >= nrhs) return;
  __shared__ double red[16];
  x += (size_t)blockIdx.y * xstride; out += (size_t)blockIdx.y * ostride;
  const int i = bd.crow[blockIdx.x], h = use_part ? bd.chuge[blockIdx.x] : -1;
  double s = 0.0;
  if (h >= 0) {
    for (int q = threadIdx.x; q < c.dP.nblk; q += TB) s += bd.part[(size_t)h * c.dP.nblk + q];
  } else {
    const int k0 = c.A.rowptr[i], k1 = c.A.rowptr[i + 1];
    int k = k0 + (int)threadIdx.x;
    for (; k + 3 * TB < k1; k += 4 * TB) {
      const double v0 = c.A.val[k], v1 = c.A.val[k + TB], v2 = c.A.val[k + 2 * TB], v3 = c.A.val[k + 3 * TB];
      const double x0 = x[c.A.col[k]], x1 = x[c.A.col[k + TB]], x2 = x[c.A.col[k + 2 * TB]], x3 = x[c.A.col[k + 3 * TB]];
      s += v0 * x0; s += v1 * x1; s += v2 * x2; s += v3 * x3;
    }
    for (; k < k1; k += TB) s += c.A.val[k] * x[c.A.col[k]];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ void __launch_bounds__(TB) k_cpl_solve(Ctx c, BdCtx bd) {
  { const State *st = c.st; if (st->stalled || !st->run) return; }
  const int r = blockIdx.x * (TB / 64) + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= bd.kc) return;
  double s = 0.0;
  if (c.plain_rhs) for (int q = lane; q < bd.kc; q += 64) { const int i = bd.crow[q]; s += bd.cap[(size_t)r * bd.kc + q] * (bd.cs[q] - c.vb[c.n + i] / c.rho[i]); }   // (S t0 - R^-1 beta)
  else for (int q = lane; q < bd.kc; q += 64) s += bd.cap[(size_t)r * bd.kc + q] * bd.cs[q];
  s = wave_sum(s);
  if (lane == 0) bd.cc[r] = s;
}
// entry j of S' v for a kc-vector v: the coupling rows' entries of column j as (position in crow, slot of the value in M), a list
// of their own (through M itself: split/rowptr -> col -> cidx -> v, four dependent loads; here three)
__device__ __forceinline__ double cpl_back_entry(const Ctx &c, const BdCtx &bd, const double *v, int j) {
  double s = 0.0;
  for (int k = bd.cbptr[j]; k < bd.cbptr[j + 1]; ++k) { const int2 e = bd.cbent[k]; s += c.M.val[e.y] * v[e.x]; }
  return s;
}
// x~ = x~0 + t - B^-1 S' c, the solve is complete
__global__ void __launch_bounds__(TB) k_blk_apply_back(Ctx c, BdCtx bd) {
  State *st = c.st;
  if (st->stalled || !st->run) return;
  __shared__ double scratch[5 * DENSE_MAX];
  __shared__ double red[16];
  const DenseP inv{c.dP.nblk, c.dP.blk, bd.binv};
  double hd[MAX_HUGE_FOLD] = {0.0, 0.0, 0.0, 0.0};
  for (int db = blockIdx.x; db < c.dP.nblk; db += gridDim.x) {
    const DenseBlk d = c.dP.blk[db];
    const double y = dense_block_mv_x<true>(inv, d, [&](int j) { return cpl_back_entry(c, bd, bd.cc, j); }, scratch);
    const int j = d.c0 + threadIdx.x;
    if ((int)threadIdx.x < d.b) {
      const double xt = c.plain_rhs ? bd.t[j] - y : c.vx[j] + (bd.t[j] - y);
      c.va[j] = xt;
      if (bd.fin_dots) for (int h = 0; h < c.nh; ++h) hd[h] += c.hcol[(size_t)h * c.n + j] * xt;
    }
    __syncthreads();
  }
  if (bd.fin_dots) fin_huge_partials(c, hd, red);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->iters[0] = 1; st->iters[1] = 0; st->done = 1;
    if (*bd.flag) st->neg_curv = 1;
  }
}
// Refresh: W_g = B^-1 S_g' for a group of 16 coupling rows [r0, r0 + 16) at once -- the one place of this path where a block
// meets several vectors, so the product runs on the matrix cores: v_mfma_f64_16x16x4_f64, D (16 rows of the block x 16
// right-hand sides) += A (16 x 4 tile of the inverse block, read transposed: the block is symmetric, so 4 rows x 16
// contiguous doubles per wave load) x B (4 x 16 tile of S_g' from LDS).  Wavefront w owns rows [32 w, 32 w + 32) of the block.
// One pass over the 50 MB of inverse blocks serves 16 columns of the capacitance matrix (the VALU kernel above: one).
// out: [16][n].
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(TB) k_blk_apply_multi(Ctx c, BdCtx bd, int r0, double *out) {
  __shared__ double xl[DENSE_MAX * 16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  static_assert(TB == 256 && DENSE_MAX == 128, "four wavefronts x two 16-row tiles cover a block");
  for (int db = blockIdx.x; db < c.dP.nblk; db += gridDim.x) {
    const DenseBlk d = c.dP.blk[db];
    const double *dv = bd.binv + d.off;
    double a[2][DENSE_MAX / 4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kt = 0; kt < DENSE_MAX / 4; ++kt) {
        const int k = 4 * kt + lk, i = 32 * w + 16 * t + li;
        a[t][kt] = (k < d.b && i < d.b) ? dv[(size_t)k * d.pitch + i] : 0.0;
      }
    for (int q = threadIdx.x; q < DENSE_MAX * 16; q += TB) xl[q] = 0.0;
    __syncthreads();
    if ((int)threadIdx.x < d.b) {
      const int j = d.c0 + threadIdx.x;
      for (int k = c.M.split[j]; k < c.M.rowptr[j + 1]; ++k) {
        const int q = bd.cidx[c.M.col[k] - c.n] - r0;
        if (q >= 0 && q < 16) xl[threadIdx.x * 16 + q] = c.M.val[k];
      }
    }
    __syncthreads();
    mfma_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < DENSE_MAX / 4; ++kt) {
      const double bv = xl[(4 * kt + lk) * 16 + li];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][kt], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][kt], bv, acc1, 0, 0, 0);
    }
    // D[row = lk + 4 r][col = li]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i0 = 32 * w + lk + 4 * r, i1 = i0 + 16;
      if (i0 < d.b) out[(size_t)li * c.n + d.c0 + i0] = acc0[r];
      if (i1 < d.b) out[(size_t)li * c.n + d.c0 + i1] = acc1[r];
    }
    __syncthreads();
  }
}
// cap <- (cap + R^-1)^-1: Gauss-Jordan without pivoting (the matrix is positive definite when K is), one launch per pivot over
// the whole machine, from one copy of the matrix into the other (no element is read after it was rewritten).  One workgroup
// walking the 2 MB in global memory took 61 us per pivot (31 ms at 501 rows); a launch takes 3.
__global__ void __launch_bounds__(TB) k_cap_diag(Ctx c, BdCtx bd) {
  const int r = blockIdx.x * TB + threadIdx.x;
  if (r >= bd.kc) return;
  const double rho = c.rho[bd.crow[r]];
  if (!(rho > 0.0)) atomicOr(bd.flag, 1);
  bd.cap[(size_t)r * bd.kc + r] += 1.0 / rho;
}
__global__ void __launch_bounds__(TB) k_cap_step(BdCtx bd, const double *src, double *dst, int p) {
  const int kc = bd.kc;
  const long long e = (long long)blockIdx.x * TB + threadIdx.x;
  if (e >= (long long)kc * kc) return;
  const int i = (int)(e / kc), j = (int)(e - (long long)i * kc);
  const double piv = src[(size_t)p * kc + p];
  if (e == 0 && !(piv > 0.0)) atomicOr(bd.flag, 1);
  const double inv = 1.0 / piv, rj = src[(size_t)p * kc + j] * inv, ci = src[(size_t)i * kc + p];
  double v;
  if (i == p) v = (j == p) ? inv : rj;
  else if (j == p) v = -ci * inv;
  else v = src[e] - ci * rj;
  dst[e] = v;
}

#include "dense_direct.h"

// Checks of the block-direct solve's fresh inverses (blk_refresh), as the dense-direct solve has them: k_blk_invert and the
// capacitance steps are Gauss-Jordan (error ~ cond^2 eps).  Per block: u = probe values, y = Binv_b u, z = B_b y with B_b = P_b +
// diag as k_blk_invert forms it; chk[0] = max |z - u| over all blocks.
__global__ void __launch_bounds__(TB) k_blk_check(Ctx c, BdCtx bd) {
  __shared__ double scratch[5 * DENSE_MAX];
  __shared__ double red[16];
  const double sigma = c.prm->sigma;
  const DenseP inv{c.dP.nblk, c.dP.blk, bd.binv};
  double worst = 0.0;
  for (int db = blockIdx.x; db < c.dP.nblk; db += gridDim.x) {
    const DenseBlk d = c.dP.blk[db];
    const double y = dense_block_mv_x(inv, d, [](int j) { return dd_probe_value(j); }, scratch);
    const int j = d.c0 + threadIdx.x;
    const bool on = (int)threadIdx.x < d.b;
    if (on) bd.t[j] = y;
    __syncthreads();
    double z = dense_block_mv_x(c.dP, d, [&](int jj) { return bd.t[jj]; }, scratch);
    if (on) {
      double dadd = sigma;
      for (int k = c.Mk.rowptr[j]; k < c.Mk.rowptr[j + 1]; ++k) {
        const int row = c.Mk.col[k] - c.n;
        if (bd.kc && bd.cidx[row] >= 0) continue;
        const double a = c.Mk.val[k]; dadd += c.rho[row] * a * a;
      }
      z += dadd * y;
      const double e = fabs(z - dd_probe_value(j));
      worst = fmax(worst, e == e ? e : 1e300);
    }
    __syncthreads();
  }
  for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long *>(bd.chk), (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))));
}
// coupled form: w = cap u, z = cap0 w (cap0: the capacitance matrix as formed, cap: its inverse); chk[1] = max |z - u|.  One workgroup.
__global__ void __launch_bounds__(TB) k_cap_check(BdCtx bd) {
  __shared__ double u[CPL_MAX], w[CPL_MAX], red[16];
  const int kc = bd.kc;
  for (int r = threadIdx.x; r < kc; r += TB) u[r] = dd_probe_value(r);
  __syncthreads();
  for (int r = threadIdx.x; r < kc; r += TB) { double s = 0.0; for (int q = 0; q < kc; ++q) s += bd.cap[(size_t)r * kc + q] * u[q]; w[r] = s; }
  __syncthreads();
  double worst = 0.0;
  for (int r = threadIdx.x; r < kc; r += TB) { double s = 0.0; for (int q = 0; q < kc; ++q) s += bd.cap0[(size_t)r * kc + q] * w[q]; const double e = fabs(s - u[r]); worst = fmax(worst, e == e ? e : 1e300); }
  for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long *>(bd.chk + 1), (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))));
}

// ---------------------------------------------------------------------------
// Elimination of slack-like variables from the linear system (launch-per-step kernels and the resident PCG).
//
// A variable y that (1) appears in exactly ONE row i of A (coefficient a), (2) has no off-diagonal entry in P and
// (3) shares its row with no other such variable couples to the rest of K = P + sigma I + A' rho A only through that row:
//     K_yy = D_y = P_yy + sigma + rho_i a^2,     K_Xy = rho_i a A(i,X)'.
// Eliminating all such y exactly (a block elimination with a DIAGONAL block) leaves, for the other variables X,
//     S = P_XX + sigma I + A_X' diag(rho~) A_X,     rho~_i = rho_i (P_yy + sigma) / D_y   (rho~_i = rho_i on rows without a y),
// i.e. the same operator with another weight on those rows; right-hand side b~_X = b_X - K_Xy b_y / D_y, back substitution
// x~_y = (b_y - rho_i a (A_X x~_X)_i) / D_y.  Why it pays: on equality rows rho_i = 1e3 rho (include/constants.h:70), so K_yy
// is huge against the Schur complement it hides and K is badly conditioned (Lasso, docs/examples/lasso.rst:41-63: the
// residual variables y with y = Ad x - b; config 3 ran 98.6 PCG iterations per ADMM iteration); S has rho~ ~ P_yy + sigma
// there.  Inertia: K > 0 <=> S > 0 and D > 0, so the convexity probe may run on the reduced operator too.
// In the kernels: the PCG vectors are zero at the eliminated variables (k_pcg_init starts them so, k_cg_B keeps w_y = 0;
// k_form_K gives their rows and columns of the resident K nothing but a diagonal entry and weighs the rows with rho~),
// k_cg_A weighs rows with rho~ (Ctx::rhoe), k_admm_finalize back-substitutes and folds b_y into the m-part of the next
// right-hand side.  The ADMM iterates themselves (x, z, y, residuals) never see any of this.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(TB) k_elim_refresh(Ctx c) {
  const double sigma = c.prm->sigma;
  for (int i = blockIdx.x * TB + threadIdx.x; i < c.m; i += gridDim.x * TB) {
    const int y = c.ecol[i];
    const double rho = c.rho[i];
    if (y < 0) { c.rhoe[i] = rho; c.ecoef[i] = 0.0; c.edinv[i] = 0.0; continue; }
    const double a = c.A.val[c.epos[i]], d0 = c.pdiag[y] + sigma, D = d0 + rho * a * a;
    c.ecoef[i] = a; c.edinv[i] = 1.0 / D; c.rhoe[i] = rho * d0 / D;
  }
}
// x~ of the eliminated variables lives in xte (per row), the PCG vectors are zero there: after the host wrote va's n-part
__global__ void __launch_bounds__(TB) k_elim_split(Ctx c) {
  for (int j = blockIdx.x * TB + threadIdx.x; j < c.n; j += gridDim.x * TB) {
    const int i = c.erow[j];
    if (i >= 0) { c.xte[i] = c.va[j]; c.va[j] = 0.0; }
  }
}

// ---------------------------------------------------------------------------
// auxiliary kernels
// ---------------------------------------------------------------------------
// Jacobi preconditioner: Minv_j = 1 / (P_jj + sigma + sum_i rho_i A_ij^2)
__global__ void __launch_bounds__(TB) k_precond(Ctx c) {
  const double sigma = c.prm->sigma;
  for (int j = blockIdx.x * TB + threadIdx.x; j < c.n; j += gridDim.x * TB) {
    double s = c.pdiag[j] + sigma;
    for (int k = c.M.split[j]; k < c.M.rowptr[j + 1]; ++k) {
      const double a = c.M.val[k];
      s += c.rhoe[c.M.col[k] - c.n] * a * a;
    }
    if (c.nelim && c.erow[j] >= 0) s = 1.0;       // (not part of the PCG system)
    c.minv[j] = 1.0 / s;
    c.g4[j].m = 1.0 / s; c.g4[c.n + j].m = 1.0 / s;
  }
}

// m-parts of the PCG input vectors from (z, z~, y, rho); rhoinv from rho.  zt_partial: zt holds A x~ WITHOUT the
// eliminated variables' terms (it was just recomputed from the PCG vector): complete it first.
__global__ void __launch_bounds__(TB) k_refresh_m(Ctx c, int zt_partial) {
  const double *y = c.xy + c.n;
  const Params prm = *c.prm;
  for (int i = blockIdx.x * TB + threadIdx.x; i < c.m; i += gridDim.x * TB) {
    const double rho = c.rho[i];
    c.rhoinv[i] = 1.0 / rho;
    const double vbn = rho * c.z[i] - y[i];
    if (c.nelim && c.ecol[i] >= 0) {
      const double a = c.ecoef[i], xt = c.xte[i];
      double ax = c.zt[i];
      if (zt_partial) c.zt[i] = ax + a * xt; else ax -= a * xt;
      c.va[c.n + i] = c.rhoe[i] * ax;
      c.vb[c.n + i] = elim_vb(c, prm, i, rho, vbn, c.xy[c.ecol[i]]);
    } else {
      c.va[c.n + i] = rho * c.zt[i];
      c.vb[c.n + i] = vbn;
    }
  }
}

// Generic SpMV used by set_iterates (z = A x) and by the kernel-level tests.
// sel: 0 = whole row, 1 = entries before split (P part), 2 = from split (A' part)
__global__ void __launch_bounds__(TB) k_spmv(DevMat Mx, const double *in, double *out, int sel) {
  LDS_DECL(1);
  for (int bi = blockIdx.x; bi < Mx.nblk; bi += gridDim.x) {
    const RowBlk b = Mx.blk[bi];
    if (IS_LONG(b)) {
      int ka = b.k0, kb = b.k1;
      if (sel == 1) kb = Mx.split[b.r0];
      if (sel == 2) ka = Mx.split[b.r0];
      const double s = long_row_dot(Mx, ka, kb, in, red);
      if (threadIdx.x == 0) out[b.r0] = s;
    } else {
      stage_products<1>(Mx, b, in, nullptr, lprod, nullptr);
      __syncthreads();
      for (int r = b.r0 + threadIdx.x; r < b.r1; r += TB) {
        int a0 = Mx.rowptr[r] - b.k0, a1 = Mx.rowptr[r + 1] - b.k0;
        if (sel == 1) a1 = Mx.split[r] - b.k0;
        if (sel == 2) a0 = Mx.split[r] - b.k0;
        out[r] = row_sum(lprod, a0, a1);
      }
    }
    __syncthreads();
  }
}

// Residuals, tolerances' norms, rho-estimate norms, objective and the cheap
// halves of both infeasibility tests in ONE launch:
//   workgroups [0, gridA)          : rows of A (A x, primal side, delta_y)
//   workgroups [gridA, gridA+gridM): rows of M (P x, A'y, dual side, delta_x)
__global__ void __launch_bounds__(TB) k_residuals(Ctx c) {
  LDS_DECL(1);
  const Params prm = *c.prm;
  const bool sc = prm.has_scaling;
  double *x = c.xy;
  if ((int)blockIdx.x < c.gridA) {
    double m_pu = 0, m_ps = 0, m_zu = 0, m_zs = 0, m_au = 0, m_as = 0, m_du = 0, m_ds = 0, lhs = 0;
    for (int bi = blockIdx.x; bi < c.A.nblk; bi += c.gridA) {
      const RowBlk b = c.A.blk[bi];
      const bool longrow = IS_LONG(b);
      if (!longrow) { stage_products<1>(c.A, b, x, nullptr, lprod, nullptr); __syncthreads(); }
      for (int i = b.r0 + (longrow ? 0 : threadIdx.x); i < b.r1; i += (longrow ? 1 : TB)) {
        double ax;
        if (longrow) ax = bi >= c.A.nwave ? huge_row_sum(c, bi, red) : long_row_dot(c.A, b.k0, b.k1, x, red);
        else ax = row_sum(lprod, c.A.rowptr[i] - b.k0, c.A.rowptr[i + 1] - b.k0);
        if (!longrow || threadIdx.x == 0) {
          const double zi = c.z[i], pr = ax + (-1.0) * zi;
          const double ei = sc ? c.Einv[i] : 1.0, e = sc ? c.E[i] : 1.0;
          m_pu = fmax(m_pu, fabs(ei * pr)); m_ps = fmax(m_ps, fabs(pr));
          m_zu = fmax(m_zu, fabs(ei * zi)); m_zs = fmax(m_zs, fabs(zi));
          m_au = fmax(m_au, fabs(ei * ax)); m_as = fmax(m_as, fabs(ax));
          // delta_y projected on the polar of the recession cone (auxil.c:375-388)
          double dy = c.dy[i];
          const double li = c.l[i], ui = c.u[i];
          if (ui > INF_BOUND) { if (li < -INF_BOUND) dy = 0.0; else dy = fmin(dy, 0.0); }
          else if (li < -INF_BOUND) dy = fmax(dy, 0.0);
          c.dxy[c.n + i] = dy;
          m_du = fmax(m_du, fabs(e * dy)); m_ds = fmax(m_ds, fabs(dy));
          lhs += ui * fmax(dy, 0.0) + li * fmin(dy, 0.0);
        }
      }
      __syncthreads();
    }
    block_max_to(m_pu, c.scal + SCI(SC_PRI_U), red); block_max_to(m_ps, c.scal + SCI(SC_PRI_S), red);
    block_max_to(m_zu, c.scal + SCI(SC_Z_U), red);   block_max_to(m_zs, c.scal + SCI(SC_Z_S), red);
    block_max_to(m_au, c.scal + SCI(SC_AX_U), red);  block_max_to(m_as, c.scal + SCI(SC_AX_S), red);
    block_max_to(m_du, c.scal + SCI(SC_DYN_U), red); block_max_to(m_ds, c.scal + SCI(SC_DYN_S), red);
    lhs = block_sum(lhs, red);
    if (threadIdx.x == 0) c.part_s0[blockIdx.x] = lhs;
    return;
  }
  const int bid = blockIdx.x - c.gridA;
  double m_du = 0, m_ds = 0, m_qu = 0, m_qs = 0, m_tu = 0, m_ts = 0, m_pu = 0, m_ps = 0;
  double m_xu = 0, m_xs = 0, obj = 0, qdx = 0;
  for (int bi = bid; bi < c.M.nblk; bi += c.gridM) {
    const RowBlk b = c.M.blk[bi];
    const bool longrow = IS_LONG(b);
    if (!longrow) { stage_products<1>(c.M, b, c.xy, nullptr, lprod, nullptr); __syncthreads(); }
    for (int j = b.r0 + (longrow ? 0 : threadIdx.x); j < b.r1; j += (longrow ? 1 : TB)) {
      double px, aty;
      if (longrow) {
        px  = long_row_dot(c.M, b.k0, c.M.split[j], c.xy, red);
        aty = long_row_dot(c.M, c.M.split[j], b.k1, c.xy, red);
      } else {
        const int a0 = c.M.rowptr[j] - b.k0, as = c.M.split[j] - b.k0, a1 = c.M.rowptr[j + 1] - b.k0;
        px = row_sum(lprod, a0, as);
        aty = row_sum(lprod, as, a1);
      }
      if (!longrow || threadIdx.x == 0) {
        const double qj = c.q[j], xj = x[j], dxj = c.dxy[j];
        double dr = qj + px;
        if (c.m > 0) dr = dr + aty;
        const double di = sc ? c.Dinv[j] : 1.0, d = sc ? c.D[j] : 1.0;
        m_du = fmax(m_du, fabs(di * dr));  m_ds = fmax(m_ds, fabs(dr));
        m_qu = fmax(m_qu, fabs(di * qj));  m_qs = fmax(m_qs, fabs(qj));
        m_tu = fmax(m_tu, fabs(di * aty)); m_ts = fmax(m_ts, fabs(aty));
        m_pu = fmax(m_pu, fabs(di * px));  m_ps = fmax(m_ps, fabs(px));
        m_xu = fmax(m_xu, fabs(d * dxj));  m_xs = fmax(m_xs, fabs(dxj));
        obj += xj * (0.5 * px + qj);
        qdx += qj * dxj;
      }
    }
    __syncthreads();
  }
  block_max_to(m_du, c.scal + SCI(SC_DUA_U), red); block_max_to(m_ds, c.scal + SCI(SC_DUA_S), red);
  block_max_to(m_qu, c.scal + SCI(SC_Q_U), red);   block_max_to(m_qs, c.scal + SCI(SC_Q_S), red);
  block_max_to(m_tu, c.scal + SCI(SC_ATY_U), red); block_max_to(m_ts, c.scal + SCI(SC_ATY_S), red);
  block_max_to(m_pu, c.scal + SCI(SC_PX_U), red);  block_max_to(m_ps, c.scal + SCI(SC_PX_S), red);
  block_max_to(m_xu, c.scal + SCI(SC_DXN_U), red); block_max_to(m_xs, c.scal + SCI(SC_DXN_S), red);
  obj = block_sum(obj, red); qdx = block_sum(qdx, red);
  if (threadIdx.x == 0) { c.part_s1[bid] = obj; c.part_s2[bid] = qdx; }
}

// Fixed-order final sums of the three partial arrays above (one workgroup).
__global__ void __launch_bounds__(TB) k_final_sums(Ctx c) {
  __shared__ double red[16];
  double s[1];
  reduce_parts<1>(c.part_s0, nullptr, nullptr, c.gridA, red, s);
  if (threadIdx.x == 0) c.scal[SCI(SC_DYLHS)] = s[0];
  reduce_parts<1>(c.part_s1, nullptr, nullptr, c.gridM, red, s);
  if (threadIdx.x == 0) c.scal[SCI(SC_OBJ)] = s[0];
  reduce_parts<1>(c.part_s2, nullptr, nullptr, c.gridM, red, s);
  if (threadIdx.x == 0) c.scal[SCI(SC_QDX)] = s[0];
}

// Second stage of the infeasibility tests (auxil.c:401-417, 456-497):
// A' dy_proj, P dx over M; A dx row test over A.
__global__ void __launch_bounds__(TB) k_certificates(Ctx c, double eps_dx, int unscaled) {
  LDS_DECL(1);
  const bool sc = c.prm->has_scaling && unscaled;
  if ((int)blockIdx.x < c.gridA) {
    double viol = 0;
    for (int bi = blockIdx.x; bi < c.A.nblk; bi += c.gridA) {
      const RowBlk b = c.A.blk[bi];
      const bool longrow = IS_LONG(b);
      if (!longrow) { stage_products<1>(c.A, b, c.dxy, nullptr, lprod, nullptr); __syncthreads(); }
      for (int i = b.r0 + (longrow ? 0 : threadIdx.x); i < b.r1; i += (longrow ? 1 : TB)) {
        double adx;
        if (longrow) adx = long_row_dot(c.A, b.k0, b.k1, c.dxy, red);
        else adx = row_sum(lprod, c.A.rowptr[i] - b.k0, c.A.rowptr[i + 1] - b.k0);
        if (!longrow || threadIdx.x == 0) {
          if (sc) adx = c.Einv[i] * adx;
          if ((c.u[i] < INF_BOUND && adx > eps_dx) || (c.l[i] > -INF_BOUND && adx < -eps_dx)) viol += 1.0;
        }
      }
      __syncthreads();
    }
    viol = block_sum(viol, red);
    if (threadIdx.x == 0 && viol > 0) atomic_max_pos(c.scal + SCI(SC_ADX_VIOL), viol);
    return;
  }
  const int bid = blockIdx.x - c.gridA;
  double m_tu = 0, m_ts = 0, m_pu = 0, m_ps = 0;
  for (int bi = bid; bi < c.M.nblk; bi += c.gridM) {
    const RowBlk b = c.M.blk[bi];
    const bool longrow = IS_LONG(b);
    if (!longrow) { stage_products<1>(c.M, b, c.dxy, nullptr, lprod, nullptr); __syncthreads(); }
    for (int j = b.r0 + (longrow ? 0 : threadIdx.x); j < b.r1; j += (longrow ? 1 : TB)) {
      double pdx, atdy;
      if (longrow) {
        pdx  = long_row_dot(c.M, b.k0, c.M.split[j], c.dxy, red);
        atdy = long_row_dot(c.M, c.M.split[j], b.k1, c.dxy, red);
      } else {
        const int a0 = c.M.rowptr[j] - b.k0, as = c.M.split[j] - b.k0, a1 = c.M.rowptr[j + 1] - b.k0;
        pdx = row_sum(lprod, a0, as);
        atdy = row_sum(lprod, as, a1);
      }
      if (!longrow || threadIdx.x == 0) {
        const double di = sc ? c.Dinv[j] : 1.0;
        m_tu = fmax(m_tu, fabs(di * atdy)); m_ts = fmax(m_ts, fabs(atdy));
        m_pu = fmax(m_pu, fabs(di * pdx));  m_ps = fmax(m_ps, fabs(pdx));
      }
    }
    __syncthreads();
  }
  block_max_to(m_tu, c.scal + SCI(SC_ATDY_U), red); block_max_to(m_ts, c.scal + SCI(SC_ATDY_S), red);
  block_max_to(m_pu, c.scal + SCI(SC_PDX_U), red);  block_max_to(m_ps, c.scal + SCI(SC_PDX_S), red);
}

// ---------------------------------------------------------------------------
// Ruiz equilibration on the device (reference src/scaling.c:44-156).  One wavefront
// per matrix row; the products are formed in the reference's order
// ((val * row factor) * column factor, then * c) so both device copies of A hold the
// same bits.  Scratch: dn (n), en (m) live in kp / dy; scalars in scal[SC_RUIZ..].
// ---------------------------------------------------------------------------
__device__ __forceinline__ double clip_scaling(double v) {
  if (v < 1e-4) v = 1.0;
  if (v > 1e4) v = 1e4;
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// dn[j] = 1/sqrt(clip(max |column j of [P;A]|)), en[i] = 1/sqrt(clip(max |row i of A|))
__global__ void __launch_bounds__(TB) k_ruiz_norms(Ctx c, double *dn, double *en) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (TB / 64);
  for (int r = w; r < c.n + c.m; r += nw) {
    const bool isM = r < c.n;
    const DevMat &X = isM ? c.M : c.A;
    const int row = isM ? r : r - c.n;
    double mx = 0.0;
    for (int k = X.rowptr[row] + lane; k < X.rowptr[row + 1]; k += 64) mx = fmax(mx, fabs(X.val[k]));
    mx = wave_max(mx);
    if (lane == 0) { if (isM) dn[row] = 1.0 / sqrt(clip_scaling(mx)); else en[row] = 1.0 / sqrt(clip_scaling(mx)); }
  }
}

// apply the pass: matrices, q, accumulated D and E
__global__ void __launch_bounds__(TB) k_ruiz_apply(Ctx c, const double *dn, const double *en,
                                                   double *Mval, double *Aval) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (TB / 64);
  for (int r = w; r < c.n + c.m; r += nw) {
    if (r < c.n) {
      const int j = r, sp = c.M.split[j];
      const double dj = dn[j];
      for (int k = c.M.rowptr[j] + lane; k < c.M.rowptr[j + 1]; k += 64) {
        const int cc = c.M.col[k];
        // P(j,cc): row factor of the stored entry, then column factor; A'(j, n+i): E_i then D_j
        if (k < sp) {
          // the stored triu entry has row = min(j,cc), column = max(j,cc)
          const double fr = j < cc ? dj : dn[cc], fc = j < cc ? dn[cc] : dj;
          Mval[k] = (Mval[k] * fr) * fc;
        } else Mval[k] = (Mval[k] * en[cc - c.n]) * dj;
      }
      if (lane == 0) { c.q[j] = c.q[j] * dj; c.D[j] = dj * c.D[j]; }
    } else {
      const int i = r - c.n;
      const double ei = en[i];
      for (int k = c.A.rowptr[i] + lane; k < c.A.rowptr[i + 1]; k += 64) Aval[k] = (Aval[k] * ei) * dn[c.A.col[k]];
      if (lane == 0) c.E[i] = ei * c.E[i];
    }
  }
}

// cost normalisation, step 1: per-workgroup partial sums of the P column norms, max |q|
__global__ void __launch_bounds__(TB) k_ruiz_cost_norms(Ctx c) {
  __shared__ double red[16];
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (TB / 64);
  double acc = 0.0, qm = 0.0;
  for (int j = w; j < c.n; j += nw) {
    double mx = 0.0;
    for (int k = c.M.rowptr[j] + lane; k < c.M.split[j]; k += 64) mx = fmax(mx, fabs(c.M.val[k]));
    mx = wave_max(mx);
    if (lane == 0) { acc += mx; qm = fmax(qm, fabs(c.q[j])); }
  }
  acc = block_sum(acc, red);
  block_max_to(qm, c.scal + SCI(SC_QDX), red);      // reuse a slot: max |q|
  if (threadIdx.x == 0) c.part_s0[blockIdx.x] = acc;
}

// step 2 (one workgroup): c_t = 1 / clip(max(mean colnorm, clip(|q|_inf))), c *= c_t
__global__ void __launch_bounds__(TB) k_ruiz_cost_scalar(Ctx c, int nparts) {
  __shared__ double red[16];
  double s[1];
  reduce_parts<1>(c.part_s0, nullptr, nullptr, nparts, red, s);
  if (threadIdx.x == 0) {
    const double mean = s[0] / (double)c.n;
    double ct = fmax(mean, clip_scaling(c.scal[SCI(SC_QDX)]));
    ct = 1.0 / clip_scaling(ct);
    c.scal[SCI(SC_OBJ)] = ct;                          // this pass's factor
    c.scal[SCI(SC_DYLHS)] = c.scal[SCI(SC_DYLHS)] * ct;  // accumulated c
    c.scal[SCI(SC_QDX)] = 0.0;
  }
}

// step 3: P <- c_t P, q <- c_t q
__global__ void __launch_bounds__(TB) k_ruiz_cost_apply(Ctx c, double *Mval) {
  const double ct = c.scal[SCI(SC_OBJ)];
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (TB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (TB / 64);
  for (int j = w; j < c.n; j += nw) {
    for (int k = c.M.rowptr[j] + lane; k < c.M.split[j]; k += 64) Mval[k] *= ct;
    if (lane == 0) c.q[j] *= ct;
  }
}

// finish: l, u <- E l, E u ; Dinv, Einv
__global__ void __launch_bounds__(TB) k_ruiz_finish(Ctx c) {
  for (int i = blockIdx.x * TB + threadIdx.x; i < c.m; i += gridDim.x * TB) {
    const double e = c.E[i];
    c.l[i] = c.l[i] * e; c.u[i] = c.u[i] * e; c.Einv[i] = 1.0 / e;
  }
  for (int j = blockIdx.x * TB + threadIdx.x; j < c.n; j += gridDim.x * TB) c.Dinv[j] = 1.0 / c.D[j];
}

__global__ void __launch_bounds__(TB) k_fill(double *p, double v, int cnt) {
  for (int i = blockIdx.x * TB + threadIdx.x; i < cnt; i += gridDim.x * TB) p[i] = v;
}

// dst[k] = src_val[map[k]] (map < 0: structural zero inside a dense block)
__global__ void __launch_bounds__(TB) k_repack(const double *src, const int *map, double *dst, long long cnt) {
  for (long long k = (long long)blockIdx.x * TB + threadIdx.x; k < cnt; k += (long long)gridDim.x * TB) {
    const int q = map[k];
    dst[k] = q >= 0 ? src[q] : 0.0;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct HostMat {         // host image of a device CSR matrix
  int nrows = 0, ncols = 0, nstream = 0, nwave = 0;
  std::vector<int> rowptr, col, split;
  std::vector<double> val;
  std::vector<RowBlk> blk;
  int *d_rowptr = nullptr, *d_col = nullptr, *d_split = nullptr;
  unsigned short *d_col16 = nullptr;
  double *d_val = nullptr;
  RowBlk *d_blk = nullptr;
};

struct hipeng {
  int device = 0;
  int n = 0, m = 0;
  hipStream_t stream = nullptr;
  HostMat A, M, Mr;                    // Mr: remainder of M outside P's dense blocks (empty: no dense blocks)
  std::vector<int> Mr_src, dP_src;     // slot in M of every Mr entry / dense entry (-1: structural zero)
  std::vector<int> hrows, hcol_src;    // huge rows of A folded into k_cg_B; slot in M of hcol_k[j] (-1: zero)
  int *d_hcol_src = nullptr;
  double *d_hcol = nullptr;
  std::vector<DenseBlk> dP_blks;
  int *d_Mr_src = nullptr, *d_dP_src = nullptr;
  DenseBlk *d_dP_blks = nullptr;
  double *d_dP_val = nullptr;
  std::vector<int> A_csc2csr;          // CSC slot of A -> CSR slot
  std::vector<int> P_toM_up, P_toM_lo; // triu(P) slot -> slots in M (lo = -1 on the diagonal)
  std::vector<int> A_toM;              // CSC slot of A -> slot in M
  std::vector<double> pdiag;
  Ctx c{};
  Params prm{};
  Params *d_prm = nullptr;
  std::vector<void *> allocs;
  std::map<int, hipGraphExec_t> graphs;
  int K = 8;
  int spec_lo = 0;           // unrolled slots below this prefetch without a look at the flags (smallest count of the last window)
  bool split = false;     // large A: vector update and operator apply as two launches (plain 8-byte gathers)
  int rlA = 8, rlM = 8;   // lanes per row segment in the PCG kernels
  bool calibrated = false;
  double ex_theta0 = 1.0;    // OSQP_AMD_EXTRAP (0 disables the extrapolated PCG start)
  bool start_dirty = true;   // [x~ | rho z~] was rewritten from outside the loop: PCG start history is void
  long long admm_total = 0;            // host copy of State::admm_done
  State *h_state = nullptr;            // pinned staging for the state read-back and the target upload
  long long *h_target = nullptr;
  int trace = 0;            // OSQP_AMD_TRACE: 1 = one line per window, 2 = every HIP call of the run loop
  hipeng_stats stats{};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  long long admm_done_seen = 0;
  bool res_on = false;       // resident PCG structures built (problem fits the register files)
  bool res_use = false;      // ... and in use
  int res_fails = 0;         // launches that gave up waiting, in a row (one sends the rest of the run_admm call to the launch-per-step
                             // kernels; the third in a row sends the engine there for good; a call without one starts the count again)
  long long res_gave_up = 0; // ... since create
  int res_cus = 248;         // CUs a process's resident launches may hold together on this device (all but one per XCD)
  long long res_slow = 0, res_slow_max = 0, res_repub = 0;   // State::res_slow / res_slow_max / res_repub added up over the windows
  long long res_slow_hist[4] = {0, 0, 0, 0}; unsigned res_slow_xcc = 0; int res_slow_last[4] = {0, 0, 0, 0};
  ResCtx rc{};
  BrCtx bc{};                // block-resident form (res_kind 2)
  int res_kind = 0;          // 0: launch-per-step only, 1: k_pcg_resident, 2: k_pcg_blockres, 3: block-direct solve (k_blk_apply / k_blk_finish), 4: dense-direct solve (dense_direct.h)
  BdCtx bd{};                // block-direct form (res_kind 3)
  DdCtx dd{};                // dense-direct form (res_kind 4)
  double *dd_init_r = nullptr; int dd_init_stride = 4; double dd_check = 0.0; bool dd_chol = false;   // what direct_disable restores; the last inverse's check; the Cholesky route was needed
  bool elim_rhs_dirty = false;   // q, the scaling, the matrices or the iterates changed since the m-part of the right-hand side was formed: with
                             // eliminated variables it carries their q_y and coefficients (elim_vb), so hipeng_run_admm forms it again first
  std::vector<double> h_rho; // host copy of rho (the capacitance matrix of the block-direct form needs the huge rows' entries)
  std::vector<int> erow, ecol, epos;   // host images of Ctx::erow / ecol / epos (empty: no variable is eliminated)
  size_t res_lds = 0;
  long long res_nnz = 0;
};

template <typename T>
static int dev_alloc(hipeng *e, T **p, size_t count) {
  void *q = nullptr;
  if (count == 0) count = 1;
  HIPCHK(hipMalloc(&q, count * sizeof(T)));
  HIPCHK(hipMemsetAsync(q, 0, count * sizeof(T), e->stream));
  e->allocs.push_back(q);
  *p = static_cast<T *>(q);
  return 0;
}

// Words that one CU polls while another writes them (exchange flags, tagged granules) live in UNCACHED device memory
// (hipDeviceMallocUncached): in ordinary memory an XCD's L2 can keep an old copy of such a line for good (a whole XCD
// then polls the old value while the others have moved on: seen about once in 1e5 exchanges on small problems, with
// every kind of load); uncached polls are also faster (143.6 against 149.8 us per ADMM iteration at config 2).
// OSQP_AMD_RESIDENT_FINE=1: fine-grained (coherent) but cacheable; =0: like everything else.
template <typename T>
static int dev_alloc_polled(hipeng *e, T **p, size_t count) {
  static const int fine = getenv("OSQP_AMD_RESIDENT_FINE") ? atoi(getenv("OSQP_AMD_RESIDENT_FINE")) : 2;
  if (!fine) return dev_alloc(e, p, count);
  void *q = nullptr;
  if (count == 0) count = 1;
  HIPCHK(hipExtMallocWithFlags(&q, count * sizeof(T), fine == 2 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
  HIPCHK(hipMemsetAsync(q, 0, count * sizeof(T), e->stream));
  e->allocs.push_back(q);
  *p = static_cast<T *>(q);
  return 0;
}

// Greedy row blocks: as many whole rows as fit `chunk` products (at least one
// row; a single row may exceed MAX_CHUNK and then takes the long-row path).
static void build_blocks(HostMat &H, int chunk, bool huge_ok, const std::vector<char> *skip = nullptr, int long_row = LONG_ROW) {
  // stream blocks first (runs of consecutive short rows, at most `chunk` products),
  // then one block per long row: the PCG kernels give each long row a wavefront
  H.blk.clear();
  std::vector<RowBlk> longs, huges;
  int r = 0;
  while (r < H.nrows) {
    if (skip && (*skip)[r]) { r++; continue; }     // rows served by another path (dense blocks of P)
    const int k0 = H.rowptr[r];
    if (H.rowptr[r + 1] - k0 >= long_row) {
      ((huge_ok && H.rowptr[r + 1] - k0 >= HUGE_ROW) ? huges : longs).push_back({r, r + 1, k0, H.rowptr[r + 1]});
      r++; continue;
    }
    int r1 = r + 1;
    while (r1 < H.nrows && !(skip && (*skip)[r1]) && H.rowptr[r1 + 1] - H.rowptr[r1] < long_row &&
           H.rowptr[r1 + 1] - k0 <= chunk && (r1 - r) < 8 * TB) r1++;
    H.blk.push_back({r, r1, k0, H.rowptr[r1]});
    r = r1;
  }
  H.nstream = (int)H.blk.size();
  H.blk.insert(H.blk.end(), longs.begin(), longs.end());
  H.nwave = (int)H.blk.size();
  H.blk.insert(H.blk.end(), huges.begin(), huges.end());
}

static int pick_chunk(long long nnz, int nrows) {
  // about one to two workgroups per CU on small problems (every workgroup of a
  // consumer kernel re-reduces one dot partial per producer workgroup, so the
  // grid is kept near the CU count), full 2048-product chunks on large ones
  int chunk = 256;
  while (chunk < MAX_CHUNK && nnz / chunk > 640) chunk <<= 1;
  (void)nrows;
  if (const char *x = getenv("OSQP_AMD_CHUNK")) { const int v = atoi(x); if (v >= 64 && v <= MAX_CHUNK) chunk = v; }   // tuning experiments
  return chunk;
}

static int upload_mat(hipeng *e, HostMat &H) {
  if (dev_alloc(e, &H.d_rowptr, H.rowptr.size())) return HIPENG_ERR_HIP;
  if (dev_alloc(e, &H.d_col, H.col.size())) return HIPENG_ERR_HIP;
  if (dev_alloc(e, &H.d_val, H.val.size())) return HIPENG_ERR_HIP;
  if (dev_alloc(e, &H.d_blk, H.blk.size())) return HIPENG_ERR_HIP;
  HIPCHK(hipMemcpyAsync(H.d_rowptr, H.rowptr.data(), H.rowptr.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  if (!H.col.empty()) {
    HIPCHK(hipMemcpyAsync(H.d_col, H.col.data(), H.col.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(H.d_val, H.val.data(), H.val.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
  }
  if (!H.blk.empty())
    HIPCHK(hipMemcpyAsync(H.d_blk, H.blk.data(), H.blk.size() * sizeof(RowBlk), hipMemcpyHostToDevice, e->stream));
  if (H.nwave > H.nstream && !H.col.empty() && *std::max_element(H.col.begin(), H.col.end()) <= 0xffff) {
    // long rows (one wavefront each in the PCG kernels) and every column id fits 16 bits: a second, narrower index array
    std::vector<unsigned short> c16(H.col.begin(), H.col.end());
    if (dev_alloc(e, &H.d_col16, c16.size())) return HIPENG_ERR_HIP;
    HIPCHK(hipMemcpyAsync(H.d_col16, c16.data(), c16.size() * sizeof(unsigned short), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));       // (the source is a local)
  }
  if (!H.split.empty()) {
    if (dev_alloc(e, &H.d_split, H.split.size())) return HIPENG_ERR_HIP;
    HIPCHK(hipMemcpyAsync(H.d_split, H.split.data(), H.split.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  }
  return 0;
}

static DevMat dev_view(const HostMat &H) {
  DevMat d;
  d.nrows = H.nrows; d.nblk = (int)H.blk.size(); d.nstream = H.nstream; d.nwave = H.nwave;
  d.rowptr = H.d_rowptr; d.col = H.d_col; d.col16 = H.d_col16; d.val = H.d_val; d.split = H.d_split; d.blk = H.d_blk;
  return d;
}

// CSC (int64) -> CSR (int32) of A, columns ascending inside each row.
static void build_A(hipeng *e, const csc *A) {
  const int n = e->n, m = e->m;
  const long long nnz = A->p[n];
  HostMat &H = e->A;
  H.nrows = m; H.ncols = n;
  H.rowptr.assign(m + 1, 0);
  for (long long k = 0; k < nnz; k++) H.rowptr[A->i[k] + 1]++;
  for (int i = 0; i < m; i++) H.rowptr[i + 1] += H.rowptr[i];
  H.col.assign(nnz, 0); H.val.assign(nnz, 0.0);
  e->A_csc2csr.assign(nnz, 0);
  std::vector<int> nxt(H.rowptr.begin(), H.rowptr.end() - 1);
  for (int j = 0; j < n; j++)
    for (long long k = A->p[j]; k < A->p[j + 1]; k++) {
      const int dst = nxt[A->i[k]]++;
      H.col[dst] = j; H.val[dst] = A->x[k];
      e->A_csc2csr[k] = dst;
    }
  build_blocks(H, pick_chunk(nnz, m), true);     // k_cg_A slices huge rows over the grid
}

// Fused row matrix M = [P_full | A'] with the reference's summation order.
static void build_M(hipeng *e, const csc *P, const csc *A) {
  const int n = e->n;
  const long long nnzP = P->p[n], nnzA = A->p[n];
  HostMat &H = e->M;
  H.nrows = n; H.ncols = n + e->m;
  std::vector<int> up(n, 0), lo(n, 0);
  for (int j = 0; j < n; j++)
    for (long long k = P->p[j]; k < P->p[j + 1]; k++) {
      const int i = (int)P->i[k];
      up[i]++;                 // entry (i,j), j >= i : upper part of row i
      if (i != j) lo[j]++;     // mirrored entry (j,i), i < j : lower part of row j
    }
  H.rowptr.assign(n + 1, 0); H.split.assign(n, 0);
  for (int j = 0; j < n; j++) {
    H.split[j] = H.rowptr[j] + up[j] + lo[j];
    H.rowptr[j + 1] = H.split[j] + (int)(A->p[j + 1] - A->p[j]);
  }
  const long long tot = H.rowptr[n];
  H.col.assign(tot, 0); H.val.assign(tot, 0.0);
  e->P_toM_up.assign(nnzP, -1); e->P_toM_lo.assign(nnzP, -1); e->A_toM.assign(nnzA, -1);
  e->pdiag.assign(n, 0.0);
  std::vector<int> nu(n), nl(n);
  for (int j = 0; j < n; j++) { nu[j] = H.rowptr[j]; nl[j] = H.rowptr[j] + up[j]; }
  // scanning columns in ascending order fills each row's upper part in
  // ascending column order; a column's own entries (ascending rows) fill the
  // lower part of row j in ascending column order
  for (int j = 0; j < n; j++)
    for (long long k = P->p[j]; k < P->p[j + 1]; k++) {
      const int i = (int)P->i[k];
      int d = nu[i]++;
      H.col[d] = j; H.val[d] = P->x[k]; e->P_toM_up[k] = d;
      if (i != j) { d = nl[j]++; H.col[d] = i; H.val[d] = P->x[k]; e->P_toM_lo[k] = d; }
      else e->pdiag[j] = P->x[k];
    }
  for (int j = 0; j < n; j++) {
    int d = H.split[j];
    for (long long k = A->p[j]; k < A->p[j + 1]; k++, d++) {
      H.col[d] = n + (int)A->i[k]; H.val[d] = A->x[k]; e->A_toM[k] = d;
    }
  }
  build_blocks(H, pick_chunk(tot, n), false);    // rows of M need their full sum inside k_cg_B
}

// Dense diagonal blocks of P (BASELINE config 5: block-diagonal covariance).  Column j
// starts a block iff no later column of triu(P) reaches above row j; a block of 32..128
// rows that is at least half full becomes a dense array for k_cg_B, the rest of M becomes the
// remainder matrix Mr.  Values of both are gathered from M on the device (k_repack), so
// every path that changes M (scaling, osqp_update_P/A) only has to repack.
static void build_dense(hipeng *e, const csc *P) {
  const int n = e->n;
  if (n == 0) return;
  std::vector<char> dense(n, 0);
  const HostMat &M = e->M;
  long long off = 0;
  if (const char *v = getenv("OSQP_AMD_DENSE_P")) if (atoi(v) == 0) goto build_remainder;
  {
  std::vector<int> lo(n), smin(n + 1, n);
  for (int j = 0; j < n; j++) {
    lo[j] = j;
    for (long long k = P->p[j]; k < P->p[j + 1]; k++) lo[j] = std::min(lo[j], (int)P->i[k]);
  }
  for (int j = n - 1; j >= 0; j--) smin[j] = std::min(smin[j + 1], lo[j]);
  int s0 = 0;
  for (int j = 1; j <= n; j++) {
    if (j < n && smin[j] < j) continue;          // column j (or a later one) still reaches into the block
    const long long b = j - s0, nnz = P->p[j] - P->p[s0];
    long long pitch = (b + 1) & ~1ll;
    if (pitch % 32 == 0 && pitch < DENSE_MAX) pitch += 2;
    if (b >= 32 && b <= DENSE_MAX && 2 * nnz - b >= (b * b) / 2 && off + b * pitch < (1ll << 31)) {
      for (int i = s0; i < j; i++) dense[i] = 1;
      e->dP_blks.push_back({s0, (int)b, (int)off, (int)pitch});
      off += b * pitch;
    }
    s0 = j;
  }
  }
build_remainder:
  // Huge rows of A (sliced over the grid in k_cg_A): their entries leave the A' part of the matrix k_cg_B streams
  // and become dense n-vectors hcol_k[j] = A(h_k, j); k_cg_B adds hcol_k[j] * t_{h_k} to row j with
  // t_{h_k} = rho * (sum of the k_cg_A partials) formed by every workgroup itself -- no k_huge_reduce launch.
  e->hrows.clear();
  for (int q = e->A.nwave; q < (int)e->A.blk.size(); q++) e->hrows.push_back(e->A.blk[q].r0);
  if ((int)e->hrows.size() > MAX_HUGE_FOLD) e->hrows.clear();        // many huge rows: keep the k_huge_reduce path
  if (const char *v = getenv("OSQP_AMD_HFOLD")) if (atoi(v) == 0) e->hrows.clear();   // tuning experiments
  if (e->dP_blks.empty() && e->hrows.empty()) return;
  std::vector<int> hidx(e->m > 0 ? e->m : 1, -1);
  for (size_t k = 0; k < e->hrows.size(); k++) hidx[e->hrows[k]] = (int)k;
  e->hcol_src.assign(e->hrows.size() * (size_t)n, -1);
  e->dP_src.assign((size_t)off, -1);
  for (const DenseBlk &d : e->dP_blks)
    for (int i = d.c0; i < d.c0 + d.b; i++)
      for (int k = M.rowptr[i]; k < M.split[i]; k++)     // P part of the row: all inside the block
        e->dP_src[(size_t)d.off + (size_t)(i - d.c0) * d.pitch + (M.col[k] - d.c0)] = k;
  HostMat &R = e->Mr;
  R.nrows = n; R.ncols = M.ncols;
  R.rowptr.assign(n + 1, 0);
  R.col.clear(); R.val.clear(); e->Mr_src.clear();
  for (int i = 0; i < n; i++) {
    for (int k = dense[i] ? M.split[i] : M.rowptr[i]; k < M.rowptr[i + 1]; k++) {
      if (k >= M.split[i] && hidx[M.col[k] - n] >= 0) { e->hcol_src[(size_t)hidx[M.col[k] - n] * n + i] = k; continue; }
      R.col.push_back(M.col[k]); R.val.push_back(M.val[k]); e->Mr_src.push_back(k);
    }
    R.rowptr[i + 1] = (int)R.col.size();
  }
  build_blocks(R, pick_chunk(std::max(1, R.rowptr[n]), n), false, e->dP_blks.empty() ? nullptr : &dense);
}

// refresh the dense rows and the remainder matrix from the values of M (device side)
static int repack_dense(hipeng *e) {
  if (e->dP_blks.empty() && e->hrows.empty()) return 0;
  const long long nr = (long long)e->Mr_src.size(), nd = (long long)e->dP_src.size(), nh = (long long)e->hcol_src.size();
  if (nr) hipLaunchKernelGGL(k_repack, dim3((unsigned)std::min<long long>(4096, (nr + TB - 1) / TB)), dim3(TB), 0, e->stream,
                             e->M.d_val, e->d_Mr_src, e->Mr.d_val, nr);
  if (nd) hipLaunchKernelGGL(k_repack, dim3((unsigned)std::min<long long>(4096, (nd + TB - 1) / TB)), dim3(TB), 0, e->stream,
                             e->M.d_val, e->d_dP_src, e->d_dP_val, nd);
  if (nh) hipLaunchKernelGGL(k_repack, dim3((unsigned)std::min<long long>(4096, (nh + TB - 1) / TB)), dim3(TB), 0, e->stream,
                             e->M.d_val, e->d_hcol_src, e->d_hcol, nh);
  HIPCHK(hipGetLastError());
  return 0;
}


// ---- resident PCG: symbolic K, row partition, register layout --------------------------------------
// Resident launches need their workgroups co-resident, one per CU.  With grids sized to the problem several engines of one process
// can be inside resident windows at the same time as long as their grids fit the device together: a per-device budget of CUs
// (all but one per XCD), taken for the span launch ... sync of a window and handed back afterwards.  (Round 2 had a mutex here:
// "one QP per stream" then meant one resident QP at a time.)  Across processes there is no such book-keeping: a launch that
// finds CUs taken waits for them -- its first wait has 20 ms -- and gives up otherwise.
struct CuBudget { std::mutex mu; std::condition_variable cv; int avail = -1; };
static CuBudget g_res_cu[16];
struct CuLease {
  CuBudget *b = nullptr; int n = 0;
  void take(int device, int total, int want) {
    b = &g_res_cu[device & 15]; n = std::min(want, total);
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->avail < 0) b->avail = total;
    b->cv.wait(lk, [&] { return b->avail >= n; });
    b->avail -= n;
  }
  void give() { if (b) { { std::lock_guard<std::mutex> lk(b->mu); b->avail += n; } b->cv.notify_all(); b = nullptr; } }
  ~CuLease() { give(); }
};
static const int RES_E_LIST[] = {8, 16, 20, 24, 32, 48, 64};

template <int E> static int res_set_lds(size_t lds) {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pcg_resident<E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return 0;
}
static void launch_dense_direct(hipeng *e);
static void launch_resident(hipeng *e) {
  if (e->res_kind == 4) { launch_dense_direct(e); return; }
  if (e->res_kind == 3) {
    const int gb = std::max(1, std::min(1024, e->c.dP.nblk));
    hipLaunchKernelGGL(k_blk_apply, dim3(gb), dim3(TB), 0, e->stream, e->c, e->bd, (const double *)e->c.init_r, e->bd.t, 1);
    if (e->bd.kc) {
      hipLaunchKernelGGL(k_cpl_dot, dim3(e->bd.kc), dim3(TB), 0, e->stream, e->c, e->bd, (const double *)e->bd.t, e->bd.cs, 1, 1, 0ll, 0ll, 1);
      hipLaunchKernelGGL(k_cpl_solve, dim3((e->bd.kc + TB / 64 - 1) / (TB / 64)), dim3(TB), 0, e->stream, e->c, e->bd);
      hipLaunchKernelGGL(k_blk_apply_back, dim3(gb), dim3(TB), 0, e->stream, e->c, e->bd);
      return;
    }
    hipLaunchKernelGGL(k_blk_finish, dim3(std::max(1, std::min(256, (e->n + TB - 1) / TB))), dim3(TB), 0, e->stream, e->c, e->bd);
    return;
  }
  if (e->res_kind == 2) {
    const dim3 g2(e->bc.nwg), b2(RES_TB);
    switch (e->c.nh) {
      case 0: hipLaunchKernelGGL(k_pcg_blockres<0>, g2, b2, 0, e->stream, e->c, e->bc); break;
      case 1: hipLaunchKernelGGL(k_pcg_blockres<1>, g2, b2, 0, e->stream, e->c, e->bc); break;
      case 2: hipLaunchKernelGGL(k_pcg_blockres<2>, g2, b2, 0, e->stream, e->c, e->bc); break;
      case 3: hipLaunchKernelGGL(k_pcg_blockres<3>, g2, b2, 0, e->stream, e->c, e->bc); break;
      default: hipLaunchKernelGGL(k_pcg_blockres<4>, g2, b2, 0, e->stream, e->c, e->bc); break;
    }
    return;
  }
  const dim3 g(e->rc.nwg), b(RES_TB);
  switch (e->rc.E) {
    case 8:  hipLaunchKernelGGL(k_pcg_resident<8>,  g, b, e->res_lds, e->stream, e->c, e->rc); break;
    case 16: hipLaunchKernelGGL(k_pcg_resident<16>, g, b, e->res_lds, e->stream, e->c, e->rc); break;
    case 20: hipLaunchKernelGGL(k_pcg_resident<20>, g, b, e->res_lds, e->stream, e->c, e->rc); break;
    case 24: hipLaunchKernelGGL(k_pcg_resident<24>, g, b, e->res_lds, e->stream, e->c, e->rc); break;
    case 32: hipLaunchKernelGGL(k_pcg_resident<32>, g, b, e->res_lds, e->stream, e->c, e->rc); break;
    case 48: hipLaunchKernelGGL(k_pcg_resident<48>, g, b, e->res_lds, e->stream, e->c, e->rc); break;
    default: hipLaunchKernelGGL(k_pcg_resident<64>, g, b, e->res_lds, e->stream, e->c, e->rc); break;
  }
}

// Host-side set-up work of the resident PCG (symbolic K, register layout) is independent per row / per workgroup: a few threads.
template <class F>
static void parallel_chunks(int count, int min_chunk, F fn) {
  int T = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *x = getenv("OSQP_AMD_SETUP_THREADS")) T = std::max(1, std::min(64, atoi(x)));
  T = std::min(T, std::max(1, count / min_chunk));
  if (T <= 1) { fn(0, count, 0, 1); return; }
  std::vector<std::thread> th;
  for (int k = 0; k < T; k++) th.emplace_back([=, &fn] { fn((int)((long long)count * k / T), (int)((long long)count * (k + 1) / T), k, T); });
  for (auto &t : th) t.join();
}

// Returns 0 (with e->res_on set when the problem qualifies) or a HIPENG error.  Not qualifying is not an error.
#define RES_NO(why) do { if (e->trace) fprintf(stderr, "[osqp_amd] resident PCG not used: %s\n", why); return 0; } while (0)
// What build_resident works out on the host before anything is uploaded (hipeng_resident_plan hands it to the CPU tests).
struct ResPlanOut {
  bool ok = false; int nwg = 0, E = 0, npad = 0; long long nnzK = 0;
  std::vector<ResWG> wg; std::vector<int> Kptr, Kcol, kdst; std::vector<unsigned short> rowpos, col; std::vector<unsigned long long> brk;
};
static int build_resident(hipeng *e, int plan_nwg = 0, ResPlanOut *po = nullptr) {
  e->res_on = e->res_use = false;
  const int n = e->n;
  int want = 1, min_n = 256;
  if (const char *x = getenv("OSQP_AMD_RESIDENT")) want = atoi(x);
  if (const char *x = getenv("OSQP_AMD_RESIDENT_MIN_N")) min_n = atoi(x);
  if (!want) RES_NO("OSQP_AMD_RESIDENT=0");
  if (n < min_n || n > RES_MAXN) RES_NO("n outside [OSQP_AMD_RESIDENT_MIN_N, 15616]");
  if (!e->hrows.empty() || e->A.nwave < (int)e->A.blk.size()) RES_NO("A has rows of 8192 or more entries");   // (their outer products alone overflow the register files)
  int nwg = plan_nwg < 0 ? -plan_nwg : plan_nwg;      // (plan mode: nwg > 0 keeps that grid, nwg < 0 sizes the grid for a machine of -nwg CUs)
  const bool fixed_grid = po && plan_nwg > 0;
  if (!po) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, e->device));
    // One CU per XCD stays free.  A grid that takes EVERY CU can be preempted in mid-launch when anything else wants a CU:
    // measured (profiles/r03_resident_slow_waits.txt) as waits of 1.05-1.10 ms, always the 32 workgroups of XCC 0 at once,
    // about once per 20-30 s of 256-workgroup launches and never with a smaller grid -- the time it takes to save and
    // restore an XCD's 32 x 570 KB of registers and LDS.  (Round 2's 1 ms wait limit turned each of them into a give-up.)
    nwg = std::min(256, prop.multiProcessorCount) - 8;
    if (const char *x = getenv("OSQP_AMD_RESIDENT_ALL_CUS")) if (atoi(x)) nwg += 8;
    e->res_cus = nwg;
    if (prop.sharedMemPerBlock < 64 * 1024) RES_NO("too little LDS");
  }
  if (nwg <= 0 || nwg > 256 || (long long)nwg * RES_MAXROWS < n) RES_NO("too few CUs");
  const HostMat &M = e->M, &A = e->A;
  const auto tb0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++)
    for (int k = M.split[i] + 1; k < M.rowptr[i + 1]; k++) if (M.col[k] <= M.col[k - 1]) RES_NO("a column of A is not sorted by row (or repeats one)");   // the merge in k_form_K wants ascending rows
  // pattern of K, row by row (sorted), with the slot of P(i,j) in M
  const size_t cap_total = (size_t)nwg * RES_PT * 64;
  std::vector<int> Kptr(n + 1, 0), Kcol, Kps;
  {
    // rows in contiguous chunks, one thread each with its own scratch; the chunks' lists are joined in order afterwards
    std::vector<std::vector<int>> cKcol(64), cKps(64), cLen(64);
    std::atomic<int> bad{0};
    int used = 1;
    parallel_chunks(n, 256, [&](int lo, int hi, int tid, int T) {
      if (tid == 0) used = T;
      std::vector<int> mark(n, -1), pslot(n, -1);
      std::vector<int> &kc = cKcol[tid], &kp = cKps[tid], &ln = cLen[tid];
      kc.reserve((size_t)(hi - lo) * 64); ln.reserve(hi - lo);
      for (int i = lo; i < hi && !bad.load(std::memory_order_relaxed); i++) {
        const size_t start = kc.size();
        auto add = [&](int j) { if (mark[j] != i) { mark[j] = i; pslot[j] = -1; kc.push_back(j); } };
        add(i);
        for (int k = M.rowptr[i]; k < M.split[i]; k++) {
          const int j = M.col[k];
          add(j);
          if (pslot[j] != -1) { bad = 1; break; }
          pslot[j] = k;
        }
        for (int k = M.split[i]; k < M.rowptr[i + 1]; k++) {
          const int row = M.col[k] - n;
          for (int q = A.rowptr[row]; q < A.rowptr[row + 1]; q++) add(A.col[q]);
        }
        if (kc.size() > cap_total) { bad = 2; break; }
        std::sort(kc.begin() + start, kc.end());
        kp.resize(kc.size());
        for (size_t q = start; q < kc.size(); q++) kp[q] = pslot[kc[q]];
        ln.push_back((int)(kc.size() - start));
      }
    });
    if (bad.load() == 1) RES_NO("P repeats an entry");
    size_t tot = 0;
    for (int k = 0; k < used; k++) tot += cKcol[k].size();
    if (bad.load() == 2 || tot > cap_total) RES_NO("K has more entries than the register files hold");
    Kcol.reserve(tot); Kps.reserve(tot);
    int row = 0;
    for (int k = 0; k < used; k++) {
      Kcol.insert(Kcol.end(), cKcol[k].begin(), cKcol[k].end());
      Kps.insert(Kps.end(), cKps[k].begin(), cKps[k].end());
      for (int len : cLen[k]) { Kptr[row + 1] = Kptr[row] + len; row++; }
    }
  }
  const long long nnzK = Kptr[n];
  const auto tb1 = std::chrono::steady_clock::now();
  // contiguous row blocks, at most RES_MAXROWS rows and C entries each: smallest C that needs <= nwg blocks
  auto packs = [&](long long C) -> int {
    int g = 0, rows = 0; long long cnt = 0;
    for (int i = 0; i < n; i++) {
      const int len = Kptr[i + 1] - Kptr[i];
      if (len > C) return 1 << 30;
      if (rows > 0 && (cnt + len > C || rows == RES_MAXROWS)) { g++; cnt = 0; rows = 0; }
      cnt += len; rows++;
    }
    return g + 1;
  };
  // The grid is sized to the problem: every exchange is an all-to-all between the participating CUs and costs 2.3 us among
  // 4, 3.6 among 16, 4.2 among 64, 6.5 among 256 (profiles/r03_exchange_probe2.txt), while a thread's E entries cost about
  // 0.06 us each per product.  So: the smallest grid that keeps E <= 16 (halving E from 16 no longer pays for the doubled
  // grid), at most `nwg_max` = the CUs of the device; beyond that E grows (20, 24, ... at config 2: 256 workgroups, E = 20).
  // OSQP_AMD_RESIDENT_NWG overrides (tests, probes).
  const int nwg_max = nwg;
  long long maxrow = 1;
  for (int i = 0; i < n; i++) maxrow = std::max<long long>(maxrow, Kptr[i + 1] - Kptr[i]);
  if (packs(std::max<long long>(nnzK, 1)) > nwg_max) RES_NO("more than 61 rows per workgroup");
  int E = 0;
  long long C = 0;
  int force_nwg = 0;
  if (const char *x = getenv("OSQP_AMD_RESIDENT_NWG")) force_nwg = std::max(0, std::min(nwg_max, atoi(x)));
  for (int cand : RES_E_LIST) {
    const long long cap = (long long)cand * RES_PT;
    if (cap < maxrow) continue;
    const int need = packs(cap);
    if (need > nwg_max) continue;
    if (!po && force_nwg > 0 && need > force_nwg) continue;
    E = cand; nwg = need;
    if (cand >= 16 || fixed_grid) break;
    // E = 8 fits: keep it only if E = 16 would not halve the grid (tiny problems: the rows, not the entries, set the grid)
    const int need16 = packs(16LL * RES_PT);
    if (need16 * 2 > need) break;
  }
  if (!E) RES_NO("a row block of K exceeds 448 x 64 entries");
  if (fixed_grid) nwg = nwg_max;               // (a plan for the CPU tests that keeps the grid it was asked for)
  else if (force_nwg > nwg) nwg = force_nwg;
  {
    // smallest block capacity that still needs no more than nwg blocks: balances the workgroups
    long long lo = maxrow, hi = (long long)E * RES_PT;
    while (lo < hi) { const long long mid = (lo + hi) / 2; if (packs(mid) <= nwg) hi = mid; else lo = mid + 1; }
    C = lo;
    if (fixed_grid) { E = 0; for (int cand : RES_E_LIST) if ((long long)cand * RES_PT >= C) { E = cand; break; } if (!E) RES_NO("a row block of K exceeds 448 x 64 entries"); }
  }
  std::vector<ResWG> wg(nwg, ResWG{n, 0, 0, 0});
  {
    int g = 0, rows = 0; long long cnt = 0; wg[0].r0 = 0;
    for (int i = 0; i < n; i++) {
      const int len = Kptr[i + 1] - Kptr[i];
      if (rows > 0 && (cnt + len > C || rows == RES_MAXROWS)) { wg[g].nr = rows; wg[g].cnt = (int)cnt; g++; wg[g].r0 = i; cnt = 0; rows = 0; }
      cnt += len; rows++;
    }
    wg[g].nr = rows; wg[g].cnt = (int)cnt;
  }
  // positions in the exchanged vector: a workgroup's rows, then its three dot partials, in lines of its own
  std::vector<unsigned short> rowpos(n, 0);
  int npad = 0;
  for (int g = 0; g < nwg; g++) {
    wg[g].pos = npad;
    for (int r = 0; r < wg[g].nr; r++) rowpos[wg[g].r0 + r] = (unsigned short)(npad + r);
    npad += ((wg[g].nr + 3 + 15) / 16) * 16;
  }
  if (npad > RES_MAXPAD) RES_NO("the exchanged vector (rows padded to whole lines) exceeds 16384 doubles");
  const size_t slots = (size_t)nwg * E * RES_PT;
  bool bank_sched = true;
  if (const char *x = getenv("OSQP_AMD_RESIDENT_BANKS")) bank_sched = atoi(x) != 0;
  std::vector<unsigned short> col(slots, 0), slot0((size_t)nwg * RES_PT, 0), segrow((size_t)nwg * (RES_MAXROWS + 1), 0);
  std::vector<unsigned char> rowl(slots, 0);
  std::vector<int> kdst((size_t)nnzK, 0);
  std::vector<unsigned long long> brk((size_t)nwg * RES_PT, 0ull);
  std::atomic<int> seg_over{0};
  parallel_chunks(nwg, 4, [&](int g_lo, int g_hi, int, int) {        // workgroups write disjoint parts of every array
  for (int g = g_lo; g < g_hi; g++) {
    const ResWG &w = wg[g];
    if (w.nr == 0) continue;
    const int base = Kptr[w.r0];
    std::vector<int> rowof(w.cnt);
    for (int r = 0; r < w.nr; r++)
      for (int q = Kptr[w.r0 + r] - base; q < Kptr[w.r0 + r + 1] - base; q++) rowof[q] = r;
    // Order of the entries inside each thread's chunk.  One wave instruction of the product loop gathers 64 doubles from
    // LDS at the columns of the lanes' k-th entries; random columns pile up to ~6 lanes on one of the 32 bank pairs.
    // A chunk that lies inside one row may be summed in any order, so its entries are dealt to the E slots such that
    // each slot spreads over the bank pairs (a greedy deal, deterministic; pairwise swaps on top of it cost 0.3 s of setup
    // at config 2 for nothing measurable).
    std::vector<int> ord((size_t)RES_PT * E, -1);
    for (int le = 0; le < w.cnt; le++) ord[le] = le;
    if (bank_sched)
    for (int wq = 0; wq < RES_PT / 64; wq++) {
      int cntb[64][32];                           // [slot][bank pair] over the 64 lanes of this wavefront
      for (int k = 0; k < E; k++) for (int b = 0; b < 32; b++) cntb[k][b] = 0;
      bool freec[64];
      for (int l = 0; l < 64; l++) {
        const int t = wq * 64 + l, a = t * E;
        freec[l] = a + E <= w.cnt && rowof[a] == rowof[a + E - 1];
        if (!freec[l]) for (int k = 0; k < E && a + k < w.cnt; k++) cntb[k][rowpos[Kcol[base + a + k]] & 31]++;
      }
      for (int k = 0; k < E; k++)                 // deal: slot by slot, every free lane gives the entry whose bank pair is emptiest
        for (int l = 0; l < 64; l++) {
          if (!freec[l]) continue;
          const int a = (wq * 64 + l) * E;
          int best = k, bc = 1 << 30;
          for (int q = k; q < E; q++) { const int c = cntb[k][rowpos[Kcol[base + ord[a + q]]] & 31]; if (c < bc) { bc = c; best = q; } }
          std::swap(ord[a + k], ord[a + best]);
          cntb[k][rowpos[Kcol[base + ord[a + k]]] & 31]++;
        }
    }
    int nseg = 0, next_row = 0;
    for (int t = 0; t < RES_PT; t++) {
      slot0[(size_t)g * RES_PT + t] = (unsigned short)nseg;
      unsigned long long b = 0;
      for (int k = 0; k < E; k++) {
        if (t * E + k >= w.cnt) break;
        const int le = ord[t * E + k];
        const size_t sl = ((size_t)g * E + k) * RES_PT + t;
        col[sl] = rowpos[Kcol[base + le]]; rowl[sl] = (unsigned char)rowof[le]; kdst[base + le] = (int)sl;
        while (next_row <= rowof[le]) segrow[(size_t)g * (RES_MAXROWS + 1) + next_row++] = (unsigned short)nseg;   // first segment of the row
        const bool last = k == E - 1 || t * E + k + 1 >= w.cnt || rowof[ord[t * E + k + 1]] != rowof[le];
        if (last) { b |= 1ull << k; nseg++; }
      }
      brk[(size_t)g * RES_PT + t] = b;
    }
    for (int r = next_row; r <= RES_MAXROWS; r++) segrow[(size_t)g * (RES_MAXROWS + 1) + r] = (unsigned short)nseg;
    if (nseg > RES_TB + RES_MAXROWS) seg_over = 1;   // cannot happen (one segment per thread plus one per row change)
  }
  });
  if (seg_over.load()) return 0;
  const auto tb2 = std::chrono::steady_clock::now();
  if (po) {                      // plan only (no device): hand the layout out
    po->ok = true; po->nwg = nwg; po->E = E; po->npad = npad; po->nnzK = nnzK;
    po->wg = wg; po->Kptr = Kptr; po->Kcol = Kcol; po->kdst = kdst; po->rowpos = rowpos; po->col = col; po->brk = brk;
    return 0;
  }
  ResCtx rc{};
  rc.nwg = nwg; rc.E = E; rc.npad = npad;
  rc.npw = (nwg + 63) / 64;
  rc.pipe = 1;
  if (const char *x = getenv("OSQP_AMD_RESIDENT_PIPE")) rc.pipe = atoi(x) != 0;
  rc.u0_direct = 1;
  rc.inject = 0;
  if (const char *x = getenv("OSQP_AMD_RESIDENT_INJECT")) rc.inject = atoi(x);
  if (const char *x = getenv("OSQP_AMD_RESIDENT_U0")) rc.u0_direct = atoi(x) != 0;
  unsigned short *d_rowpos = nullptr;
  ResWG *d_wg = nullptr; unsigned short *d_col = nullptr, *d_slot0 = nullptr, *d_segrow = nullptr; unsigned char *d_rowl = nullptr;
  int *d_krp = nullptr, *d_kcj = nullptr, *d_kps = nullptr, *d_kdst = nullptr; unsigned long long *d_brk = nullptr;
  if (dev_alloc(e, &d_wg, wg.size()) || dev_alloc(e, &rc.val, slots) || dev_alloc(e, &d_col, slots) || dev_alloc(e, &d_rowl, slots) ||
      dev_alloc(e, &d_krp, Kptr.size()) || dev_alloc(e, &d_kcj, Kcol.size()) || dev_alloc(e, &d_kps, Kps.size()) || dev_alloc(e, &d_kdst, kdst.size()) ||
      dev_alloc(e, &d_brk, brk.size()) || dev_alloc(e, &d_slot0, slot0.size()) ||
      dev_alloc(e, &d_segrow, segrow.size()) || dev_alloc(e, &d_rowpos, rowpos.size()) || dev_alloc(e, &rc.ubuf, (size_t)2 * rc.npad) ||
      dev_alloc_polled(e, &rc.flags, (size_t)nwg * RES_FSTRIDE) || dev_alloc(e, &rc.dbg, (size_t)nwg * RES_DBG) || dev_alloc_polled(e, &rc.sbuf, (size_t)2 * nwg * RES_GSTRIDE)) return HIPENG_ERR_HIP;
#define UP(dst, src) HIPCHK(hipMemcpyAsync(dst, (src).data(), (src).size() * sizeof((src)[0]), hipMemcpyHostToDevice, e->stream))
  UP(d_wg, wg); UP(d_rowpos, rowpos); UP(d_col, col); UP(d_rowl, rowl); UP(d_krp, Kptr); UP(d_kcj, Kcol); UP(d_kps, Kps); UP(d_kdst, kdst); UP(d_brk, brk); UP(d_slot0, slot0); UP(d_segrow, segrow);
#undef UP
  HIPCHK(hipStreamSynchronize(e->stream));       // the sources are locals
  {
    double *d_u0pos = nullptr;
    if (dev_alloc(e, &d_u0pos, (size_t)rc.npad)) return HIPENG_ERR_HIP;       // (zeroed: the padding positions are read, never written)
    e->c.u0map = d_rowpos; e->c.u0pos = d_u0pos;
  }
  rc.wg = d_wg; rc.rowpos = d_rowpos; rc.col = d_col; rc.rowl = d_rowl; rc.krp = d_krp; rc.kcj = d_kcj; rc.kps = d_kps; rc.kdst = d_kdst; rc.brk = d_brk; rc.slot0 = d_slot0; rc.segrow = d_segrow;
  e->rc = rc;
  e->res_lds = ((size_t)rc.npad + RES_TB + RES_MAXROWS + 3 + 16 + 768 + 5 * 64 + 96) * sizeof(double);   // + phase stamps of the TIMELINE build
  int rcode = 0;
  switch (E) {
    case 8: rcode = res_set_lds<8>(e->res_lds); break;   case 16: rcode = res_set_lds<16>(e->res_lds); break;
    case 20: rcode = res_set_lds<20>(e->res_lds); break; case 24: rcode = res_set_lds<24>(e->res_lds); break;
    case 32: rcode = res_set_lds<32>(e->res_lds); break; case 48: rcode = res_set_lds<48>(e->res_lds); break;
    default: rcode = res_set_lds<64>(e->res_lds); break;
  }
  if (rcode) RES_NO("the launch's LDS request was refused");
  e->res_nnz = nnzK;
  e->res_on = e->res_use = true;
  if (e->trace) {
    const auto tb3 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[osqp_amd] resident PCG: nnz(K)=%lld, %d workgroups x %d threads x %d entries, %zu B LDS; host: pattern %.0f ms, layout %.0f ms, upload %.0f ms\n",
            nnzK, nwg, RES_TB, E, e->res_lds, ms(tb0, tb1), ms(tb1, tb2), ms(tb2, tb3));
  }
  return 0;
}


// Block-direct form (k_blk_invert / k_blk_apply / k_blk_finish): same eligibility as the block-resident PCG.
static bool blocks_eligible(hipeng *e) {
  if (e->dP_blks.empty()) return false;
  int next = 0;
  for (const DenseBlk &d : e->dP_blks) { if (d.c0 != next || d.b > DENSE_MAX) return false; next += d.b; }
  if (next != e->n) return false;
  // rows of A: one entry each, except the folded huge rows (all of the huge ones must be folded)
  if ((int)e->A.blk.size() - e->A.nwave != (int)e->hrows.size()) return false;
  std::vector<char> ishuge(std::max(1, e->m), 0);
  for (int hr : e->hrows) ishuge[hr] = 1;
  for (int i = 0; i < e->m; i++) if (!ishuge[i] && e->A.rowptr[i + 1] - e->A.rowptr[i] != 1) return false;
  return true;
}
// Coupled form of the block-direct solve: the blocks tile 0..n and at most CPL_MAX rows of A have two or more entries (any
// length, any pattern).  `rows` receives them.
static bool blocks_coupled_eligible(hipeng *e, std::vector<int> &rows) {
  if (e->dP_blks.empty()) return false;
  int next = 0;
  for (const DenseBlk &d : e->dP_blks) { if (d.c0 != next || d.b > DENSE_MAX) return false; next += d.b; }
  if (next != e->n) return false;
  int cap = CPL_MAX;
  if (const char *x = getenv("OSQP_AMD_BLOCK_COUPLED_MAX")) cap = std::max(0, std::min(CPL_MAX, atoi(x)));
  rows.clear();
  for (int i = 0; i < e->m; i++) if (e->A.rowptr[i + 1] - e->A.rowptr[i] >= 2) { if ((int)rows.size() >= cap) return false; rows.push_back(i); }
  return !rows.empty();
}
static int build_blockdirect(hipeng *e) {
  int want = 1;
  if (const char *x = getenv("OSQP_AMD_RESIDENT")) want = atoi(x);
  if (const char *x = getenv("OSQP_AMD_RESIDENT_BLOCKS")) want = want && atoi(x);
  if (const char *x = getenv("OSQP_AMD_BLOCK_DIRECT")) want = want && atoi(x);
  if (!want) return 0;
  std::vector<int> crows;
  const bool plain = blocks_eligible(e);
  if (!plain && !blocks_coupled_eligible(e, crows)) return 0;
  BdCtx bd{};
  if (!plain) {
    int *d_crow = nullptr, *d_cidx = nullptr, *d_chuge = nullptr;
    const size_t kc = crows.size();
    std::vector<int> chuge(kc, -1);
    for (size_t r = 0; r < kc; r++) for (size_t h = 0; h < e->hrows.size(); h++) if (e->hrows[h] == crows[r]) chuge[r] = (int)h;
    std::vector<int> cidx((size_t)std::max(1, e->m), -1);
    for (size_t r = 0; r < kc; r++) cidx[crows[r]] = (int)r;
    if (dev_alloc(e, &d_crow, kc) || dev_alloc(e, &d_chuge, kc) || dev_alloc(e, &d_cidx, cidx.size()) || dev_alloc(e, &bd.cs, kc) || dev_alloc(e, &bd.cc, kc) ||
        dev_alloc(e, &bd.cap, kc * kc) || dev_alloc(e, &bd.cap2, kc * kc) || dev_alloc(e, &bd.cap0, kc * kc) || dev_alloc(e, &bd.wm, (size_t)16 * e->n)) return HIPENG_ERR_HIP;
    HIPCHK(hipMemcpyAsync(d_crow, crows.data(), kc * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_cidx, cidx.data(), cidx.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_chuge, chuge.data(), kc * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));          // (the sources are locals)
    bd.kc = (int)kc; bd.crow = d_crow; bd.cidx = d_cidx; bd.chuge = d_chuge;
    {
      std::vector<int> cbptr((size_t)e->n + 1, 0);
      std::vector<int2> cbent;
      for (int j = 0; j < e->n; j++) {
        for (int k = e->M.split[j]; k < e->M.rowptr[j + 1]; k++) { const int q = cidx[e->M.col[k] - e->n]; if (q >= 0) cbent.push_back(int2{q, k}); }
        cbptr[j + 1] = (int)cbent.size();
      }
      int *d_ptr = nullptr; int2 *d_ent = nullptr;
      if (dev_alloc(e, &d_ptr, cbptr.size()) || dev_alloc(e, &d_ent, std::max<size_t>(1, cbent.size()))) return HIPENG_ERR_HIP;
      HIPCHK(hipMemcpyAsync(d_ptr, cbptr.data(), cbptr.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
      if (!cbent.empty()) HIPCHK(hipMemcpyAsync(d_ent, cbent.data(), cbent.size() * sizeof(int2), hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      bd.cbptr = d_ptr; bd.cbent = d_ent;
    }
  }
  const size_t nd = e->dP_src.size(), n = (size_t)e->n, nb = e->dP_blks.size();
  if (dev_alloc(e, &bd.binv, nd) || dev_alloc(e, &bd.t, n) || dev_alloc(e, &bd.wh, (size_t)MAX_HUGE_FOLD * n) ||
      dev_alloc(e, &bd.cinv, (size_t)MAX_HUGE_FOLD * MAX_HUGE_FOLD) || dev_alloc(e, &bd.part, (size_t)MAX_HUGE_FOLD * nb) || dev_alloc(e, &bd.flag, 4) || dev_alloc(e, &bd.chk, (size_t)2)) return HIPENG_ERR_HIP;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_blk_invert), hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_MAX * DENSE_MAX * (int)sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    return 0;                        // not an error: the block-resident PCG takes over
  }
  {
    // A x~ of the folded huge rows from the kernel that writes x~ (instead of a k_huge_dot launch per ADMM iteration): every
    // huge row folded, and that kernel's grid within the gridA partial slots k_admm_finalize sums
    const int fin_grid = bd.kc ? std::max(1, std::min(1024, (int)nb)) : std::max(1, std::min(256, (e->n + TB - 1) / TB));
    bd.fin_dots = !e->hrows.empty() && (int)e->A.blk.size() - e->A.nwave == (int)e->hrows.size() && fin_grid <= e->c.gridA;
    if (const char *x = getenv("OSQP_AMD_BLOCK_FIN_DOTS")) bd.fin_dots = bd.fin_dots && atoi(x) != 0;
  }
  {
    // Opt-in (OSQP_AMD_BLOCK_PLAIN=1), not the default: exact algebra, 46 -> 37 us per ADMM iteration at config 5, and x~ as accurate
    // entry by entry -- but a' x~ = a' t0 - c a' W is then a difference of two numbers of size 1e4: 1e-12 of noise in the budget row's
    // constraint value, times rho_h in y_h (1e-10), which moves the dual residual norm by 1e-7 relative, the next rho estimate with
    // it, and the trajectory by 1e-7: config 5's golden objective is then off by 2.4e-6 (bar 1e-6).  tools/plain_check.py.
    int plain = 0;
    if (const char *x = getenv("OSQP_AMD_BLOCK_PLAIN")) plain = atoi(x) != 0;
    e->c.plain_rhs = plain;
    e->c.plain_skip = (plain && bd.kc) ? bd.cidx : nullptr;
  }
  e->bd = bd;
  // the solve kernels read the residual as a plain n-vector
  e->dd_init_r = e->c.init_r; e->dd_init_stride = e->c.init_stride;       // (what the launch-per-step kernels use: direct_disable puts them back)
  e->c.init_r = e->c.r; e->c.init_stride = 1;
  e->res_kind = 3; e->res_on = e->res_use = true;
  if (e->trace) fprintf(stderr, "[osqp_amd] block-direct solve: %zu dense blocks inverted explicitly, %d %s of A as a Woodbury term\n", nb,
                        bd.kc ? bd.kc : (int)e->hrows.size(), bd.kc ? "coupling rows" : "huge rows");
  return 0;
}
static int elem_grid(int count);
// A direct form whose fresh inverse cannot be trusted leaves: this engine goes on with the launch-per-step PCG kernels for good.
static void direct_disable(hipeng *e, const char *what, const char *why, double err) {
  if (e->trace) fprintf(stderr, "[osqp_amd] %s solve dropped (%s, check %.2e): the PCG kernels take over\n", what, why, err);
  for (auto &g : e->graphs) (void)hipGraphExecDestroy(g.second);      // (they hold the Ctx by value)
  e->graphs.clear();
  e->c.init_r = e->dd_init_r; e->c.init_stride = e->dd_init_stride; e->c.fin_wave_rows = 0; e->c.plain_rhs = 0; e->c.plain_skip = nullptr; e->c.vd = nullptr;
  const bool was_dense = e->res_kind == 4;
  e->res_kind = 0; e->res_on = e->res_use = false;
  e->calibrated = false; e->spec_lo = 0; e->start_dirty = true;
  if (was_dense) hipLaunchKernelGGL(k_precond, dim3(elem_grid(e->n)), dim3(TB), 0, e->stream, e->c);
}
// The verdict on the inverses blk_refresh has just formed (k_blk_check, k_cap_check): Gauss-Jordan has an error ~ cond^2 eps; fine on
// the blocks this form was built for (cond ~ 20), not on every block-diagonal P.  A pivot that was not positive stays what it was: the
// sign of an indefinite block (State::neg_curv).
#define BLK_CHECK 1e-6
static int blk_check(hipeng *e) {
  double chk[2] = {0.0, 0.0}; int flag = 0;
  HIPCHK(hipMemcpyAsync(chk, e->bd.chk, sizeof(chk), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(&flag, e->bd.flag, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->dd_check = std::max(chk[0], chk[1]);
  if (e->trace) fprintf(stderr, "[osqp_amd] block-direct refresh: inverse blocks off by %.2e, capacitance inverse by %.2e (probe vector of size 1..2)\n", chk[0], chk[1]);
  if (!flag && !(e->dd_check <= BLK_CHECK)) direct_disable(e, "block-direct", chk[0] > BLK_CHECK ? "the inverse blocks failed their check" : "the capacitance inverse failed its check", e->dd_check);
  return 0;
}
// New rho, sigma or matrix values: invert the blocks again, W = B^-1 A_h', capacitance matrix R_h^-1 + A_h W and its inverse.
static int blk_refresh(hipeng *e) {
  const Ctx &c = e->c;
  const int nh = c.nh, nb = c.dP.nblk, n = e->n;
  HIPCHK(hipMemsetAsync(e->bd.flag, 0, 4 * sizeof(int), e->stream));
  int bmax = 1;
  for (const DenseBlk &d : e->dP_blks) bmax = std::max(bmax, d.b);
  hipLaunchKernelGGL(k_blk_invert, dim3(std::min(nb, 1024)), dim3(INV_TB), (size_t)bmax * bmax * sizeof(double), e->stream, e->c, e->bd);
  HIPCHK(hipMemsetAsync(e->bd.chk, 0, 2 * sizeof(double), e->stream));
  hipLaunchKernelGGL(k_blk_check, dim3(std::min(nb, 1024)), dim3(TB), 0, e->stream, e->c, e->bd);
  if (e->bd.kc) {
    // capacitance matrix, rows r0 .. r0 + 15 = S (B^-1 S_g')': one block pass on the matrix cores and 16 kc row dots per group,
    // then the diagonal and kc pivot steps; everything on the stream
    const int kc = e->bd.kc;
    for (int r0 = 0; r0 < kc; r0 += 16) {
      hipLaunchKernelGGL(k_blk_apply_multi, dim3(std::min(nb, 1024)), dim3(TB), 0, e->stream, e->c, e->bd, r0, e->bd.wm);
      hipLaunchKernelGGL(k_cpl_dot, dim3(kc, 16), dim3(TB), 0, e->stream, e->c, e->bd, (const double *)e->bd.wm, e->bd.cap + (size_t)r0 * kc, 0, 0,
                         (long long)n, (long long)kc, std::min(16, kc - r0));
    }
    hipLaunchKernelGGL(k_cap_diag, dim3((kc + TB - 1) / TB), dim3(TB), 0, e->stream, e->c, e->bd);
    HIPCHK(hipMemcpyAsync(e->bd.cap0, e->bd.cap, (size_t)kc * kc * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    double *src = e->bd.cap, *dst = e->bd.cap2;
    const unsigned gs = (unsigned)(((long long)kc * kc + TB - 1) / TB);
    for (int p = 0; p < kc; p++) { hipLaunchKernelGGL(k_cap_step, dim3(gs), dim3(TB), 0, e->stream, e->bd, (const double *)src, dst, p); std::swap(src, dst); }
    if (src != e->bd.cap) HIPCHK(hipMemcpyAsync(e->bd.cap, src, (size_t)kc * kc * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    hipLaunchKernelGGL(k_cap_check, dim3(1), dim3(TB), 0, e->stream, e->bd);
    HIPCHK(hipGetLastError());
    return blk_check(e);
  }
  std::vector<double> C((size_t)MAX_HUGE_FOLD * MAX_HUGE_FOLD, 0.0), part((size_t)MAX_HUGE_FOLD * nb);
  for (int h = 0; h < nh; h++) {
    hipLaunchKernelGGL(k_blk_apply, dim3(std::min(nb, 1024)), dim3(TB), 0, e->stream, e->c, e->bd, (const double *)(c.hcol + (size_t)h * n), e->bd.wh + (size_t)h * n, 0);
    HIPCHK(hipMemcpyAsync(part.data(), e->bd.part, part.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int h2 = 0; h2 < nh; h2++) { double s = 0.0; for (int q = 0; q < nb; q++) s += part[(size_t)h2 * nb + q]; C[(size_t)h2 * MAX_HUGE_FOLD + h] = s; }   // (A_h2 W_h)
  }
  HIPCHK(hipGetLastError());
  bool bad = false;
  for (int h = 0; h < nh; h++) {
    const double rho = e->h_rho.size() > (size_t)c.hrow[h] ? e->h_rho[(size_t)c.hrow[h]] : 0.0;
    if (!(rho > 0.0)) bad = true; else C[(size_t)h * MAX_HUGE_FOLD + h] += 1.0 / rho;
  }
  // inverse of the nh x nh capacitance matrix (symmetric positive definite when K is): Gauss-Jordan on the host
  std::vector<double> I((size_t)MAX_HUGE_FOLD * MAX_HUGE_FOLD, 0.0);
  for (int h = 0; h < nh; h++) I[(size_t)h * MAX_HUGE_FOLD + h] = 1.0;
  for (int p = 0; p < nh && !bad; p++) {
    const double piv = C[(size_t)p * MAX_HUGE_FOLD + p];
    if (!(piv > 0.0)) { bad = true; break; }
    for (int j = 0; j < nh; j++) { C[(size_t)p * MAX_HUGE_FOLD + j] /= piv; I[(size_t)p * MAX_HUGE_FOLD + j] /= piv; }
    for (int i = 0; i < nh; i++) if (i != p) {
      const double f = C[(size_t)i * MAX_HUGE_FOLD + p];
      for (int j = 0; j < nh; j++) { C[(size_t)i * MAX_HUGE_FOLD + j] -= f * C[(size_t)p * MAX_HUGE_FOLD + j]; I[(size_t)i * MAX_HUGE_FOLD + j] -= f * I[(size_t)p * MAX_HUGE_FOLD + j]; }
    }
  }
  if (bad) { const int one = 1; for (double &v : I) v = 0.0; HIPCHK(hipMemcpyAsync(e->bd.flag, &one, sizeof(int), hipMemcpyHostToDevice, e->stream)); }
  HIPCHK(hipMemcpyAsync(e->bd.cinv, I.data(), I.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
  return blk_check(e);
}

// Block-resident form (k_pcg_blockres): P = dense diagonal blocks only, rows of A single-entry or folded huge rows.
static int build_blockres(hipeng *e) {
  const int n = e->n;
  int want = 1;
  if (const char *x = getenv("OSQP_AMD_RESIDENT")) want = atoi(x);
  if (const char *x = getenv("OSQP_AMD_RESIDENT_BLOCKS")) want = want && atoi(x);
  if (!want || e->dP_blks.empty()) return 0;
  // the blocks tile 0..n
  int next = 0;
  for (const DenseBlk &d : e->dP_blks) { if (d.c0 != next || d.b > DENSE_MAX) return 0; next += d.b; }
  if (next != n) return 0;
  // rows of A: one entry each, except the folded huge rows (all of the huge ones must be folded)
  if ((int)e->A.blk.size() - e->A.nwave != (int)e->hrows.size()) return 0;
  std::vector<char> ishuge(std::max(1, e->m), 0);
  for (int hr : e->hrows) ishuge[hr] = 1;
  for (int i = 0; i < e->m; i++) if (!ishuge[i] && e->A.rowptr[i + 1] - e->A.rowptr[i] != 1) return 0;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, e->device));
  const int nwg = std::min(256, prop.multiProcessorCount) - 8;       // one CU per XCD stays free (see build_resident)
  e->res_cus = nwg;
  const int nb = (int)e->dP_blks.size();
  if (nb > 2 * nwg) RES_NO("more than two dense blocks per CU");
  // contiguous blocks, at most two and at most 256 rows per workgroup, spread evenly
  std::vector<int> blk0(nwg + 1, nb);
  {
    int q = 0;
    for (int g = 0; g < nwg; g++) {
      blk0[g] = q;
      const int upto = (int)(((long long)nb * (g + 1) + nwg - 1) / nwg);
      int rows = 0, cnt = 0;
      while (q < nb && q < upto && cnt < 2 && rows + e->dP_blks[q].b <= 256) { rows += e->dP_blks[q].b; q++; cnt++; }
    }
    blk0[nwg] = q;
    if (q != nb) RES_NO("the dense blocks do not fit two per workgroup");
  }
  BrCtx bc{};
  bc.nwg = nwg;
  int *d_blk0 = nullptr;
  if (dev_alloc(e, &d_blk0, blk0.size()) || dev_alloc_polled(e, &bc.sbuf, (size_t)2 * nwg * RES_GSTRIDE) || dev_alloc(e, &bc.dbg, (size_t)nwg * RES_DBG)) return HIPENG_ERR_HIP;
  HIPCHK(hipMemcpyAsync(d_blk0, blk0.data(), blk0.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  bc.blk0 = d_blk0;
  e->bc = bc;
  e->res_kind = 2; e->res_on = e->res_use = true;
  if (e->trace) fprintf(stderr, "[osqp_amd] block-resident PCG: %d dense blocks over %d workgroups, %d huge rows of A as rank-one terms\n", nb, nwg, (int)e->hrows.size());
  return 0;
}

static int elem_grid(int cnt) {
  int g = (cnt + TB - 1) / TB;
  return std::max(1, std::min(g, MAX_PARTS));
}

// Everything derived from (P, A, rho, sigma): the Jacobi preconditioner.
static int dd_refresh(hipeng *e);
static void refresh_operator(hipeng *e) {
  e->elim_rhs_dirty = true;
  if (e->c.nelim) hipLaunchKernelGGL(k_elim_refresh, dim3(elem_grid(e->m)), dim3(TB), 0, e->stream, e->c);
  // (the Jacobi preconditioner: one thread per column -- 1.3 ms on the Lasso's 1 500-entry columns -- and of no use to a dense-direct
  // engine; direct_disable forms it when such an engine goes back to the PCG kernels)
  if (e->res_kind != 4) hipLaunchKernelGGL(k_precond, dim3(elem_grid(e->n)), dim3(TB), 0, e->stream, e->c);
  if (e->res_kind == 1) hipLaunchKernelGGL(k_form_K, dim3(std::min(2048, (e->n + 3) / 4)), dim3(TB), 0, e->stream, e->c, e->rc);
  if (e->res_kind == 3 && blk_refresh(e)) fprintf(stderr, "osqp_amd: the block-direct solve could not be refreshed\n");
  if (e->res_kind == 4 && dd_refresh(e)) fprintf(stderr, "osqp_amd: the dense-direct solve could not be refreshed\n");
}

// The PCG start vector history is void (cold/warm start from the host, new rho, new matrices):
// start from [x~ | rho z~] as it stands and extrapolate again once a few iterations are on record.
static int reset_start(hipeng *e) {
  if (e->c.vx != e->c.va)
    HIPCHK(hipMemcpyAsync(e->c.vx, e->c.va, (size_t)std::max(1, e->n + e->m) * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(hipMemsetAsync(&e->c.st->hist_r, 0, sizeof(int), e->stream));
  return 0;
}

static int upload_vec(hipeng *e, double *dst, const c_float *src, size_t cnt) {
  if (cnt == 0 || !src) return 0;
  HIPCHK(hipMemcpyAsync(dst, src, cnt * sizeof(double), hipMemcpyHostToDevice, e->stream));
  return 0;
}

static int push_params(hipeng *e) {
  HIPCHK(hipMemcpyAsync(e->d_prm, &e->prm, sizeof(Params), hipMemcpyHostToDevice, e->stream));
  // the source is host-pageable and may be reused right away
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// Which variables can be eliminated from the linear system (see k_elim_refresh): exactly one entry in their column of A,
// no off-diagonal entry in P, at most one such variable per row, the row not a huge one.  Launch-per-step engines and the
// resident PCG (k_form_K forms the reduced operator then); not the block forms.  OSQP_AMD_ELIM=0 switches it off.
static int build_elim(hipeng *e, const csc *P, const csc *A) {
  Ctx &c = e->c;
  c.nelim = 0; c.erow = c.ecol = c.epos = nullptr; c.rhoe = c.rho; c.ecoef = c.edinv = c.xte = nullptr;
  const int n = e->n, m = e->m;
  if (const char *x = getenv("OSQP_AMD_ELIM")) if (!atoi(x)) return 0;
  if (e->res_kind >= 2 || m == 0 || n == 0) return 0;      // (the block forms of the portfolio family hold their own operator)
  std::vector<char> coupled(n, 0);
  for (int j = 0; j < n; j++)
    for (long long k = P->p[j]; k < P->p[j + 1]; k++) if (P->i[k] != j) { coupled[j] = 1; coupled[(int)P->i[k]] = 1; }
  std::vector<int> erow(n, -1), ecol(m, -1), epos(m, 0);
  int cnt = 0;
  for (int j = 0; j < n; j++) {
    if (coupled[j] || A->p[j + 1] - A->p[j] != 1) continue;
    const long long k = A->p[j];
    const int i = (int)A->i[k];
    if (ecol[i] >= 0) continue;
    if (e->A.rowptr[i + 1] - e->A.rowptr[i] >= HUGE_ROW) continue;
    erow[j] = i; ecol[i] = j; epos[i] = e->A_csc2csr[k]; cnt++;
  }
  if (cnt == 0) return 0;
  int *d_erow = nullptr, *d_ecol = nullptr, *d_epos = nullptr;
  if (dev_alloc(e, &d_erow, (size_t)n) || dev_alloc(e, &d_ecol, (size_t)m) || dev_alloc(e, &d_epos, (size_t)m) ||
      dev_alloc(e, &c.rhoe, (size_t)m) || dev_alloc(e, &c.ecoef, (size_t)m) || dev_alloc(e, &c.edinv, (size_t)m) || dev_alloc(e, &c.xte, (size_t)m)) return HIPENG_ERR_HIP;
  HIPCHK(hipMemcpyAsync(d_erow, erow.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(d_ecol, ecol.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(d_epos, epos.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  c.erow = d_erow; c.ecol = d_ecol; c.epos = d_epos; c.nelim = cnt;
  e->erow = erow; e->ecol = ecol; e->epos = epos;
  if (e->trace) fprintf(stderr, "[osqp_amd] %d of %d variables eliminated from the linear system (one row of A each, no coupling in P)\n", cnt, n);
  return 0;
}

#include "dense_direct_host.h"

extern "C" int hipeng_create(hipeng **out, const csc *P, const csc *A, const c_float *q,
                             const c_float *l, const c_float *u, const c_float *rho_vec,
                             const hipeng_params *prm, int device) {
  if (!out || !P || !A || !prm) return HIPENG_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "osqp_amd: no HIP device available -- the HIP engine has no CPU fallback\n");
    return HIPENG_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return HIPENG_ERR_ARG;
  if (P->n > 0x7ffffff0LL || A->m > 0x7ffffff0LL || P->p[P->n] + A->p[A->n] + P->p[P->n] > 0x7ffffff0LL)
    return HIPENG_ERR_ARG;   // int32 device indices
  hipeng *e = new (std::nothrow) hipeng();
  if (!e) return HIPENG_ERR_ALLOC;
  e->device = device; e->n = (int)P->n; e->m = (int)A->m;
  const int n = e->n, m = e->m;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&e->ev0));
  HIPCHK(hipEventCreate(&e->ev1));
  build_A(e, A);
  build_M(e, P, A);
  build_dense(e, P);
  auto pick_rl = [](const HostMat &H) {
    int maxrows = 1;
    for (const RowBlk &b : H.blk) maxrows = std::max(maxrows, b.r1 - b.r0);
    int rl = 8;
    while (rl > 1 && TB / rl < maxrows) rl >>= 1;   // one pass over the block's rows if possible
    return rl;
  };
  e->rlA = pick_rl(e->A); e->rlM = pick_rl(e->M);
  {
    // when most of A sits in long rows, a 32-byte record gather per entry dominates k_cg_A:
    // do the vector update in its own (tiny) launch and gather plain 8-byte u instead
    long long lnnz = 0;
    for (size_t q = (size_t)e->A.nstream; q < (size_t)e->A.nwave; q++) lnnz += e->A.blk[q].k1 - e->A.blk[q].k0;
    e->split = 2 * lnnz >= (long long)e->A.val.size() && !e->A.val.empty();
  }
  if (const char *sp = getenv("OSQP_AMD_SPLIT")) e->split = atoi(sp) != 0;
  if (upload_mat(e, e->A) || upload_mat(e, e->M)) return HIPENG_ERR_HIP;
  Ctx &c = e->c;
  c.n = n; c.m = m;
  c.A = dev_view(e->A); c.M = dev_view(e->M);
  c.Mk = c.M; c.dP = DenseP{0, nullptr, nullptr}; c.nh = 0; c.hcol = nullptr;
  if (!e->dP_blks.empty() || !e->hrows.empty()) {
    if (upload_mat(e, e->Mr)) return HIPENG_ERR_HIP;
    if (dev_alloc(e, &e->d_Mr_src, e->Mr_src.size()) || dev_alloc(e, &e->d_dP_src, e->dP_src.size()) ||
        dev_alloc(e, &e->d_dP_blks, e->dP_blks.size()) || dev_alloc(e, &e->d_dP_val, e->dP_src.size()) ||
        dev_alloc(e, &e->d_hcol_src, e->hcol_src.size()) || dev_alloc(e, &e->d_hcol, e->hcol_src.size())) return HIPENG_ERR_HIP;
    if (!e->Mr_src.empty())
      HIPCHK(hipMemcpyAsync(e->d_Mr_src, e->Mr_src.data(), e->Mr_src.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    if (!e->dP_src.empty()) {
      HIPCHK(hipMemcpyAsync(e->d_dP_src, e->dP_src.data(), e->dP_src.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipMemcpyAsync(e->d_dP_blks, e->dP_blks.data(), e->dP_blks.size() * sizeof(DenseBlk), hipMemcpyHostToDevice, e->stream));
    }
    if (!e->hcol_src.empty())
      HIPCHK(hipMemcpyAsync(e->d_hcol_src, e->hcol_src.data(), e->hcol_src.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    c.Mk = dev_view(e->Mr);
    if (!e->dP_blks.empty()) c.dP = DenseP{(int)e->dP_blks.size(), e->d_dP_blks, e->d_dP_val};
    c.nh = (int)e->hrows.size(); c.hcol = e->d_hcol;
    for (int k = 0; k < MAX_HUGE_FOLD; k++) c.hrow[k] = k < c.nh ? e->hrows[k] : 0;
    if (repack_dense(e)) return HIPENG_ERR_HIP;
  }
  {
    if (const char *t = getenv("OSQP_AMD_TRACE")) e->trace = std::max(1, atoi(t));
  }
  const int cap = MAX_PARTS;
  c.gridM = std::min(cap, std::max(1, c.M.nblk));
  c.gridA = std::min(cap, std::max(std::max(1, c.A.nblk), std::min(elem_grid(n), 256)));
#define DA(field, cnt) if (dev_alloc(e, &c.field, (size_t)(cnt))) return HIPENG_ERR_HIP
  DA(xy, n + m); DA(z, m); DA(zt, m); DA(va, n + m); DA(vb, n + m);
  DA(q, n); DA(l, m); DA(u, m); DA(rho, m); DA(rhoinv, m); DA(minv, n); DA(pdiag, n);
  DA(r, n); DA(zz, n); DA(kp, n); DA(pt0, n + m); DA(pt1, n + m);
  DA(dxy, n + m); DA(dy, m); DA(cvec, n);
  DA(pdir, n); DA(ut, n + m); DA(g4, 2 * n);
  DA(vx, n + m); DA(vold, n + m);

  DA(D, n); DA(Dinv, n); DA(E, m); DA(Einv, m);
  const int np = std::max(std::max(c.gridM, c.gridA), 2048);   // also scratch for the scaling kernels
  DA(part_rz, np); DA(part_rr, np); DA(part_bb, np); DA(part_pkp, np);
  DA(part_s0, np); DA(part_s1, np); DA(part_s2, np); DA(part_gam, np); DA(part_del, np);
  DA(scal, SC_COUNT * 16);
  DA(part_h, (size_t)std::max(1, c.A.nblk - c.A.nwave) * c.gridA);
  DA(st, 1);
#ifdef OSQP_AMD_TIMELINE
  DA(tl, TL_CAP);
#endif
#undef DA
  {
    c.init_r = &c.g4[n].r; c.init_stride = 4; c.init_z = c.ut; c.fin_rr = c.part_rr; c.fin_cnt = c.gridM;
    HIPCHK(hipHostMalloc((void **)&e->h_state, sizeof(State), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&e->h_target, sizeof(long long), hipHostMallocDefault));
  }
  if (dev_alloc(e, &e->d_prm, 1)) return HIPENG_ERR_HIP;
  c.prm = e->d_prm;
  e->prm.sigma = prm->sigma; e->prm.alpha = prm->alpha;
  e->prm.eps_rel = prm->pcg_eps_rel; e->prm.eps_abs = prm->pcg_eps_abs;
  e->prm.pcg_max_iter = (int)prm->pcg_max_iter; e->prm.cinv = 1.0;
  e->prm.use_cvec = 0; e->prm.has_scaling = 0; e->prm.pad0 = 0;
  e->prm.ex_theta = 1.0; e->prm.ex_h0 = 5; e->prm.ex_h1 = 12; e->prm.no_restart = 0;
  if (const char *x = getenv("OSQP_AMD_EXTRAP_SCHED")) sscanf(x, "%d,%d", &e->prm.ex_h0, &e->prm.ex_h1);
  if (const char *x = getenv("OSQP_AMD_EXTRAP")) e->prm.ex_theta = atof(x);
  if (e->prm.ex_theta == 0.0) c.vx = c.va;   // plain warm start
  e->ex_theta0 = e->prm.ex_theta;
  if (prm->pcg_eps_rel < 0.5e-12) e->prm.ex_theta = 0.0;
  if (push_params(e)) return HIPENG_ERR_HIP;
  if (upload_vec(e, c.q, q, n) || upload_vec(e, c.l, l, m) || upload_vec(e, c.u, u, m) ||
      upload_vec(e, c.pdiag, e->pdiag.data(), n)) return HIPENG_ERR_HIP;
  e->stats.kernels_per_pcg_iter = 2;
  *out = e;
  if (int rc = build_blockdirect(e)) return rc;
  if (!e->res_on) if (int rc = build_blockres(e)) return rc;
  if (int rc = build_elim(e, P, A)) return rc;
  if (int rc = build_dense_direct(e, P, A, true)) return rc;          // small reduced systems: ahead of the resident PCG
  if (!e->res_on) { if (int rc = build_resident(e)) return rc; if (e->res_on) e->res_kind = 1; }
  if (int rc = build_dense_direct(e, P, A, false)) return rc;
  if (rho_vec) { int rc = hipeng_upload_rho(e, rho_vec); if (rc) return rc; }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" void hipeng_destroy(hipeng *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (auto &g : e->graphs) (void)hipGraphExecDestroy(g.second);
  for (void *p : e->allocs) (void)hipFree(p);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->h_state) (void)hipHostFree(e->h_state);
  if (e->h_target) (void)hipHostFree(e->h_target);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

extern "C" int hipeng_sync(hipeng *e) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int hipeng_set_params(hipeng *e, const hipeng_params *prm) {
  if (!e || !prm) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const bool sigma_changed = prm->sigma != e->prm.sigma;
  e->prm.sigma = prm->sigma; e->prm.alpha = prm->alpha;
  e->prm.eps_rel = prm->pcg_eps_rel; e->prm.eps_abs = prm->pcg_eps_abs;
  e->prm.pcg_max_iter = (int)prm->pcg_max_iter; e->prm.no_restart = prm->no_restart ? 1 : 0;
  // Extrapolated start vectors pay while PCG does real work.  When the caller asks for more
  // than the default accuracy (tight eps_abs/eps_rel tighten pcg_eps_rel, see osqp_solve) every
  // digit of the solve counts and the residual-based stop from a closer start leaves a
  // slightly larger error: plain warm start below 1e-12 (the stop equality-row problems get by default).
  e->prm.ex_theta = prm->pcg_eps_rel >= 0.5e-12 ? e->ex_theta0 : 0.0;
  if (push_params(e)) return HIPENG_ERR_HIP;
  if (sigma_changed) refresh_operator(e);
  return 0;
}

extern "C" int hipeng_set_scaling(hipeng *e, const c_float *D, const c_float *E, c_float cc) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  if (D && (e->m == 0 || E)) {
    std::vector<double> Di(e->n), Ei(e->m);
    for (int j = 0; j < e->n; j++) Di[j] = 1.0 / D[j];
    for (int i = 0; i < e->m; i++) Ei[i] = 1.0 / E[i];
    if (upload_vec(e, e->c.D, D, e->n) || upload_vec(e, e->c.E, E, e->m) ||
        upload_vec(e, e->c.Dinv, Di.data(), e->n) || upload_vec(e, e->c.Einv, Ei.data(), e->m))
      return HIPENG_ERR_HIP;
    e->prm.has_scaling = 1; e->prm.cinv = 1.0 / cc;
    if (push_params(e)) return HIPENG_ERR_HIP;   // also fences the temporaries above
  } else {
    e->prm.has_scaling = 0; e->prm.cinv = 1.0;
    if (push_params(e)) return HIPENG_ERR_HIP;
  }
  return 0;
}

// Ruiz equilibration of the resident (raw) problem, `passes` sweeps, entirely on the
// device; afterwards the scaled q, l, u, the matrix values (triu(P) and A in the caller's
// CSC order) and D, E, c are copied back for the host mirrors of the workspace.
extern "C" int hipeng_ruiz_scale(hipeng *e, c_int passes, c_float *D, c_float *E, c_float *cost,
                                 c_float *q, c_float *l, c_float *u, c_float *Px, c_float *Ax) {
  if (!e || passes < 0) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const Ctx &c = e->c;
  const int n = e->n, m = e->m;
  double *dn = c.kp, *en = c.dy;                 // scratch vectors of length n and m
  const int gw = std::max(1, std::min(2048, (n + m + 3) / 4));
  const int gn = std::max(1, std::min(1024, (n + 3) / 4));
  HIPCHK(hipMemsetAsync(c.scal, 0, SC_COUNT * 16 * sizeof(double), e->stream));
  hipLaunchKernelGGL(k_fill, dim3(elem_grid(n)), dim3(TB), 0, e->stream, c.D, 1.0, n);
  hipLaunchKernelGGL(k_fill, dim3(elem_grid(std::max(m, 1))), dim3(TB), 0, e->stream, c.E, 1.0, m);
  hipLaunchKernelGGL(k_fill, dim3(1), dim3(TB), 0, e->stream, c.scal + SCI(SC_DYLHS), 1.0, 1);
  for (c_int p = 0; p < passes; p++) {
    hipLaunchKernelGGL(k_ruiz_norms, dim3(gw), dim3(TB), 0, e->stream, c, dn, en);
    hipLaunchKernelGGL(k_ruiz_apply, dim3(gw), dim3(TB), 0, e->stream, c, dn, en, e->M.d_val, e->A.d_val);
    hipLaunchKernelGGL(k_ruiz_cost_norms, dim3(gn), dim3(TB), 0, e->stream, c);
    hipLaunchKernelGGL(k_ruiz_cost_scalar, dim3(1), dim3(TB), 0, e->stream, c, gn);
    hipLaunchKernelGGL(k_ruiz_cost_apply, dim3(gn), dim3(TB), 0, e->stream, c, e->M.d_val);
  }
  hipLaunchKernelGGL(k_ruiz_finish, dim3(elem_grid(std::max(n, m))), dim3(TB), 0, e->stream, c);
  HIPCHK(hipGetLastError());
  if (repack_dense(e)) return HIPENG_ERR_HIP;
  // host mirrors
  double cc = 1.0;
  HIPCHK(hipMemcpyAsync(&cc, c.scal + SCI(SC_DYLHS), sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->M.val.data(), e->M.d_val, e->M.val.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (D) HIPCHK(hipMemcpyAsync(D, c.D, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (E && m) HIPCHK(hipMemcpyAsync(E, c.E, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (q) HIPCHK(hipMemcpyAsync(q, c.q, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (l && m) HIPCHK(hipMemcpyAsync(l, c.l, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (u && m) HIPCHK(hipMemcpyAsync(u, c.u, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (cost) *cost = cc;
  if (Px) for (size_t k = 0; k < e->P_toM_up.size(); k++) Px[k] = e->M.val[e->P_toM_up[k]];
  if (Ax) for (size_t k = 0; k < e->A_toM.size(); k++) Ax[k] = e->M.val[e->A_toM[k]];
  for (size_t k = 0; k < e->A_toM.size(); k++) e->A.val[e->A_csc2csr[k]] = e->M.val[e->A_toM[k]];
  for (int j = 0; j < n; j++) e->pdiag[j] = 0.0;
  for (size_t k = 0; k < e->P_toM_up.size(); k++) if (e->P_toM_lo[k] < 0) e->pdiag[e->M.col[e->P_toM_up[k]]] = e->M.val[e->P_toM_up[k]];
  if (upload_vec(e, c.pdiag, e->pdiag.data(), n)) return HIPENG_ERR_HIP;
  e->prm.has_scaling = passes > 0 ? 1 : 0; e->prm.cinv = 1.0 / cc;
  if (push_params(e)) return HIPENG_ERR_HIP;
  e->elim_rhs_dirty = true;
  return 0;
}

extern "C" int hipeng_upload_q(hipeng *e, const c_float *q) {
  if (!e || !q) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  if (upload_vec(e, e->c.q, q, e->n)) return HIPENG_ERR_HIP;
  HIPCHK(hipStreamSynchronize(e->stream));
  e->elim_rhs_dirty = true;
  return 0;
}

extern "C" int hipeng_upload_bounds(hipeng *e, const c_float *l, const c_float *u) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  if (upload_vec(e, e->c.l, l, e->m) || upload_vec(e, e->c.u, u, e->m)) return HIPENG_ERR_HIP;
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int hipeng_upload_rho(hipeng *e, const c_float *rho_vec) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  if (e->m > 0) {
    if (!rho_vec) return HIPENG_ERR_ARG;
    if (upload_vec(e, e->c.rho, rho_vec, e->m)) return HIPENG_ERR_HIP;
    e->h_rho.assign(rho_vec, rho_vec + e->m);
    if (e->c.nelim) hipLaunchKernelGGL(k_elim_refresh, dim3(elem_grid(e->m)), dim3(TB), 0, e->stream, e->c);
    hipLaunchKernelGGL(k_refresh_m, dim3(elem_grid(e->m)), dim3(TB), 0, e->stream, e->c, 0); e->start_dirty = true;
  }
  refresh_operator(e);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  e->calibrated = false; e->spec_lo = 0;   // the PCG iteration count usually jumps after a rho change
  return 0;
}

extern "C" int hipeng_upload_matrices(hipeng *e, const csc *P, const csc *A) {
  if (!e || !P || !A) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const int n = e->n;
  if (P->p[n] != (long long)e->P_toM_up.size() || A->p[n] != (long long)e->A_toM.size())
    return HIPENG_ERR_ARG;
  for (long long k = 0; k < P->p[n]; k++) {
    e->M.val[e->P_toM_up[k]] = P->x[k];
    if (e->P_toM_lo[k] >= 0) e->M.val[e->P_toM_lo[k]] = P->x[k];
  }
  for (int j = 0; j < n; j++) {
    e->pdiag[j] = 0.0;
    for (long long k = P->p[j]; k < P->p[j + 1]; k++) if (P->i[k] == j) e->pdiag[j] = P->x[k];
  }
  for (long long k = 0; k < A->p[n]; k++) {
    e->M.val[e->A_toM[k]] = A->x[k];
    e->A.val[e->A_csc2csr[k]] = A->x[k];
  }
  if (upload_vec(e, e->M.d_val, e->M.val.data(), e->M.val.size()) ||
      upload_vec(e, e->A.d_val, e->A.val.data(), e->A.val.size()) ||
      upload_vec(e, e->c.pdiag, e->pdiag.data(), n)) return HIPENG_ERR_HIP;
  if (repack_dense(e)) return HIPENG_ERR_HIP;
  refresh_operator(e);
  // z~ = A x~ for the new A so that the PCG warm start stays consistent
  if (e->m > 0) {
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), dim3(TB), 0, e->stream, e->c.A, e->c.va, e->c.zt, 0);
    hipLaunchKernelGGL(k_refresh_m, dim3(elem_grid(e->m)), dim3(TB), 0, e->stream, e->c, 1); e->start_dirty = true;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// The resident matrix values changed in place (device-side rescaling): rebuild what depends
// on them -- Jacobi preconditioner, z~ = A x~ and the m-parts of the PCG input vectors.
extern "C" int hipeng_matrices_changed(hipeng *e) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  if (repack_dense(e)) return HIPENG_ERR_HIP;
  refresh_operator(e);
  if (e->m > 0) {
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), dim3(TB), 0, e->stream, e->c.A, e->c.va, e->c.zt, 0);
    hipLaunchKernelGGL(k_refresh_m, dim3(elem_grid(e->m)), dim3(TB), 0, e->stream, e->c, 1); e->start_dirty = true;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int hipeng_cold_start(hipeng *e) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const size_t nm = (size_t)(e->n + e->m) * sizeof(double);
  HIPCHK(hipMemsetAsync(e->c.xy, 0, std::max<size_t>(nm, 8), e->stream));
  HIPCHK(hipMemsetAsync(e->c.va, 0, std::max<size_t>(nm, 8), e->stream));
  HIPCHK(hipMemsetAsync(e->c.vb, 0, std::max<size_t>(nm, 8), e->stream));
  HIPCHK(hipMemsetAsync(e->c.z, 0, std::max<size_t>(e->m, 1) * sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(e->c.zt, 0, std::max<size_t>(e->m, 1) * sizeof(double), e->stream));
  if (e->c.nelim) HIPCHK(hipMemsetAsync(e->c.xte, 0, (size_t)e->m * sizeof(double), e->stream));
  e->calibrated = false; e->spec_lo = 0;   // the first solve from zero needs far more PCG iterations than the steady state
  e->start_dirty = true;
  e->elim_rhs_dirty = true;                // (vb = rho z - y = 0 is NOT the whole m-part when a row carries an eliminated variable: -rho a (-q_y) / D_y)
  return 0;
}

extern "C" int hipeng_set_iterates(hipeng *e, const c_float *x, const c_float *y) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const int n = e->n, m = e->m;
  if (y && upload_vec(e, e->c.xy + n, y, m)) return HIPENG_ERR_HIP;
  if (x) {
    if (upload_vec(e, e->c.xy, x, n)) return HIPENG_ERR_HIP;
    // PCG warm start x~ = x ; z = A x ; z~ = z   (osqp_warm_start, osqp.c:960-963)
    HIPCHK(hipMemcpyAsync(e->c.va, e->c.xy, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    if (e->c.nelim) hipLaunchKernelGGL(k_elim_split, dim3(elem_grid(n)), dim3(TB), 0, e->stream, e->c);
    e->start_dirty = true;
    if (m > 0) {
      hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), dim3(TB), 0, e->stream, e->c.A, e->c.xy, e->c.z, 0);
      HIPCHK(hipMemcpyAsync(e->c.zt, e->c.z, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    }
  }
  if (m > 0) hipLaunchKernelGGL(k_refresh_m, dim3(elem_grid(m)), dim3(TB), 0, e->stream, e->c, 0);
  e->start_dirty = true;
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int hipeng_set_z(hipeng *e, const c_float *z) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const int m = e->m;
  if (m > 0 && z) {
    if (upload_vec(e, e->c.z, z, m)) return HIPENG_ERR_HIP;
    HIPCHK(hipMemcpyAsync(e->c.zt, e->c.z, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    hipLaunchKernelGGL(k_refresh_m, dim3(elem_grid(m)), dim3(TB), 0, e->stream, e->c, 0); e->start_dirty = true;
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// ---- graphs ---------------------------------------------------------------
#define TR2(e, ...) do { if ((e)->trace >= 2) { fprintf(stderr, "[osqp_amd:t2] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
static void launch_init(hipeng *e, int bench = 0) {
  if (e->c.vd) hipLaunchKernelGGL(k_vd, dim3(elem_grid(e->n + e->m)), dim3(TB), 0, e->stream, e->c);
  if (e->c.dP.nblk) hipLaunchKernelGGL(k_pcg_init<true>, dim3(e->c.gridM), dim3(TB), 0, e->stream, e->c, bench);
  else hipLaunchKernelGGL(k_pcg_init<false>, dim3(e->c.gridM), dim3(TB), 0, e->stream, e->c, bench);
}

static void launch_cg_A(hipeng *e, int it, int flags) {
  hipLaunchKernelGGL(k_cg_A, dim3(e->c.gridA), dim3(TB), 0, e->stream, e->c, it, flags);
  const int nhuge = e->c.nh ? 0 : e->c.A.nblk - e->c.A.nwave;   // folded into k_cg_B when c.nh > 0
  if (nhuge > 0 && !(flags & 16)) hipLaunchKernelGGL(k_huge_reduce, dim3(nhuge), dim3(TB), 0, e->stream, e->c, flags & 4);
}
static void launch_cg_B(hipeng *e, int it, int flags) {
  const bool d = e->c.dP.nblk > 0, f = e->c.nh > 0;
  if (d && f) hipLaunchKernelGGL((k_cg_B<true, true>), dim3(e->c.gridM), dim3(TB), 0, e->stream, e->c, it, flags);
  else if (d) hipLaunchKernelGGL((k_cg_B<true, false>), dim3(e->c.gridM), dim3(TB), 0, e->stream, e->c, it, flags);
  else if (f) hipLaunchKernelGGL((k_cg_B<false, true>), dim3(e->c.gridM), dim3(TB), 0, e->stream, e->c, it, flags);
  else hipLaunchKernelGGL((k_cg_B<false, false>), dim3(e->c.gridM), dim3(TB), 0, e->stream, e->c, it, flags);
}

static void launch_pcg_iter(hipeng *e, int it, int flags, int spec_lo = 0) {
  const int hi = spec_lo << 8;     // slots below spec_lo prefetch without looking at the flags first (k_cg_A)
  if (e->split) { launch_cg_A(e, it, (flags & 4) | 16 | hi); launch_cg_A(e, it, (flags & 4) | 32 | hi); }
  else launch_cg_A(e, it, (flags & 4) | hi);
  launch_cg_B(e, it, (flags & 4) | hi);
}

// One ADMM iteration as a graph: k_pcg_init, the operator on u0 ("pre" pass), K unrolled PCG
// iterations, k_admm_finalize.  If K was too small, k_admm_finalize raises `stalled` and the NEXT
// launch of the same graph continues the solve (its init and pre pass return at once).
// R such segments back to back form one graph (a graph launch has a fixed cost of its own on the GPU
// timeline); a segment whose predecessor stalled continues it, segments beyond admm_target return at once.
static int get_graph(hipeng *e, int K, int R, int spec_lo, hipGraphExec_t *out) {
  const int key = (K * 64 + R) * 1024 + spec_lo;
  auto f = e->graphs.find(key);
  if (f != e->graphs.end()) { *out = f->second; return 0; }
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  TR2(e, "get_graph K=%d R=%d: begin capture", K, R);
  HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
  for (int seg = 0; seg < R; seg++) {
    launch_init(e);
    if (K == 0) launch_resident(e);                   // the whole linear solve in one launch (K in registers)
    else { launch_cg_A(e, -1, 8); launch_cg_B(e, -1, 0); }   // operator apply on u0 (+ first convergence test), then w0 and the first dots
    for (int it = 0; it < K; it++) launch_pcg_iter(e, it, 0, spec_lo);
    if (e->c.A.nblk > e->c.A.nwave && !(K == 0 && e->res_kind == 3 && e->bd.fin_dots))      // (block-direct: the kernel that wrote x~ left the partials)
      hipLaunchKernelGGL(k_huge_dot, dim3(e->c.gridA), dim3(TB), 0, e->stream, e->c, (const double *)nullptr, 0);
    {
      // dense-direct engines (one wavefront per long row in this kernel): enough workgroups that a wavefront has one or two rows --
      // a row is a chain of dependent round trips (the dot, then the row's own scalars), and at gridA <= 1024 each wavefront walked three
      int fg = e->c.gridA;
      if (e->c.fin_wave_rows) fg = std::max(fg, std::min(4096, (e->c.A.nwave - e->c.A.nstream + 3) / 4 + 1));
      if (const char *x = getenv("OSQP_AMD_FIN_GRID")) fg = std::max(1, atoi(x));
      hipLaunchKernelGGL(k_admm_finalize, dim3(fg), dim3(TB), 0, e->stream, e->c);
    }
  }
  HIPCHK(hipStreamEndCapture(e->stream, &g));
  TR2(e, "get_graph: captured, instantiate");
  HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  TR2(e, "get_graph: instantiated");
  HIPCHK(hipGraphDestroy(g));
  e->graphs[key] = ge;
  *out = ge;
  return 0;
}

static int read_state(hipeng *e, State *s) {
  HIPCHK(hipMemcpyAsync(e->h_state, e->c.st, sizeof(State), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  *s = *e->h_state;
  return 0;
}

static int next_K(int want, int cap) {
  // even unroll counts: step 2 up to 32, step 4 up to 96, then +25 %
  int K = 2;
  while (K < want) K = (K < 32) ? K + 2 : (K < 96 ? K + 4 : ((K + K / 4 + 1) & ~1));
  if (K > cap) K = std::max(2, (cap + 1) & ~1);
  return K;
}

// A resident launch gave up (State::res_fail): say what the workgroups left behind, clean up, count the strike.
// The record is meant to be decisive: for the workgroup whose wait timed out first it prints the word it last read from the
// workgroup it missed NEXT TO the word that is in memory now (read by the host), where the missed workgroup itself stood,
// when it published (its give-up clock minus the ticks it had waited) and both XCC ids -- "store landed but was never seen",
// "store never landed" and "the other one was late" read differently.
static int resident_gave_up(hipeng *e, const State &s, bool quiet) {
  const int nwg = e->res_kind == 2 ? e->bc.nwg : e->rc.nwg;
  int *ddbg = e->res_kind == 2 ? e->bc.dbg : e->rc.dbg;
  std::vector<int> d((size_t)nwg * RES_DBG, 0);
  bool have = ddbg && hipMemcpy(d.data(), ddbg, d.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  const bool injected = have && e->res_kind == 1 && e->rc.inject > 0 && d[(size_t)RES_DBG * (nwg - 1) + 3] == 4;
  if (!quiet || !injected) {
    fprintf(stderr, "osqp_amd: a resident PCG launch gave up waiting for its workgroups%s; %s with the launch-per-step kernels "
                    "[wait %d (1 flags, 2 granules) of exchange %d: workgroup %d missed workgroup %d]\n",
            injected ? " (injected by OSQP_AMD_RESIDENT_INJECT)" : " (GPU shared with another stream or process?)",
            e->res_fails >= 2 ? "continuing for good" : "finishing this call", s.res_dbg[0], s.res_dbg[1], s.res_dbg[2], s.res_dbg[3]);
    if (have) {
      std::map<long long, int> hist;
      for (int g = 0; g < nwg; g++) hist[((long long)d[(size_t)RES_DBG * g] << 16) | ((d[(size_t)RES_DBG * g + 1] & 255) << 8) | (d[(size_t)RES_DBG * g + 3] & 255)]++;
      fprintf(stderr, "osqp_amd:   workgroups by (exchange, mode, how they left: 1/2 own flag/granule wait timed out, 3 saw the give-up word, 4 injected, 0 never waited):");
      for (auto &h : hist) fprintf(stderr, " (%lld, %lld, %lld) x %d;", h.first >> 16, (h.first >> 8) & 255, h.first & 255, h.second);
      fprintf(stderr, " epoch base %u\n", s.res_epoch);
      const int wg = s.res_dbg[2], ow = s.res_dbg[3];
      if (wg >= 0 && wg < nwg) {
        const int *w = &d[(size_t)RES_DBG * wg];
        fprintf(stderr, "osqp_amd:   waiter %d (XCC %d): waited %.1f us, %d rounds, last read 0x%08x from workgroup %d (expected tag 0x%08x), gave up at clock 0x%08x\n",
                wg, w[7], w[6] * 0.01, w[9], (unsigned)w[5], w[4], s.res_epoch + (unsigned)w[0], (unsigned)w[8]);
        if (ow >= 0 && ow < nwg) {
          const int *o = &d[(size_t)RES_DBG * ow];
          unsigned long long memword = 0; bool got = false;
          if (s.res_dbg[0] == 1 && e->res_kind == 1) { unsigned f = 0; got = hipMemcpy(&f, e->rc.flags + (size_t)ow * RES_FSTRIDE, 4, hipMemcpyDeviceToHost) == hipSuccess; memword = f; }
          else {
            const double *sb = e->res_kind == 2 ? e->bc.sbuf : e->rc.sbuf;
            const unsigned tag = s.res_epoch + (unsigned)w[0];
            int gi = 0; while (gi < 15 && !((w[10] >> gi) & 1)) gi++;
            got = hipMemcpy(&memword, reinterpret_cast<const unsigned long long *>(sb) + ((size_t)(tag & 1u) * nwg + ow) * RES_GSTRIDE + gi, 8, hipMemcpyDeviceToHost) == hipSuccess;
            memword >>= 32;
          }
          fprintf(stderr, "osqp_amd:   owner %d (XCC %d): stood in exchange %d (left: %d), published at clock 0x%08x (its give-up 0x%08x minus %.1f us of its own wait); "
                          "the word in memory now: %s0x%08llx\n", ow, o[7], o[0], o[3], (unsigned)(o[8] - o[6]), (unsigned)o[8], o[6] * 0.01, got ? "" : "(unreadable) ", memword);
        }
      }
    }
  }
  e->res_use = false;
  e->res_fails += 1;
  e->res_gave_up += 1;
  HIPCHK(hipMemsetAsync(&e->c.st->res_fail, 0, sizeof(int), e->stream));
  // the launch that gave up left flags and granules with tags beyond the epoch it started from (it never wrote the
  // epoch back): the next resident launch must not take them for its own -- start it far beyond (tags compare modulo 2^32)
  e->h_state->res_epoch = s.res_epoch + (1u << 20);
  HIPCHK(hipMemcpyAsync(&e->c.st->res_epoch, &e->h_state->res_epoch, sizeof(unsigned), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemsetAsync(e->c.st->res_dbg, 0, sizeof(int) * 4, e->stream));
  if (ddbg) HIPCHK(hipMemsetAsync(ddbg, 0, d.size() * sizeof(int), e->stream));
  return 0;
}

extern "C" int hipeng_run_admm(hipeng *e, c_int count) {
  if (!e || count < 0) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  State s;
  if (e->trace >= 2) {
    if (const char *mp = getenv("OSQP_AMD_TRACE_MAPS")) {     // address map, to read a crash stack
      static bool dumped = false;
      if (!dumped) {
        dumped = true;
        if (FILE *in = fopen("/proc/self/maps", "r")) {
          if (FILE *out = fopen(mp, "w")) { char buf[4096]; size_t k; while ((k = fread(buf, 1, sizeof buf, in)) > 0) fwrite(buf, 1, k, out); fclose(out); }
          fclose(in);
        }
      }
    }
  }
  // admm_done only moves inside this function: the host keeps its own copy, so a call starts
  // without a device round trip.  The device stops starting iterations at admm_target.
  if (count == 0) return 0;
  if (e->elim_rhs_dirty) {
    if (e->c.nelim && e->m > 0) hipLaunchKernelGGL(k_refresh_m, dim3(elem_grid(e->m)), dim3(TB), 0, e->stream, e->c, 0);
    e->elim_rhs_dirty = false;
  }
  // A launch that gave up (a wait timed out) took the rest of that call to the launch-per-step kernels.  Apart from a GPU
  // that is shared for good, there is a rare transient (about one exchange in 1e5 on small, fast problems: a workgroup's
  // flag or granule stays invisible to some CUs): the next call tries resident launches again, three strikes end that.
  if (e->res_on && !e->res_use && e->res_fails < 3) e->res_use = true;
  bool had_fail = false;
  const long long start = e->admm_total, target = start + (long long)count;
  *e->h_target = target;
  HIPCHK(hipMemcpyAsync(&e->c.st->admm_target, e->h_target, sizeof(long long), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemsetAsync(&e->c.st->iters_max, 0, sizeof(int), e->stream));
  HIPCHK(hipMemsetAsync(&e->c.st->iters_min, 0x7f, sizeof(int), e->stream));
  if (e->start_dirty) { if (reset_start(e)) return HIPENG_ERR_HIP; e->start_dirty = false; }
  long long remaining = count;
  const int cap = std::max(2, e->prm.pcg_max_iter);
  static const int khead = getenv("OSQP_AMD_KHEAD") ? atoi(getenv("OSQP_AMD_KHEAD")) : 1;   // unrolled iterations beyond the last window's maximum
  int guard = 0, call_max = 0;
  s.admm_done = start; s.iters_last = 0; s.iters_total = 0; s.forced = 0; s.neg_curv_seen = 0;
  static const int segs = getenv("OSQP_AMD_GRAPH_SEGS") ? std::max(1, std::min(32, atoi(getenv("OSQP_AMD_GRAPH_SEGS")))) : 5;
  while (remaining > 0) {
    // after a reset (cold start, new rho, new matrices) the PCG iteration count jumps: two
    // iterations to see where it lands, then whole windows
    const bool resident = e->res_use;
    // A resident launch needs every CU for itself (one workgroup per CU, all co-resident): resident windows of
    // different engines of this process on one device take turns.  (Across processes there is no such lock: a
    // launch that finds CUs taken times out and the engine falls back, see k_pcg_resident.)
    CuLease lease;
    if (resident && e->res_kind < 3) lease.take(e->device, e->res_cus, e->res_kind == 2 ? e->bc.nwg : e->rc.nwg);
    const long long burst = std::min<long long>(remaining, (e->calibrated || resident) ? 128 : 2);
    TR2(e, "launch burst=%lld K=%d", burst, e->K);
    // `burst` ADMM iterations = burst / segs graphs of `segs` segments + single-segment graphs for the rest
    // (graphs with many nodes only when the unroll is short: keeps instantiation cheap for long PCG runs)
    const int R = (resident || e->K <= 48) ? segs : 1;
    const int gK = resident ? 0 : e->K, gS = resident ? 0 : e->spec_lo;
    hipGraphExec_t gR = nullptr, g1 = nullptr;
    if (burst >= R && R > 1 && get_graph(e, gK, R, gS, &gR)) return HIPENG_ERR_HIP;
    if ((!gR || burst % R) && get_graph(e, gK, 1, gS, &g1)) return HIPENG_ERR_HIP;
    long long left = burst;
    for (; gR && left >= R; left -= R) { HIPCHK(hipGraphLaunch(gR, e->stream)); e->stats.graph_launches += 1; }
    for (; left > 0; left--) { HIPCHK(hipGraphLaunch(g1, e->stream)); e->stats.graph_launches += 1; }
    if (read_state(e, &s)) return HIPENG_ERR_HIP;
    lease.give();
    e->res_slow += s.res_slow; e->res_slow_max = std::max<long long>(e->res_slow_max, s.res_slow_max); e->res_repub += s.res_repub;
    if (s.res_slow || s.res_repub) {
      for (int k = 0; k < 4; k++) { e->res_slow_hist[k] += s.res_slow_hist[k]; e->res_slow_last[k] = s.res_slow_last[k]; }
      e->res_slow_xcc |= s.res_slow_xcc;
      if (e->trace) fprintf(stderr, "[osqp_amd] resident waits of this window: %d took more than 50 us (50-200 us: %d, 200-500: %d, 0.5-2 ms: %d, longer: %d; longest %.1f us; XCC mask 0x%02x; "
                                    "latest: exchange %d, workgroup %d missed workgroup %d, wait %d), %d re-publications\n", s.res_slow, s.res_slow_hist[0], s.res_slow_hist[1], s.res_slow_hist[2],
                            s.res_slow_hist[3], s.res_slow_max * 0.01, s.res_slow_xcc, s.res_slow_last[0], s.res_slow_last[1], s.res_slow_last[2], s.res_slow_last[3], s.res_repub);
      HIPCHK(hipMemsetAsync(&e->c.st->res_slow, 0, 12 * sizeof(int), e->stream));
    }
    if (s.res_fail) { had_fail = true; if (resident_gave_up(e, s, !e->trace)) return HIPENG_ERR_HIP; }
    const long long done_now = s.admm_done - start;
    const bool stalls = count - done_now > remaining - burst;       // some launches were continuations
    remaining = count - done_now;
    call_max = std::max(call_max, s.iters_max);
    const int in_flight = s.stalled ? std::max(s.iters[0], s.iters[1]) : 0;
    if (e->trace) fprintf(stderr, "[osqp_amd] window: K=%d burst=%lld done=%lld iters_max=%d last=%d stalled=%d(%d) launches=%lld\n", e->K, burst, done_now, s.iters_max, s.iters_last, s.stalled, in_flight, (long long)e->stats.graph_launches);
    // next unroll count: the last window's maximum (the last solve's count right after a reset,
    // where the counts are still falling) plus a little head-room; a solve that needs more simply
    // takes a second graph launch
    int base = e->calibrated ? std::max(s.iters_max, in_flight) : std::max(s.iters_last, in_flight);
    if (stalls && base <= e->K) base = e->K + 2;
    e->K = next_K(std::max(2, base + khead), cap);
    // slots every solve of the last window needed (two below its smallest count, in steps of 4: few distinct graphs)
    e->spec_lo = (e->calibrated && s.iters_min < (1 << 20) && !stalls) ? std::min(1000, std::max(0, ((s.iters_min - 2) / 4) * 4)) : 0;
    e->calibrated = true;
    if (remaining > 0) {
      HIPCHK(hipMemsetAsync(&e->c.st->iters_max, 0, sizeof(int), e->stream));
      HIPCHK(hipMemsetAsync(&e->c.st->iters_min, 0x7f, sizeof(int), e->stream));
    }
    if (++guard > 1000000) { fprintf(stderr, "osqp_amd: the ADMM run loop did not terminate\n"); return HIPENG_ERR_HIP; }
  }
  if (!had_fail && e->res_use) e->res_fails = 0;     // strikes count in a row: a call of clean resident windows starts over
  e->admm_total = s.admm_done;
  e->stats.admm_done = (c_int)(s.admm_done - start);
  e->stats.pcg_iters_total = (c_int)s.iters_total;
  e->stats.pcg_iters_last = s.iters_last;
  e->stats.pcg_iters_max = call_max;
  e->stats.pcg_forced = s.forced;
  e->stats.neg_curvature = s.neg_curv_seen;
  return 0;
}

extern "C" int hipeng_reset_stats(hipeng *e) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  State s;
  if (read_state(e, &s)) return HIPENG_ERR_HIP;
  s.forced = 0; s.iters_total = 0; s.iters_max = 0; s.iters_last = 0; s.neg_curv_seen = 0; s.neg_curv = 0;
  HIPCHK(hipMemcpyAsync(e->c.st, &s, sizeof(State), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->stats = hipeng_stats{};
  e->stats.kernels_per_pcg_iter = 2;
  e->calibrated = false;
  e->K = 8; e->spec_lo = 0;
  return 0;
}

extern "C" int hipeng_get_stats(hipeng *e, hipeng_stats *st) {
  if (!e || !st) return HIPENG_ERR_ARG;
  *st = e->stats;
  st->resident = (e->res_on && e->res_fails < 3) ? 1 : 0;     // (a launch that gave up suspends the mode for the rest of that call only)
  return 0;
}

// ---- residual scalars -----------------------------------------------------
static void fill_scalars(const double *h, hipeng_scalars *o) {
  o->pri_res_u = h[SCI(SC_PRI_U)]; o->pri_res_s = h[SCI(SC_PRI_S)];
  o->z_u = h[SCI(SC_Z_U)]; o->z_s = h[SCI(SC_Z_S)]; o->Ax_u = h[SCI(SC_AX_U)]; o->Ax_s = h[SCI(SC_AX_S)];
  o->dua_res_u = h[SCI(SC_DUA_U)]; o->dua_res_s = h[SCI(SC_DUA_S)];
  o->q_u = h[SCI(SC_Q_U)]; o->q_s = h[SCI(SC_Q_S)]; o->Aty_u = h[SCI(SC_ATY_U)]; o->Aty_s = h[SCI(SC_ATY_S)];
  o->Px_u = h[SCI(SC_PX_U)]; o->Px_s = h[SCI(SC_PX_S)];
  o->obj_scaled = h[SCI(SC_OBJ)];
  o->dy_norm_u = h[SCI(SC_DYN_U)]; o->dy_norm_s = h[SCI(SC_DYN_S)]; o->dy_lhs = h[SCI(SC_DYLHS)];
  o->dx_norm_u = h[SCI(SC_DXN_U)]; o->dx_norm_s = h[SCI(SC_DXN_S)]; o->q_dx = h[SCI(SC_QDX)];
  o->Atdy_u = h[SCI(SC_ATDY_U)]; o->Atdy_s = h[SCI(SC_ATDY_S)];
  o->Pdx_u = h[SCI(SC_PDX_U)]; o->Pdx_s = h[SCI(SC_PDX_S)];
  o->Adx_viol = h[SCI(SC_ADX_VIOL)];
}

extern "C" int hipeng_residuals(hipeng *e, hipeng_scalars *out) {
  if (!e || !out) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemsetAsync(e->c.scal, 0, SC_COUNT * 16 * sizeof(double), e->stream));
  if (e->c.A.nblk > e->c.A.nwave) hipLaunchKernelGGL(k_huge_dot, dim3(e->c.gridA), dim3(TB), 0, e->stream, e->c, (const double *)e->c.xy, 1);
  hipLaunchKernelGGL(k_residuals, dim3(e->c.gridA + e->c.gridM), dim3(TB), 0, e->stream, e->c);
  hipLaunchKernelGGL(k_final_sums, dim3(1), dim3(TB), 0, e->stream, e->c);
  HIPCHK(hipGetLastError());
  double h[SCI(SC_COUNT)];
  HIPCHK(hipMemcpyAsync(h, e->c.scal, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  fill_scalars(h, out);
  return 0;
}

extern "C" int hipeng_certificates(hipeng *e, c_float eps_dx, int unscaled, hipeng_scalars *io) {
  if (!e || !io) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  // the five output slots (one 128-byte line each, SC_ATDY_U .. SC_ADX_VIOL) start from zero on every call:
  // the exact and the approximate check of one iteration both come here
  HIPCHK(hipMemsetAsync(e->c.scal + SCI(SC_ATDY_U), 0, (size_t)(SCI(SC_ADX_VIOL) - SCI(SC_ATDY_U) + 1) * sizeof(double), e->stream));
  hipLaunchKernelGGL(k_certificates, dim3(e->c.gridA + e->c.gridM), dim3(TB), 0, e->stream, e->c,
                     (double)eps_dx, unscaled);
  HIPCHK(hipGetLastError());
  double h[SCI(SC_COUNT)];
  HIPCHK(hipMemcpyAsync(h, e->c.scal, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  io->Atdy_u = h[SCI(SC_ATDY_U)]; io->Atdy_s = h[SCI(SC_ATDY_S)];
  io->Pdx_u = h[SCI(SC_PDX_U)]; io->Pdx_s = h[SCI(SC_PDX_S)]; io->Adx_viol = h[SCI(SC_ADX_VIOL)];
  return 0;
}

extern "C" int hipeng_download(hipeng *e, c_float *x, c_float *y, c_float *z, c_float *dx,
                               c_float *dy, int dy_projected) {
  if (!e) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const size_t n = e->n, m = e->m;
  if (x && n) HIPCHK(hipMemcpyAsync(x, e->c.xy, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (y && m) HIPCHK(hipMemcpyAsync(y, e->c.xy + n, m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (z && m) HIPCHK(hipMemcpyAsync(z, e->c.z, m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (dx && n) HIPCHK(hipMemcpyAsync(dx, e->c.dxy, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (dy && m) HIPCHK(hipMemcpyAsync(dy, dy_projected ? e->c.dxy + n : e->c.dy, m * sizeof(double),
                                     hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// ---- plugin-boundary solve --------------------------------------------------
extern "C" int hipeng_kkt_solve(hipeng *e, c_float *b) {
  if (!e || !b) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const int n = e->n, m = e->m;
  // reduced right-hand side: cvec = b1, vb(m-part) = rho . b2 ; start from x~ = 0
  if (hipeng_cold_start(e)) return HIPENG_ERR_HIP;
  if (upload_vec(e, e->c.cvec, b, n)) return HIPENG_ERR_HIP;
  e->prm.use_cvec = 1;                       // (before the m-parts are formed: an eliminated variable's b_y comes from cvec then)
  if (push_params(e)) return HIPENG_ERR_HIP;
  if (m > 0) {
    // stage b2 in z, y = 0  =>  k_refresh_m writes vb = rho*z - y = rho.b2
    if (upload_vec(e, e->c.z, b + n, m)) return HIPENG_ERR_HIP;
    hipLaunchKernelGGL(k_refresh_m, dim3(elem_grid(m)), dim3(TB), 0, e->stream, e->c, 0); e->start_dirty = true;
  }
  int rc = hipeng_run_admm(e, 1);
  e->prm.use_cvec = 0;
  if (push_params(e)) return HIPENG_ERR_HIP;
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(b, e->c.va, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (m > 0) HIPCHK(hipMemcpyAsync(b + n, e->c.zt, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  std::vector<double> xte;
  if (e->c.nelim) {          // x~ of the eliminated variables comes from the back substitution (the PCG vector is zero there)
    xte.resize((size_t)m);
    HIPCHK(hipMemcpyAsync(xte.data(), e->c.xte, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->c.nelim) for (int j = 0; j < n; j++) if (e->erow[(size_t)j] >= 0) b[j] = xte[(size_t)e->erow[(size_t)j]];
  return 0;
}

// ---- kernel-level entry points for tests and the roofline measurement -------
extern "C" int hipeng_spmv(hipeng *e, int which, const c_float *x, c_float *y) {
  if (!e || !x || !y || which < 0 || which > 2) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const int n = e->n, m = e->m;
  double *in = e->c.pt0, *out = e->c.pt1;   // scratch of n+m doubles each
  HIPCHK(hipMemsetAsync(in, 0, (size_t)std::max(1, n + m) * sizeof(double), e->stream));
  if (which == 0) {
    if (upload_vec(e, in, x, n)) return HIPENG_ERR_HIP;
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), dim3(TB), 0, e->stream, e->c.A, in, out, 0);
    HIPCHK(hipMemcpyAsync(y, out, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  } else {
    if (which == 1) { if (upload_vec(e, in + n, x, m)) return HIPENG_ERR_HIP; }
    else if (upload_vec(e, in, x, n)) return HIPENG_ERR_HIP;
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.M.nblk))), dim3(TB), 0, e->stream, e->c.M, in, out,
                       which == 1 ? 2 : 1);
    HIPCHK(hipMemcpyAsync(y, out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// Device-to-device SpMV with the engine's matrices (which as in hipeng_spmv), on the engine's stream, no sync:
// the building block of the row-partitioned multi-GPU variant (osqp_amd/rowpart.py), whose vectors live in
// torch tensors.  d_x / d_y are device pointers of n (A x: in, A' y / P x: out) or m doubles.
extern "C" int hipeng_spmv_dev(hipeng *e, int which, const double *d_x, double *d_y) {
  if (!e || !d_x || !d_y || which < 0 || which > 2) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  const int n = e->n, m = e->m;
  if (which == 0) {
    if (m > 0) hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.A.nblk))), dim3(TB), 0, e->stream, e->c.A, d_x, d_y, 0);
  } else if (which == 1) {
    // the fused row matrix reads y at column ids n + i: stage it behind n unused slots
    double *in = e->c.pt0;
    if (m > 0) HIPCHK(hipMemcpyAsync(in + n, d_x, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.M.nblk))), dim3(TB), 0, e->stream, e->c.M, (const double *)in, d_y, 2);
  } else {
    // k_spmv stages the products of whole rows of [P | A'] (the A' entries gather at column ids up to n + m - 1)
    // before it sums the P part: the input must be n + m long, so x is staged too
    double *in = e->c.pt0;
    HIPCHK(hipMemcpyAsync(in, d_x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    hipLaunchKernelGGL(k_spmv, dim3(std::min(MAX_PARTS, std::max(1, e->c.M.nblk))), dim3(TB), 0, e->stream, e->c.M, (const double *)in, d_y, 1);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int hipeng_time_kernel(hipeng *e, int which, int reps, double *usec) {
  if (!e || !usec || reps <= 0 || which < 0 || which > 8) return HIPENG_ERR_ARG;
  if (which == 8 && !(e->res_on && e->res_fails < 3)) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  CuLease lease;
  if (which == 8 && e->res_kind < 3) lease.take(e->device, e->res_cus, e->res_kind == 2 ? e->bc.nwg : e->rc.nwg);
  auto one = [&](int it) {
    // the first two kernels of a resident ADMM iteration: right-hand side + start residual, then the whole linear solve
    // (without k_admm_finalize the iterates do not move: every repetition solves the same system from the same start)
    if (which == 8) { launch_init(e, 1); launch_resident(e); return; }
    if (which == 5) { launch_init(e, 1); return; }   // first kernel of an ADMM iteration
    if (which == 6) { launch_pcg_iter(e, it, 4); return; }   // one whole PCG iteration (all its launches, in loop order)

    if (which == 7) { hipLaunchKernelGGL(k_fill, dim3(1), dim3(TB), 0, e->stream, e->c.cvec, 0.0, 0); return; }   // empty dependent launch
    if (which == 0) launch_cg_A(e, it, 4);
    else if (which == 3) launch_cg_A(e, it, 4 | 16);
    else if (which == 4) launch_cg_A(e, it, 4 | 32);
    else launch_cg_B(e, it, 4);
  };
  // the launches are captured into one graph so the measurement is not bound by
  // the host's eager launch rate (~3-4 us per launch)
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < reps; i++) one(i);
  HIPCHK(hipStreamEndCapture(e->stream, &g));
  HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  HIPCHK(hipGraphDestroy(g));
  HIPCHK(hipGraphLaunch(ge, e->stream));            // warm-up replay
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  HIPCHK(hipGraphLaunch(ge, e->stream));
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipEventSynchronize(e->ev1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  HIPCHK(hipGraphExecDestroy(ge));
  *usec = 1e3 * (double)ms / reps;
  if (which == 8) {
    // a resident launch of the timed run may have given up: the number is void then; leave the engine as the run loop would
    State s;
    if (read_state(e, &s)) return HIPENG_ERR_HIP;
    if (s.res_fail) {
      if (resident_gave_up(e, s, false)) return HIPENG_ERR_HIP;
      return HIPENG_ERR_HIP;
    }
  }
  return 0;
}

// For the CPU tests (no device needed): the host side of the resident PCG set-up -- symbolic K, row partition, positions
// in the exchanged vector, register layout -- for a machine of `nwg` CUs.  stats: [0] the problem qualifies, [1] entries of K
// per thread, [2] length of the exchanged vector, [3] nnz(K), [4] workgroups that own rows, [5] most rows per workgroup,
// [6] most entries per workgroup, [7] 448.  Optional outputs (NULL to skip): Kptr (n + 1), Kcol / kdst (cap entries at most:
// column and register slot of every entry of K, row by row), rowpos (n), slotcol (nwg * E * 448: the position held in every
// slot), wg4 (4 ints per workgroup: first row, rows, entries, first position).
extern "C" int hipeng_resident_plan(const csc *P, const csc *A, int nwg, long long stats[8], int *Kptr, int *Kcol, int *kdst, long long cap,
                                    unsigned short *rowpos, unsigned short *slotcol, int *wg4) {
  if (!P || !A || !stats || P->n > 0x7ffffff0LL || A->m > 0x7ffffff0LL) return HIPENG_ERR_ARG;
  hipeng tmp;
  tmp.n = (int)P->n; tmp.m = (int)A->m;
  build_A(&tmp, A); build_M(&tmp, P, A); build_dense(&tmp, P);
  ResPlanOut po;
  const int rc = build_resident(&tmp, nwg, &po);
  for (int k = 0; k < 8; k++) stats[k] = 0;
  if (rc) return rc;
  stats[0] = po.ok; stats[7] = RES_PT;
  if (!po.ok) return 0;
  stats[1] = po.E; stats[2] = po.npad; stats[3] = po.nnzK;
  for (const ResWG &w : po.wg) { if (w.nr) stats[4]++; stats[5] = std::max<long long>(stats[5], w.nr); stats[6] = std::max<long long>(stats[6], w.cnt); }
  if (Kptr) std::copy(po.Kptr.begin(), po.Kptr.end(), Kptr);
  if (po.nnzK <= cap) { if (Kcol) std::copy(po.Kcol.begin(), po.Kcol.end(), Kcol); if (kdst) std::copy(po.kdst.begin(), po.kdst.end(), kdst); }
  else if (Kcol || kdst) return HIPENG_ERR_ARG;
  if (rowpos) std::copy(po.rowpos.begin(), po.rowpos.end(), rowpos);
  if (slotcol) std::copy(po.col.begin(), po.col.end(), slotcol);
  if (wg4) for (size_t g = 0; g < po.wg.size(); g++) { wg4[4 * g] = po.wg[g].r0; wg4[4 * g + 1] = po.wg[g].nr; wg4[4 * g + 2] = po.wg[g].cnt; wg4[4 * g + 3] = po.wg[g].pos; }
  return 0;
}

#ifdef OSQP_AMD_TIMELINE
// make TIMELINE=1 builds only: copy out (and reset) the kernel start marks; returns their number
extern "C" long long hipeng_timeline(hipeng *e, unsigned long long *out, long long cap) {
  if (!e || !out) return -1;
  if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) return -1;
  unsigned long long cnt = 0;
  if (hipMemcpy(&cnt, e->c.tl, sizeof cnt, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const long long k = (long long)std::min<unsigned long long>(std::min<unsigned long long>(cnt, TL_CAP - 2), (unsigned long long)cap);
  if (k > 0 && hipMemcpy(out, e->c.tl + 1, (size_t)k * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (hipMemset(e->c.tl, 0, sizeof(unsigned long long)) != hipSuccess) return -1;
  return k;
}
#endif

// 1 if the vector update and the operator apply of k_cg_A run as two launches (A dominated by long rows)
extern "C" int hipeng_is_split(hipeng *e) { return e && e->split ? 1 : 0; }
extern "C" int hipeng_row_eliminated(hipeng *e, c_int i) { return e && !e->ecol.empty() && i >= 0 && i < (c_int)e->ecol.size() && e->ecol[(size_t)i] >= 0 ? 1 : 0; }
extern "C" c_int hipeng_elim_count(hipeng *e) { return e ? e->c.nelim : 0; }

// Resident PCG: out[0] structures built, [1] in use, [2] entries of K per thread, [3] workgroups, [4] nnz(K),
// [5] LDS bytes per workgroup, [6] PCG iterations of the most recent linear solve, [7] pipelined phase switched off for this K,
// [8] true-residual checks that failed since create, [9] form: 1 k_pcg_resident, 2 k_pcg_blockres,
// [10] launches that gave up waiting (the third one ends the mode for this engine)
extern "C" int hipeng_resident_info(hipeng *e, long long out[16]) {
  if (!e || !out) return HIPENG_ERR_ARG;
  HIPCHK(hipSetDevice(e->device));
  State s;
  if (read_state(e, &s)) return HIPENG_ERR_HIP;
  out[0] = e->res_on; out[1] = e->res_on && e->res_fails < 3; out[2] = e->rc.E; out[3] = e->rc.nwg; out[4] = e->res_nnz; out[5] = (long long)e->res_lds;
  out[6] = std::max(s.iters[0], s.iters[1]); out[7] = s.res_pipe_off; out[8] = s.res_chk_fail; out[9] = e->res_kind; out[10] = e->res_gave_up; out[11] = e->res_slow + s.res_slow;
  out[12] = std::max<long long>(e->res_slow_max, s.res_slow_max); out[13] = e->res_repub + s.res_repub; out[14] = e->res_fails; out[15] = e->res_kind == 3 ? e->bd.kc : 0;
  if (e->res_kind == 2) { out[2] = 64; out[3] = e->bc.nwg; out[4] = 0; out[5] = 0; }
  if (e->res_kind == 3) { out[2] = 0; out[3] = e->c.dP.nblk; out[4] = 0; out[5] = 0; }
  if (e->res_kind == 4) { out[2] = 0; out[3] = e->dd.na; out[4] = (long long)e->dd.nap * e->dd.nap; out[5] = e->dd_chol ? 1 : 0; out[15] = e->dd.nb2; }
  return 0;
}

// Resident PCG, for the tests: the rows of K as the resident kernel holds them.  row/col/val receive nnz(K) triplets
// (row by row, columns ascending; values read from the slots the threads load them from); returns the count or a negative code.
extern "C" long long hipeng_resident_dump(hipeng *e, int *row, int *col, double *val, long long cap) {
  if (!e || e->res_kind != 1 || !row || !col || !val) return HIPENG_ERR_ARG;
  if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) return HIPENG_ERR_HIP;
  const ResCtx &rc = e->rc;
  const size_t slots = (size_t)rc.nwg * rc.E * RES_PT;
  const long long nnz = e->res_nnz;
  if (nnz > cap) return HIPENG_ERR_ARG;
  std::vector<double> v(slots); std::vector<int> krp(e->n + 1), kcj((size_t)nnz), kdst((size_t)nnz);
  if (hipMemcpy(v.data(), rc.val, slots * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(krp.data(), rc.krp, krp.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(kcj.data(), rc.kcj, kcj.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(kdst.data(), rc.kdst, kdst.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return HIPENG_ERR_HIP;
  long long k = 0;
  for (int i = 0; i < e->n; i++)
    for (int q = krp[i]; q < krp[i + 1]; q++) { row[k] = i; col[k] = kcj[q]; val[k] = v[kdst[q]]; k++; }
  return k;
}

extern "C" int hipeng_kernel_bytes(hipeng *e, int which, double *bytes) {
  if (!e || !bytes) return HIPENG_ERR_ARG;
  const double n = e->n, m = e->m;
  const double nnzA = (double)e->A.val.size();
  const double nnzPtriu = (double)e->P_toM_up.size();
  // device layout: fp64 values + int32 indices (12 B per stored entry), one
  // int32 row pointer per row, fp64 vectors; gathers counted once per vector
  // k_cg_A: A stream; u,w,p,s,r,Minv,x read + p,s,r,x,u written (12n); rho read, t written (2m)
  if (which == 0)      *bytes = nnzA * 12 + (m + 1) * 4 + 8 * (12 * n + 2 * m);
  // k_cg_B: [P|A'] stream; [u|t] and r read, w written.  P is counted as its stored upper
  // triangle (each stored entry serves both triangles) although the fused row
  // matrix M holds both, so the figure does not reward the expanded layout.
  else if (which == 1) *bytes = (nnzA + nnzPtriu) * 12 + (n + 1) * 4 + 8 * ((n + m) + 2 * n);
  else                 *bytes = 0;
  return 0;
}

// ---- ONE QP over several GPUs by rows, driven from C (include/osqp_amd_rowpart.h) ----
#include "rowpart_native.h"
