// dense_direct.h -- included by engine.hip after the elimination and block forms (it uses Ctx, State, k_blk-style helpers).
//
// Dense-direct solve (res_kind 4): problems whose reduced matrix K = P + sigma I + A' rho~ A is SMALL AND DENSE -- BASELINE
// config 3, the Lasso QP: 5 000 feature variables tied together by 10 000 data rows of 750 entries.  There the PCG streams the
// 90 MB of A twice per iteration, 22 times per ADMM iteration; the matrix it inverts piecemeal is only 5 000 x 5 000.
// So K is formed as a dense matrix, inverted explicitly on the matrix cores whenever rho / sigma / the matrices change, and one
// linear solve is ONE pass over the 210 MB of K^-1 (as the block-direct solve of the portfolio family, with one block).
//
//   unknowns of the engine's reduced system (variables not eliminated by k_elim_*):
//     B2  "sparse" variables: not in any dense row of A, no off-diagonal entry in P, at most DD_NBR neighbours, pairwise not
//         adjacent (Lasso: the 5 000 bound variables t_j, which meet x_j only).  K_B2,B2 is diagonal: they leave the system by an
//         exact sparse Schur complement (private to this solver: r_a -= K_aB D^-1 r_B before, v_B = D^-1 (r_B - K_Ba v_a) after);
//     a   everything else (n_a <= DD_MAX): the dense Schur complement  S = K_aa - K_aB D^-1 K_Ba.
//   formation (dd_refresh):  S = R' diag(w) R  over the dense rows of A (R: n_d x n_a, built from the CSR rows; one TN GEMM on
//     v_mfma_f64_16x16x4_f64) + the short rows, P and sigma scattered with atomics + the Schur terms; then the explicit inverse by
//     blocked Gauss-Jordan (128-wide pivot blocks inverted in LDS, every other flop in the same GEMM kernel).
//   solve (launch_dense_direct): k_dd_gather -> k_dd_gemv (v_a = S^-1 r_a, HBM-bound) -> k_dd_finish, applied to the residual
//     r0 = b - K x~0 that k_pcg_init forms (refinement form, as the block-direct solve).
#define DD_NB 128
#define DD_MAX 8192
#define DD_NBR 8
#define DD_DENSE_ROW 32
#ifndef DD_CH
#define DD_CH 2           // MFMA steps per prefetched chunk of k_dd_gemm_tn
#endif

struct DdCtx {
  int na, nap, nb2, nd;
  const int *vidx;          // [n] variable -> dense index (>= 0), -(b + 2) for B2 variable b, -1: eliminated by the engine
  const int *alist, *blist; // [na], [nb2] variable of a dense index / of a B2 index
  const int *bnbr;          // [nb2][DD_NBR] dense indices of a B2 variable's neighbours (-1: none)
  double *bval, *bdiag;     // [nb2][DD_NBR] K(b, neighbour), [nb2] K(b, b)
  const int *aptr; const int2 *aadj;   // the same adjacency from the dense side: for dense index a the pairs {b, slot} with bnbr[b][slot] == a
  double *S;                // [nap][nap] Schur complement, then its inverse
  double *S0;               // [nap][nap] the Schur complement as formed (both triangles): the check of every fresh inverse multiplies by it
  double *R, *dw;           // [nd][nap] dense rows of A over the dense unknowns; [nd] their weights rho~
  const int *drow;          // [nd] row of A
  const char *isdense;      // [m] 1: the row is in R
  const int *sptr, *spos;   // per variable j: the slots in M of its entries in SHORT rows of A (sptr[n + 1], spos[])
  double *rr, *vv;          // [nap] reduced residual, solution
  double *rowpart, *colpart; // [tiles][128] partial products of k_dd_symv_tiles (null: k_dd_gemv is used)
  double *D, *Bp, *T;       // inversion: [128][128] scratch, two packed panels [128][nap]
  double *Tp, *X2;          // Cholesky route: a transposed block row [nap][128], nap x nap scratch
  int *flag;                // [0] a pivot was not positive
};

// C = beta C + alpha T' diag(w) B on v_mfma_f64_16x16x4_f64.  T: [K][ldt] (so the A operand A[i][k] = T[k][i] is 4 rows x 16
// contiguous doubles per wave load), B: [K][ldb], C: [M][ldc].  Workgroup = 128 x 128 of C, wavefront = 64 x 64 = 4 x 4 tiles.
// Tiles whose rows lie in [sr0, sr1) or columns in [sc0, sc1) are left alone (the pivot block row / column of a sweep step).
__global__ void __launch_bounds__(TB, 2) k_dd_gemm_tn(double *C, int ldc, const double *T, int ldt, const double *B, int ldb, const double *w,
                                                   int M, int N, int K, double alpha, double beta, int sr0, int sr1, int sc0, int sc1, int lower, int kmode,
                                                   int ksl = 0, long long cz = 0) {
  const int row0 = blockIdx.y * 128, col0 = blockIdx.x * 128;      // (tried: an XCD-contiguous workgroup -> tile map; 24.9 -> 23.9 TFLOP/s, the sweep 21.6 -> 26.9 ms)
  if ((row0 >= sr0 && row0 < sr1) || (col0 >= sc0 && col0 < sc1)) return;
  if (lower && col0 > row0) return;            // symmetric result: the tiles on and below the diagonal only (k_dd_mirror fills the rest)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  const int wr = row0 + (wv >> 1) * 64, wc = col0 + (wv & 1) * 64;
  mfma_d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  // K in chunks of DD_CH MFMA steps (4 k each): the operand loads of the NEXT chunk are issued before the MFMAs of this one
  double a[DD_CH][4], b[DD_CH][4], an[DD_CH][4], bn[DD_CH][4];
  auto load = [&](int k0, double (&aa)[DD_CH][4], double (&bb)[DD_CH][4]) {
#pragma unroll
    for (int s = 0; s < DD_CH; ++s) {
      const int kk = k0 + 4 * s + lk;
      const bool kv = kk < K;
      const double wk = (w && kv) ? w[kk] : 1.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = wr + 16 * t + li, c = wc + 16 * t + li;
        aa[s][t] = (kv && r < M) ? T[(size_t)kk * ldt + r] : 0.0;
        bb[s][t] = (kv && c < N) ? B[(size_t)kk * ldb + c] * wk : 0.0;
      }
    }
  };
  // kmode 1: B is lower triangular (B[k][j] = 0 for k < j): k starts at the column tile; 2: T and B both are: k starts at the row tile
  // ksl > 0: split K -- slice blockIdx.z takes k in [z ksl, (z + 1) ksl) and writes its partial product to C + z cz (summed in a fixed
  // order afterwards, k_dd_sum_slices): skinny products (128 rows, k up to n) fill the machine that way
  int kbeg = kmode == 1 ? col0 : (kmode == 2 ? row0 : 0);
  if (ksl > 0) { kbeg = max(kbeg, (int)blockIdx.z * ksl); K = min(K, ((int)blockIdx.z + 1) * ksl); C += (size_t)blockIdx.z * cz; }
  load(kbeg, a, b);
  for (int k0 = kbeg; k0 < K; k0 += 4 * DD_CH) {
    if (k0 + 4 * DD_CH < K) load(k0 + 4 * DD_CH, an, bn);
#pragma unroll
    for (int s = 0; s < DD_CH; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < DD_CH; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[s][t] = an[s][t]; b[s][t] = bn[s][t]; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {              // D[row = lk + 4 r][col = li] of the tile
        const int row = wr + 16 * i + lk + 4 * r, col = wc + 16 * j + li;
        if (row < M && col < N) {
          double *p = C + (size_t)row * ldc + col;
          *p = beta == 0.0 ? alpha * acc[i][j][r] : beta * *p + alpha * acc[i][j][r];
        }
      }
}

// The same product with both operands staged through LDS (two buffers of 16 k): every value is loaded from global memory once per
// workgroup, with 16-byte coalesced loads, instead of once per wavefront with 8-byte ones; the next chunk's loads are in registers
// while the 64 MFMAs of this one run.  Rows of the LDS tiles are padded to 144 doubles (288 dwords = 32 mod 64 banks: the four k of
// an operand fetch fall on disjoint bank halves).
#define DD_KC 16
#define DD_LP 144
__global__ void __launch_bounds__(TB, 2) k_dd_gemm_tn_lds(double *C, int ldc, const double *T, int ldt, const double *B, int ldb, const double *w,
                                                       int M, int N, int K, double alpha, double beta, int sr0, int sr1, int sc0, int sc1, int lower, int kmode,
                                                       int ksl = 0, long long cz = 0) {
  const int row0 = blockIdx.y * 128, col0 = blockIdx.x * 128;
  if ((row0 >= sr0 && row0 < sr1) || (col0 >= sc0 && col0 < sc1)) return;
  if (lower && col0 > row0) return;
  extern __shared__ __attribute__((aligned(16))) double sm[];           // As[2][16][144], Bs[2][16][144]
  double *As = sm, *Bs = sm + 2 * DD_KC * DD_LP;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, lk = lane >> 4;
  const int wr = (wv >> 1) * 64, wc = (wv & 1) * 64;                     // the wavefront's 64 x 64 inside the tile
  int kbeg = kmode == 1 ? col0 : (kmode == 2 ? row0 : 0);
  if (ksl > 0) { kbeg = max(kbeg, (int)blockIdx.z * ksl); K = min(K, ((int)blockIdx.z + 1) * ksl); C += (size_t)blockIdx.z * cz; }
  mfma_d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  // global -> registers: thread t takes the double2 at (k row = p / 64, column pair = p % 64) for p = t + 256 u, u < 4, of each operand
  double2 ga[4], gb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = t + 256 * u, kr = p >> 6, c2 = (p & 63) * 2, kk = k0 + kr;
      const bool kv = kk < K;
      const double wk = (w && kv) ? w[kk] : 1.0;
      double2 a = double2{0.0, 0.0}, b = double2{0.0, 0.0};
      if (kv && row0 + c2 + 1 < M) a = *reinterpret_cast<const double2 *>(T + (size_t)kk * ldt + row0 + c2);
      else if (kv && row0 + c2 < M) a.x = T[(size_t)kk * ldt + row0 + c2];
      if (kv && col0 + c2 + 1 < N) b = *reinterpret_cast<const double2 *>(B + (size_t)kk * ldb + col0 + c2);
      else if (kv && col0 + c2 < N) b.x = B[(size_t)kk * ldb + col0 + c2];
      ga[u] = a; gb[u] = double2{b.x * wk, b.y * wk};
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = t + 256 * u, kr = p >> 6, c2 = (p & 63) * 2;
      *reinterpret_cast<double2 *>(As + (buf * DD_KC + kr) * DD_LP + c2) = ga[u];
      *reinterpret_cast<double2 *>(Bs + (buf * DD_KC + kr) * DD_LP + c2) = gb[u];
    }
  };
  gload(kbeg);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < K; k0 += DD_KC) {
    const bool more = k0 + DD_KC < K;
    if (more) gload(k0 + DD_KC);
    const double *Ab = As + buf * DD_KC * DD_LP, *Bb = Bs + buf * DD_KC * DD_LP;
#pragma unroll
    for (int s = 0; s < DD_KC / 4; ++s) {
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = Ab[(4 * s + lk) * DD_LP + wr + 16 * q + li]; b[q] = Bb[(4 * s + lk) * DD_LP + wc + 16 * q + li]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) { sstore(buf ^ 1); __syncthreads(); buf ^= 1; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wr + 16 * i + lk + 4 * r, col = col0 + wc + 16 * j + li;
        if (row < M && col < N) {
          double *p = C + (size_t)row * ldc + col;
          *p = beta == 0.0 ? alpha * acc[i][j][r] : beta * *p + alpha * acc[i][j][r];
        }
      }
}

// every product of this file goes through here: the LDS-staged kernel, or (OSQP_AMD_DENSE_GEMM_LDS=0) the one that feeds the MFMAs from global memory
static void dd_gemm(hipStream_t stream, dim3 grid, double *C, int ldc, const double *T, int ldt, const double *B, int ldb, const double *w,
                    int M, int N, int K, double alpha, double beta, int sr0, int sr1, int sc0, int sc1, int lower, int kmode, int ksl = 0, long long cz = 0) {
  static int lds = -1;
  constexpr size_t bytes = (size_t)4 * DD_KC * DD_LP * sizeof(double);
  if (lds < 0) {
    lds = 1;
    if (const char *x = getenv("OSQP_AMD_DENSE_GEMM_LDS")) lds = atoi(x) != 0;
    if (lds && hipFuncSetAttribute(reinterpret_cast<const void *>(k_dd_gemm_tn_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { (void)hipGetLastError(); lds = 0; }
  }
  if (lds) hipLaunchKernelGGL(k_dd_gemm_tn_lds, grid, dim3(TB), bytes, stream, C, ldc, T, ldt, B, ldb, w, M, N, K, alpha, beta, sr0, sr1, sc0, sc1, lower, kmode, ksl, cz);
  else hipLaunchKernelGGL(k_dd_gemm_tn, grid, dim3(TB), 0, stream, C, ldc, T, ldt, B, ldb, w, M, N, K, alpha, beta, sr0, sr1, sc0, sc1, lower, kmode, ksl, cz);
}

// D = (pivot block kb of A)^-1 in LDS, by Gauss-Jordan in 32 x 32 sub-blocks (no pivoting: the Schur complements of a positive
// definite matrix are positive definite; a non-positive pivot raises the flag).  Per sub-block: its own 32 x 32 inverse element by
// element (1024 threads = one element each, 32 steps of two barriers), then the block step on the rest of the 128 x 128 array in
// place: row panel <- D_s x row panel (a thread per column, the column in registers), everything else -= column panel x new row panel
// (3 x 3 outputs per thread), column panel <- -column panel x D_s (a thread per row).  128 element-wise steps over the whole array
// (the first version) took 0.25 ms per pivot block, 10 of the 21.6 ms of a sweep at n = 5120.
#define DD_SB 32
__global__ void __launch_bounds__(INV_TB) k_dd_pivot(const double *A, int lda, int kb, double *D, int *flag) {
  extern __shared__ __attribute__((aligned(16))) double bl[];       // 128 x 128
  const int t = threadIdx.x;
  {
    const int j = t & (DD_NB - 1), i0 = t / DD_NB;
    const double *src = A + (size_t)kb * DD_NB * lda + (size_t)kb * DD_NB;
    for (int i = i0; i < DD_NB; i += INV_TB / DD_NB) bl[i * DD_NB + j] = src[(size_t)i * lda + j];
  }
  __syncthreads();
  for (int q0 = 0; q0 < DD_NB; q0 += DD_SB) {
    // (1) the sub-block's inverse, in place
    {
      const int i = t >> 5, j = t & 31;
      double *S = bl + q0 * DD_NB + q0;
      for (int p = 0; p < DD_SB; ++p) {
        const double piv = S[p * DD_NB + p], ci = S[i * DD_NB + p], rj = S[p * DD_NB + j], cur = S[i * DD_NB + j];
        if (!(piv > 0.0) && t == 0) atomicOr(flag, 1);
        const double inv = 1.0 / piv;
        double v;
        if (i == p) v = (j == p) ? inv : rj * inv;
        else if (j == p) v = -ci * inv;
        else v = cur - ci * (rj * inv);
        __syncthreads();
        S[i * DD_NB + j] = v;
        __syncthreads();
      }
    }
    // (2) row panel <- D_s x row panel, columns outside the sub-block: thread (column j, row group g) with the column in registers
    {
      const int j = t & (DD_NB - 1), g = t / DD_NB;                 // g = 0..7: rows g, g + 8, g + 16, g + 24 of the panel
      const bool out = j < q0 || j >= q0 + DD_SB;
      double col[DD_SB];
#pragma unroll
      for (int c = 0; c < DD_SB; ++c) col[c] = bl[(q0 + c) * DD_NB + j];
      __syncthreads();
      if (out) {
#pragma unroll
        for (int u = 0; u < DD_SB / 8; ++u) {
          const int r = g + 8 * u;
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < DD_SB; ++c) s += bl[(q0 + r) * DD_NB + q0 + c] * col[c];
          bl[(q0 + r) * DD_NB + j] = s;
        }
      }
      __syncthreads();
    }
    // (3) everything outside the sub-block's rows and columns -= (old column panel) x (new row panel): 96 x 96 outputs, 3 x 3 per thread
    {
      const int ti = t >> 5, tj = t & 31;
      int oi[3], oj[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) { int a = ti + 32 * u; oi[u] = a < q0 ? a : a + DD_SB; a = tj + 32 * u; oj[u] = a < q0 ? a : a + DD_SB; }
      double acc[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
      for (int c = 0; c < DD_SB; ++c) {
        double ca[3], rb[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) { ca[u] = bl[oi[u] * DD_NB + q0 + c]; rb[u] = bl[(q0 + c) * DD_NB + oj[u]]; }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int v = 0; v < 3; ++v) acc[u][v] += ca[u] * rb[v];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) bl[oi[u] * DD_NB + oj[v]] -= acc[u][v];
      __syncthreads();
    }
    // (4) column panel <- -(column panel) x D_s, rows outside the sub-block: thread (row i, column group g) with the row in registers
    {
      const int i = t & (DD_NB - 1), g = t / DD_NB;
      const bool out = i < q0 || i >= q0 + DD_SB;
      double row[DD_SB];
#pragma unroll
      for (int r = 0; r < DD_SB; ++r) row[r] = bl[i * DD_NB + q0 + r];
      __syncthreads();
      if (out) {
#pragma unroll
        for (int u = 0; u < DD_SB / 8; ++u) {
          const int c = g + 8 * u;
          double s = 0.0;
#pragma unroll
          for (int r = 0; r < DD_SB; ++r) s += row[r] * bl[(q0 + r) * DD_NB + q0 + c];
          bl[i * DD_NB + q0 + c] = -s;
        }
      }
      __syncthreads();
    }
  }
  {
    const int j = t & (DD_NB - 1), i0 = t / DD_NB;
    for (int i = i0; i < DD_NB; i += INV_TB / DD_NB) D[i * DD_NB + j] = bl[i * DD_NB + j];
  }
}
// Wt = column panel kb of A transposed ([128][n], from the lower triangle: rows below the pivot block from the column panel itself,
// columns left of it from the row panel -- the matrix is symmetric and only its lower tiles are kept up to date): 64 rows per workgroup
__global__ void __launch_bounds__(TB) k_dd_panel(const double *A, int lda, int n, int kb, double *Wt) {
  __shared__ double tile[64][DD_NB + 1];
  const int i0 = blockIdx.x * 64, p0 = kb * DD_NB;
  if (i0 >= p0) {                        // at or below the pivot block: A[i][p0 + r], transposed through LDS
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) {
      const int i = q / DD_NB, r = q % DD_NB;
      tile[i][r] = i0 + i < n ? A[(size_t)(i0 + i) * lda + p0 + r] : 0.0;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) {
      const int r = q / 64, i = q % 64;
      if (i0 + i < n) Wt[(size_t)r * n + i0 + i] = tile[i][r];
    }
  } else {                               // left of it: A[i][p0 + r] = A[p0 + r][i], rows of the row panel
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) {
      const int r = q / 64, i = q % 64;
      Wt[(size_t)r * n + i0 + i] = A[(size_t)(p0 + r) * lda + i0 + i];
    }
  }
}
// the swept pivot block and panel back into the lower triangle: A_kk = -D, A[i][p0 + c] = V[c][i] below, A[p0 + r][j] = V[r][j] left
__global__ void __launch_bounds__(TB) k_dd_store_panel(double *A, int lda, int n, int kb, const double *Vt, const double *D) {
  __shared__ double tile[DD_NB][64 + 1];
  const int i0 = blockIdx.x * 64, p0 = kb * DD_NB;
  if (i0 >= p0 + DD_NB) {
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int r = q / 64, i = q % 64; tile[r][i] = i0 + i < n ? Vt[(size_t)r * n + i0 + i] : 0.0; }
    __syncthreads();
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int i = q / DD_NB, r = q % DD_NB; if (i0 + i < n) A[(size_t)(i0 + i) * lda + p0 + r] = tile[r][i]; }
  } else if (i0 + 64 <= p0) {
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int r = q / 64, i = q % 64; A[(size_t)(p0 + r) * lda + i0 + i] = Vt[(size_t)r * n + i0 + i]; }
  } else {                               // the pivot block itself (two workgroups, 64 columns each)
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int r = q / 64, i = q % 64; A[(size_t)(p0 + r) * lda + i0 + i] = -D[r * DD_NB + (i0 - p0) + i]; }
  }
}
// upper = lower' (and the whole matrix negated when neg: the sweep leaves -A^-1)
__global__ void __launch_bounds__(TB) k_dd_mirror(double *A, int n, int neg) {
  __shared__ double tile[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const double sg = neg ? -1.0 : 1.0;
  for (int q = threadIdx.x; q < 64 * 64; q += TB) { const int i = q / 64, j = q % 64; tile[i][j] = sg * A[(size_t)(bi * 64 + i) * n + bj * 64 + j]; }
  __syncthreads();
  if (neg) for (int q = threadIdx.x; q < 64 * 64; q += TB) { const int i = q / 64, j = q % 64; if (bi != bj || j <= i) A[(size_t)(bi * 64 + i) * n + bj * 64 + j] = tile[i][j]; }
  for (int q = threadIdx.x; q < 64 * 64; q += TB) { const int j = q / 64, i = q % 64; if (bi != bj || j < i) A[(size_t)(bj * 64 + j) * n + bi * 64 + i] = tile[i][j]; }
}

// A (n x n, n a multiple of 128, symmetric positive definite, lower tiles valid) <- A^-1 in place on `stream`, by symmetric block sweeps
// (Beaton's sweep operator in blocks: half the flops of Gauss-Jordan, everything stays symmetric).  Per pivot block k:
//   D = A_kk^-1 (LDS);  W = column panel k (packed transposed, Wt);  V = W D (Vt = D Wt: one 128 x n GEMM);
//   A_ij -= V_i W_j' for the lower tiles outside block row / column k (one GEMM);  panel <- V;  A_kk <- -D.
// All pivots swept: A = -(A^-1); mirrored and negated at the end.
// (Tried: the pivot inversion of block k + 1 on a second stream beside the bulk of step k's trailing update, which is then issued in
// two parts -- look-ahead.  21.6 -> 24.9 ms at n = 5120, 3.5 -> 14.5 ms at n = 1280: two event hand-overs per step and the stream's
// creation cost more than the 0.25 ms pivot kernels they were to hide.)
static int dd_invert_sweep(hipStream_t stream, double *A, int n, double *D, double *Wt, double *Vt, int *flag) {
  static bool lds_set = false;
  if (!lds_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dd_pivot), hipFuncAttributeMaxDynamicSharedMemorySize, DD_NB * DD_NB * (int)sizeof(double)) != hipSuccess) return HIPENG_ERR_HIP;
    lds_set = true;
  }
  const int nb = n / DD_NB;
  for (int kb = 0; kb < nb; kb++) {
    const int p0 = kb * DD_NB, p1 = p0 + DD_NB;
    hipLaunchKernelGGL(k_dd_pivot, dim3(1), dim3(INV_TB), DD_NB * DD_NB * sizeof(double), stream, (const double *)A, n, kb, D, flag);
    hipLaunchKernelGGL(k_dd_panel, dim3(n / 64), dim3(TB), 0, stream, (const double *)A, n, n, kb, Wt);
    // Vt[c][i] = sum_r D[r][c] Wt[r][i]  (D symmetric)
    dd_gemm(stream, dim3(nb, 1), Vt, n, (const double *)D, DD_NB, (const double *)Wt, n, (const double *)nullptr,
                       DD_NB, n, DD_NB, 1.0, 0.0, -1, -1, p0, p1, 0, 0);
    // A[i][j] -= sum_c Vt[c][i] Wt[c][j]
    dd_gemm(stream, dim3(nb, nb), A, n, (const double *)Vt, n, (const double *)Wt, n, (const double *)nullptr,
                       n, n, DD_NB, -1.0, 1.0, p0, p1, p0, p1, 1, 0);
    hipLaunchKernelGGL(k_dd_store_panel, dim3(n / 64), dim3(TB), 0, stream, A, n, n, kb, (const double *)Vt, (const double *)D);
  }
  hipLaunchKernelGGL(k_dd_mirror, dim3(n / 64, n / 64), dim3(TB), 0, stream, A, n, 1);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- the same inverse by blocked Cholesky: A = L L',  X = L^-1 (blocked triangular inversion),  A^-1 = X' X -----------------------------
// Error cond(A) eps where the sweeps above have cond(A)^2 eps (tools/dd_illcond.py); the same n^3 flops, all but the 128 x 128 diagonal
// work in k_dd_gemm_tn.  Nothing is ever multiplied by an explicit inverse of a diagonal block: panels are SOLVED against it.
//
// L_kk = chol(A_kk) in LDS (lower; the upper triangle of the block is zeroed): 1024 threads, column p scaled, trailing rank-one update
__global__ void __launch_bounds__(INV_TB) k_dd_chol(double *A, int lda, int kb, int *flag) {
  extern __shared__ __attribute__((aligned(16))) double bl[];       // 128 x 128
  __shared__ double colp[DD_NB];
  constexpr int RS = INV_TB / DD_NB, RU = DD_NB / RS;
  const int t = threadIdx.x, j = t & (DD_NB - 1), i0 = t / DD_NB;
  double *blk = A + (size_t)kb * DD_NB * lda + (size_t)kb * DD_NB;
  for (int i = i0; i < DD_NB; i += RS) bl[i * DD_NB + j] = blk[(size_t)i * lda + j];
  __syncthreads();
  for (int p = 0; p < DD_NB; ++p) {
    const double d = bl[p * DD_NB + p];
    if (!(d > 0.0) && t == 0) atomicOr(flag, 1);
    const double inv = 1.0 / sqrt(d > 0.0 ? d : 1.0);
    __syncthreads();
    if (t < DD_NB) { const double v = t >= p ? bl[t * DD_NB + p] * inv : 0.0; colp[t] = v; bl[t * DD_NB + p] = v; }      // column p of L (l_pp = sqrt(d))
    __syncthreads();
    if (j > p) {
      const double lj = colp[j];
#pragma unroll
      for (int u = 0; u < RU; ++u) { const int i = i0 + RS * u; if (i >= j) bl[i * DD_NB + j] -= colp[i] * lj; }
    }
    __syncthreads();
  }
  for (int i = i0; i < DD_NB; i += RS) blk[(size_t)i * lda + j] = j <= i ? bl[i * DD_NB + j] : 0.0;
}
// Solve L_kk z = v for `cnt` vectors stored packed as V[c][v0 + q] (component c = 0..127 of vector q), in place: one thread per vector,
// the vector in LDS (its own column: no bank conflict), L_kk packed lower in LDS (its entries are read by all lanes at once: broadcast).
// z_c = (v_c - sum_{t<c} L[c][t] z_t) / L[c][c].  Serves the Cholesky panel (Z_i = W_i L^-T: every row of the panel is a vector), the
// block rows of the triangular inverse (X = L_ii^-1 Y: every column is one) and the diagonal blocks (vectors = the identity).
__global__ void __launch_bounds__(64) k_dd_trsm(const double *A, int lda, int kb, double *V, int ldv, int v0, int cnt) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *Lp = sm, *ws = sm + DD_NB * (DD_NB + 1) / 2;                  // Lp[c (c + 1) / 2 + t], ws[c][64]
  const double *blk = A + (size_t)kb * DD_NB * lda + (size_t)kb * DD_NB;
  // (sixteen loads in flight per lane: one at a time these two loops were 270 of the kernel's 364 us)
  for (int q0 = threadIdx.x; q0 < DD_NB * DD_NB; q0 += 64 * 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int q = q0 + 64 * u, c = q / DD_NB, t = q % DD_NB; v[u] = t <= c ? blk[(size_t)c * lda + t] : 0.0; }
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int q = q0 + 64 * u, c = q / DD_NB, t = q % DD_NB; if (t <= c) Lp[c * (c + 1) / 2 + t] = v[u]; }
  }
  const int qv = blockIdx.x * 64 + threadIdx.x;
  const bool on = qv < cnt;
  double *col = V + v0 + qv;
  for (int c0 = 0; c0 < DD_NB; c0 += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = on ? col[(size_t)(c0 + u) * ldv] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) ws[(c0 + u) * 64 + threadIdx.x] = v[u];
  }
  __syncthreads();
  for (int c = 0; c < DD_NB; ++c) {
    const double *Lr = Lp + c * (c + 1) / 2;
    double s0 = ws[c * 64 + threadIdx.x], s1 = 0.0;
    int t2 = 0;
    for (; t2 + 1 < c; t2 += 2) { s0 -= Lr[t2] * ws[t2 * 64 + threadIdx.x]; s1 -= Lr[t2 + 1] * ws[(t2 + 1) * 64 + threadIdx.x]; }
    if (t2 < c) s0 -= Lr[t2] * ws[t2 * 64 + threadIdx.x];
    ws[c * 64 + threadIdx.x] = (s0 + s1) / Lr[c];
  }
  if (on) for (int c = 0; c < DD_NB; ++c) col[(size_t)c * ldv] = ws[c * 64 + threadIdx.x];
}
// mode 0: A[i][p0 + c] = V[c][i] for the rows i >= p1 (the Cholesky panel back into the matrix);
// mode 1: A[p0 + r][j] = -V[r][j] for the columns j < p0 (a block row of the triangular inverse);
// mode 2: A_kk = V[r][c] (its diagonal block);  mode 3: V = the 128 x 128 identity
__global__ void __launch_bounds__(TB) k_dd_put(double *A, int lda, int n, int kb, double *V, int ldv, int mode) {
  __shared__ double tile[DD_NB][64 + 1];
  const int i0 = blockIdx.x * 64, p0 = kb * DD_NB, p1 = p0 + DD_NB;
  if (mode == 0) {
    if (i0 < p1) return;
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int r = q / 64, i = q % 64; tile[r][i] = i0 + i < n ? V[(size_t)r * ldv + i0 + i] : 0.0; }
    __syncthreads();
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int i = q / DD_NB, r = q % DD_NB; if (i0 + i < n) A[(size_t)(i0 + i) * lda + p0 + r] = tile[r][i]; }
  } else if (mode == 1) {
    if (i0 >= p0) return;
    for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int r = q / 64, i = q % 64; A[(size_t)(p0 + r) * lda + i0 + i] = -V[(size_t)r * ldv + i0 + i]; }
  } else if (mode == 2) {
    for (int q = blockIdx.x * TB + threadIdx.x; q < DD_NB * DD_NB; q += gridDim.x * TB) A[(size_t)(p0 + q / DD_NB) * lda + p0 + q % DD_NB] = V[(size_t)(q / DD_NB) * ldv + q % DD_NB];
  } else {
    for (int q = blockIdx.x * TB + threadIdx.x; q < DD_NB * DD_NB; q += gridDim.x * TB) V[(size_t)(q / DD_NB) * ldv + q % DD_NB] = (q / DD_NB == q % DD_NB) ? 1.0 : 0.0;
  }
}
// Y[r][j] = sum over the K slices of part[z][r][j] (fixed order), r < 128, j < cols
__global__ void __launch_bounds__(TB) k_dd_sum_slices(double *Y, int ldy, const double *part, long long cz, int nz, int cols) {
  for (long long q = (long long)blockIdx.x * TB + threadIdx.x; q < (long long)DD_NB * cols; q += (long long)gridDim.x * TB) {
    const int r = (int)(q / cols), j = (int)(q % cols);
    double s = 0.0;
    for (int z = 0; z < nz; ++z) s += part[(size_t)z * cz + (size_t)r * ldy + j];
    Y[(size_t)r * ldy + j] = s;
  }
}
// Tp[k][r] = A[p0 + r][k] for k < p0: block row kb of L transposed ([p0][128]), the T operand of  Y = L_i,0:i X_0:i,0:i
__global__ void __launch_bounds__(TB) k_dd_rowT(const double *A, int lda, int kb, double *Tp) {
  __shared__ double tile[DD_NB][64 + 1];
  const int k0 = blockIdx.x * 64, p0 = kb * DD_NB;
  if (k0 >= p0) return;
  for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int r = q / 64, k = q % 64; tile[r][k] = A[(size_t)(p0 + r) * lda + k0 + k]; }
  __syncthreads();
  for (int q = threadIdx.x; q < 64 * DD_NB; q += TB) { const int k = q / DD_NB, r = q % DD_NB; Tp[(size_t)(k0 + k) * DD_NB + r] = tile[r][k]; }
}

// A (n x n, n a multiple of 128, symmetric positive definite, lower tiles valid) <- A^-1 (both triangles) on `stream`.
// Wt, Yt: [128][n] panels; Tp: [n][128]; X2: n x n scratch.
static int dd_invert_chol(hipStream_t stream, double *A, int n, double *Wt, double *Yt, double *Tp, double *X2, int *flag) {
  static bool lds_set = false;
  const size_t trsm_lds = ((size_t)DD_NB * (DD_NB + 1) / 2 + (size_t)DD_NB * 64) * sizeof(double);
  if (!lds_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dd_chol), hipFuncAttributeMaxDynamicSharedMemorySize, DD_NB * DD_NB * (int)sizeof(double)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_dd_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsm_lds) != hipSuccess) return HIPENG_ERR_HIP;
    lds_set = true;
  }
  const int nb = n / DD_NB;
  const double *nul = nullptr;
  // (a) A = L L': per block column the diagonal factor, the panel solved against it, the trailing lower tiles updated
  for (int kb = 0; kb < nb; kb++) {
    const int p1 = (kb + 1) * DD_NB;
    hipLaunchKernelGGL(k_dd_chol, dim3(1), dim3(INV_TB), DD_NB * DD_NB * sizeof(double), stream, A, n, kb, flag);
    if (p1 >= n) break;
    hipLaunchKernelGGL(k_dd_panel, dim3(n / 64), dim3(TB), 0, stream, (const double *)A, n, n, kb, Wt);            // Wt[c][i] = A[i][p0 + c]  (rows i >= p0 are used)
    hipLaunchKernelGGL(k_dd_trsm, dim3((n - p1 + 63) / 64), dim3(64), trsm_lds, stream, (const double *)A, n, kb, Wt, n, p1, n - p1);
    dd_gemm(stream, dim3(nb, nb), A, n, (const double *)Wt, n, (const double *)Wt, n, nul,
                       n, n, DD_NB, -1.0, 1.0, 0, p1, 0, p1, 1, 0);
    hipLaunchKernelGGL(k_dd_put, dim3(n / 64), dim3(TB), 0, stream, A, n, n, kb, Wt, n, 0);
  }
  // (b) X = L^-1 in place, block row by block row:  X_i,0:i = -L_ii^-1 (L_i,0:i X_0:i,0:i),  X_ii = L_ii^-1
  for (int kb = 0; kb < nb; kb++) {
    const int p0 = kb * DD_NB;
    if (kb > 0) {
      hipLaunchKernelGGL(k_dd_rowT, dim3(n / 64), dim3(TB), 0, stream, (const double *)A, n, kb, Tp);
      // 128 rows x p0 columns, k up to p0: on kb workgroups alone this was 1 ms per block row; k is split into slices of 512
      const int nz = (p0 + 511) / 512;
      dd_gemm(stream, dim3(kb, 1, nz), X2, n, (const double *)Tp, DD_NB, (const double *)A, n, nul,
                         DD_NB, p0, p0, 1.0, 0.0, -1, -1, -1, -1, 0, 1, 512, (long long)DD_NB * n);
      hipLaunchKernelGGL(k_dd_sum_slices, dim3(std::min(1024, (DD_NB * p0 + TB - 1) / TB)), dim3(TB), 0, stream, Yt, n, (const double *)X2, (long long)DD_NB * n, nz, p0);
      hipLaunchKernelGGL(k_dd_trsm, dim3((p0 + 63) / 64), dim3(64), trsm_lds, stream, (const double *)A, n, kb, Yt, n, 0, p0);
    }
    hipLaunchKernelGGL(k_dd_put, dim3(16), dim3(TB), 0, stream, A, n, n, kb, Wt, n, 3);                            // Wt[0:128][0:128] = I
    hipLaunchKernelGGL(k_dd_trsm, dim3(2), dim3(64), trsm_lds, stream, (const double *)A, n, kb, Wt, n, 0, DD_NB);
    if (kb > 0) hipLaunchKernelGGL(k_dd_put, dim3(n / 64), dim3(TB), 0, stream, A, n, n, kb, Yt, n, 1);
    hipLaunchKernelGGL(k_dd_put, dim3(16), dim3(TB), 0, stream, A, n, n, kb, Wt, n, 2);
  }
  // (c) A^-1 = X' X on the lower tiles (k from the row tile on: X is lower triangular), then both triangles
  dd_gemm(stream, dim3(nb, nb), X2, n, (const double *)A, n, (const double *)A, n, nul,
                     n, n, n, 1.0, 0.0, -1, -1, -1, -1, 1, 2);
  HIPCHK(hipMemcpyAsync(A, X2, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(k_dd_mirror, dim3(n / 64, n / 64), dim3(TB), 0, stream, A, n, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- formation ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TB) k_dd_fill_R(Ctx c, DdCtx dd) {
  const int d = blockIdx.x, i = dd.drow[d];
  double *row = dd.R + (size_t)d * dd.nap;
  for (int k = c.A.rowptr[i] + (int)threadIdx.x; k < c.A.rowptr[i + 1]; k += TB) {
    const int a = dd.vidx[c.A.col[k]];
    if (a >= 0) row[a] = c.A.val[k];
  }
  if (threadIdx.x == 0) dd.dw[d] = c.rhoe[i];
}
// The short rows of A, P and sigma: one thread per unknown j adds up its OWN row of S (or its own B2 record) -- every entry has one
// writer and a fixed summation order (the order of M's row j, then of each row of A), so the formed matrix, its inverse and the
// iterates are bit-reproducible; no atomics.
__global__ void __launch_bounds__(TB) k_dd_scatter(Ctx c, DdCtx dd) {
  const double sigma = c.prm->sigma;
  for (int j = blockIdx.x * TB + threadIdx.x; j < c.n; j += gridDim.x * TB) {
    const int cj = dd.vidx[j];
    if (cj == -1) continue;
    double *row = cj >= 0 ? dd.S + (size_t)cj * dd.nap : nullptr;
    const int b = -cj - 2;
    double diag = sigma;                                    // (B2: the diagonal entry; dense: added to row[cj] below)
    for (int k = c.M.rowptr[j]; k < c.M.split[j]; ++k) {    // P
      const int col = c.M.col[k], cq = dd.vidx[col];
      if (col == j) diag += c.M.val[k];
      else if (row && cq >= 0) row[cq] += c.M.val[k];
    }
    for (int kk = dd.sptr[j]; kk < dd.sptr[j + 1]; ++kk) {     // the SHORT rows of A that hold j (a list of their slots in M: walking all of
      const int k = dd.spos[kk];                               // column j's entries to skip the dense rows was 1.1 ms on the Lasso's 1 500-entry columns)
      const int i = c.M.col[k] - c.n;
      const double wa = c.rhoe[i] * c.M.val[k];
      for (int q = c.A.rowptr[i]; q < c.A.rowptr[i + 1]; ++q) {
        const int col = c.A.col[q], cq = dd.vidx[col];
        const double v = wa * c.A.val[q];
        if (col == j) diag += v;
        else if (cq < 0) continue;
        else if (row) row[cq] += v;
        else for (int s2 = 0; s2 < DD_NBR; ++s2) if (dd.bnbr[b * DD_NBR + s2] == cq) { dd.bval[b * DD_NBR + s2] += v; break; }
      }
    }
    if (row) row[cj] += diag; else dd.bdiag[b] = diag;
  }
}
// S -= K_aB D^-1 K_Ba, row a by its own thread (through the adjacency seen from the dense side); the padding rows get a unit diagonal
__global__ void __launch_bounds__(TB) k_dd_schur(DdCtx dd) {
  for (int b = blockIdx.x * TB + threadIdx.x; b < dd.nb2; b += gridDim.x * TB) if (!(dd.bdiag[b] > 0.0)) atomicOr(dd.flag, 1);
  for (int a = blockIdx.x * TB + threadIdx.x; a < dd.nap; a += gridDim.x * TB) {
    double *row = dd.S + (size_t)a * dd.nap;
    if (a >= dd.na) { row[a] = 1.0; continue; }
    for (int k = dd.aptr[a]; k < dd.aptr[a + 1]; ++k) {
      const int b = dd.aadj[k].x;
      const double f = dd.bval[b * DD_NBR + dd.aadj[k].y] / dd.bdiag[b];
      for (int t = 0; t < DD_NBR; ++t) { const int at = dd.bnbr[b * DD_NBR + t]; if (at >= 0) row[at] -= f * dd.bval[b * DD_NBR + t]; }
    }
  }
}

// ---- solve ----------------------------------------------------------------------------------------------------------------
// rr = r_a - K_aB D^-1 r_B: the residual on the dense unknowns, minus what the B2 unknowns' rows contribute (fixed order: no atomics)
__global__ void __launch_bounds__(TB) k_dd_gather(Ctx c, DdCtx dd) {
  { const State *st = c.st; if (st->stalled || !st->run) return; }
  for (int a = blockIdx.x * TB + threadIdx.x; a < dd.nap; a += gridDim.x * TB) {
    double v = 0.0;
    if (a < dd.na) {
      v = c.init_r[dd.alist[a]];
      for (int k = dd.aptr[a]; k < dd.aptr[a + 1]; ++k) { const int b = dd.aadj[k].x; v -= dd.bval[b * DD_NBR + dd.aadj[k].y] * (c.init_r[dd.blist[b]] / dd.bdiag[b]); }
    }
    dd.rr[a] = v;
  }
}
// y = Mx x for a symmetric nap x nap matrix: one wavefront per row, eight 16-byte loads in flight per lane, the vector in LDS.
// With Mx = S^-1 this is the pass that bounds a solve (210 MB at n_a = 5 000).
__global__ void __launch_bounds__(TB) k_dd_gemv(Ctx c, int nap, const double *Mx, const double *x, double *y, int gated) {
  if (gated) { const State *st = c.st; if (st->stalled || !st->run) return; }
  extern __shared__ __attribute__((aligned(16))) double xs[];
  for (int q = threadIdx.x; q < nap; q += TB) xs[q] = x[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wv; row < nap; row += gridDim.x * 4) {
    const double2 *src = reinterpret_cast<const double2 *>(Mx + (size_t)row * nap);
    const double2 *x2 = reinterpret_cast<const double2 *>(xs);
    double s0 = 0.0, s1 = 0.0;
    int q = lane;
    for (; q + 7 * 64 < nap / 2; q += 8 * 64) {
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[q + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const double2 xx = x2[q + 64 * u]; s0 += v[u].x * xx.x; s1 += v[u].y * xx.y; }
    }
    for (; q < nap / 2; q += 64) { const double2 v = src[q], xx = x2[q]; s0 += v.x * xx.x; s1 += v.y * xx.y; }
    const double s = wave_sum(s0 + s1);
    if (lane == 0) y[row] = s;
  }
}
// The same product from the LOWER 128 x 128 tiles only (the inverse is symmetric bit for bit: k_dd_mirror): one workgroup per tile
// (I, J), J <= I, reads its 128 KB once and leaves two partial vectors -- the tile times x_J (for y_I) and, off the diagonal, its
// transpose times x_I (for y_J); k_dd_symv_sum adds each y_I's partials in a fixed order.  Half the bytes of k_dd_gemv: 105 MB
// instead of 210 at n_a = 5 000.  Used from 2 048 unknowns up (below that a solve is launch-bound and this form has one more launch).
// (Tried: lanes as 8 x 8 groups -- a row's product with x_J then takes three DPP steps among eight neighbours instead of a wavefront sum, but
// every load instruction touches eight rows: 8 870 -> 8 420 it/s at config 3.  The wavefront-per-row form stays.)
__global__ void __launch_bounds__(TB) k_dd_symv_tiles(Ctx c, int nap, const double *Mx, const double *x, double *rowpart, double *colpart, int gated) {
  if (gated) { const State *st = c.st; if (st->stalled || !st->run) return; }
  const int I = blockIdx.y, J = blockIdx.x;
  if (J > I) return;
  __shared__ double xI[DD_NB], xJ[DD_NB], cred[4][DD_NB];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t < DD_NB) { xI[t] = x[I * DD_NB + t]; xJ[t] = x[J * DD_NB + t]; }
  __syncthreads();
  const size_t tile = (size_t)I * (I + 1) / 2 + J;
  const double2 *base = reinterpret_cast<const double2 *>(Mx + (size_t)I * DD_NB * nap + (size_t)J * DD_NB) + lane;
  const double xj0 = xJ[2 * lane], xj1 = xJ[2 * lane + 1];
  double c0 = 0.0, c1 = 0.0;
  for (int r0 = 32 * wv; r0 < 32 * wv + 32; r0 += 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(r0 + u) * (nap / 2)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double xi = xI[r0 + u];
      c0 += v[u].x * xi; c1 += v[u].y * xi;
      const double s = wave_sum(v[u].x * xj0 + v[u].y * xj1);
      if (lane == 0) rowpart[tile * DD_NB + r0 + u] = s;
    }
  }
  if (I != J) {
    cred[wv][2 * lane] = c0; cred[wv][2 * lane + 1] = c1;
    __syncthreads();
    if (t < DD_NB) colpart[tile * DD_NB + t] = (cred[0][t] + cred[1][t]) + (cred[2][t] + cred[3][t]);
  }
}
// entry q of the product from the tiles' partials.  Term k < nb of y_I: the row partial of tile (I, k) for k <= I, the column partial of
// tile (k, I) above; eight loads in flight, added in the order of k (one load at a time this sum was a 13.8 us kernel at nb = 40)
__device__ __forceinline__ double dd_symv_entry(const double *rowpart, const double *colpart, int nb, int q) {
  const int I = q / DD_NB, r = q % DD_NB;
  double s = 0.0;
  for (int k0 = 0; k0 < nb; k0 += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      v[u] = k >= nb ? 0.0 : (k <= I ? rowpart[((size_t)I * (I + 1) / 2 + k) * DD_NB + r] : colpart[((size_t)k * (k + 1) / 2 + I) * DD_NB + r]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  return s;
}
// Check of a fresh inverse: u = a fixed vector of +-(1 .. 2); after y = S^-1 u and z = S0 y (the matrix as formed), err = max |z - u|
__device__ __forceinline__ double dd_probe_value(int a) { const unsigned h = (unsigned)a * 2654435761u; return ((h >> 9) & 1 ? -1.0 : 1.0) * (1.0 + (double)((h >> 12) & 1023) / 1024.0); }
__global__ void __launch_bounds__(TB) k_dd_probe_fill(DdCtx dd) {
  for (int a = blockIdx.x * TB + threadIdx.x; a < dd.nap; a += gridDim.x * TB) dd.rr[a] = a < dd.na ? dd_probe_value(a) : 0.0;
}
__global__ void __launch_bounds__(TB) k_dd_probe_err(DdCtx dd, const double *z, double *err) {
  __shared__ double red[16];
  double m = 0.0;
  for (int a = threadIdx.x; a < dd.na; a += TB) { const double d = fabs(z[a] - dd_probe_value(a)); m = fmax(m, d == d ? d : 1e300); }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *err = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// x~ = x~0 + v on the unknowns of the reduced system (the engine's eliminated variables are k_admm_finalize's); solve complete
__global__ void __launch_bounds__(TB) k_dd_finish(Ctx c, DdCtx dd) {
  State *st = c.st;
  if (st->stalled || !st->run) return;
  for (int j = blockIdx.x * TB + threadIdx.x; j < c.n; j += gridDim.x * TB) {
    const int cj = dd.vidx[j];
    if (cj == -1) continue;
    // (with the tiles' partials the product's entries are summed here, where they are used: no launch of their own)
    const int nb = dd.nap / DD_NB;
    auto entry = [&](int a) { return dd.rowpart ? dd_symv_entry(dd.rowpart, dd.colpart, nb, a) : dd.vv[a]; };
    double v;
    if (cj >= 0) v = entry(cj);
    else {
      const int b = -cj - 2;
      double s = c.init_r[j];
      for (int q = 0; q < DD_NBR; ++q) { const int a = dd.bnbr[b * DD_NBR + q]; if (a >= 0) s -= dd.bval[b * DD_NBR + q] * entry(a); }
      v = s / dd.bdiag[b];
    }
    c.va[j] = c.vx[j] + v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->iters[0] = 1; st->iters[1] = 0; st->done = 1;      // (a pivot that was not positive never gets here: dd_refresh drops the inverse)
  }
}
