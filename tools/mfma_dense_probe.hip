// mfma_dense_probe.hip -- evidence for the "MFMA on the dense diagonal blocks of P" question (BASELINE config 5:
// 400 symmetric blocks of 125 x 125, ONE right-hand side per PCG iteration).
//
// Four kernels compute y_b = P_b x_b for all blocks, one workgroup (4 wavefronts) per block:
//   k_valu      the product path of engine.hip (dense_block_mv): pitch 125, 8-byte loads, VALU column walk
//   k_valu_wide the same walk on blocks padded to pitch 128 with 16-byte loads (one 1-KiB row per wave instruction)
//   k_mfma      v_mfma_f64_16x16x4_f64: A = a 16 x 4 tile of P_b, B = x_b broadcast over the 16 columns (a GEMV uses
//               1/16 of the instruction: 15 of the 16 result columns are copies), pitch 128
//   k_stream    reads the same bytes with 16-byte loads and only sums them: the bandwidth ceiling of this launch shape
// and print time per launch and GB/s over the bytes of the stored blocks.  Run plain for the timings, and under
//   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... / --pmc FETCH_SIZE
// for the counters (tools/profile_mfma.sh).  All results are checked against k_valu.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define TB 256
#define BMAX 128
typedef double double4v __attribute__((ext_vector_type(4)));

// ---- product path (copy of dense_block_mv's algorithm, pitch = b) ----
__global__ void __launch_bounds__(TB) k_valu(const double *P, const double *x, double *y, int b, int nblk) {
  __shared__ double scratch[5 * BMAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int db = blockIdx.x; db < nblk; db += gridDim.x) {
    const double *dv = P + (size_t)db * b * b;
    const bool h0 = lane < b, h1 = lane + 64 < b;
    double v0[BMAX / 4], v1[BMAX / 4];
#pragma unroll
    for (int q = 0; q < BMAX / 4; ++q) {
      const int j = w + 4 * q;
      const double *row = dv + (size_t)j * b;
      v0[q] = (j < b && h0) ? row[lane] : 0.0;
      v1[q] = (j < b && h1) ? row[lane + 64] : 0.0;
    }
    double *xl = scratch + 4 * BMAX;
    if ((int)threadIdx.x < BMAX) xl[threadIdx.x] = (int)threadIdx.x < b ? x[(size_t)db * BMAX + threadIdx.x] : 0.0;
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < BMAX / 4; ++q) { const double u = xl[min(w + 4 * q, BMAX - 1)]; a0 += v0[q] * u; a1 += v1[q] * u; }
    scratch[w * BMAX + lane] = a0; scratch[w * BMAX + 64 + lane] = a1;
    __syncthreads();
    const int r = threadIdx.x;
    if (r < b) y[(size_t)db * BMAX + r] = (scratch[r] + scratch[BMAX + r]) + (scratch[2 * BMAX + r] + scratch[3 * BMAX + r]);
    __syncthreads();
  }
}

// ---- padded pitch 128, 16-byte loads: lane owns columns 2*lane, 2*lane+1 ----
__global__ void __launch_bounds__(TB) k_valu_wide(const double *P, const double *x, double *y, int nblk) {
  __shared__ double scratch[5 * BMAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int db = blockIdx.x; db < nblk; db += gridDim.x) {
    const double2 *dv = reinterpret_cast<const double2 *>(P + (size_t)db * BMAX * BMAX);
    double2 v[BMAX / 4];
#pragma unroll
    for (int q = 0; q < BMAX / 4; ++q) v[q] = dv[(size_t)(w + 4 * q) * (BMAX / 2) + lane];
    double *xl = scratch + 4 * BMAX;
    if ((int)threadIdx.x < BMAX) xl[threadIdx.x] = x[(size_t)db * BMAX + threadIdx.x];
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < BMAX / 4; ++q) { const double u = xl[w + 4 * q]; a0 += v[q].x * u; a1 += v[q].y * u; }
    scratch[w * BMAX + 2 * lane] = a0; scratch[w * BMAX + 2 * lane + 1] = a1;
    __syncthreads();
    const int r = threadIdx.x;
    if (r < BMAX) y[(size_t)db * BMAX + r] = (scratch[r] + scratch[BMAX + r]) + (scratch[2 * BMAX + r] + scratch[3 * BMAX + r]);
    __syncthreads();
  }
}

// ---- v_mfma_f64_16x16x4_f64: wave w owns rows [32w, 32w+32) = two 16-row tiles; symmetric P lets the A tile
//      A[i][k] = P[row0+i][k0+k] be read as P[k0+k][row0+i]: 4 rows x 16 contiguous doubles per wave load ----
__global__ void __launch_bounds__(TB) k_mfma(const double *P, const double *x, double *y, int nblk) {
  __shared__ double xl[BMAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  for (int db = blockIdx.x; db < nblk; db += gridDim.x) {
    const double *dv = P + (size_t)db * BMAX * BMAX;
    double a[2][BMAX / 4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kt = 0; kt < BMAX / 4; ++kt) a[t][kt] = dv[(size_t)(4 * kt + lk) * BMAX + 32 * w + 16 * t + li];
    if ((int)threadIdx.x < BMAX) xl[threadIdx.x] = x[(size_t)db * BMAX + threadIdx.x];
    __syncthreads();
    double4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int kt = 0; kt < BMAX / 4; ++kt) {
      const double bv = xl[4 * kt + lk];                  // B[k][col] = x[k0 + k] for every column
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][kt], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][kt], bv, acc1, 0, 0, 0);
    }
    if (li == 0) {     // column 0 of D: rows (lane >> 4) + 4 * reg
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[(size_t)db * BMAX + 32 * w + lk + 4 * r] = acc0[r];
        y[(size_t)db * BMAX + 32 * w + 16 + lk + 4 * r] = acc1[r];
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(TB) k_stream(const double *P, double *y, int nblk) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int db = blockIdx.x; db < nblk; db += gridDim.x) {
    const double2 *dv = reinterpret_cast<const double2 *>(P + (size_t)db * BMAX * BMAX);
    double2 v[BMAX / 4];
#pragma unroll
    for (int q = 0; q < BMAX / 4; ++q) v[q] = dv[(size_t)(w + 4 * q) * (BMAX / 2) + lane];
    double s = 0;
#pragma unroll
    for (int q = 0; q < BMAX / 4; ++q) s += v[q].x + v[q].y;
    if (s == 12345.678) y[db] = s;
  }
}

int main(int argc, char **argv) {
  const int nblk = 400, b = 125, reps = argc > 1 ? atoi(argv[1]) : 200;
  std::vector<double> hP((size_t)nblk * b * b), hPp((size_t)nblk * BMAX * BMAX, 0.0), hx((size_t)nblk * BMAX, 0.0);
  unsigned long long lcg = 88172645463325252ULL;
  auto rnd = [&]() { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((lcg >> 11) & 0xFFFFF) / 1048576.0 - 0.5; };
  for (int k = 0; k < nblk; k++) {
    for (int i = 0; i < b; i++) for (int j = i; j < b; j++) { const double v = rnd() + (i == j ? 8.0 : 0.0); hP[(size_t)k * b * b + i * b + j] = hP[(size_t)k * b * b + j * b + i] = v; }
    for (int i = 0; i < b; i++) for (int j = 0; j < b; j++) hPp[(size_t)k * BMAX * BMAX + i * BMAX + j] = hP[(size_t)k * b * b + i * b + j];
    for (int i = 0; i < b; i++) hx[(size_t)k * BMAX + i] = rnd();
  }
  double *dP, *dPp, *dx, *dy[4];
  CHK(hipMalloc(&dP, hP.size() * 8)); CHK(hipMalloc(&dPp, hPp.size() * 8)); CHK(hipMalloc(&dx, hx.size() * 8));
  for (int k = 0; k < 4; k++) { CHK(hipMalloc(&dy[k], hx.size() * 8)); CHK(hipMemset(dy[k], 0, hx.size() * 8)); }
  CHK(hipMemcpy(dP, hP.data(), hP.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dPp, hPp.data(), hPp.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const char *names[4] = {"k_valu (pitch 125, 8 B loads)", "k_valu_wide (pitch 128, 16 B loads)", "k_mfma (f64 16x16x4, pitch 128)", "k_stream (read only, 16 B loads)"};
  const double bytes[4] = {(double)hP.size() * 8, (double)hPp.size() * 8, (double)hPp.size() * 8, (double)hPp.size() * 8};
  for (int k = 0; k < 4; k++) {
    auto launch = [&]() {
      if (k == 0) hipLaunchKernelGGL(k_valu, dim3(nblk), dim3(TB), 0, 0, (const double *)dP, (const double *)dx, dy[0], b, nblk);
      if (k == 1) hipLaunchKernelGGL(k_valu_wide, dim3(nblk), dim3(TB), 0, 0, (const double *)dPp, (const double *)dx, dy[1], nblk);
      if (k == 2) hipLaunchKernelGGL(k_mfma, dim3(nblk), dim3(TB), 0, 0, (const double *)dPp, (const double *)dx, dy[2], nblk);
      if (k == 3) hipLaunchKernelGGL(k_stream, dim3(nblk), dim3(TB), 0, 0, (const double *)dPp, dy[3], nblk);
    };
    for (int r = 0; r < 5; r++) launch();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) launch();
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    printf("%-40s %7.2f us per launch  %7.1f GB/s on %.1f MB of blocks\n", names[k], us, bytes[k] / us / 1e3, bytes[k] / 1e6);
  }
  std::vector<double> h0(hx.size()), h1(hx.size()), h2(hx.size());
  CHK(hipMemcpy(h0.data(), dy[0], h0.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(h1.data(), dy[1], h1.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(h2.data(), dy[2], h2.size() * 8, hipMemcpyDeviceToHost));
  double e1m = 0, e2m = 0, ref = 0;
  for (int k = 0; k < nblk; k++) for (int i = 0; i < b; i++) {
    const size_t q = (size_t)k * BMAX + i;
    e1m = fmax(e1m, fabs(h1[q] - h0[q])); e2m = fmax(e2m, fabs(h2[q] - h0[q])); ref = fmax(ref, fabs(h0[q]));
  }
  printf("max |y_wide - y_valu| = %.3e, max |y_mfma - y_valu| = %.3e (max |y| = %.3e)\n", e1m, e2m, ref);
  return (e1m < 1e-12 * ref && e2m < 1e-12 * ref) ? 0 : 5;
}
