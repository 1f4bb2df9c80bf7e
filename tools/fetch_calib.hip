// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// patterns of the OSQP engine kernels (MI355X_MICROARCH.md, HBM section: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Each kernel moves a known number of bytes; run once under
//     rocprofv3 --pmc FETCH_SIZE  --output-format csv -d <dir> -- tools/fetch_calib
//     rocprofv3 --pmc WRITE_SIZE  --output-format csv -d <dir> -- tools/fetch_calib
// and compare the counter of every kernel with the "bytes" line this program prints.
//   k_stream4 / k_stream8 / k_stream16 : coalesced streaming reads, 4 / 8 / 16 B per lane
//   k_gather8  : random 8-byte gathers from a 64 MiB table (the `u[col]` gather of the SpMV kernels)
//   k_gather32 : random 32-byte record gathers (the G4 {r,w,s,Minv} records of k_cg_A)
//   k_store8   : coalesced 8-byte stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_stream4(const float *a, size_t n, float *out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_stream8(const double *a, size_t n, double *out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_stream16(const double2 *a, size_t n, double *out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}
__global__ void k_gather8(const double *tab, const int *idx, size_t n, double *out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += tab[idx[i]];
  if (s == 12345.678) out[0] = s;
}
struct __attribute__((aligned(32))) G4 { double r, w, s, m; };
__global__ void k_gather32(const G4 *tab, const int *idx, size_t n, double *out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const G4 g = tab[idx[i]]; s += g.r + g.w + g.s + g.m; }
  if (s == 12345.678) out[0] = s;
}
__global__ void k_store8(double *a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}

int main() {
  const size_t BYTES = 64ull << 20;                 // 64 MiB per stream / table: well past the 8 x 4 MiB of L2
  const size_t NG = 4ull << 20;                     // gathers per launch
  double *buf = nullptr, *out = nullptr; G4 *tab32 = nullptr; int *idx8 = nullptr, *idx32 = nullptr;
  CHK(hipMalloc(&buf, BYTES)); CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&tab32, BYTES));
  CHK(hipMalloc(&idx8, NG * sizeof(int))); CHK(hipMalloc(&idx32, NG * sizeof(int)));
  CHK(hipMemset(buf, 0, BYTES)); CHK(hipMemset(tab32, 0, BYTES));
  std::vector<int> h8(NG), h32(NG);
  unsigned long long lcg = 88172645463325252ULL;
  for (size_t i = 0; i < NG; i++) {
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; h8[i] = (int)((lcg >> 24) % (BYTES / 8));
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; h32[i] = (int)((lcg >> 24) % (BYTES / 32));
  }
  CHK(hipMemcpy(idx8, h8.data(), NG * sizeof(int), hipMemcpyHostToDevice));
  CHK(hipMemcpy(idx32, h32.data(), NG * sizeof(int), hipMemcpyHostToDevice));
  const dim3 g(2048), b(256);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_stream4, g, b, 0, 0, (const float *)buf, BYTES / 4, (float *)out);
    hipLaunchKernelGGL(k_stream8, g, b, 0, 0, (const double *)buf, BYTES / 8, out);
    hipLaunchKernelGGL(k_stream16, g, b, 0, 0, (const double2 *)buf, BYTES / 16, out);
    hipLaunchKernelGGL(k_gather8, g, b, 0, 0, (const double *)buf, (const int *)idx8, NG, out);
    hipLaunchKernelGGL(k_gather32, g, b, 0, 0, (const G4 *)tab32, (const int *)idx32, NG, out);
    hipLaunchKernelGGL(k_store8, g, b, 0, 0, buf, BYTES / 8);
    CHK(hipDeviceSynchronize());
  }
  printf("bytes k_stream4 read %zu\nbytes k_stream8 read %zu\nbytes k_stream16 read %zu\n", BYTES, BYTES, BYTES);
  printf("bytes k_gather8 read %zu index stream + %zu gathered (8 B each; %zu B if a 32-B sector, %zu B if a 64-B half line, %zu B if a 128-B line is fetched per gather)\n",
         NG * 4, NG * 8, NG * 32, NG * 64, NG * 128);
  printf("bytes k_gather32 read %zu index stream + %zu gathered (32 B each; %zu B at 64 B, %zu B at 128 B per gather)\n", NG * 4, NG * 32, NG * 64, NG * 128);
  printf("bytes k_store8 written %zu\n", BYTES);
  return 0;
}
