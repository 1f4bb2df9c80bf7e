// Probe 2: what the in-launch exchanges of k_pcg_resident / k_pcg_blockres cost as a function of the number of
// participating workgroups P (one 512-thread workgroup per CU), and which poll form is cheapest.
//   vec : every workgroup publishes `rows` doubles (sc1 stores, drain, one flag), everybody needs all P x rows in LDS.
//         poll forms: 0 = four wavefronts, one atomic load per lane per look (round 2's form)
//                     1 = ceil(P/64) wavefronts, one buffer_load_dword sc1 per lane, only the flags still missing
//                     2 = ONE wavefront, up to four buffer_load_dword sc1 in flight per lane, only the flags still missing
//   scal: every workgroup publishes NG 8-byte granules {32 bits, tag}; everybody needs all of them.
//         poll forms: 0 = four wavefronts, NG atomic loads per lane one after the other (round 2's form)
//                     1 = ceil(P/64) wavefronts, NG buffer_load_dwordx2 sc1 in flight per lane, only what is still missing
//                     2 = ONE wavefront, slot by slot (up to four workgroups per lane), NG loads in flight per slot
//   line: two workgroups write neighbouring 8-byte words of ONE 128-byte line with sc1 stores while others read the
//         line with sc1 loads in between; counts lost writes (the reader of iteration k must see both words of iteration k).
// Every value is checked.  Polled words in uncached memory (hipDeviceMallocUncached) unless `cached` is set.
// build: hipcc -O3 --offload-arch=gfx950 tools/exchange_probe2.hip -o tools/exchange_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define SC1 16
#define VOL ((int)0x80000000)
#define TPB 512
#define LIMIT 2000000LL     // 20 ms of the 100 MHz clock

static __device__ inline __amdgpu_buffer_rsrc_t rsrc(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
static __device__ inline double expected(int j, int it) { return (double)j * 0.25 + (double)it * 3.0 + 1.0; }

struct Args {
  double *ubuf; unsigned *flags; unsigned long long *gran; unsigned *err;   // err[0] mismatches, err[1] timeouts, err[2] slow looks
  int P, rows, seg, iters, form, ng, work;
};

// ---- vector exchange ----
__global__ __launch_bounds__(TPB) void k_vec(Args a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int s_fail;
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, P = a.P, npad = a.P * a.seg;
  if (t == 0) s_fail = 0;
  __syncthreads();
  unsigned bad = 0;
  double acc = 0.0;
  for (int it = 0; it < a.iters; ++it) {
    const unsigned tag = (unsigned)it + 1u;
    const int par = it & 1;
    const __amdgpu_buffer_rsrc_t rs = rsrc(a.ubuf + (size_t)par * npad, (size_t)npad * 8);
    if (wv == 0) {
      if (lane < a.rows) {
        const double v = expected(g * a.rows + lane, it);
        u32x2 d; d.x = (unsigned)__double2loint(v); d.y = (unsigned)__double2hiint(v);
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, (g * a.seg + lane) * 8, 0, SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(a.flags + (size_t)g * 32, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int npw = a.form == 0 ? 4 : (a.form == 1 ? (P + 63) / 64 : 1);
    if (wv < npw) {
      const long long t0 = wall_clock64();
      unsigned rounds = 0;
      if (a.form == 0) {
        const int o = t;
        while (true) {
          bool ok = true;
          if (o < P) { const unsigned f = __hip_atomic_load(a.flags + (size_t)o * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = (int)(f - tag) >= 0; }
          if (__all(ok)) break;
          if ((++rounds & 15u) == 0 && wall_clock64() - t0 > LIMIT) { s_fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else {
        const __amdgpu_buffer_rsrc_t rf = rsrc(a.flags, (size_t)P * 128);
        unsigned pend = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int o = t + npw * 64 * q; if (o < P && o != g) pend |= 1u << q; }
        while (true) {
          unsigned f[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) if (pend & (1u << q)) f[q] = __builtin_amdgcn_raw_buffer_load_b32(rf, (t + npw * 64 * q) * 128, 0, SC1 | VOL);
#pragma unroll
          for (int q = 0; q < 4; ++q) if ((pend & (1u << q)) && (int)(f[q] - tag) >= 0) pend &= ~(1u << q);
          if (__all(pend == 0)) break;
          asm volatile("" ::: "memory");
          if ((++rounds & 15u) == 0 && wall_clock64() - t0 > LIMIT) { s_fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    if (s_fail) break;
    const int half = npad >> 1;
    for (int i2 = t; i2 < half; i2 += TPB) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i2 * 16, 0, SC1);
      lds[2 * i2] = __hiloint2double((int)v.y, (int)v.x); lds[2 * i2 + 1] = __hiloint2double((int)v.w, (int)v.z);
    }
    __syncthreads();
    for (int j = t; j < P * a.rows; j += TPB) { const int o = j / a.rows, r = j % a.rows; bad += lds[o * a.seg + r] != expected(j, it); }
    if (a.work) { double s = 0.0; for (int k = 0; k < 16; ++k) s += lds[(t * 16 + k * 37) % (P * a.seg)] * 0.5; acc += s; }
    __syncthreads();
  }
  if (bad) atomicAdd(a.err, bad);
  if (s_fail && t == 0) atomicAdd(a.err + 1, 1u);
  if (acc == 12345.678) a.err[3] = 1;
}

// ---- scalar exchange (granules) ----
template <int NG>
__global__ __launch_bounds__(TPB) void k_scal(Args a) {
  __shared__ unsigned sv[256 * NG];
  __shared__ int s_fail;
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, P = a.P;
  if (t == 0) s_fail = 0;
  __syncthreads();
  unsigned bad = 0;
  for (int it = 0; it < a.iters; ++it) {
    const unsigned tag = (unsigned)it + 1u;
    const int par = it & 1;
    unsigned long long *gb = a.gran + (size_t)par * P * 16;
    if (wv == 0 && lane < NG) __hip_atomic_store(gb + (size_t)g * 16 + lane, ((unsigned long long)tag << 32) | (unsigned)(g * 64 + lane * 7 + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int npw = a.form == 0 ? 4 : (a.form == 1 ? (P + 63) / 64 : 1);
    if (wv < npw) {
      const long long t0 = wall_clock64();
      unsigned rounds = 0;
      if (a.form == 0) {
        const int o = t;
        unsigned pend = o < P ? (1u << NG) - 1u : 0u;
        while (true) {
#pragma unroll
          for (int q = 0; q < NG; ++q)
            if (pend & (1u << q)) {
              const unsigned long long v = __hip_atomic_load(gb + (size_t)o * 16 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((unsigned)(v >> 32) == tag) { sv[o * NG + q] = (unsigned)v; pend &= ~(1u << q); }
            }
          if (__all(pend == 0)) break;
          asm volatile("" ::: "memory");
          if ((++rounds & 15u) == 0 && wall_clock64() - t0 > LIMIT) { s_fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else {
        const __amdgpu_buffer_rsrc_t rg = rsrc(gb, (size_t)P * 128);
        unsigned pend[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int o = t + npw * 64 * q; pend[q] = o < P ? (1u << NG) - 1u : 0u; }
        while (true) {
          bool all = true;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (npw * 64 * q >= P) break;
            const int o = t + npw * 64 * q;
            u32x2 v[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) if (pend[q] & (1u << k)) v[k] = __builtin_amdgcn_raw_buffer_load_b64(rg, (o * 16 + k) * 8, 0, SC1 | VOL);
#pragma unroll
            for (int k = 0; k < NG; ++k) if ((pend[q] & (1u << k)) && v[k].y == tag) { sv[o * NG + k] = v[k].x; pend[q] &= ~(1u << k); }
            all = all && pend[q] == 0;
          }
          if (__all(all)) break;
          asm volatile("" ::: "memory");
          if ((++rounds & 15u) == 0 && wall_clock64() - t0 > LIMIT) { s_fail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
    if (s_fail) break;
    for (int j = t; j < P * NG; j += TPB) { const int o = j / NG, k = j % NG; bad += sv[j] != (unsigned)(o * 64 + k * 7 + it); }
    __syncthreads();
  }
  if (bad) atomicAdd(a.err, bad);
  if (s_fail && t == 0) atomicAdd(a.err + 1, 1u);
}

// ---- two writers per 128-byte line ----
// Workgroups 2k and 2k+1 (different XCDs under round-robin placement) own the words 0 and 1 of line k of the buffer of the
// iteration's parity.  Per iteration: write-through store of {it} into the own word, drain, flag; wait for every flag;
// every workgroup then reads EVERY line (sc1) and checks both words.  `readers`: 1 = as said (the lines are resident in
// the readers' L2s when the next write of the same parity arrives), 0 = only the two owners read their line.
__global__ __launch_bounds__(TPB) void k_line(Args a, int readers) {
  __shared__ int s_fail;
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, P = a.P;
  if (t == 0) s_fail = 0;
  __syncthreads();
  unsigned lost = 0;
  unsigned long long *base = reinterpret_cast<unsigned long long *>(a.ubuf);
  for (int it = 0; it < a.iters; ++it) {
    const unsigned tag = (unsigned)it + 1u;
    const int par = it & 1;
    unsigned long long *lines = base + (size_t)par * (P / 2) * 16;
    if (t == 0) {
      __hip_atomic_store(lines + (size_t)(g / 2) * 16 + (g & 1), (unsigned long long)tag * 1000003ull + (unsigned)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(a.flags + (size_t)g * 32, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wv == 0) {
      const __amdgpu_buffer_rsrc_t rf = rsrc(a.flags, (size_t)P * 128);
      unsigned pend = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int o = lane + 64 * q; if (o < P) pend |= 1u << q; }
      const long long t0 = wall_clock64();
      unsigned rounds = 0;
      while (true) {
        unsigned f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) if (pend & (1u << q)) f[q] = __builtin_amdgcn_raw_buffer_load_b32(rf, (lane + 64 * q) * 128, 0, SC1 | VOL);
#pragma unroll
        for (int q = 0; q < 4; ++q) if ((pend & (1u << q)) && (int)(f[q] - tag) >= 0) pend &= ~(1u << q);
        if (__all(pend == 0)) break;
        asm volatile("" ::: "memory");
        if ((++rounds & 15u) == 0 && wall_clock64() - t0 > LIMIT) { s_fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_fail) break;
    for (int k = t; k < P / 2; k += TPB) {
      if (!readers && k != g / 2) continue;
      const unsigned long long w0 = __hip_atomic_load(lines + (size_t)k * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long w1 = __hip_atomic_load(lines + (size_t)k * 16 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lost += w0 != (unsigned long long)tag * 1000003ull + (unsigned)(2 * k);
      lost += w1 != (unsigned long long)tag * 1000003ull + (unsigned)(2 * k + 1);
    }
    __syncthreads();
  }
  if (lost) atomicAdd(a.err, lost);
  if (s_fail && t == 0) atomicAdd(a.err + 1, 1u);
}

static void *alloc(size_t bytes, bool uncached) {
  void *p = nullptr;
  if (uncached) CK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)); else CK(hipMalloc(&p, bytes));
  CK(hipMemset(p, 0, bytes));
  return p;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const char *what = argc > 2 ? argv[2] : "all";
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned *err = (unsigned *)alloc(64, false);
  const int Ps[] = {4, 8, 16, 32, 64, 128, 256};
  CK(hipFuncSetAttribute((const void *)k_vec, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  auto run = [&](auto launch, const char *label) {
    float best = 1e30f; unsigned h[4] = {0, 0, 0, 0}, tot[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(err, 0, 64));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost)); tot[0] += h[0]; tot[1] += h[1];
    }
    printf("%s: %.3f us/exchange  wrong=%u timeouts=%u\n", label, best * 1e3 / iters, tot[0], tot[1]);
    fflush(stdout);
    return tot[1] == 0;
  };
  if (!strcmp(what, "all") || !strcmp(what, "vec")) {
    for (int ubuf_uc = 0; ubuf_uc < 2; ++ubuf_uc)
      for (int rows : {16, 45})
        for (int P : Ps)
          for (int form = 0; form < 3; ++form) {
            if (form == 1 && P <= 64) continue;                  // (same as form 2 there)
            if (ubuf_uc && (form != 2 || (P != 16 && P != 256))) continue;
            const int seg = ((rows + 3 + 15) / 16) * 16, npad = P * seg;
            double *ubuf = (double *)alloc((size_t)2 * npad * 8, ubuf_uc);
            unsigned *flags = (unsigned *)alloc((size_t)256 * 128, true);
            Args a{ubuf, flags, nullptr, err, P, rows, seg, iters, form, 0, 1};
            char lab[160]; snprintf(lab, sizeof lab, "vec  P=%3d rows=%2d (vector %5.1f KB, %s) poll form %d", P, rows, npad * 8 / 1024.0, ubuf_uc ? "uncached" : "hipMalloc", form);
            const bool ok = run([&] { hipLaunchKernelGGL(k_vec, dim3(P), dim3(TPB), (size_t)npad * 8, 0, a); }, lab);
            CK(hipFree(ubuf)); CK(hipFree(flags));
            if (!ok) { printf("timeout: stopping\n"); return 2; }
          }
  }
  if (!strcmp(what, "all") || !strcmp(what, "scal")) {
    for (int ng : {6, 8})
      for (int P : Ps)
        for (int form = 0; form < 3; ++form) {
          if (form == 1 && P <= 64) continue;
          unsigned long long *gran = (unsigned long long *)alloc((size_t)2 * 256 * 128, true);
          Args a{nullptr, nullptr, gran, err, P, 0, 0, iters, form, ng, 0};
          char lab[160]; snprintf(lab, sizeof lab, "scal P=%3d granules=%d poll form %d", P, ng, form);
          const bool ok = run([&] { if (ng == 6) hipLaunchKernelGGL(k_scal<6>, dim3(P), dim3(TPB), 0, 0, a); else hipLaunchKernelGGL(k_scal<8>, dim3(P), dim3(TPB), 0, 0, a); }, lab);
          CK(hipFree(gran));
          if (!ok) { printf("timeout: stopping\n"); return 2; }
        }
  }
  if (!strcmp(what, "all") || !strcmp(what, "line")) {
    const int it2 = argc > 3 ? atoi(argv[3]) : 200000;
    for (int uc = 0; uc < 2; ++uc)
      for (int readers = 1; readers >= 0; --readers) {
        const int P = 256;
        double *ubuf = (double *)alloc((size_t)2 * (P / 2) * 128, uc);
        unsigned *flags = (unsigned *)alloc((size_t)256 * 128, true);
        Args a{ubuf, flags, nullptr, err, P, 0, 0, it2, 2, 0, 0};
        CK(hipMemset(err, 0, 64));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_line, dim3(P), dim3(TPB), 0, 0, a, readers); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[4]; CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
        printf("line two writers per 128-byte line, %s, %s: %d iterations x 128 lines, %.2f us each: wrong words seen %u, timeouts %u\n", uc ? "uncached" : "hipMalloc",
               readers ? "every workgroup reads every line" : "only the owners read their line", it2, ms * 1e3 / it2, h[0], h[1]);
        fflush(stdout);
        CK(hipFree(ubuf)); CK(hipFree(flags));
        if (h[1]) { printf("timeout: stopping\n"); return 2; }
      }
  }
  return 0;
}
