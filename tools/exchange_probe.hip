// Probe: what does one all-to-all vector exchange per iteration cost INSIDE a persistent launch on MI355X?
// 256 workgroups (one per CU) each publish `per` doubles per iteration; every workgroup needs all of them in LDS
// before its next step (the pattern of a CG iteration whose operator rows live in registers).
//   mode 0: no exchange (loop + barriers + the emulated row work): the floor
//   mode 1: data-tagged granules {low word, tag, high word, tag} (two 8-byte granules per double), sc1 stores, sc1 sweep until every tag matches
//   mode 2: sc1 payload (16-byte stores of two doubles), drain, one flag per workgroup, poll flags, sc1 payload loads
//   mode 4: mode 2 without the drain (payload words self-tagged in their two low mantissa bits, flag stored right behind them)
//   mode 3: the tag in the two low mantissa bits of every double (8-byte sc1 stores, sc1 sweep until every word carries it)
// Every word is checked against its expected value; `uneven` makes some workgroups late on some iterations.
// build: hipcc -O3 --offload-arch=gfx950 tools/exchange_probe.hip -o tools/exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define AUX_SC1 16

static __device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

static __device__ inline double expected(int j, int it) { return (double)j * 0.25 + (double)it * 3.0 + 1.0; }

struct Args {
    void *buf;                 // mode 1: 2 x G granules of 16 B; mode 2: 2 x G doubles (G even)
    unsigned *flags;           // mode 2: 256 words (64-B apart)
    int G, per, iters, mode, uneven, work, fstride;
    const unsigned short *cols; // emulated operator columns: 512 x E entries per workgroup (shared by all)
    double *out;
    unsigned *err;             // [0] mismatches, [1] timeouts
};

constexpr int TPB = 512;
constexpr int E = 16;          // emulated operator entries per thread
constexpr int KMAX = 24;       // granules per thread per sweep (G <= 512 * KMAX)

template <int MODE>
__global__ __launch_bounds__(TPB) void k_probe(Args a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];   // G doubles + TPB partials
    const int g = blockIdx.x, t = threadIdx.x, G = a.G;
    double *red = lds + G;
    volatile int &s_fail = *(volatile int *)(lds + G + TPB);
    if (t == 0) s_fail = 0;
    unsigned short col[E];
    double val[E];
#pragma unroll
    for (int k = 0; k < E; ++k) { col[k] = a.cols[t * E + k]; val[k] = 1.0 / (double)(1 + ((t * E + k) & 15)); }
    double acc = 0.0;
    unsigned bad = 0;
    const long long t_limit = 2000000;   // 20 ms of the 100 MHz clock per wait
    __syncthreads();

    for (int it = 0; it < a.iters; ++it) {
        const unsigned tag = (unsigned)it + 1u;
        const int par = it & 1;
        if (a.uneven && ((g * 7 + it) % 13) == 0) {           // a late workgroup
            long long t0 = wall_clock64();
            while (wall_clock64() - t0 < 150) { }
        }
        if (MODE == 1) {
            __amdgpu_buffer_rsrc_t rs = make_rsrc((const char *)a.buf + (size_t)par * G * 16, (unsigned)G * 16u);
            if (t < a.per) {
                int j = g * a.per + t;
                double v = expected(j, it);
                u32x4 w;
                w.x = (unsigned)__double2loint(v); w.y = tag; w.z = (unsigned)__double2hiint(v); w.w = tag;   // two 8-byte granules: a 16-byte store may land in halves
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, j * 16, 0, AUX_SC1);
            }
            // sweep: thread t takes granules t, t+512, ...
            u32x4 r[KMAX];
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (t + k * TPB < G) pending |= 1u << k;
            long long t0 = wall_clock64();
            while (true) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (pending & (1u << k)) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t + k * TPB) * 16, 0, AUX_SC1);
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if ((pending & (1u << k)) && r[k].y == tag && r[k].w == tag) {
                        lds[t + k * TPB] = __hiloint2double((int)r[k].z, (int)r[k].x);
                        pending &= ~(1u << k);
                    }
                if (!pending) break;
                asm volatile("" ::: "memory");
                if (wall_clock64() - t0 > t_limit) { s_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        } else if (MODE == 2) {
            __amdgpu_buffer_rsrc_t rs = make_rsrc((const char *)a.buf + (size_t)par * G * 8, (unsigned)G * 8u);
            if (t < a.per / 2) {
                int j = g * a.per + 2 * t;
                double v0 = expected(j, it), v1 = expected(j + 1, it);
                u32x4 w;
                w.x = (unsigned)__double2loint(v0); w.y = (unsigned)__double2hiint(v0);
                w.z = (unsigned)__double2loint(v1); w.w = (unsigned)__double2hiint(v1);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, j * 8, 0, AUX_SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(a.flags + g * a.fstride, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < 64) {                                    // one wave polls the 256 flags
                long long t0 = wall_clock64();
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned f = __hip_atomic_load(a.flags + (t + 64 * k) * a.fstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (f >= tag);
                    }
                    if (__all(ok)) break;
                    if (wall_clock64() - t0 > t_limit) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __syncthreads();
            // payload: G doubles, 16 B per lane
            u32x4 r[KMAX / 2];
#pragma unroll
            for (int k = 0; k < KMAX / 2; ++k)
                if (t + k * TPB < G / 2) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t + k * TPB) * 16, 0, AUX_SC1);
#pragma unroll
            for (int k = 0; k < KMAX / 2; ++k)
                if (t + k * TPB < G / 2) {
                    int j2 = t + k * TPB;
                    lds[2 * j2] = __hiloint2double((int)r[k].y, (int)r[k].x);
                    lds[2 * j2 + 1] = __hiloint2double((int)r[k].w, (int)r[k].z);
                }
        } else if (MODE == 5) {   // mode 2 with the 256 flags packed into 1 KB: one 16-byte poll load per lane
            __amdgpu_buffer_rsrc_t rs = make_rsrc((const char *)a.buf + (size_t)par * G * 8, (unsigned)G * 8u);
            if (t < a.per / 2) {
                int j = g * a.per + 2 * t;
                double v0 = expected(j, it), v1 = expected(j + 1, it);
                u32x4 w;
                w.x = (unsigned)__double2loint(v0); w.y = (unsigned)__double2hiint(v0);
                w.z = (unsigned)__double2loint(v1); w.w = (unsigned)__double2hiint(v1);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, j * 8, 0, AUX_SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(a.flags + g, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < 64) {                                    // one wave polls the 256 flags
                long long t0 = wall_clock64();
                while (true) {
                    __amdgpu_buffer_rsrc_t fr = make_rsrc(a.flags, 1024u);
                    u32x4 f4 = __builtin_amdgcn_raw_buffer_load_b128(fr, t * 16, 0, AUX_SC1);
                    bool ok = f4.x >= tag && f4.y >= tag && f4.z >= tag && f4.w >= tag;
                    asm volatile("" ::: "memory");
                    if (__all(ok)) break;
                    if (wall_clock64() - t0 > t_limit) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __syncthreads();
            // payload: G doubles, 16 B per lane
            u32x4 r[KMAX / 2];
#pragma unroll
            for (int k = 0; k < KMAX / 2; ++k)
                if (t + k * TPB < G / 2) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t + k * TPB) * 16, 0, AUX_SC1);
#pragma unroll
            for (int k = 0; k < KMAX / 2; ++k)
                if (t + k * TPB < G / 2) {
                    int j2 = t + k * TPB;
                    lds[2 * j2] = __hiloint2double((int)r[k].y, (int)r[k].x);
                    lds[2 * j2 + 1] = __hiloint2double((int)r[k].w, (int)r[k].z);
                }
        } else if (MODE == 3) {
            // the tag rides in the two low mantissa bits of every double: no flag, no drain, 8 bytes per value
            __amdgpu_buffer_rsrc_t rs = make_rsrc((const char *)a.buf + (size_t)par * G * 8, (unsigned)G * 8u);
            const unsigned tg = tag & 3u;
            if (t < a.per) {
                int j = g * a.per + t;
                double v = expected(j, it);
                u32x2 w;
                w.x = ((unsigned)__double2loint(v) & ~3u) | tg; w.y = (unsigned)__double2hiint(v);
                __builtin_amdgcn_raw_buffer_store_b64(w, rs, j * 8, 0, AUX_SC1);
            }
            u32x4 r[KMAX / 2];
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < KMAX / 2; ++k) if (t + k * TPB < G / 2) pending |= 1u << k;
            long long t0 = wall_clock64();
            while (true) {
#pragma unroll
                for (int k = 0; k < KMAX / 2; ++k)
                    if (pending & (1u << k)) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t + k * TPB) * 16, 0, AUX_SC1);
#pragma unroll
                for (int k = 0; k < KMAX / 2; ++k)
                    if ((pending & (1u << k)) && (r[k].x & 3u) == tg && (r[k].z & 3u) == tg) {
                        int j2 = t + k * TPB;
                        lds[2 * j2] = __hiloint2double((int)r[k].y, (int)(r[k].x & ~3u));
                        lds[2 * j2 + 1] = __hiloint2double((int)r[k].w, (int)(r[k].z & ~3u));
                        pending &= ~(1u << k);
                    }
                if (!pending) break;
                asm volatile("" ::: "memory");
                if (wall_clock64() - t0 > t_limit) { s_fail = 1; break; }
            }
        } else if (MODE == 4) {
            // mode 2 without the drain: payload words carry the tag in their two low mantissa bits, the flag is stored
            // right behind them; the sweep after the flags re-reads the few words that have not landed yet
            __amdgpu_buffer_rsrc_t rs = make_rsrc((const char *)a.buf + (size_t)par * G * 8, (unsigned)G * 8u);
            const unsigned tg = tag & 3u;
            if (t < a.per) {
                int j = g * a.per + t;
                double v = expected(j, it);
                u32x2 w;
                w.x = ((unsigned)__double2loint(v) & ~3u) | tg; w.y = (unsigned)__double2hiint(v);
                __builtin_amdgcn_raw_buffer_store_b64(w, rs, j * 8, 0, AUX_SC1);
            }
            if (t == 0) __hip_atomic_store(a.flags + g * 16, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < 64) {
                long long t0 = wall_clock64();
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned f = __hip_atomic_load(a.flags + (t + 64 * k) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (f >= tag);
                    }
                    if (__all(ok)) break;
                    if (wall_clock64() - t0 > t_limit) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __syncthreads();
            u32x4 r[KMAX / 2];
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < KMAX / 2; ++k) if (t + k * TPB < G / 2) pending |= 1u << k;
            long long t0 = wall_clock64();
            while (true) {
#pragma unroll
                for (int k = 0; k < KMAX / 2; ++k)
                    if (pending & (1u << k)) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t + k * TPB) * 16, 0, AUX_SC1);
#pragma unroll
                for (int k = 0; k < KMAX / 2; ++k)
                    if ((pending & (1u << k)) && (r[k].x & 3u) == tg && (r[k].z & 3u) == tg) {
                        int j2 = t + k * TPB;
                        lds[2 * j2] = __hiloint2double((int)r[k].y, (int)(r[k].x & ~3u));
                        lds[2 * j2 + 1] = __hiloint2double((int)r[k].w, (int)(r[k].z & ~3u));
                        pending &= ~(1u << k);
                    }
                if (!pending) break;
                bad += 1u << 20;            // count re-reads in the high bits of the mismatch counter
                asm volatile("" ::: "memory");
                if (wall_clock64() - t0 > t_limit) { s_fail = 1; break; }
            }
        } else {
            for (int j = t; j < G; j += TPB) lds[j] = expected(j, it);
        }
        __syncthreads();
        if (s_fail) break;
        // check every word
        if (MODE) for (int j = t; j < G; j += TPB) bad += (lds[j] != expected(j, it));   // expected() values have zero low mantissa bits
        // emulated operator rows: E products per thread from LDS, then a two-level sum
        if (a.work) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < E; ++k) s += val[k] * lds[col[k]];
            red[t] = s;
            __syncthreads();
            if (t < 40) { double r = 0.0; for (int k = 0; k < 12; ++k) r += red[t * 12 + k]; acc += r; }
        }
        __syncthreads();
    }
    if (bad) atomicAdd(a.err, bad);
    if (s_fail && t == 0) atomicAdd(a.err + 1, 1u);
    if (t < 40) a.out[g * 40 + t] = acc;
}

int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 10000;
    int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int WG = 256;
    int per = (n + WG - 1) / WG + 3;          // own slice + three dot partials
    per = (per + 1) & ~1;
    int G = per * WG;
    if (G > TPB * KMAX) { fprintf(stderr, "G too large\n"); return 1; }
    printf("n=%d per=%d G=%d (%.1f KB of doubles, %.1f KB of granules)\n", n, per, G, G * 8 / 1024.0, G * 16 / 1024.0);
    void *buf; unsigned *flags, *err; double *out; unsigned short *cols;
    CK(hipMalloc(&buf, (size_t)2 * G * 16));
    CK(hipMalloc(&flags, WG * 1024));
    CK(hipMalloc(&err, 8));
    CK(hipMalloc(&out, WG * 40 * 8));
    std::vector<unsigned short> hc(TPB * E);
    unsigned s = 12345;
    for (auto &c : hc) { s = s * 1664525u + 1013904223u; c = (unsigned short)((s >> 8) % (unsigned)n); }
    CK(hipMalloc(&cols, hc.size() * 2));
    CK(hipMemcpy(cols, hc.data(), hc.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t lds = (size_t)(G + TPB + 2) * 8;
    CK(hipFuncSetAttribute((const void *)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_probe<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int fstride = argc > 3 ? atoi(argv[3]) : 16;      // flag spacing in 4-byte words (16, 32, 64, 128: no difference measured)
    for (int work = 0; work < 2; ++work)
        for (int mode = 0; mode < 6; ++mode)
            for (int uneven = 0; uneven < 2; ++uneven) {
                if (mode == 0 && uneven) continue;
                float best = 1e30f; unsigned herr[2] = {0, 0};
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(buf, 0, (size_t)2 * G * 16));
                    CK(hipMemset(flags, 0, WG * 1024));
                    CK(hipMemset(err, 0, 8));
                    Args a{buf, flags, G, per, iters, mode, uneven, work, fstride, cols, out, err};
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(WG), dim3(TPB), lds, 0, a);
                    else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(WG), dim3(TPB), lds, 0, a);
                    else if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(WG), dim3(TPB), lds, 0, a);
                    else if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(WG), dim3(TPB), lds, 0, a);
                    else if (mode == 4) hipLaunchKernelGGL(k_probe<4>, dim3(WG), dim3(TPB), lds, 0, a);
                    else hipLaunchKernelGGL(k_probe<5>, dim3(WG), dim3(TPB), lds, 0, a);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
                    herr[0] += h[0]; herr[1] += h[1];
                }
                printf("fstride=%d work=%d mode=%d uneven=%d: %.3f us/iteration  mismatches=%u re-reads=%u timeouts=%u\n", fstride, work, mode, uneven,
                       best * 1e3 / iters, herr[0] & 0xFFFFFu, herr[0] >> 20, herr[1]);
                fflush(stdout);
                if (herr[1]) { printf("timeout: stopping\n"); return 2; }
            }
    return 0;
}
