// Streaming-update ceiling for one big tensor (the 23.4 M-element word-embedding table): 4 fp32 read streams (p, g, m, v) and 3
// fp32 write streams, the access pattern of adamw_multi_kernel, in a few variants of chunking / unrolling / cache hints.
//   hipcc --offload-arch=gfx950 -O3 -o adam_bench adam_bench.hip && ./adam_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int CHUNK, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr) {
    const long base = (long)blockIdx.x * CHUNK;
    const long end = base + CHUNK < n ? base + CHUNK : n;
    for (long i0 = base + threadIdx.x * 4; i0 < end; i0 += 1024L * UNROLL) {
        f4 pv[UNROLL], gv[UNROLL], mv[UNROLL], vv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long i = i0 + 1024L * u;
            if (i < end) {
                if (NT) {
                    pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + i)); gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g + i));
                    mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m + i)); vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v + i));
                } else {
                    pv[u] = *reinterpret_cast<const f4*>(p + i); gv[u] = *reinterpret_cast<const f4*>(g + i);
                    mv[u] = *reinterpret_cast<const f4*>(m + i); vv[u] = *reinterpret_cast<const f4*>(v + i);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long i = i0 + 1024L * u;
            if (i < end) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mv[u][j] = 0.9f * mv[u][j] + 0.1f * gv[u][j];
                    vv[u][j] = 0.999f * vv[u][j] + 0.001f * gv[u][j] * gv[u][j];
                    pv[u][j] -= lr * (mv[u][j] / (sqrtf(vv[u][j]) + 1e-8f)) + lr * 0.01f * pv[u][j];
                }
                if (NT) {
                    __builtin_nontemporal_store(pv[u], reinterpret_cast<f4*>(p + i)); __builtin_nontemporal_store(mv[u], reinterpret_cast<f4*>(m + i));
                    __builtin_nontemporal_store(vv[u], reinterpret_cast<f4*>(v + i));
                } else {
                    *reinterpret_cast<f4*>(p + i) = pv[u]; *reinterpret_cast<f4*>(m + i) = mv[u]; *reinterpret_cast<f4*>(v + i) = vv[u];
                }
            }
        }
    }
}

// grid-stride form: a fixed number of workgroups, each walking the tensor with a stride of the whole grid
template <int UNROLL>
__global__ __launch_bounds__(256) void adam_gs_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, float lr) {
    const long stride = (long)gridDim.x * 1024;
    for (long i0 = (long)blockIdx.x * 1024 + threadIdx.x * 4; i0 < n; i0 += stride * UNROLL) {
        f4 pv[UNROLL], gv[UNROLL], mv[UNROLL], vv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long i = i0 + stride * u;
            if (i < n) { pv[u] = *reinterpret_cast<const f4*>(p + i); gv[u] = *reinterpret_cast<const f4*>(g + i); mv[u] = *reinterpret_cast<const f4*>(m + i); vv[u] = *reinterpret_cast<const f4*>(v + i); }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long i = i0 + stride * u;
            if (i < n) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mv[u][j] = 0.9f * mv[u][j] + 0.1f * gv[u][j];
                    vv[u][j] = 0.999f * vv[u][j] + 0.001f * gv[u][j] * gv[u][j];
                    pv[u][j] -= lr * (mv[u][j] / (sqrtf(vv[u][j]) + 1e-8f)) + lr * 0.01f * pv[u][j];
                }
                *reinterpret_cast<f4*>(p + i) = pv[u]; *reinterpret_cast<f4*>(m + i) = mv[u]; *reinterpret_cast<f4*>(v + i) = vv[u];
            }
        }
    }
}

__global__ void copy_kernel(const f4* __restrict__ a, f4* __restrict__ b, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) b[i] = a[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename F>
static float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 10 * 1e3f;
}

int main() {
    const long n = 30522L * 768;
    float *p, *g, *m, *v, *scratch;
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4));
    CK(hipMalloc(&scratch, 512L << 20));
    CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
    const double bytes = (double)n * 28.0;
    auto flush = [&]() { hipMemsetAsync(scratch, 1, 512L << 20, 0); };      // push the tensors out of the 256 MB MALL between runs
    auto report = [&](const char* name, float us) { printf("%-44s %8.1f us  %6.2f TB/s (28 B/element)\n", name, us, bytes / us / 1e6); };
    float t_flush = timeit([&]() { flush(); });
    printf("tensor: %ld elements (%.1f MB per stream); memset of 512 MB: %.1f us\n", n, n * 4 / 1e6, t_flush);
    const dim3 B(256);
#define RUNK(name, CH, UN, NTV) { auto k = adam_kernel<CH, UN, NTV>; const unsigned gr = (unsigned)((n + CH - 1) / CH); \
        float us = timeit([&]() { flush(); hipLaunchKernelGGL(k, dim3(gr), B, 0, 0, p, g, m, v, n, 1e-4f); }) - t_flush; report(name, us); }
#define RUNG(name, UN, GR) { auto k = adam_gs_kernel<UN>; \
        float us = timeit([&]() { flush(); hipLaunchKernelGGL(k, dim3(GR), B, 0, 0, p, g, m, v, n, 1e-4f); }) - t_flush; report(name, us); }
    RUNK("chunk 16384, unroll 1 (shipped form)", 16384, 1, false)
    RUNK("chunk 16384, unroll 2", 16384, 2, false)
    RUNK("chunk 16384, unroll 4", 16384, 4, false)
    RUNK("chunk 4096, unroll 1", 4096, 1, false)
    RUNK("chunk 4096, unroll 4", 4096, 4, false)
    RUNK("chunk 65536, unroll 2", 65536, 2, false)
    RUNK("chunk 16384, unroll 2, nontemporal", 16384, 2, true)
    RUNK("chunk 4096, unroll 4, nontemporal", 4096, 4, true)
    RUNG("grid-stride 2048 wgs, unroll 1", 1, 2048)
    RUNG("grid-stride 2048 wgs, unroll 2", 2, 2048)
    RUNG("grid-stride 4096 wgs, unroll 4", 4, 4096)
    {   // plain copy for scale: 2 streams
        float us = timeit([&]() { flush(); hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, (const f4*)p, (f4*)m, n / 4); }) - t_flush;
        printf("%-44s %8.1f us  %6.2f TB/s (8 B/element)\n", "copy p -> m", us, (double)n * 8 / us / 1e6);
    }
    return 0;
}
