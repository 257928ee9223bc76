// L2 -> LDS staging throughput per CU as a function of workgroup shape and DMA depth (global_load_lds_dwordx4), with the
// access pattern of the GEMM K-loop: a workgroup stages ROWS_A rows of an activation panel (row stride lda, one 128-byte
// chunk per row and K-step) and ROWS_B rows of a weight panel per step.  No MFMA, no LDS reads: the ceiling of the load path.
//   hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.hip && ./dma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// NTH threads; per step each thread issues PIECES x 16-byte LDS-DMA pieces (PIECES * NTH * 16 bytes per step).
// DEPTH stages may be in flight: wait vmcnt(PIECES * (DEPTH - 1)) then a raw barrier (DEPTH == 1: vmcnt(0)).
template <int NTH, int ROWS_A, int ROWS_B, int DEPTH>
__global__ __launch_bounds__(NTH) void dma_kernel(const char* A, const char* B, int lda, int ldb, int tiles_n, int nk, int rot_bytes, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PA = ROWS_A * 128 / (NTH * 16), PB = ROWS_B * 128 / (NTH * 16), PIECES = PA + PB;
    constexpr int STAGE = (ROWS_A + ROWS_B) * 128;
    const int tid = threadIdx.x, wave = tid >> 6;
    int bid = blockIdx.x;
    {   // XCD-aware contiguous runs, like the GEMM
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int sw = (tid >> 3) & 7;
    const int chunk = ((tid & 7) ^ sw) * 16;
    auto issue = [&](int kt, int buf) {
        unsigned char* st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = tile_m * ROWS_A + (tid >> 3) + (NTH / 8) * i;
            const char* src = A + (size_t)row * lda + kt * 128 + chunk;
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(st + i * (NTH * 16) + wave * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = tile_n * ROWS_B + (tid >> 3) + (NTH / 8) * i;
            const char* src = B + (size_t)row * ldb + kt * 128 + chunk;
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(st + ROWS_A * 128 + i * (NTH * 16) + wave * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (d < nk) issue(d, d);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (DEPTH == 1 || kt + DEPTH - 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + DEPTH < nk) issue(kt + DEPTH, buf);       // refill the stage everybody is done with
        buf = buf + 1 == DEPTH ? 0 : buf + 1;
    }
    if (tid == 0 && sink && smem[threadIdx.x] == 123) sink[0] = 1;
}

template <int NTH, int RA, int RB, int DEPTH>
float run(const char* A, const char* B, int M, int N, int K, int wgs_per_cu_hint, int reps) {
    const int tiles_m = M / RA, tiles_n = N / RB, nk = K / 64;
    const int lds = DEPTH * (RA + RB) * 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<NTH, RA, RB, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((dma_kernel<NTH, RA, RB, DEPTH>), dim3(tiles_m * tiles_n), dim3(NTH), lds, 0, A, B, K * 2, K * 2, tiles_n, nk, 0, (int*)nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dma_kernel<NTH, RA, RB, DEPTH>), dim3(tiles_m * tiles_n), dim3(NTH), lds, 0, A, B, K * 2, K * 2, tiles_n, nk, 0, (int*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)tiles_m * tiles_n * nk * (RA + RB) * 128;
    const int wgs = tiles_m * tiles_n;
    const int slots = 256 * wgs_per_cu_hint;
    const int rounds = (wgs + slots - 1) / slots;
    printf("  %3dx%-3d x%d wg/cu depth %d: %4d wgs (%d round%s) %7.1f us  %6.1f TB/s chip  %6.1f GB/s per CU (busy-CU estimate %6.1f)  step %.3f us\n", RA, RB, wgs_per_cu_hint, DEPTH,
           wgs, rounds, rounds > 1 ? "s" : " ", us, bytes / us / 1e6, bytes / us / 1e3 / 256, bytes / us / 1e3 / 256 * (double)(rounds * slots) / wgs,
           us / (rounds * nk));
    if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
    return (float)us;
}

int main() {
    const int M = 7296 + 384;  // a little slack so every tile shape divides
    const int Mp = 7680;
    (void)M;
    size_t abytes = (size_t)Mp * 3072 * 2, bbytes = (size_t)3072 * 3072 * 2;
    char *A, *B;
    hipMalloc(&A, abytes); hipMalloc(&B, bbytes);
    hipMemset(A, 1, abytes); hipMemset(B, 1, bbytes);
    for (int K : {768, 3072}) {
        for (int N : {768, 3072}) {
            printf("M=7680 N=%d K=%d  (A panel %.1f MB, W %.1f MB)\n", N, K, 7680.0 * K * 2 / 1e6, (double)N * K * 2 / 1e6);
            run<512, 128, 128, 1>(A, B, Mp, N, K, 2, 20);
            run<512, 128, 128, 2>(A, B, Mp, N, K, 2, 20);
            run<512, 256, 128, 1>(A, B, Mp, N, K, 1, 20);
            run<512, 256, 128, 2>(A, B, Mp, N, K, 1, 20);
            run<512, 256, 128, 3>(A, B, Mp, N, K, 1, 20);
            run<512, 256, 256, 1>(A, B, Mp, N, K, 1, 20);
            run<512, 256, 256, 2>(A, B, Mp, N, K, 1, 20);
            run<512, 192, 192, 2>(A, B, Mp, N, K, 1, 20);
            run<512, 192, 192, 3>(A, B, Mp, N, K, 1, 20);
            run<512, 192, 256, 2>(A, B, Mp, N, K, 1, 20);
            run<256, 128, 128, 1>(A, B, Mp, N, K, 2, 20);
            run<256, 128, 128, 2>(A, B, Mp, N, K, 2, 20);
        }
    }
    return 0;
}
