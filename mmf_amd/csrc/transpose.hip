// mmf_amd :: multi-tensor bf16 transpose.  The input-gradient GEMM of nn.Linear, dX = dY W (autograd of
// mmf/modules/hf_layers.py:169-180 etc.), reads W [out, in] along its rows as the reduction — a k-major operand.  In the
// isolated microbenchmark the GEMM kernel runs ~18 % faster when both operands are row operands (tools/gemm_vs_library.py),
// so the bf16 weight shadows keep a transposed twin W^T [in, out] (functional.DGRAD_NT) that the optimizer step refreshes
// with this kernel (340 MB per step in 54 us = 6.3 TB/s: at the HBM roofline).  The kernel: one launch for up to MMF_MT_MAX matrices, 64x64 tiles through LDS, 16-byte accesses on both
// sides.  ~340 MB of traffic per step for VisualBERT-base (85 M weights), ~0.07 ms.
#include "common.h"
#include "mmf_amd.h"

namespace {

constexpr int TT = 64;          // tile edge
constexpr int TP = TT + 8;      // padded LDS row (elements): keeps the 16-byte alignment, spreads the column reads over banks

__global__ __launch_bounds__(256) void transpose_multi_kernel(mmf_transpose_list d) {
    __shared__ __attribute__((aligned(16))) bf16 tile[TT][TP];
    const int which = blockIdx.y;
    const int R = d.rows[which], Cn = d.cols[which];
    const int tiles_c = Cn / TT, ntiles = (R / TT) * tiles_c;
    const bf16* __restrict__ src = reinterpret_cast<const bf16*>(d.src[which]);
    bf16* __restrict__ dst = reinterpret_cast<bf16*>(d.dst[which]);
    const int t = threadIdx.x, sub = t >> 3, c8 = (t & 7) * 8;
    for (int tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        const int r0 = (tix / tiles_c) * TT, c0 = (tix % tiles_c) * TT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = sub + 32 * i;
            *reinterpret_cast<bf16x8*>(&tile[r][c8]) = *reinterpret_cast<const bf16x8*>(src + (size_t)(r0 + r) * Cn + c0 + c8);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = sub + 32 * i;       // source column = destination row
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[c8 + j][c];
            // (the twins are read by the NEXT step's backward pass: non-temporal, they would only push the bf16 shadows the next forward reads out of the
            // Infinity Cache; -0.04 ms per step, profiles/r04_store_policy.txt section 11)
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(dst + (size_t)(c0 + c) * R + r0 + c8));
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int mmf_transpose_bf16_multi(const mmf_transpose_list* d, void* stream) {
    MMF_CHECK_ARG(d && d->n > 0 && d->n <= MMF_MT_MAX, "transpose_bf16_multi: bad descriptor");
    int mx = 0;
    for (int i = 0; i < d->n; ++i) {
        MMF_CHECK_ARG(d->src[i] && d->dst[i] && d->rows[i] > 0 && d->cols[i] > 0 && (d->rows[i] % TT) == 0 && (d->cols[i] % TT) == 0,
                      "transpose_bf16_multi: matrices must be non-empty with both dimensions multiples of 64");
        const int nt = (d->rows[i] / TT) * (d->cols[i] / TT);
        mx = nt > mx ? nt : mx;
    }
    const int gx = mx < 1024 ? mx : 1024;
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(gx, d->n), dim3(256), 0, (hipStream_t)stream, *d);
    MMF_CHECK_LAUNCH();
    return 0;
}
