// mmf_amd :: the small kernels M4C adds around the shared encoder (mmf/models/m4c.py): L2 row normalisation of the
// appearance / FastText / PHOC features, the previous-prediction gather from [answer vocabulary ; OCR tokens], the OCR
// pointer-network scores with their backward, and the decoding BCE loss with its step mask.  All of them move a few
// hundred KB per step at the TextVQA shape (B = 32..128, 100 objects, 50 OCR tokens, 12 decoding steps): one wave per
// row, fp32 arithmetic, no LDS staging.  Reference call sites are cited at each C entry point in include/mmf_amd.h.
#include "common.h"
#include "mmf_amd.h"

namespace {

// ------------------------------------------------------------------------------------------------
// F.normalize(x, dim=-1): y = x / max(||x||_2, eps).  Columns may start at any element offset inside a wider
// destination row (the concatenated OCR feature, m4c.py:235-237), so accesses are per element.
// ------------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const XT* __restrict__ x, int ldx, bf16* __restrict__ y, int ldy,
                                                          float* __restrict__ inv, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const XT* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = (float)xr[c]; s += v * v; }
    s = wave_sum(s);
    const float r = 1.f / fmaxf(sqrtf(s), eps);
    bf16* yr = y + (size_t)row * ldy;
    for (int c = lane; c < D; c += 64) yr[c] = (bf16)((float)xr[c] * r);
    if (lane == 0) inv[row] = r;
}
// dx = (g - y * <g, y>) * inv   (y = the normalised row, inv = 1 / ||x||)
template <typename T>   // bf16 rows (throughput path) or fp32 rows (mmf_amd.fp32_training())
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ y, int ldy,
                                                          const float* __restrict__ inv, T* __restrict__ dx, int lddx, int rows,
                                                          int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* gr = g + (size_t)row * ldg;
    const T* yr = y + (size_t)row * ldy;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot += (float)gr[c] * (float)yr[c];
    dot = wave_sum(dot);
    const float r = inv[row];
    T* dr = dx + (size_t)row * lddx;
    for (int c = lane; c < D; c += 64) dr[c] = (T)(((float)gr[c] - (float)yr[c] * dot) * r);
}
// inv[r] = 1 / max(||x[r, :D]||, eps): the factor the fp32 forward (mmf_l2norm_rows_f32) applied, recomputed for its backward
__global__ __launch_bounds__(256) void l2norm_inv_f32_kernel(const float* __restrict__ x, int ldx, float* __restrict__ inv, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = xr[c]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(s), eps);
}

// ------------------------------------------------------------------------------------------------
// out[r] = idx[r] < rows_a ? a[idx[r]] : b[idx[r] - rows_a]   (bf16 rows of H, H % 8 == 0)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather2_kernel(const bf16* __restrict__ a, int64_t rows_a, const bf16* __restrict__ b,
                                                       int64_t rows_b, const int64_t* __restrict__ idx, bf16* __restrict__ out, int n,
                                                       int H) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    int64_t ix = idx[r];
    ix = ix < 0 ? 0 : (ix >= rows_a + rows_b ? rows_a + rows_b - 1 : ix);
    const bf16* src = ix < rows_a ? a + (size_t)ix * H : b + (size_t)(ix - rows_a) * H;
    for (int c = lane * 8; c < H; c += 512)
        *reinterpret_cast<bf16x8*>(out + (size_t)r * H + c) = *reinterpret_cast<const bf16x8*>(src + c);
}

// ------------------------------------------------------------------------------------------------
// OCR pointer network scores (OcrPtrNet.forward, m4c.py:474-493)
//   out[b, t, n] = scale * <q[b, t, :], k[b, n, :]> + mask_add[b, n]
// one workgroup per (b, t); each wave walks the OCR tokens n = wave, wave + 4, ...
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ptr_scores_fwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                              const float* __restrict__ mask_add, float* __restrict__ out, int ldo,
                                                              int T, int N, int HQ, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bt = blockIdx.x, b = bt / T;
    const bf16* qr = q + (size_t)bt * HQ;
    for (int n = wave; n < N; n += 4) {
        const bf16* kr = k + ((size_t)b * N + n) * HQ;
        float s = 0.f;
        for (int c = lane * 8; c < HQ; c += 512) {
            const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qr + c);
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)qv[e] * (float)kv[e];
        }
        s = wave_sum(s);
        if (lane == 0) out[(size_t)bt * ldo + n] = s * scale + (mask_add ? mask_add[(size_t)b * N + n] : 0.f);
    }
}
// dq[b, t, :] = scale * sum_n ds[b, t, n] * k[b, n, :]          (workgroup per (b, t), thread per feature)
template <typename RT>
__global__ __launch_bounds__(256) void ptr_scores_dq_kernel(const float* __restrict__ ds, int ldd, const RT* __restrict__ k,
                                                             RT* __restrict__ dq, int T, int N, int HQ, float scale) {
    const int bt = blockIdx.x, b = bt / T;
    const float* dr = ds + (size_t)bt * ldd;
    for (int c = threadIdx.x; c < HQ; c += 256) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += dr[n] * (float)k[((size_t)b * N + n) * HQ + c];
        dq[(size_t)bt * HQ + c] = (RT)(acc * scale);
    }
}
// dk[b, n, :] = scale * sum_t ds[b, t, n] * q[b, t, :]          (workgroup per (b, n))
template <typename RT>
__global__ __launch_bounds__(256) void ptr_scores_dk_kernel(const float* __restrict__ ds, int ldd, const RT* __restrict__ q,
                                                             RT* __restrict__ dk, int T, int N, int HQ, float scale) {
    const int bn = blockIdx.x, b = bn / N, n = bn - b * N;
    for (int c = threadIdx.x; c < HQ; c += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += ds[((size_t)b * T + t) * ldd + n] * (float)q[((size_t)b * T + t) * HQ + c];
        dk[(size_t)bn * HQ + c] = (RT)(acc * scale);
    }
}

// ------------------------------------------------------------------------------------------------
// M4CDecodingBCEWithMaskLoss (mmf/modules/losses.py:581-592):
//   loss = sum_r w[r] * sum_n bce(x[r, n], t[r, n]) / max(sum_r w[r], 1)
// ------------------------------------------------------------------------------------------------
constexpr int BCEM_BLOCKS = 128;
__global__ __launch_bounds__(256) void bce_rowmask_partial_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                                   const float* __restrict__ w, float* __restrict__ partial,
                                                                   int64_t n, int N) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float xv = x[i], tv = t[i];
        s += w[i / N] * (fmaxf(xv, 0.f) - xv * tv + log1pf(__expf(-fabsf(xv))));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void bce_rowmask_final_kernel(const float* __restrict__ partial, int nparts, const float* __restrict__ w,
                                                                int rows, float* __restrict__ loss, float* __restrict__ count) {
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += partial[i];
    for (int i = threadIdx.x; i < rows; i += 64) c += w[i];
    s = wave_sum(s);
    c = fmaxf(wave_sum(c), 1.f);
    if (threadIdx.x == 0) { loss[0] = s / c; count[0] = c; }
}
__global__ __launch_bounds__(256) void bce_rowmask_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                               const float* __restrict__ w, const float* __restrict__ count,
                                                               const float* __restrict__ gloss, float* __restrict__ d, int64_t n, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = (gloss ? gloss[0] : 1.f) / count[0];
    d[i] = g * w[i / N] * (1.f / (1.f + __expf(-x[i])) - t[i]);
}

}  // namespace

extern "C" {

int mmf_l2norm_rows_fwd(const void* x, int x_f32, int ldx, void* y, int ldy, float* inv_norm, int rows, int D, float eps, void* stream) {
    MMF_CHECK_ARG(x && y && inv_norm && rows > 0 && D > 0 && ldx >= D && ldy >= D, "l2norm_rows_fwd: bad operand");
    const dim3 grid((rows + 3) / 4);
    if (x_f32)
        hipLaunchKernelGGL(l2norm_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (bf16*)y, ldy, inv_norm, rows, D, eps);
    else
        hipLaunchKernelGGL(l2norm_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (bf16*)y, ldy, inv_norm, rows, D, eps);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_l2norm_rows_bwd(const void* g, int ldg, const void* y, int ldy, const float* inv_norm, void* dx, int lddx, int rows, int D,
                        void* stream) {
    MMF_CHECK_ARG(g && y && inv_norm && dx && rows > 0 && D > 0 && ldg >= D && ldy >= D && lddx >= D, "l2norm_rows_bwd: bad operand");
    hipLaunchKernelGGL(l2norm_bwd_kernel<bf16>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)g, ldg, (const bf16*)y, ldy,
                       inv_norm, (bf16*)dx, lddx, rows, D);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_l2norm_rows_f32_bwd(const float* g, int ldg, const float* y, int ldy, const float* x, int ldx, float* inv_ws, float* dx, int lddx, int rows,
                            int D, float eps, void* stream) {
    MMF_CHECK_ARG(g && y && x && inv_ws && dx && rows > 0 && D > 0 && ldg >= D && ldy >= D && ldx >= D && lddx >= D, "l2norm_rows_f32_bwd: bad operand");
    hipLaunchKernelGGL(l2norm_inv_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, inv_ws, rows, D, eps);
    hipLaunchKernelGGL(l2norm_bwd_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, ldg, y, ldy, inv_ws, dx, lddx, rows, D);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_gather_rows2(const void* a, int64_t rows_a, const void* b, int64_t rows_b, const int64_t* idx, void* out, int n, int H,
                     void* stream) {
    MMF_CHECK_ARG(a && idx && out && n > 0 && rows_a > 0 && rows_b >= 0 && (b || rows_b == 0) && (H % 8) == 0, "gather_rows2: bad operand");
    hipLaunchKernelGGL(gather2_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, rows_a, (const bf16*)b, rows_b,
                       idx, (bf16*)out, n, H);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_ptr_scores_fwd(const void* q, const void* k, const float* mask_add, float* out, int ldo, int B, int T, int N, int HQ,
                       float scale, void* stream) {
    MMF_CHECK_ARG(q && k && out && B > 0 && T > 0 && N > 0 && HQ > 0 && (HQ % 8) == 0 && ldo >= N, "ptr_scores_fwd: bad operand");
    hipLaunchKernelGGL(ptr_scores_fwd_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, (const bf16*)q, (const bf16*)k, mask_add, out,
                       ldo, T, N, HQ, scale);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_ptr_scores_bwd(const float* dscores, int ldd, const void* q, const void* k, void* dq, void* dk, int B, int T, int N, int HQ,
                       float scale, void* stream) {
    MMF_CHECK_ARG(dscores && q && k && dq && dk && B > 0 && T > 0 && N > 0 && HQ > 0 && ldd >= N, "ptr_scores_bwd: bad operand");
    hipLaunchKernelGGL(ptr_scores_dq_kernel<bf16>, dim3(B * T), dim3(256), 0, (hipStream_t)stream, dscores, ldd, (const bf16*)k, (bf16*)dq, T, N, HQ, scale);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(ptr_scores_dk_kernel<bf16>, dim3(B * N), dim3(256), 0, (hipStream_t)stream, dscores, ldd, (const bf16*)q, (bf16*)dk, T, N, HQ, scale);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_ptr_scores_f32_bwd(const float* dscores, int ldd, const float* q, const float* k, float* dq, float* dk, int B, int T, int N, int HQ,
                           float scale, void* stream) {
    MMF_CHECK_ARG(dscores && q && k && dq && dk && B > 0 && T > 0 && N > 0 && HQ > 0 && ldd >= N, "ptr_scores_f32_bwd: bad operand");
    hipLaunchKernelGGL(ptr_scores_dq_kernel<float>, dim3(B * T), dim3(256), 0, (hipStream_t)stream, dscores, ldd, k, dq, T, N, HQ, scale);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(ptr_scores_dk_kernel<float>, dim3(B * N), dim3(256), 0, (hipStream_t)stream, dscores, ldd, q, dk, T, N, HQ, scale);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_bce_rowmask_ws_floats(void) { return BCEM_BLOCKS; }
int mmf_bce_rowmask_fwd(const float* scores, const float* targets, const float* row_weight, float* loss, float* count, float* ws,
                        int rows, int N, void* stream) {
    MMF_CHECK_ARG(scores && targets && row_weight && loss && count && ws && rows > 0 && N > 0, "bce_rowmask_fwd: bad operand");
    hipLaunchKernelGGL(bce_rowmask_partial_kernel, dim3(BCEM_BLOCKS), dim3(256), 0, (hipStream_t)stream, scores, targets, row_weight, ws,
                       (int64_t)rows * N, N);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_rowmask_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, BCEM_BLOCKS, row_weight, rows, loss, count);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_bce_rowmask_bwd(const float* scores, const float* targets, const float* row_weight, const float* count, const float* gloss,
                        float* dscores, int rows, int N, void* stream) {
    MMF_CHECK_ARG(scores && targets && row_weight && count && dscores && rows > 0 && N > 0, "bce_rowmask_bwd: bad operand");
    const int64_t n = (int64_t)rows * N;
    hipLaunchKernelGGL(bce_rowmask_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, targets,
                       row_weight, count, gloss, dscores, n, N);
    MMF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
