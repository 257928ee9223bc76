// mmf_amd :: the kernels UNITER's remaining pretraining heads add around the shared encoder (mmf/models/uniter.py:36-39 default task list
// mlm, itm, mrc, mrfr, wra):
//   MRFR  masked region feature regression, mmf/models/transformers/heads/mrfr.py:58-93: the mean-squared-error loss between the fp32
//         prediction (the tied image-embedding weight applied transposed, an MFMA GEMM) and the original features of the masked regions,
//         and its gradient as the bf16 operand of the projection's dgrad / wgrad GEMMs;
//   WRA   word-region alignment, mmf/models/transformers/heads/wra.py:36-83 over mmf/modules/ot.py:15-110: per sample the cosine cost
//         matrix between the text rows and the region rows of the joint sequence, 50 IPOT iterations for the transport plan (a constant
//         of the backward pass, `ipot` runs under no_grad), the distance trace(C T), the signed mean over the batch, and the backward
//         through the cost matrix and the two L2 normalisations.
// One workgroup per sample, fp32 arithmetic throughout (the reference casts to fp32 "for stability", wra.py:74); the transport plan and
// its kernel matrix live in LDS for all 50 iterations (2 x 66 KB of the CU's 160 KB).  A few MFLOP per sample: latency-bound, not a
// matrix-core problem (the cost matrix of 128 x 100 x 768 is tiled through LDS on the vector ALUs).
#include "common.h"
#include "mmf_amd.h"

namespace {

// ------------------------------------------------------------------------------------------------
// F.mse_loss(pred, target, reduction="mean")
// ------------------------------------------------------------------------------------------------
constexpr int MSE_BLOCKS = 256;
// `row_label` (optional): only rows with label == 1 count (ViLBERT's masked region regression, vilbert.py:1139-1148: the sum over the
// masked regions divided by max(number of their elements, 1)); ws[MSE_BLOCKS + b] then carries block b's count of selected rows.
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ target, int ldt,
                                                           const int64_t* __restrict__ row_label, float* __restrict__ ws, int rows, int cols) {
    __shared__ float red[8];
    const int64_t n = (int64_t)rows * cols;
    float s = 0.f, cnt = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)MSE_BLOCKS * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        if (row_label && row_label[r] != 1) continue;
        const float d = pred[r * ldp + c] - target[r * ldt + c];
        s += d * d;
        cnt += 1.f;
    }
    s = wave_sum(s); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) { ws[blockIdx.x] = red[0] + red[1] + red[2] + red[3]; ws[MSE_BLOCKS + blockIdx.x] = red[4] + red[5] + red[6] + red[7]; }
}
__global__ __launch_bounds__(64) void mse_final_kernel(const float* __restrict__ ws, float* __restrict__ loss, float* __restrict__ count, float inv_n, int masked) {
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < MSE_BLOCKS; i += 64) { s += ws[i]; c += ws[MSE_BLOCKS + i]; }
    s = wave_sum(s); c = wave_sum(c);
    if (threadIdx.x == 0) {
        if (masked) { const float den = fmaxf(c, 1.f); loss[0] = s / den; if (count) count[0] = den; }
        else { loss[0] = s * inv_n; if (count) count[0] = c; }
    }
}
// d = gloss * 2 (pred - target) / n  as bf16 [rows, ldd] (pad columns zeroed)
template <typename RT>   // bf16: the GEMM operand of the throughput path; float: mmf_amd.fp32_training()
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ target, int ldt,
                                                       const int64_t* __restrict__ row_label, const float* __restrict__ count,
                                                       const float* __restrict__ gloss, RT* __restrict__ d, int ldd, int rows, int cols, float two_over_n) {
    const int64_t n = (int64_t)rows * ldd;
    const float g = (gloss ? gloss[0] : 1.f) * (row_label ? 2.f / count[0] : two_over_n);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ldd;
        const int c = (int)(i - r * ldd);
        const bool on = c < cols && (!row_label || row_label[r] == 1);
        d[i] = (RT)(on ? g * (pred[r * ldp + c] - target[r * ldt + c]) : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// WRA
// ------------------------------------------------------------------------------------------------
constexpr int WRA_MAX = 128;          // text rows / region rows per sample (UNITER: <= 128 tokens, <= 100 regions)
constexpr int WRA_LD = WRA_MAX + 1;   // LDS row stride of the [N][M] plan / kernel matrices
constexpr int WRA_KC = 32;            // feature chunk of the cost-matrix tiles
constexpr int WRA_TLD = WRA_KC + 1;

struct WraArgs {
    const bf16* seq; int ld;          // joint sequence [B, S, H] (row stride ld), text rows [0, M), region rows [M, M + N)
    int S, H, M, N, B;
    const float* txt_pad;             // [B, M] 1 = padding
    const float* img_pad;             // [B, N]
    const int64_t* label;             // [B] is_correct (1 matched, 0 mismatched)
    float* xinv; float* yinv;         // [B, M], [B, N]: 1 / max(||row||, eps), saved for the backward
    float* plan;                      // [B, N, M] transport plan (masked), saved for the backward
    float* cost;                      // [B, M, N] workspace
    float* dist;                      // [B]
    float beta, eps;
    int iters;
};

// cost[m][n] = joint_pad ? 0 : 1 - <x_m, y_n> / (max(|x_m|, eps) max(|y_n|, eps)); also kern[n][m] = joint_pad ? 0 : exp(-cost / beta) into LDS
DEVI void wra_cost_tile(const WraArgs& a, int b, float* xs, float* ys, float (&acc)[8][8], int tid) {
    const int ty = tid >> 4, tx = tid & 15;         // 16 x 16 threads, 8 x 8 outputs each: 128 x 128
    const bf16* base = a.seq + (size_t)b * a.S * a.ld;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < a.H; k0 += WRA_KC) {
        __syncthreads();
        for (int e = tid; e < WRA_MAX * WRA_KC; e += 256) {      // stage the chunk of both operands as fp32 (rows beyond M / N: zeros)
            const int r = e / WRA_KC, k = e - r * WRA_KC;
            const bool kin = (k0 + k) < a.H;
            xs[r * WRA_TLD + k] = (r < a.M && kin) ? (float)base[(size_t)r * a.ld + k0 + k] : 0.f;
            ys[r * WRA_TLD + k] = (r < a.N && kin) ? (float)base[(size_t)(a.M + r) * a.ld + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < WRA_KC; ++k) {
            float xv[8], yv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = xs[(ty * 8 + i) * WRA_TLD + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) yv[j] = ys[(tx * 8 + j) * WRA_TLD + k];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(xv[i], yv[j], acc[i][j]);
        }
    }
}

// row norms of the text / region rows of sample b: inv[r] = 1 / max(||row||, eps)
DEVI void wra_norms(const WraArgs& a, int b, float* xinv_s, float* yinv_s, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const bf16* base = a.seq + (size_t)b * a.S * a.ld;
    for (int r = wave; r < a.M + a.N; r += 4) {
        const bf16* row = base + (size_t)r * a.ld;
        float s = 0.f;
        for (int c = lane; c < a.H; c += 64) { const float v = (float)row[c]; s += v * v; }
        s = wave_sum(s);
        const float inv = 1.f / fmaxf(sqrtf(s), a.eps);
        if (lane == 0) { if (r < a.M) xinv_s[r] = inv; else yinv_s[r - a.M] = inv; }
    }
}

__global__ __launch_bounds__(256) void wra_fwd_kernel(WraArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* kern = reinterpret_cast<float*>(smem);            // [N][WRA_LD]  A = exp(-C^T / beta)
    float* plan = kern + WRA_MAX * WRA_LD;                    // [N][WRA_LD]  T
    float* xinv_s = plan + WRA_MAX * WRA_LD;                  // [128]
    float* yinv_s = xinv_s + WRA_MAX;
    float* sigma = yinv_s + WRA_MAX;                          // [M]
    float* delta = sigma + WRA_MAX;                           // [N]
    float* xpad = delta + WRA_MAX;
    float* ypad = xpad + WRA_MAX;
    float* red = ypad + WRA_MAX;                              // [8]
    // the cost tiles alias the plan matrix (not needed before the iterations start)
    float* xs = plan;
    float* ys = plan + WRA_MAX * WRA_TLD;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int M = a.M, N = a.N;

    for (int i = tid; i < WRA_MAX; i += 256) {
        xpad[i] = i < M ? a.txt_pad[(size_t)b * M + i] : 1.f;
        ypad[i] = i < N ? a.img_pad[(size_t)b * N + i] : 1.f;
    }
    wra_norms(a, b, xinv_s, yinv_s, tid);
    float acc[8][8];
    wra_cost_tile(a, b, xs, ys, acc, tid);       // (begins and ends its chunk loop with barriers: the norms and pads are visible after it)
    __syncthreads();
    float xl = 0.f, yl = 0.f;
    for (int i = 0; i < M; ++i) xl += 1.f - xpad[i];
    for (int i = 0; i < N; ++i) yl += 1.f - ypad[i];
    {
        const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = ty * 8 + i, n = tx * 8 + j;
                if (m < M && n < N) {
                    const bool jp = xpad[m] != 0.f || ypad[n] != 0.f;
                    const float c = jp ? 0.f : 1.f - acc[i][j] * xinv_s[m] * yinv_s[n];
                    a.cost[((size_t)b * M + m) * N + n] = c;
                    kern[n * WRA_LD + m] = jp ? 0.f : expf(-c / a.beta);
                }
            }
    }
    __syncthreads();
    // IPOT (ot.py:38-84): sigma = 1 / x_len (0 on padding), T = 1 (0 on joint padding); per iteration Q = A o T,
    // delta = 1 / (y_len Q sigma + y_mask), sigma = 1 / (x_len delta Q + x_mask), T = delta Q sigma
    for (int e = tid; e < N * WRA_LD; e += 256) {
        const int n = e / WRA_LD, m = e - n * WRA_LD;
        if (m < M) plan[e] = (xpad[m] != 0.f || ypad[n] != 0.f) ? 0.f : 1.f;
    }
    for (int m = tid; m < M; m += 256) sigma[m] = xpad[m] != 0.f ? 0.f : 1.f / xl;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int it = 0; it < a.iters; ++it) {
        for (int n = wave; n < N; n += 4) {             // delta[n]: one wave per region row
            float s = 0.f;
            for (int m = lane; m < M; m += 64) s += kern[n * WRA_LD + m] * plan[n * WRA_LD + m] * sigma[m];
            s = wave_sum(s);
            if (lane == 0) delta[n] = 1.f / (yl * s + ypad[n] * 1e4f);
        }
        __syncthreads();
        for (int m = tid; m < M; m += 256) {             // sigma[m]: a column walk (row stride 129 words: conflict-free across lanes)
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += delta[n] * kern[n * WRA_LD + m] * plan[n * WRA_LD + m];
            sigma[m] = 1.f / (xl * s + xpad[m] * 1e4f);
        }
        __syncthreads();
        for (int e = tid; e < N * WRA_LD; e += 256) {
            const int n = e / WRA_LD, m = e - n * WRA_LD;
            if (m < M) plan[e] = delta[n] * kern[e] * plan[e] * sigma[m];
        }
        __syncthreads();
    }
    // distance = trace(C T) = sum_{m, n} C[m][n] T[n][m]; the masked plan is saved for the backward
    float d = 0.f;
    for (int e = tid; e < N * M; e += 256) {
        const int n = e / M, m = e - n * M;
        const bool jp = xpad[m] != 0.f || ypad[n] != 0.f;
        const float t = jp ? 0.f : plan[n * WRA_LD + m];
        a.plan[((size_t)b * N + n) * M + m] = t;
        d += a.cost[((size_t)b * M + m) * N + n] * t;
    }
    d = wave_sum(d);
    if (lane == 0) red[wave] = d;
    for (int m = tid; m < M; m += 256) a.xinv[(size_t)b * M + m] = xinv_s[m];
    for (int n = tid; n < N; n += 256) a.yinv[(size_t)b * N + n] = yinv_s[n];
    __syncthreads();
    if (tid == 0) a.dist[b] = red[0] + red[1] + red[2] + red[3];
}

// loss = (sum over label == 1 of dist - sum over label == 0 of dist) / (#label 1 + #label 0)   (wra.py:77-80)
__global__ __launch_bounds__(64) void wra_loss_kernel(const float* __restrict__ dist, const int64_t* __restrict__ label, int B, float* __restrict__ loss,
                                                       float* __restrict__ count) {
    float s = 0.f, c = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        if (label[b] == 1) { s += dist[b]; c += 1.f; }
        else if (label[b] == 0) { s -= dist[b]; c += 1.f; }
    }
    s = wave_sum(s); c = wave_sum(c);
    if (threadIdx.x == 0) { loss[0] = s / c; count[0] = c; }
}

// Backward: d dist / d C[m][n] = T[n][m] (the plan is a constant), C = 1 - xh yh^T with xh = x xinv, yh = y yinv:
//   G[m][n] = -g_b T[n][m];  dxh = G yh;  dyh = G^T xh;  dx = xinv (dxh - xh <xh, dxh>)  (dx = dxh / eps for a row shorter than eps).
// The projection term needs no pass over the features: <xh_m, dxh_m> = sum_n G[m][n] <xh_m, yh_n> = sum_n G[m][n] (1 - C[m][n]), from
// the saved cost matrix (G vanishes wherever the cost was masked).  One workgroup per (sample, 64-feature slab): the slab of xh / yh
// sits in LDS, every output row is a short dot-product walk.  Deterministic (no atomics).
constexpr int WRA_FS = 64;
__global__ __launch_bounds__(256) void wra_bwd_kernel(WraArgs a, const float* __restrict__ gloss, const float* __restrict__ count, bf16* __restrict__ dseq,
                                                       int ldd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xh = reinterpret_cast<float*>(smem);               // [M][WRA_FS + 1]
    float* yh = xh + WRA_MAX * (WRA_FS + 1);                  // [N][WRA_FS + 1]
    float* dots = yh + WRA_MAX * (WRA_FS + 1);                // [M + N]
    const int b = blockIdx.x, f0 = blockIdx.y * WRA_FS, tid = threadIdx.x;
    const int M = a.M, N = a.N;
    const int64_t lab = a.label[b];
    const float sign = lab == 1 ? 1.f : (lab == 0 ? -1.f : 0.f);
    const float g = -(gloss ? gloss[0] : 1.f) * sign / count[0];
    const bf16* base = a.seq + (size_t)b * a.S * a.ld;
    const float* plan = a.plan + (size_t)b * N * M;           // T[n][m]
    const float* cost = a.cost + (size_t)b * M * N;           // C[m][n]
    for (int e = tid; e < (M + N) * WRA_FS; e += 256) {
        const int r = e / WRA_FS, k = e - r * WRA_FS;
        const float v = (f0 + k < a.H) ? (float)base[(size_t)r * a.ld + f0 + k] : 0.f;
        if (r < M) xh[r * (WRA_FS + 1) + k] = v * a.xinv[(size_t)b * M + r];
        else yh[(r - M) * (WRA_FS + 1) + k] = v * a.yinv[(size_t)b * N + r - M];
    }
    for (int r = tid; r < M + N; r += 256) {
        float d = 0.f;
        if (r < M) { for (int n = 0; n < N; ++n) d += plan[(size_t)n * M + r] * (1.f - cost[(size_t)r * N + n]); }
        else { const int n = r - M; for (int m = 0; m < M; ++m) d += plan[(size_t)n * M + m] * (1.f - cost[(size_t)m * N + n]); }
        dots[r] = g * d;
    }
    __syncthreads();
    // thread -> (row r of the M + N outputs, 16-feature quarter of the slab)
    for (int o = tid; o < (M + N) * 4; o += 256) {
        const int r = o >> 2, q = (o & 3) * 16;
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        if (r < M) {
            for (int n = 0; n < N; ++n) {
                const float w = g * plan[(size_t)n * M + r];
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fmaf(w, yh[n * (WRA_FS + 1) + q + k], acc[k]);
            }
        } else {
            const float* trow = plan + (size_t)(r - M) * M;
            for (int m = 0; m < M; ++m) {
                const float w = g * trow[m];
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fmaf(w, xh[m * (WRA_FS + 1) + q + k], acc[k]);
            }
        }
        const float* self = r < M ? xh + r * (WRA_FS + 1) + q : yh + (r - M) * (WRA_FS + 1) + q;
        const float inv = r < M ? a.xinv[(size_t)b * M + r] : a.yinv[(size_t)b * N + r - M];
        const bool clamped = inv >= 1.f / a.eps;                              // ||row|| <= eps: y = x / eps, no projection term
        const float dot = clamped ? 0.f : dots[r];
        bf16* out = dseq + ((size_t)b * a.S + r) * ldd + f0 + q;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (f0 + q + k < a.H) out[k] = (bf16)(inv * (acc[k] - self[k] * dot));
    }
}

}  // namespace

extern "C" {

int mmf_mse_ws_floats(void) { return 2 * MSE_BLOCKS; }
int mmf_mse_fwd(const float* pred, int ldp, const float* target, int ldt, const int64_t* row_label, float* loss, float* count, float* ws, int rows, int cols,
                void* stream) {
    MMF_CHECK_ARG(pred && target && loss && ws && rows > 0 && cols > 0 && ldp >= cols && ldt >= cols, "mse_fwd: bad operand");
    MMF_CHECK_ARG(!row_label || count, "mse_fwd: the masked form needs `count` (the denominator, reused by the backward)");
    hipLaunchKernelGGL(mse_partial_kernel, dim3(MSE_BLOCKS), dim3(256), 0, (hipStream_t)stream, pred, ldp, target, ldt, row_label, ws, rows, cols);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, loss, count, 1.f / ((float)rows * (float)cols), row_label ? 1 : 0);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_mse_bwd(const float* pred, int ldp, const float* target, int ldt, const int64_t* row_label, const float* count, const float* gloss, void* dpred, int ldd,
                int rows, int cols, void* stream) {
    MMF_CHECK_ARG(pred && target && dpred && rows > 0 && cols > 0 && ldd >= cols && (ldd % 8) == 0, "mse_bwd: bad operand (ldd % 8 == 0)");
    MMF_CHECK_ARG(!row_label || count, "mse_bwd: the masked form needs the forward's `count`");
    const int64_t n = (int64_t)rows * ldd;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(mse_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, ldp, target, ldt, row_label, count, gloss,
                       reinterpret_cast<bf16*>(dpred), ldd, rows, cols, 2.f / ((float)rows * (float)cols));
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_mse_f32_bwd(const float* pred, int ldp, const float* target, int ldt, const int64_t* row_label, const float* count, const float* gloss, float* dpred, int ldd,
                    int rows, int cols, void* stream) {
    MMF_CHECK_ARG(pred && target && dpred && rows > 0 && cols > 0 && ldd >= cols && (ldd % 4) == 0, "mse_f32_bwd: bad operand (ldd % 4 == 0)");
    MMF_CHECK_ARG(!row_label || count, "mse_f32_bwd: the masked form needs the forward's `count`");
    const int64_t n = (int64_t)rows * ldd;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(mse_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, ldp, target, ldt, row_label, count, gloss, dpred, ldd, rows,
                       cols, 2.f / ((float)rows * (float)cols));
    MMF_CHECK_LAUNCH();
    return 0;
}

static int wra_fill(WraArgs& a, const mmf_wra_desc* d) {
    MMF_CHECK_ARG(d && d->seq && d->txt_pad && d->img_pad && d->label && d->xinv && d->yinv && d->plan && d->cost && d->dist, "wra: null operand");
    MMF_CHECK_ARG(d->B > 0 && d->M > 0 && d->N > 0 && d->M <= WRA_MAX && d->N <= WRA_MAX, "wra: 1 .. 128 text rows and region rows per sample");
    MMF_CHECK_ARG(d->S >= d->M + d->N && d->H > 0 && d->ld >= d->H, "wra: sequence extents");
    a.seq = reinterpret_cast<const bf16*>(d->seq); a.ld = d->ld; a.S = d->S; a.H = d->H; a.M = d->M; a.N = d->N; a.B = d->B;
    a.txt_pad = d->txt_pad; a.img_pad = d->img_pad; a.label = d->label; a.xinv = d->xinv; a.yinv = d->yinv; a.plan = d->plan; a.cost = d->cost;
    a.dist = d->dist; a.beta = d->beta > 0.f ? d->beta : 0.5f; a.eps = d->eps > 0.f ? d->eps : 1e-5f; a.iters = d->iterations > 0 ? d->iterations : 50;
    return 0;
}
int mmf_wra_fwd(const mmf_wra_desc* d, float* loss, float* count, void* stream) {
    WraArgs a;
    if (int rc = wra_fill(a, d)) return rc;
    MMF_CHECK_ARG(loss && count, "wra_fwd: null loss / count");
    constexpr int lds = (2 * WRA_MAX * WRA_LD + 6 * WRA_MAX + 8) * (int)sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wra_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { mmf_amd_set_error(hipGetErrorString(e)); return 2; }
        attr = true;
    }
    hipLaunchKernelGGL(wra_fwd_kernel, dim3(a.B), dim3(256), lds, (hipStream_t)stream, a);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(wra_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a.dist, a.label, a.B, loss, count);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_wra_bwd(const mmf_wra_desc* d, const float* gloss, const float* count, void* dseq, int ldd, void* stream) {
    WraArgs a;
    if (int rc = wra_fill(a, d)) return rc;
    MMF_CHECK_ARG(count && dseq && ldd >= d->H, "wra_bwd: bad operand");
    constexpr int lds = (2 * WRA_MAX * (WRA_FS + 1) + 2 * WRA_MAX) * (int)sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wra_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { mmf_amd_set_error(hipGetErrorString(e)); return 2; }
        attr = true;
    }
    const dim3 grid(a.B, (a.H + WRA_FS - 1) / WRA_FS);
    hipLaunchKernelGGL(wra_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, a, gloss, count, reinterpret_cast<bf16*>(dseq), ldd);
    MMF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
