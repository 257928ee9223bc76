// mmf_amd :: the kernels ViLBERT's `dynamic_attention` adds around the visual self-attention
// (mmf/models/vilbert.py:174-176, 199-212):
//     pool   = (txt_embedding * txt_mask).sum(1) / txt_mask.sum(1)                 masked mean of the text stream   [B, H_t]
//     gate_q = 1 + sigmoid(dyLinear_q(pool)),  gate_k = 1 + sigmoid(dyLinear_k(pool))                                 [B, H_v]
//     q = query(x) * gate_q.unsqueeze(1),  k = key(x) * gate_k.unsqueeze(1)
// The two Linear layers are ordinary GEMM launches; here: the masked mean with its backward, and the per-sample column
// scale of the packed Q|K columns with its backward (gradient of the gate = column sums of dQ o Q over the sample's rows).
// A few MB per call at the VQA2 shape (B = 32, 101 regions x 1024, 128 tokens x 768): plain coalesced loops, fp32 sums.
#include "common.h"
#include "mmf_amd.h"

namespace {

// pool[b][c] = sum_t x[b][t][c] * mask[b][t] / sum_t mask[b][t]; one thread per column
__global__ __launch_bounds__(256) void masked_mean_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ mask,
                                                               float* __restrict__ pool, int T, int H) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const bf16* xb = x + (size_t)b * T * H + c;
    const float* mb = mask + (size_t)b * T;
    float acc = 0.f, cnt = 0.f;
    for (int t = 0; t < T; ++t) {
        const float m = mb[t];
        acc += (float)xb[(size_t)t * H] * m;
        cnt += m;
    }
    pool[(size_t)b * H + c] = acc / cnt;       // an all-zero mask divides by zero like the reference does
}
// dx[b][t][c] = dpool[b][c] * mask[b][t] / sum_t mask[b][t]
template <typename RT>   // bf16 rows (throughput path) or fp32 rows (mmf_amd.fp32_training())
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const float* __restrict__ dpool, const float* __restrict__ mask,
                                                               RT* __restrict__ dx, int T, int H) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const float* mb = mask + (size_t)b * T;
    float cnt = 0.f;
    for (int t = 0; t < T; ++t) cnt += mb[t];
    const float g = dpool[(size_t)b * H + c] / cnt;
    RT* db = dx + (size_t)b * T * H + c;
    for (int t = 0; t < T; ++t) db[(size_t)t * H] = (RT)(g * mb[t]);
}

// x[g * rpg + r][c] *= gate[g][c] for c < C (C % 8 == 0, ld % 8 == 0): 8 columns per thread
__global__ __launch_bounds__(256) void rowgroup_scale_kernel(bf16* __restrict__ x, int ld, const float* __restrict__ gate, int rpg,
                                                              int C, int rows) {
    const int per_row = C >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * per_row) return;
    const int row = (int)(i / per_row), c = (int)(i - (int64_t)row * per_row) << 3;
    const float* g = gate + (size_t)(row / rpg) * C + c;
    bf16x8* p = reinterpret_cast<bf16x8*>(x + (size_t)row * ld + c);
    bf16x8 v = *p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] * g[e]);
    *p = v;
}
// Backward of the scale, in place: with y = x * gate (y is what the attention saw, dy its gradient),
//   dgate[g][c] = sum_r dy[r][c] * x[r][c] = (sum_r dy[r][c] * y[r][c]) / gate[g][c],   dx = dy * gate  (written over dy).
// One workgroup per (group, 1024 columns): 128 threads x 8 columns, looping over the group's rows.
__global__ __launch_bounds__(128) void rowgroup_scale_bwd_kernel(bf16* __restrict__ dy, const bf16* __restrict__ y, int ld,
                                                                  const float* __restrict__ gate, float* __restrict__ dgate,
                                                                  int rpg, int C) {
    const int g = blockIdx.y, c = (blockIdx.x * 128 + threadIdx.x) << 3;
    if (c >= C) return;
    float gt[8], acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gt[e] = gate[(size_t)g * C + c + e]; acc[e] = 0.f; }
    for (int r = 0; r < rpg; ++r) {
        const size_t off = (size_t)(g * rpg + r) * ld + c;
        bf16x8 d = *reinterpret_cast<const bf16x8*>(dy + off);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] += (float)d[e] * (float)v[e];
            d[e] = (bf16)((float)d[e] * gt[e]);
        }
        *reinterpret_cast<bf16x8*>(dy + off) = d;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dgate[(size_t)g * C + c + e] = acc[e] / gt[e];
}

// gate[b][c] = 1 + sigmoid(z[b][c]) written into the packed [B, ldg] gate at column offset `col0` (the Q half or the K half);
// backward: dz = dgate * s (1 - s) with s = gate - 1   (vilbert.py:206-209: 1 + sigmoid(dyLinear_{q,k}(pool)))
__global__ __launch_bounds__(256) void gate_sigmoid_fwd_kernel(const float* __restrict__ z, float* __restrict__ gate, int ldg, int col0, int C, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / C;
    const int c = (int)(i - b * C);
    gate[b * ldg + col0 + c] = 1.f + 1.f / (1.f + __expf(-z[i]));
}
__global__ __launch_bounds__(256) void gate_sigmoid_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate, int ldg, int col0, int C,
                                                                float* __restrict__ dz, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / C;
    const int c = (int)(i - b * C);
    const float s = gate[b * ldg + col0 + c] - 1.f;
    dz[i] = dgate[b * ldg + col0 + c] * s * (1.f - s);
}

// the same backward of the per-sample column scale on fp32 rows: one thread per column, looping over the group's rows
__global__ __launch_bounds__(256) void rowgroup_scale_f32_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, int ld,
                                                                      const float* __restrict__ gate, float* __restrict__ dgate, int rpg, int C) {
    const int g = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float gt = gate[(size_t)g * C + c];
    float acc = 0.f;
    for (int r = 0; r < rpg; ++r) {
        const size_t off = (size_t)(g * rpg + r) * ld + c;
        const float d = dy[off];
        acc += d * y[off];
        dy[off] = d * gt;
    }
    dgate[(size_t)g * C + c] = acc / gt;
}

}  // namespace


// ------------------------------------------------------------------------------------------------
// ViLBERT masked-region NCE loss (`visual_target: 2`, mmf/models/vilbert.py:1158-1227): for every region r with image_label == 1 the
// prediction pred[r] is scored against its own target feature and K negative target features picked by flat indices into the [B * R]
// regions, score[r][j] = <sample_j, pred[r]> (torch.bmm, :1221), loss = CrossEntropyLoss(score, class 0) = mean over those regions of
// (logsumexp_j score - score_0).  One workgroup per region: a wave per sample for the dot products (rows of N fp32, 16-byte loads),
// the K + 1 scores stay in LDS for the log-sum-exp; backward: d pred[r] = g / count * (sum_j softmax_j sample_j - sample_0), written as
// the zero-padded bf16 operand of the image-prediction decoder's gradient GEMMs (rows without label: zeros).
// ------------------------------------------------------------------------------------------------
constexpr int NCE_MAXK = 1024;
DEVI f32x4 nce_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DEVI void nce_store4(bf16* p, f32x4 v) { p[0] = (bf16)v[0]; p[1] = (bf16)v[1]; p[2] = (bf16)v[2]; p[3] = (bf16)v[3]; }
DEVI void nce_store4(float* p, f32x4 v) { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3]; }
__global__ __launch_bounds__(256) void nce_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const int64_t* __restrict__ neg,
                                                       const int64_t* __restrict__ label, float* __restrict__ scores, float* __restrict__ lse,
                                                       float* __restrict__ rowloss, int M, int N, int K) {
    __shared__ float sc[NCE_MAXK + 1];
    __shared__ float red[4];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (label[r] != 1) { if (threadIdx.x == 0) { rowloss[r] = 0.f; lse[r] = 0.f; } return; }      // (uniform per workgroup)
    const float* p = pred + (size_t)r * N;
    for (int j = wave; j <= K; j += 4) {
        long src = j == 0 ? r : neg[(size_t)r * K + (j - 1)];
        src = src < 0 ? 0 : (src >= M ? M - 1 : src);
        const float* t = target + (size_t)src * N;
        float s = 0.f;
        for (int c = lane * 4; c < N; c += 256) {
            const f32x4 a = nce_load4(p + c), b = nce_load4(t + c);
            s += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
        }
        s = wave_sum(s);
        if (lane == 0) { sc[j] = s; scores[(size_t)r * (K + 1) + j] = s; }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = threadIdx.x; j <= K; j += 256) m = fmaxf(m, sc[j]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float e = 0.f;
    for (int j = threadIdx.x; j <= K; j += 256) e += expf(sc[j] - m);
    e = wave_sum(e);
    if (lane == 0) red[wave] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l = m + logf((red[0] + red[1]) + (red[2] + red[3]));
        lse[r] = l;
        rowloss[r] = l - sc[0];
    }
}
__global__ __launch_bounds__(256) void nce_finalize_kernel(const float* __restrict__ rowloss, const int64_t* __restrict__ label, float* __restrict__ loss,
                                                            float* __restrict__ count, int M) {
    __shared__ float rs[4], rc[4];
    float s = 0.f, c = 0.f;
    for (int r = threadIdx.x; r < M; r += 256) if (label[r] == 1) { s += rowloss[r]; c += 1.f; }
    s = wave_sum(s); c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float cs = (rc[0] + rc[1]) + (rc[2] + rc[3]);
        count[0] = cs;
        loss[0] = ((rs[0] + rs[1]) + (rs[2] + rs[3])) / cs;        // no labelled region: 0 / 0 = NaN, like CrossEntropyLoss over nothing
    }
}
template <typename RT>   // bf16: the GEMM operand of the throughput path; float: mmf_amd.fp32_training()
__global__ __launch_bounds__(256) void nce_bwd_kernel(const float* __restrict__ target, const int64_t* __restrict__ neg, const int64_t* __restrict__ label,
                                                       const float* __restrict__ scores, const float* __restrict__ lse, const float* __restrict__ count,
                                                       const float* __restrict__ gloss, RT* __restrict__ d, int ldd, int M, int N, int K) {
    __shared__ float w[NCE_MAXK + 1];
    __shared__ long srcs[NCE_MAXK + 1];
    const int r = blockIdx.x;
    RT* dr = d + (size_t)r * ldd;
    if (label[r] != 1) {
        for (int c = threadIdx.x; c < ldd; c += 256) dr[c] = (RT)0.f;
        return;
    }
    const float g = gloss[0] / count[0], l = lse[r];
    for (int j = threadIdx.x; j <= K; j += 256) {
        w[j] = g * (expf(scores[(size_t)r * (K + 1) + j] - l) - (j == 0 ? 1.f : 0.f));
        long src = j == 0 ? r : neg[(size_t)r * K + (j - 1)];
        srcs[j] = src < 0 ? 0 : (src >= M ? M - 1 : src);
    }
    __syncthreads();
    for (int c = threadIdx.x * 4; c < ldd; c += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (c < N) {
            for (int j = 0; j <= K; ++j) acc += w[j] * nce_load4(target + (size_t)srcs[j] * N + c);
        }
        nce_store4(dr + c, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// ViLBERT `in_batch_pairs` / `fast_mode` batch expansion (mmf/models/vilbert.py:678-725): every text of the batch against every image.
//   mode 0 (the image side, `x.unsqueeze(0).expand(reps, ...)`):  out[i * Bs + j] = x[j]      (the whole batch tiled `reps` times)
//   mode 1 (the text side,  `x.unsqueeze(1).expand(.., reps, ...)`): out[i * reps + j] = x[i]  (every sample repeated `reps` times)
// n = elements per sample; 16-byte accesses.  Backward: the sum over the broadcast index, accumulated in fp32.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_batch_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, long Bs, long reps, long n8, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;            // over reps * Bs * n8 chunks of 8 elements
    if (i >= reps * Bs * n8) return;
    const long sample = i / n8, c = i - sample * n8;
    const long src = mode == 0 ? sample % Bs : sample / reps;
    reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(x)[src * n8 + c];
}
// fp32 activations (mmf_amd.fp32_training() / fp32_inference(): the reference's default arithmetic): 16-byte chunks of four floats
__global__ __launch_bounds__(256) void expand_batch_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long Bs, long reps, long n4, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= reps * Bs * n4) return;
    const long sample = i / n4, c = i - sample * n4;
    const long src = mode == 0 ? sample % Bs : sample / reps;
    reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(x)[src * n4 + c];
}
__global__ __launch_bounds__(256) void reduce_batch_f32_kernel(const float* __restrict__ g, float* __restrict__ dx, long Bs, long reps, long n4, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Bs * n4) return;
    const long s = i / n4, c = i - s * n4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long r = 0; r < reps; ++r) {       // fixed order: the sum torch.expand's backward forms, reproducible
        const long sample = mode == 0 ? r * Bs + s : s * reps + r;
        acc += reinterpret_cast<const f32x4*>(g)[sample * n4 + c];
    }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
}
__global__ __launch_bounds__(256) void reduce_batch_kernel(const bf16* __restrict__ g, bf16* __restrict__ dx, long Bs, long reps, long n8, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;            // over Bs * n8 chunks
    if (i >= Bs * n8) return;
    const long s = i / n8, c = i - s * n8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long r = 0; r < reps; ++r) {
        const long sample = mode == 0 ? r * Bs + s : s * reps + r;
        const uint4 v = reinterpret_cast<const uint4*>(g)[sample * n8 + c];
        const bf16* e = reinterpret_cast<const bf16*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (float)e[k];
    }
    uint4 o;
    bf16* e = reinterpret_cast<bf16*>(&o);
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = (bf16)acc[k];
    reinterpret_cast<uint4*>(dx)[i] = o;
}

extern "C" {

int mmf_gate_sigmoid_fwd(const float* z, float* gate, int ldg, int col0, int B, int C, void* stream) {
    MMF_CHECK_ARG(z && gate && B > 0 && C > 0 && col0 >= 0 && col0 + C <= ldg, "gate_sigmoid_fwd: bad operand");
    const int64_t n = (int64_t)B * C;
    hipLaunchKernelGGL(gate_sigmoid_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, gate, ldg, col0, C, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_gate_sigmoid_bwd(const float* dgate, const float* gate, int ldg, int col0, float* dz, int B, int C, void* stream) {
    MMF_CHECK_ARG(dgate && gate && dz && B > 0 && C > 0 && col0 >= 0 && col0 + C <= ldg, "gate_sigmoid_bwd: bad operand");
    const int64_t n = (int64_t)B * C;
    hipLaunchKernelGGL(gate_sigmoid_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dgate, gate, ldg, col0, C, dz, n);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_masked_mean_fwd(const void* x, const float* mask, float* pool, int B, int T, int H, void* stream) {
    MMF_CHECK_ARG(x && mask && pool && B > 0 && T > 0 && H > 0, "masked_mean_fwd: bad operand");
    hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, mask, pool, T, H);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_masked_mean_bwd(const float* dpool, const float* mask, void* dx, int B, int T, int H, void* stream) {
    MMF_CHECK_ARG(dpool && mask && dx && B > 0 && T > 0 && H > 0, "masked_mean_bwd: bad operand");
    hipLaunchKernelGGL(masked_mean_bwd_kernel<bf16>, dim3((H + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dpool, mask, (bf16*)dx, T, H);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_masked_mean_f32_bwd(const float* dpool, const float* mask, float* dx, int B, int T, int H, void* stream) {
    MMF_CHECK_ARG(dpool && mask && dx && B > 0 && T > 0 && H > 0, "masked_mean_f32_bwd: bad operand");
    hipLaunchKernelGGL(masked_mean_bwd_kernel<float>, dim3((H + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dpool, mask, dx, T, H);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_rowgroup_scale_f32_bwd(float* dy, const float* y, int ld, const float* gate, float* dgate, int groups, int rows_per_group, int C, void* stream) {
    MMF_CHECK_ARG(dy && y && gate && dgate && groups > 0 && rows_per_group > 0 && C > 0 && C <= ld, "rowgroup_scale_f32_bwd: bad operand");
    hipLaunchKernelGGL(rowgroup_scale_f32_bwd_kernel, dim3((C + 255) / 256, groups), dim3(256), 0, (hipStream_t)stream, dy, y, ld, gate, dgate,
                       rows_per_group, C);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_rowgroup_scale(void* x, int ld, const float* gate, int groups, int rows_per_group, int C, void* stream) {
    MMF_CHECK_ARG(x && gate && groups > 0 && rows_per_group > 0 && C > 0, "rowgroup_scale: bad operand");
    MMF_CHECK_ARG((C % 8) == 0 && (ld % 8) == 0 && C <= ld, "rowgroup_scale: C and ld must be multiples of 8 with C <= ld");
    const int rows = groups * rows_per_group;
    const int64_t n = (int64_t)rows * (C / 8);
    hipLaunchKernelGGL(rowgroup_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16*)x, ld, gate,
                       rows_per_group, C, rows);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_rowgroup_scale_bwd(void* dy, const void* y, int ld, const float* gate, float* dgate, int groups, int rows_per_group, int C,
                           void* stream) {
    MMF_CHECK_ARG(dy && y && gate && dgate && groups > 0 && rows_per_group > 0 && C > 0, "rowgroup_scale_bwd: bad operand");
    MMF_CHECK_ARG((C % 8) == 0 && (ld % 8) == 0 && C <= ld, "rowgroup_scale_bwd: C and ld must be multiples of 8 with C <= ld");
    hipLaunchKernelGGL(rowgroup_scale_bwd_kernel, dim3((C / 8 + 127) / 128, groups), dim3(128), 0, (hipStream_t)stream, (bf16*)dy,
                       (const bf16*)y, ld, gate, dgate, rows_per_group, C);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_nce_fwd(const float* pred, const float* target, const int64_t* neg, const int64_t* label, float* scores, float* lse, float* rowloss, float* loss,
                float* count, int M, int N, int K, void* stream) {
    MMF_CHECK_ARG(pred && target && neg && label && scores && lse && rowloss && loss && count, "nce_fwd: null operand");
    MMF_CHECK_ARG(M > 0 && N > 0 && (N % 4) == 0 && K > 0 && K <= NCE_MAXK, "nce_fwd: N % 4 == 0, 1 <= K <= 1024");
    hipLaunchKernelGGL(nce_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, pred, target, neg, label, scores, lse, rowloss, M, N, K);
    hipLaunchKernelGGL(nce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rowloss, label, loss, count, M);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_nce_bwd(const float* target, const int64_t* neg, const int64_t* label, const float* scores, const float* lse, const float* count, const float* gloss,
                void* dpred, int ldd, int M, int N, int K, void* stream) {
    MMF_CHECK_ARG(target && neg && label && scores && lse && count && gloss && dpred, "nce_bwd: null operand");
    MMF_CHECK_ARG(M > 0 && N > 0 && (N % 4) == 0 && K > 0 && K <= NCE_MAXK && ldd >= N && (ldd % 8) == 0, "nce_bwd: bad shape (ldd: a multiple of 8 covering N)");
    hipLaunchKernelGGL(nce_bwd_kernel<bf16>, dim3(M), dim3(256), 0, (hipStream_t)stream, target, neg, label, scores, lse, count, gloss, (bf16*)dpred, ldd, M, N, K);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_nce_f32_bwd(const float* target, const int64_t* neg, const int64_t* label, const float* scores, const float* lse, const float* count, const float* gloss,
                    float* dpred, int ldd, int M, int N, int K, void* stream) {
    MMF_CHECK_ARG(target && neg && label && scores && lse && count && gloss && dpred, "nce_f32_bwd: null operand");
    MMF_CHECK_ARG(M > 0 && N > 0 && (N % 4) == 0 && K > 0 && K <= NCE_MAXK && ldd >= N && (ldd % 4) == 0, "nce_f32_bwd: bad shape (ldd: a multiple of 4 covering N)");
    hipLaunchKernelGGL(nce_bwd_kernel<float>, dim3(M), dim3(256), 0, (hipStream_t)stream, target, neg, label, scores, lse, count, gloss, dpred, ldd, M, N, K);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_expand_batch_bf16(const void* x, void* out, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream) {
    MMF_CHECK_ARG(x && out && Bs > 0 && reps > 0 && n > 0 && (n % 8) == 0 && (mode == 0 || mode == 1), "expand_batch: bad operand (n % 8 == 0, mode 0 / 1)");
    const long total = reps * Bs * (n / 8);
    hipLaunchKernelGGL(expand_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, (long)Bs,
                       (long)reps, (long)(n / 8), mode);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_expand_batch_f32(const float* x, float* out, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream) {
    MMF_CHECK_ARG(x && out && Bs > 0 && reps > 0 && n > 0 && (n % 4) == 0 && (mode == 0 || mode == 1), "expand_batch_f32: bad operand (n % 4 == 0, mode 0 / 1)");
    const long total = reps * Bs * (n / 4);
    hipLaunchKernelGGL(expand_batch_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long)Bs, (long)reps, (long)(n / 4), mode);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_reduce_batch_f32(const float* g, float* dx, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream) {
    MMF_CHECK_ARG(g && dx && Bs > 0 && reps > 0 && n > 0 && (n % 4) == 0 && (mode == 0 || mode == 1), "reduce_batch_f32: bad operand (n % 4 == 0, mode 0 / 1)");
    const long total = Bs * (n / 4);
    hipLaunchKernelGGL(reduce_batch_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, dx, (long)Bs, (long)reps, (long)(n / 4), mode);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_reduce_batch_bf16(const void* g, void* dx, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream) {
    MMF_CHECK_ARG(g && dx && Bs > 0 && reps > 0 && n > 0 && (n % 8) == 0 && (mode == 0 || mode == 1), "reduce_batch: bad operand (n % 8 == 0, mode 0 / 1)");
    const long total = Bs * (n / 8);
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)g, (bf16*)dx, (long)Bs,
                       (long)reps, (long)(n / 8), mode);
    MMF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
