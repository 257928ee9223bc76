// mmf_amd fp32 TRAINING kernels (round 3): what the fp32-accurate path (fp32_path.hip) needs beyond its forward kernels to run
// the reference's default arithmetic — fp32 forward AND backward (mmf/trainers/core/training_loop.py:199-211: autocast only under
// `training.fp16`) — for the VisualBERT training step.  The contractions are mmf_gemm_f32 (forward / dgrad / weight-gradient
// layouts) and mmf_attention_f32_fwd / _bwd in fp32_path.hip; here:
//   mmf_layernorm_f32_bwd     autograd of nn.LayerNorm (hf_layers.py:248,290; embeddings.py:456; visual_bert.py:328)
//   mmf_colsum_f32            bias gradients (autograd of `+ bias`, hf_layers.py:169-180) and the LayerNorm column sums
//   mmf_dropout_f32           nn.Dropout forward / backward (hf_layers.py:249,291; embeddings.py:457; visual_bert.py:400) on fp32 rows
//   mmf_scatter_add_rows_f32  autograd of the embedding gathers (embeddings.py:339-343) and of the pooling gather (visual_bert.py:389-398)
// All memory-bound row kernels: 16-byte accesses, one wave per row where a row reduction is needed.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmf_amd.h"
#include "common.h"

namespace {

// dx = rstd (gamma dy - mean_j(gamma dy) - xhat mean_j(gamma dy xhat)); per-workgroup partial column sums of dy xhat (-> dgamma) and
// dy (-> dbeta) in partials[blk][2][H].  One wave per row, the row in registers (H <= 2048), a workgroup walks rows blk*4 + w, + 4 grid.
__global__ __launch_bounds__(256) void ln_f32_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma, float* __restrict__ dx,
                                                          float* __restrict__ partials, int rows, int H) {
    __shared__ float red[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[8], ab[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) ag[c] = ab[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.x * 4 + wave; row < rows; row += 4 * gridDim.x) {
        const float* xr = x + (size_t)row * H;
        const float* dr = dy + (size_t)row * H;
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[8], gd[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int col = (lane + 64 * c) * 4;
            if (col < H) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + col), dv = *reinterpret_cast<const f32x4*>(dr + col);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[c][e] = (xv[e] - mu) * rs;
                    gd[c][e] = dv[e] * gm[e];
                    s1 += gd[c][e];
                    s2 += gd[c][e] * xh[c][e];
                    ag[c][e] += dv[e] * xh[c][e];
                    ab[c][e] += dv[e];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
        float* or_ = dx + (size_t)row * H;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int col = (lane + 64 * c) * 4;
            if (col < H) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (gd[c][e] - c1 - xh[c][e] * c2);
                *reinterpret_cast<f32x4*>(or_ + col) = o;
            }
        }
    }
#pragma unroll 1
    for (int qn = 0; qn < 2; ++qn) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int col = (lane + 64 * c) * 4;
            if (col < H) *reinterpret_cast<f32x4*>(&red[wave][col]) = qn == 0 ? ag[c] : ab[c];
        }
        __syncthreads();
        for (int col = threadIdx.x; col < H; col += 256)
            partials[((size_t)blockIdx.x * 2 + qn) * H + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    }
}

// out[rb][n] = sum over the rows of slice rb of x[row][n]: thread = column, coalesced across the workgroup; fixed order
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, int ld, int rows, int N, int rows_per_blk, float* __restrict__ out, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        a0 += x[(size_t)r * ld + n]; a1 += x[(size_t)(r + 1) * ld + n]; a2 += x[(size_t)(r + 2) * ld + n]; a3 += x[(size_t)(r + 3) * ld + n];
    }
    for (; r < r1; ++r) a0 += x[(size_t)r * ld + n];
    float* o = out + (size_t)blockIdx.y * N + n;
    const float t = (a0 + a1) + (a2 + a3);
    *o = accumulate ? *o + t : t;
}

__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, long n, DropoutCfg drop) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint32_t key = drop_key(drop);
    const long e0 = i * 4;
    if (e0 + 3 < n) {
        const f32x4 sc = drop_scale4(key, (uint32_t)e0, drop.thr16, drop.scale);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + e0);
        v *= sc;
        *reinterpret_cast<f32x4*>(y + e0) = v;
    } else {
        for (long e = e0; e < n; ++e) y[e] = x[e] * drop_scale1(key, (uint32_t)e, drop.thr16, drop.scale);
    }
}

// out[dst(r)][:] += g[src(r)][:] for r < rows:  src(r) = (r / grp) * grp_stride + grp_off + r % grp  (grp > 0: the text or visual rows of
// the joint [B, S, H] buffer), dst(r) = idx[r] (+ r * dst_stride: the pooling gather's row b S + index[b]); rows with idx == skip
// (the padding row of nn.Embedding) are dropped.  fp32 atomics: few rows collide, summation order across colliding rows is free.
__global__ __launch_bounds__(256) void scatter_add_rows_f32_kernel(const float* __restrict__ g, int ld, int rows, int H, int grp, int grp_stride, int grp_off,
                                                                    const int64_t* __restrict__ idx, long dst_stride, long skip, int NT,
                                                                    float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long ix = idx[r];
    if (ix == skip) return;
    const long drow = ix + (long)r * dst_stride;
    if (ix < 0 || (dst_stride == 0 && ix >= NT)) return;     // (the forward gather of the same indices has already flagged them: mmf_amd_take_index_error)
    const size_t srow = grp > 0 ? (size_t)(r / grp) * grp_stride + grp_off + (r % grp) : (size_t)r;
    const float* src = g + srow * ld;
    float* dst = out + (size_t)drow * ldo;
    for (int col = lane * 4; col < H; col += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + col + e, v[e]);
    }
}

// Touched-row exchange of an embedding-table gradient (data parallelism, mmf/trainers/core/device.py:104-110 reduces the dense table): the ranks
// exchange only the rows their batches touched — ids [M] and rows [M, H] gathered over all ranks — and every rank rebuilds the SUM with this kernel
// from the ids in STABLE-sorted order: one wave per sorted position that starts a segment (the previous id differs) walks its segment and adds
// the rows in that fixed order — no atomics, so every rank computes bit-identical sums — and writes out[id] = sum (rows nobody touched are
// left alone: the local dense gradient holds zeros there).  ids < 0 mark duplicates removed on the sending rank; they sort first and are skipped.
__global__ __launch_bounds__(256) void segment_sum_rows_f32_kernel(const int64_t* __restrict__ sorted_ids, const int64_t* __restrict__ perm,
                                                                    const float* __restrict__ rows, float* __restrict__ out, int M, int H, long V) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    const long id = sorted_ids[i];
    if (id < 0 || id >= V) return;
    if (i > 0 && sorted_ids[i - 1] == id) return;
    for (int col = lane * 4; col < H; col += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = i; j < M && sorted_ids[j] == id; ++j) acc += *reinterpret_cast<const f32x4*>(rows + (size_t)perm[j] * H + col);
        *reinterpret_cast<f32x4*>(out + (size_t)id * H + col) = acc;
    }
}

// d[b][n] = gloss (sigmoid(x) - t) / B: autograd of mean(BCEWithLogits) * num_labels (losses.py:246-251), fp32 out
__global__ __launch_bounds__(256) void bce_f32_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ gloss,
                                                           float* __restrict__ d, long n, int B) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    d[i] = gloss[0] * (1.f / (1.f + expf(-x[i])) - t[i]) / (float)B;
}

}  // namespace

extern "C" {

int mmf_segment_sum_rows_f32(const int64_t* sorted_ids, const int64_t* perm, const float* rows, float* out, int M, int H, int64_t V, void* stream) {
    MMF_CHECK_ARG(sorted_ids && perm && rows && out && M > 0 && H > 0 && (H % 4) == 0 && V > 0, "segment_sum_rows_f32: bad operand");
    MMF_CHECK_ARG((((uintptr_t)rows | (uintptr_t)out) & 15) == 0, "segment_sum_rows_f32: 16-byte alignment");
    hipLaunchKernelGGL(segment_sum_rows_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, sorted_ids, perm, rows, out, M, H, (long)V);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_bce_logits_f32_bwd(const float* scores, const float* targets, const float* gloss, float* dscores, int B, int N, void* stream) {
    MMF_CHECK_ARG(scores && targets && gloss && dscores && B > 0 && N > 0, "bce_logits_f32_bwd: bad operand");
    const long n = (long)B * N;
    hipLaunchKernelGGL(bce_f32_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, targets, gloss, dscores, n, B);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_layernorm_f32_bwd_blocks(int rows) { const int b = (rows + 3) / 4; return b < 512 ? b : 512; }

int mmf_layernorm_f32_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma,
                          float* dbeta, float* partials, int rows, int H, void* stream) {
    MMF_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && partials, "layernorm_f32_bwd: null operand");
    MMF_CHECK_ARG(rows > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "layernorm_f32_bwd: H % 4 == 0, H <= 2048");
    MMF_CHECK_ARG((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma) & 15) == 0, "layernorm_f32_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const int grid = mmf_layernorm_f32_bwd_blocks(rows);
    hipLaunchKernelGGL(ln_f32_bwd_kernel, dim3(grid), dim3(256), 0, s, dy, x, mean, rstd, gamma, dx, partials, rows, H);
    MMF_CHECK_LAUNCH();
    // partials is [grid][2][H] = a [grid, 2H] matrix: its column sums are dgamma | dbeta
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((H + 255) / 256, 1), dim3(256), 0, s, partials, 2 * H, grid, H, grid, dgamma, 0);
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((H + 255) / 256, 1), dim3(256), 0, s, partials + H, 2 * H, grid, H, grid, dbeta, 0);
    MMF_CHECK_LAUNCH();
    return 0;
}

// out[n] (+)= sum_rows x[row][n]; `ws` (>= mmf_colsum_f32_slices(rows) * N floats) holds the per-slice sums of the first pass
int mmf_colsum_f32_slices(int rows) { const int s = (rows + 255) / 256; return s < 1 ? 1 : (s > 64 ? 64 : s); }
int mmf_colsum_f32(const float* x, int ld, int rows, int N, float* out, int accumulate, float* ws, void* stream) {
    MMF_CHECK_ARG(x && out && ws && rows > 0 && N > 0 && ld >= N, "colsum_f32: bad operand");
    hipStream_t s = (hipStream_t)stream;
    const int slices = mmf_colsum_f32_slices(rows);
    const int rpb = (rows + slices - 1) / slices;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 255) / 256, slices), dim3(256), 0, s, x, ld, rows, N, rpb, ws, 0);
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 255) / 256, 1), dim3(256), 0, s, ws, N, slices, N, slices, out, accumulate);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_dropout_f32(const float* x, float* y, long n, uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, void* stream) {
    MMF_CHECK_ARG(x && y && n > 0 && drop_thr16 > 0, "dropout_f32: bad operand (thr16 == 0 means no dropout: do not call)");
    MMF_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "dropout_f32: 16-byte alignment");
    const long n4 = (n + 3) / 4;
    hipLaunchKernelGGL(dropout_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n4, n,
                       DropoutCfg{drop_key, drop_thr16, drop_scale, drop_seed});
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_scatter_add_rows_f32(const float* g, int ld, int rows, int H, int grp, int grp_stride, int grp_off, const int64_t* idx, long dst_stride,
                             long skip, int NT, float* out, int ldo, void* stream) {
    MMF_CHECK_ARG(g && idx && out && rows > 0 && H > 0 && (H % 4) == 0 && ld >= H && ldo >= H && (ld % 4) == 0, "scatter_add_rows_f32: bad operand");
    MMF_CHECK_ARG((((uintptr_t)g) & 15) == 0, "scatter_add_rows_f32: 16-byte alignment");
    hipLaunchKernelGGL(scatter_add_rows_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, ld, rows, H, grp, grp_stride, grp_off, idx,
                       dst_stride, skip, NT, out, ldo);
    MMF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
