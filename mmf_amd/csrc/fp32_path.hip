// mmf_amd :: the fp32-accurate forward path (gfx950).
//
// The reference's default arithmetic is fp32 (mmf/trainers/core/training_loop.py:199: autocast only under
// `training.fp16`), and BASELINE.json's north_star asks for outputs within 1e-3 of it in fp32.  The bf16 kernels of
// gemm.hip / attention.hip / rowops.hip are the throughput path (5e-2 bound); the kernels here keep every activation
// in fp32 and contract on the fp32-input matrix cores (`v_mfma_f32_32x32x2_f32`: exact fp32 products, fp32 accumulate,
// bitwise a k-ordered fmaf chain, 157 TFLOP/s dense peak = 1/16 of the bf16 MFMA rate), so the same models evaluate to
// fp32 round-off of the reference.  Forward only: evaluation / inference and parity checking, not training.
//
//   mmf_gemm_f32           nn.Linear forward + fused bias / table adds / GELU / tanh / residual   (hf_layers.py:169-180,
//                          HF BertSelfOutput / BertIntermediate / BertOutput at hf_layers.py:248,289,290, embeddings.py:352,
//                          visual_bert.py:146,328-330)
//   mmf_attention_f32_fwd  BertSelfAttentionJit.forward without the [B,A,S,S] tensors               (hf_layers.py:161-213)
//   mmf_layernorm_f32_fwd  nn.LayerNorm(eps=1e-12)                                                   (hf_layers.py:248,290, embeddings.py:456)
#include "common.h"
#include "mmf_amd.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------------
// GEMM: C[m][n] = epilogue(sum_k A[m][k] * B[n][k]), everything fp32.
// 128x128 (or 64x128) tile per 256-thread workgroup, BK = 16; wave w owns a 64x64 (32x64) quadrant = 2x2 (1x2) MFMA tiles of 32x32.  Operands go global -> registers (16-byte loads along K) -> LDS as K-MAJOR images [k][row] (row stride 132
// floats: the four k-quads of a wave's stores land in disjoint bank groups), so that the MFMA operand of lane l —
// A[row = l & 31][k = l >> 5] — is a conflict-free ds_read_b32 of consecutive rows.  Register double buffering: the next
// K-step's global loads are in flight while the MFMAs of the current one run; one barrier per K-step.
// ------------------------------------------------------------------------------------------------
constexpr int GBM = 128, GBN = 128, GBK = 16, GLD = 132;

struct GemmF32 {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias; const float* coladd; const float* rowtab; const int64_t* rowidx; int rowtab_ld;
    int act;
    const float* resid; int ldr;
    int grp_in, grp_pad, grp_off;
};

DEVI float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

DEVI f32x4 ld_tile4(const float* base, int row, int rows, int ld, int k, int K) {
    // 4 consecutive k of one row, zero outside the matrix (K % 4 == 0 and ld % 4 == 0 are checked by the host)
    if (row < rows && k < K) return *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + k);
    return f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int MI>     // MI row tiles of 32 per wave: BM = 64 * MI (128x128 tiles, or 64x128 when 128-row tiles would leave CUs idle)
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32 g) {
    constexpr int BM = 64 * MI;
    __shared__ float As[2][GBK][GLD];
    __shared__ float Bs[2][GBK][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * GBN;
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * 64;
    // staging role of this thread: rows r0 (and r0 + 64) of the tile, k-quad kq
    const int r0 = tid >> 2, kq = (tid & 3) * 4;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (g.K + GBK - 1) / GBK;
    f32x4 ra[MI], rb[2];
#pragma unroll
    for (int h = 0; h < MI; ++h) ra[h] = ld_tile4(g.A, m0 + r0 + 64 * h, g.M, g.lda, kq, g.K);
    rb[0] = ld_tile4(g.B, n0 + r0, g.N, g.ldb, kq, g.K);
    rb[1] = ld_tile4(g.B, n0 + r0 + 64, g.N, g.ldb, kq, g.K);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int h = 0; h < MI; ++h) As[0][kq + j][r0 + 64 * h] = ra[h][j];
#pragma unroll
        for (int h = 0; h < 2; ++h) Bs[0][kq + j][r0 + 64 * h] = rb[h][j];
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            const int k = (kt + 1) * GBK + kq;
#pragma unroll
            for (int h = 0; h < MI; ++h) ra[h] = ld_tile4(g.A, m0 + r0 + 64 * h, g.M, g.lda, k, g.K);
            rb[0] = ld_tile4(g.B, n0 + r0, g.N, g.ldb, k, g.K);
            rb[1] = ld_tile4(g.B, n0 + r0 + 64, g.N, g.ldb, k, g.K);
        }
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            float a[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = As[cur][k][wm + 32 * i + (lane & 31)];
            const float b0 = Bs[cur][k][wn + (lane & 31)];
            const float b1 = Bs[cur][k][wn + 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b1, acc[i][1], 0, 0, 0);
            }
        }
        if (more) {
            const int nxt = cur ^ 1;   // last read in iteration kt - 1, which every wave left through the barrier below
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int h = 0; h < MI; ++h) As[nxt][kq + j][r0 + 64 * h] = ra[h][j];
#pragma unroll
                for (int h = 0; h < 2; ++h) Bs[nxt][kq + j][r0 + 64 * h] = rb[h][j];
            }
        }
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA — col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + (lane & 31);
        if (n >= g.N) continue;
        float cadd = 0.f;
        if (g.bias) cadd += g.bias[n];
        if (g.coladd) cadd += g.coladd[n];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = acc[i][j][r] + cadd;
                if (g.rowtab) v += g.rowtab[(size_t)g.rowidx[m] * g.rowtab_ld + n];
                if (g.act == 1) v = gelu_exact(v);
                else if (g.act == 3) v = tanhf(v);
                if (g.resid) v += g.resid[(size_t)m * g.ldr + n];
                const int orow = g.grp_in > 0 ? m + (m / g.grp_in) * g.grp_pad + g.grp_off : m;
                g.C[(size_t)orow * g.ldc + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// attention forward, fp32.  Workgroup = one (batch, head) x 128 queries; wave = 32 queries x ALL keys (Sk <= 256):
// scores on the fp32 MFMA (A = Q rows, B = K rows), exact two-pass softmax in registers, the normalised probabilities
// of one 32-key tile go through a per-wave LDS patch to become the A operand of P.V (B = V rows, read straight from
// L2: lanes 0..31 of an MFMA operand read 128 contiguous bytes of one V row).
// ------------------------------------------------------------------------------------------------
struct AttnF32 {
    const float* q; const float* k; const float* v; float* o;
    int ldq, ldk, ldv, ldo;
    const float* mask;
    int B, heads, Sq, Sk;
    float scale;
};

constexpr int ATT_MAXT = 8;   // key tiles of 32: Sk <= 256

template <int D>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const AttnF32 a) {
    __shared__ float Ps[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int half = lane >> 5, l31 = lane & 31;
    const int nt = (a.Sk + 31) >> 5;

    // Q operand: A[i = query l31][k = e]: aq[j] = Q[q0 + l31][2 j + half]
    // (rows are read as 16-byte quads — each lane pulls its whole 4 D-byte row, lanes l and l + 32 the same one — and the lane keeps
    // the even (half 0) or odd (half 1) elements: half as many, four times as wide load instructions as element gathers)
    float aq[D / 2];
    {
        const int qr = min(q0 + l31, a.Sq - 1);
        const f32x4* qp = reinterpret_cast<const f32x4*>(a.q + ((size_t)b * a.Sq + qr) * a.ldq + h * D);
#pragma unroll
        for (int i = 0; i < D / 4; ++i) {
            const f32x4 t = qp[i];
            aq[2 * i] = half ? t[1] : t[0];
            aq[2 * i + 1] = half ? t[3] : t[2];
        }
    }

    f32x16 sc[ATT_MAXT];
#pragma unroll
    for (int t = 0; t < ATT_MAXT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
        if (t < nt) {
            const int key = 32 * t + l31;
            const int kr = min(key, a.Sk - 1);
            const f32x4* kp = reinterpret_cast<const f32x4*>(a.k + ((size_t)b * a.Sk + kr) * a.ldk + h * D);
            float bk[D / 2];
#pragma unroll
            for (int i = 0; i < D / 4; ++i) {
                const f32x4 t = kp[i];
                bk[2 * i] = half ? t[1] : t[0];
                bk[2 * i + 1] = half ? t[3] : t[2];
            }
#pragma unroll
            for (int j = 0; j < D / 2; ++j) sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j], bk[j], sc[t], 0, 0, 0);
            // this lane's column = key; rows = 16 queries
            const float madd = (key < a.Sk) ? (a.mask ? a.mask[(size_t)b * a.Sk + key] : 0.f) : -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[t][r] = sc[t][r] * a.scale + madd;
        }
    }

    // softmax over keys: row r of this lane's half lives in the 32 lanes of the half, across the nt tiles
    float inv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < ATT_MAXT; ++t) if (t < nt) m = fmaxf(m, sc[t][r]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < ATT_MAXT; ++t) if (t < nt) { const float p = expf(sc[t][r] - m); sc[t][r] = p; s += p; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        inv[r] = s;
    }

    f32x16 oc[D / 32];
#pragma unroll
    for (int eb = 0; eb < D / 32; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oc[eb][r] = 0.f;

#pragma unroll
    for (int t = 0; t < ATT_MAXT; ++t) {
        if (t < nt) {     // nt is uniform over the workgroup: every wave meets the same barriers
            __syncthreads();     // the patch's previous tile has been consumed
#pragma unroll
            for (int r = 0; r < 16; ++r) Ps[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = sc[t][r] / inv[r];
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int kin = 2 * kk + half;              // key inside the tile = the MFMA's k index of this lane
                const float pa = Ps[wave][l31][kin];        // A[i = query l31][k = key]
                const int kr = min(32 * t + kin, a.Sk - 1); // (keys past Sk carry probability 0)
                const float* vp = a.v + ((size_t)b * a.Sk + kr) * a.ldv + h * D + l31;
#pragma unroll
                for (int eb = 0; eb < D / 32; ++eb)
                    oc[eb] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, vp[32 * eb], oc[eb], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = q0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (qr < a.Sq) {
            float* op = a.o + ((size_t)b * a.Sq + qr) * a.ldo + h * D + l31;
#pragma unroll
            for (int eb = 0; eb < D / 32; ++eb) op[32 * eb] = oc[eb][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward, fp32 in / fp32 out: one wave per row, the row held in registers (H <= 2048), two-pass statistics
// (mean, then the biased variance of the centred values), eps inside the square root.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_f32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, int rows, int H,
                                                          float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * H;
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = (lane + 64 * c) * 4;
        if (col < H) { v[c] = *reinterpret_cast<const f32x4*>(xr + col); s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]); }
        else v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = (lane + 64 * c) * 4;
        if (col < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    float* yr = y + (size_t)row * H;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = (lane + 64 * c) * 4;
        if (col < H) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + col), bt = *reinterpret_cast<const f32x4*>(beta + col);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * gm[e] + bt[e];
            *reinterpret_cast<f32x4*>(yr + col) = o;
        }
    }
}

}  // namespace

extern "C" int mmf_gemm_f32(const mmf_gemm_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->A && d->B && d->C, "gemm_f32: null operand");
    MMF_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm_f32: empty problem");
    MMF_CHECK_ARG(d->a_f32 && d->b_f32 && d->out_f32, "gemm_f32: operands and output are fp32 (a_f32 = b_f32 = out_f32 = 1)");
    MMF_CHECK_ARG(!d->a_kmajor && !d->b_kmajor, "gemm_f32: forward form only (row operands)");
    MMF_CHECK_ARG((d->K % 4) == 0 && (d->lda % 4) == 0 && (d->ldb % 4) == 0 && d->lda >= d->K && d->ldb >= d->K,
                  "gemm_f32: K, lda, ldb must be multiples of 4 and lda, ldb >= K");
    MMF_CHECK_ARG((((uintptr_t)d->A | (uintptr_t)d->B) & 15) == 0, "gemm_f32: A and B must be 16-byte aligned");
    MMF_CHECK_ARG(d->ldc >= d->N, "gemm_f32: ldc < N");
    MMF_CHECK_ARG(d->act == 0 || d->act == 1 || d->act == 3, "gemm_f32: act must be 0 (none), 1 (gelu) or 3 (tanh)");
    MMF_CHECK_ARG(!d->U && !d->aux && d->drop_thr16 == 0 && !d->splitk_ws && !d->rowsum_out && d->beta == 0.f,
                  "gemm_f32: forward-only inference epilogue (no saved derivative, dropout, split-K, row sums or beta)");
    MMF_CHECK_ARG(!d->rowtab || (d->rowidx && d->rowtab_ld >= d->N), "gemm_f32: rowtab needs rowidx and rowtab_ld >= N");
    MMF_CHECK_ARG(!d->resid || d->ldr >= d->N, "gemm_f32: ldr < N");
    MMF_CHECK_ARG(d->grp_in >= 0, "gemm_f32: grp_in < 0");
    GemmF32 g;
    g.A = (const float*)d->A; g.B = (const float*)d->B; g.C = (float*)d->C;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
    g.bias = d->bias; g.coladd = d->coladd; g.rowtab = d->rowtab; g.rowidx = d->rowidx; g.rowtab_ld = d->rowtab_ld;
    g.act = d->act; g.resid = (const float*)d->resid; g.ldr = d->ldr;
    g.grp_in = d->grp_in; g.grp_pad = d->grp_pad; g.grp_off = d->grp_off;
    // 128-row tiles unless they would leave the chip short of work (fewer than two tiles per CU): then 64-row tiles — e.g.
    // M = 7296, N = 768: 342 tiles of 128x128 on 256 CUs (1.34 rounds) become 684 of 64x128
    const int nt = (d->N + GBN - 1) / GBN;
    const bool small = (long)nt * ((d->M + GBM - 1) / GBM) < 512;
    const int bm = small ? 64 : GBM;
    const dim3 grid(nt, (d->M + bm - 1) / bm);
    MMF_CHECK_ARG(grid.y <= 65535u, "gemm_f32: M too large for one launch");
    if (small) hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(gemm_f32_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, g);
    MMF_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmf_attention_f32_fwd(const mmf_attn_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->q && d->k && d->v && d->ctx, "attention_f32_fwd: null operand");
    MMF_CHECK_ARG(d->B > 0 && d->heads > 0 && d->Sq > 0 && d->Sk > 0, "attention_f32_fwd: empty problem");
    MMF_CHECK_ARG(d->head_dim == 0 || d->head_dim == 64, "attention_f32_fwd: head_dim must be 64");
    MMF_CHECK_ARG(d->Sk <= 32 * ATT_MAXT, "attention_f32_fwd: Sk <= 256");
    MMF_CHECK_ARG(d->drop_thr16 == 0 && d->causal_tail == 0 && !d->ctx_f32 && !d->lse && d->q_batch_rows == 0 &&
                  d->kv_batch_rows == 0 && d->mask_batch_stride == 0,
                  "attention_f32_fwd: inference form only (no dropout, prefix-LM tail, lse, K|V cache strides)");
    const int HD = d->heads * 64;
    MMF_CHECK_ARG(d->ldq >= HD && d->ldk >= HD && d->ldv >= HD && d->ldo >= HD, "attention_f32_fwd: leading dimension < heads * 64");
    MMF_CHECK_ARG((d->ldq % 4) == 0 && (d->ldk % 4) == 0 && (((uintptr_t)d->q | (uintptr_t)d->k) & 15) == 0,
                  "attention_f32_fwd: q and k rows are read as 16-byte quads (pointers 16-byte aligned, ldq / ldk multiples of 4)");
    MMF_CHECK_ARG((size_t)d->B * d->heads <= 65535u, "attention_f32_fwd: B * heads too large for one launch");
    AttnF32 a;
    a.q = (const float*)d->q; a.k = (const float*)d->k; a.v = (const float*)d->v; a.o = (float*)d->ctx;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
    a.mask = d->mask; a.B = d->B; a.heads = d->heads; a.Sq = d->Sq; a.Sk = d->Sk; a.scale = d->scale;
    hipLaunchKernelGGL(attn_f32_fwd_kernel<64>, dim3((d->Sq + 127) / 128, d->B * d->heads), dim3(256), 0, (hipStream_t)stream, a);
    MMF_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmf_layernorm_f32_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int H, float eps,
                                     void* stream) {
    MMF_CHECK_ARG(x && gamma && beta && y, "layernorm_f32_fwd: null operand");
    MMF_CHECK_ARG(rows > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "layernorm_f32_fwd: H % 4 == 0, H <= 2048");
    MMF_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_f32_fwd: 16-byte alignment");
    hipLaunchKernelGGL(ln_f32_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, H, eps);
    MMF_CHECK_LAUNCH();
    return 0;
}
