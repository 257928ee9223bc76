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
// attention forward, fp32, on v_mfma_f32_16x16x4_f32.  Workgroup = one (batch, head) x 128 queries, 8 waves; wave = 16 queries x ALL
// keys (Sk <= 16 * MAXT).  Round 3 rewrite (was: 4 waves x 32 queries on the 32x32x2 MFMA with every operand pulled from L2 per MFMA,
// 215 us per VisualBERT layer; see DESIGN §2a for the measured numbers of this form):
//   * K and V rows of the (batch, head) are staged ONCE per workgroup in LDS, row-major with a row stride of D + 4 floats: the
//     16-byte operand reads `K[key = lane % 16][16 m + 4 (lane / 16) ..]` and the scalar reads `V[key][16 eb + lane % 16]` are both
//     bank-conflict-free at that stride, and no transpose is needed;
//   * the scores are computed TRANSPOSED, S^T = K Q^T (A = K rows from LDS, B = Q rows held in registers), with the MFMA's k-slot
//     (lane / 16) mapped to the feature quad 4 (lane / 16) + c: one 16-byte LDS read feeds four MFMAs;
//   * in that layout the accumulator register r of key tile t holds P[q = lane % 16][key = 16 t + 4 (lane / 16) + r] — exactly the A
//     operand layout of the P.V product when ITS k-slot is mapped to key 16 t + 4 (lane / 16) + c: the probabilities never leave the
//     registers (no LDS patch, no barrier between the two products), and a query's softmax statistics are a reduction over the
//     lane's own registers plus two cross-lane steps;
//   * 16-query wave tiles keep the whole score row in 64 (head_dim 64, 256 keys) accumulator registers: two waves per SIMD, so one
//     wave's softmax (VALU) runs under the other's MFMAs.
// Templated on head_dim (64: Sk <= 256; 128: Sk <= 128 — ViLBERT's image stream and co-attention); Sq != Sk (cross attention) and
// the prefix-LM tail of M4C (mmf_attn_desc.causal_tail) are handled.  exp(x) = exp2(x log2 e) on v_exp_f32.
// ------------------------------------------------------------------------------------------------
struct AttnF32 {
    const float* q; const float* k; const float* v; float* o;
    int ldq, ldk, ldv, ldo;
    const float* mask;
    int B, heads, Sq, Sk;
    float scale;
    int cfrom;     // first key of the causal tail (== Sk: none)
};

template <int D, int MAXT>
__global__ __launch_bounds__(512) void attn_f32_fwd_kernel(const AttnF32 a) {
    constexpr int RS = D + 4;                     // LDS row stride in floats (== 4 mod 64 for D = 64 and 128)
    constexpr int NM = D / 16;                    // feature blocks of 16
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ float att_smem[];
    float* Ks = att_smem;                         // [16 * MAXT][RS]
    float* Vs = Ks + 16 * MAXT * RS;              // [16 * MAXT][RS]
    float* Ms = Vs + 16 * MAXT * RS;              // [16 * MAXT]  additive key mask x log2(e); -inf past Sk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int q0 = blockIdx.x * 128 + wave * 16;
    const int nt = (a.Sk + 15) >> 4;

    // stage K and V rows (rows past Sk repeat the last one: their probabilities are exactly 0) and the key mask
    {
        constexpr int QPR = D / 4;                // 16-byte quads per row
        const int quads = nt * 16 * QPR;
        for (int idx = threadIdx.x; idx < quads; idx += 512) {
            const int key = idx / QPR, qd = idx - key * QPR;
            const int kr = min(key, a.Sk - 1);
            const f32x4 kq = *reinterpret_cast<const f32x4*>(a.k + ((size_t)b * a.Sk + kr) * a.ldk + h * D + 4 * qd);
            const f32x4 vq = *reinterpret_cast<const f32x4*>(a.v + ((size_t)b * a.Sk + kr) * a.ldv + h * D + 4 * qd);
            *reinterpret_cast<f32x4*>(Ks + key * RS + 4 * qd) = kq;
            *reinterpret_cast<f32x4*>(Vs + key * RS + 4 * qd) = vq;
        }
        for (int key = threadIdx.x; key < nt * 16; key += 512)
            Ms[key] = key < a.Sk ? (a.mask ? a.mask[(size_t)b * a.Sk + key] * LOG2E : 0.f) : -INFINITY;
    }
    // Q operand (B[k][j = query]): qv[m][c] = Q[q0 + j][16 m + 4 g + c]
    f32x4 qv[NM];
    {
        const int qr = min(q0 + j, a.Sq - 1);
        const float* qp = a.q + ((size_t)b * a.Sq + qr) * a.ldq + h * D + 4 * g;
#pragma unroll
        for (int m = 0; m < NM; ++m) qv[m] = *reinterpret_cast<const f32x4*>(qp + 16 * m);
    }
    __syncthreads();
    if (q0 >= a.Sq) return;                       // (no barrier below)

    const float sl2 = a.scale * LOG2E;
    const int qme = q0 + j;                       // this lane's query
    f32x4 sc[MAXT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* kr = Ks + (16 * t + j) * RS + 4 * g;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const f32x4 kq = *reinterpret_cast<const f32x4*>(kr + 16 * m);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[c], qv[m][c], acc, 0, 0, 0);
            }
            // acc[r] = S[query q0 + j][key 16 t + 4 g + r]
            const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + 16 * t + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * t + 4 * g + r;
                float madd = mk[r];
                if (key >= a.cfrom && key < a.Sk) madd = (qme >= a.cfrom && key <= qme) ? 0.f : -10000.f * LOG2E;
                const float x = acc[r] * sl2 + madd;
                sc[t][r] = x;
                mx = fmaxf(mx, x);
            }
        }
    }
    // softmax of query q0 + j: its keys live in this lane's registers and in the three other lane groups
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float p = __builtin_amdgcn_exp2f(sc[t][r] - mx); sc[t][r] = p; sum += p; }
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    // O = P V with the un-normalised probabilities straight from the score registers
    f32x4 oc[NM];
#pragma unroll
    for (int eb = 0; eb < NM; ++eb) oc[eb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < nt) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* vr = Vs + (16 * t + 4 * g + c) * RS + j;
#pragma unroll
                for (int eb = 0; eb < NM; ++eb) oc[eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[t][c], vr[16 * eb], oc[eb], 0, 0, 0);
            }
        }
    }
    // oc[eb][r] = O[query q0 + 4 g + r][feature 16 eb + j]: that query's sum lives in lane 4 g + r (of every group)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float inv = 1.0f / __shfl(sum, 4 * g + r, 64);
        const int qr = q0 + 4 * g + r;
        if (qr < a.Sq) {
            float* op = a.o + ((size_t)b * a.Sq + qr) * a.ldo + h * D + j;
#pragma unroll
            for (int eb = 0; eb < NM; ++eb) op[16 * eb] = oc[eb][r] * inv;
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


// ------------------------------------------------------------------------------------------------
// Small fp32 row operators of the widened models (ViLBERT, UNITER) on the fp32 path.
// ------------------------------------------------------------------------------------------------
// dst[r][0..KP) = src[r][0..K) followed by zeros: operands whose contraction length is not a multiple of 4 (the 5-d box geometry of
// ViLBERT, the 7-d one of UNITER) become 16-byte rows for the fp32 GEMM.
__global__ __launch_bounds__(256) void pad_rows_f32_kernel(const float* __restrict__ src, int K, float* __restrict__ dst, int KP, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long r = i / KP;
    const int c = (int)(i - r * KP);
    dst[i] = c < K ? src[r * K + c] : 0.f;
}
// op 0: a * b, 1: max(a, 0), 3: a + b   (the op codes of mmf_eltwise)
__global__ __launch_bounds__(256) void eltwise_f32_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    y[i] = op == 0 ? x * b[i] : (op == 1 ? fmaxf(x, 0.f) : x + b[i]);
}
// pool[b][c] = sum_t x[b][t][c] mask[b][t] / sum_t mask[b][t]   (ViLBERT dynamic_attention, vilbert.py:204-205)
__global__ __launch_bounds__(256) void masked_mean_f32_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ pool, int T, int H) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const float* xb = x + (size_t)b * T * H + c;
    const float* mb = mask + (size_t)b * T;
    float acc = 0.f, cnt = 0.f;
    for (int t = 0; t < T; ++t) {
        const float m = mb[t];
        acc += xb[(size_t)t * H] * m;
        cnt += m;
    }
    pool[(size_t)b * H + c] = acc / cnt;
}
// x[row][c] *= gate[row / rpg][c] for c < C   (the Q | K gates, vilbert.py:211-212)
__global__ __launch_bounds__(256) void rowgroup_scale_f32_kernel(float* __restrict__ x, int ld, const float* __restrict__ gate, int rpg, int C, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long row = i / C;
    const int c = (int)(i - row * C);
    x[row * ld + c] *= gate[(row / rpg) * C + c];
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

template <int D, int MAXT>
static int launch_attn_f32(const AttnF32& a, hipStream_t s) {
    constexpr int lds = (2 * 16 * MAXT * (D + 4) + 16 * MAXT) * (int)sizeof(float);
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_fwd_kernel<D, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once = true;
    }
    hipLaunchKernelGGL((attn_f32_fwd_kernel<D, MAXT>), dim3((a.Sq + 127) / 128, a.B * a.heads), dim3(512), lds, s, a);
    MMF_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmf_attention_f32_fwd(const mmf_attn_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->q && d->k && d->v && d->ctx, "attention_f32_fwd: null operand");
    MMF_CHECK_ARG(d->B > 0 && d->heads > 0 && d->Sq > 0 && d->Sk > 0, "attention_f32_fwd: empty problem");
    const int hd = d->head_dim ? d->head_dim : 64;
    MMF_CHECK_ARG(hd == 64 || hd == 128, "attention_f32_fwd: head_dim must be 64 or 128");
    MMF_CHECK_ARG(d->Sk <= (hd == 64 ? 256 : 128), "attention_f32_fwd: Sk <= 256 (head_dim 64) / 128 (head_dim 128)");
    MMF_CHECK_ARG(d->drop_thr16 == 0 && !d->ctx_f32 && !d->lse && d->q_batch_rows == 0 && d->kv_batch_rows == 0 && d->mask_batch_stride == 0,
                  "attention_f32_fwd: inference form only (no dropout, lse, K|V cache strides)");
    MMF_CHECK_ARG(d->causal_tail >= 0 && d->causal_tail <= d->Sk && (d->causal_tail == 0 || d->Sq == d->Sk),
                  "attention_f32_fwd: a causal tail needs self-attention (Sq == Sk)");
    const int HD = d->heads * hd;
    MMF_CHECK_ARG(d->ldq >= HD && d->ldk >= HD && d->ldv >= HD && d->ldo >= HD, "attention_f32_fwd: leading dimension < heads * head_dim");
    MMF_CHECK_ARG((d->ldq % 4) == 0 && (d->ldk % 4) == 0 && (d->ldv % 4) == 0 && (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) == 0,
                  "attention_f32_fwd: q / k / v rows are read as 16-byte quads (pointers 16-byte aligned, leading dimensions multiples of 4)");
    MMF_CHECK_ARG((size_t)d->B * d->heads <= 65535u, "attention_f32_fwd: B * heads too large for one launch");
    AttnF32 a;
    a.q = (const float*)d->q; a.k = (const float*)d->k; a.v = (const float*)d->v; a.o = (float*)d->ctx;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
    a.mask = d->mask; a.B = d->B; a.heads = d->heads; a.Sq = d->Sq; a.Sk = d->Sk; a.scale = d->scale;
    a.cfrom = d->Sk - d->causal_tail;
    return hd == 64 ? launch_attn_f32<64, 16>(a, (hipStream_t)stream) : launch_attn_f32<128, 8>(a, (hipStream_t)stream);
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

extern "C" int mmf_pad_rows_f32(const float* src, int K, float* dst, int KP, int rows, void* stream) {
    MMF_CHECK_ARG(src && dst && rows > 0 && K > 0 && KP >= K, "pad_rows_f32: bad operand");
    const long n = (long)rows * KP;
    hipLaunchKernelGGL(pad_rows_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, K, dst, KP, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmf_eltwise_f32(int op, const float* a, const float* b, float* y, long n, void* stream) {
    MMF_CHECK_ARG(a && y && n > 0 && (op == 0 || op == 1 || op == 3) && (op == 1 || b), "eltwise_f32: bad operand (op 0 mul, 1 relu, 3 add)");
    hipLaunchKernelGGL(eltwise_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, a, b, y, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmf_masked_mean_f32(const float* x, const float* mask, float* pool, int B, int T, int H, void* stream) {
    MMF_CHECK_ARG(x && mask && pool && B > 0 && T > 0 && H > 0, "masked_mean_f32: bad operand");
    hipLaunchKernelGGL(masked_mean_f32_kernel, dim3((H + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, mask, pool, T, H);
    MMF_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmf_rowgroup_scale_f32(float* x, int ld, const float* gate, int groups, int rows_per_group, int C, void* stream) {
    MMF_CHECK_ARG(x && gate && groups > 0 && rows_per_group > 0 && C > 0 && C <= ld, "rowgroup_scale_f32: bad operand");
    const long n = (long)groups * rows_per_group * C;
    hipLaunchKernelGGL(rowgroup_scale_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, gate, rows_per_group, C, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
