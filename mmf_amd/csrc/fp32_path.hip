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
// GEMM: C[m][n] = epilogue(sum_k A(m,k) * B(n,k)), everything fp32, on v_mfma_f32_16x16x4_f32 (round 3 rewrite; the round-2 kernel
// ran the 32x32x2 MFMA from k-major LDS images with one scalar LDS read per operand register: 80-98 TFLOP/s).
// 64x128 tile per 256-thread workgroup (the template also builds 128x128: measured no better), BK = 16; wave w owns a 32x64 quadrant
// = 2x4 MFMA blocks of 16x16.
// Both operand tiles live in LDS ROW-major, [row][k] with XOR-swizzled 16-byte quads (gemm_f32_lds), whatever their layout in memory:
//   * a row operand (k contiguous in memory) is copied 16 bytes at a time;
//   * a k-major operand (the dgrad's weight, both operands of a weight gradient) is transposed by the staging stores, with the lanes
//     of a wave laid out 16 k x 4 row-quads (2-way conflicts on these scalar stores at worst).
// The MFMA's k-slot (lane / 16) is mapped to the k QUAD 4 (lane / 16) + c, so ONE 16-byte LDS read per 16-row block feeds four MFMAs
// (the operand rows of a wave are 8 ds_read_b128 per K-step for 64 MFMAs).  The product is formed TRANSPOSED (MFMA A operand = the
// B tile's rows): the accumulator register r of lane l then holds C[m = block row l % 16][n = 4 (l / 16) + r], four consecutive
// columns — bias / residual / saved-derivative reads and the C stores are 16 bytes per lane.
// Register double buffering: the next K-step's global loads are in flight while the MFMAs of the current one run; one barrier per
// K-step; split-K over gridDim.z into fp32 slabs (weight gradients: K = B S rows, few output tiles) summed by splitk_f32_reduce.
// ------------------------------------------------------------------------------------------------
constexpr int GBN = 128, GBK = 16, GRS = GBK;        // (BK = 32 measured no better: 77-93 vs 81-96 TFLOP/s)
// LDS tile address of (row, k): rows of GBK floats, NO padding, the 16-byte quad index XOR-swizzled by (row >> 1) & 3 — the layout for which
// the four non-contiguous 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS) AND the 8-lane groups of ds_write_b128 are conflict-free
// (a padded stride of 20 floats measured SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles: its quads collide inside those lane groups).
DEVI int gemm_f32_lds(int row, int k) { return row * GRS + ((((k >> 2) ^ (row >> 1)) & 3) << 2) + (k & ~15) + (k & 3); }

struct GemmF32 {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias; const float* coladd; const float* rowtab; const int64_t* rowidx; int rowtab_ld;
    int act;
    float* U; const float* aux;
    const float* resid; int ldr;
    DropoutCfg drop;
    float beta;
    int grp_in, grp_pad, grp_off;
    int ksplit;        // K-steps per z-slice (0: no split); slabs of M * N floats at C
    int vec_ok;        // ldc, ldr, rowtab_ld multiples of 4 and 16-byte aligned bases: 16-byte epilogue accesses
};

DEVI float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
DEVI float gelu_exact_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// One operand tile's share of a K-step: global -> registers.  ROWS = tile rows (64 or 128); a pass of the 256 threads covers GPR rows.
constexpr int GKQ = GBK / 4, GPR = 1024 / GBK;     // k-quads per tile row; rows per pass (64 at BK = 16, 32 at BK = 32)
// Every load is issued unconditionally from a clamped (always valid) address and zeroed by a select afterwards: no control flow around the
// loads, so the compiler's s_waitcnt insertion can COUNT (vmcnt(n) for the older register set while the younger set stays in flight).  With
// the bounds checks as branches it fell back to vmcnt(0) at every stash: the full memory latency was exposed once per K-step on every
// wave (MFMA pipe 62 % busy, the waves 66 % of their cycles in s_waitcnt; profiles/r03_fp32_gemm_pmc.txt).
template <bool KM, int ROWS>
DEVI void gemm_f32_fetch(f32x4 (&r)[ROWS / GPR], const float* __restrict__ base, int row0, int rows, int ld, int k0, int K, int tid) {
    if constexpr (!KM) {     // [row][k]: thread = row (tid / GKQ) + GPR h, k-quad 4 (tid % GKQ)
        const int kc = min(k0 + (tid % GKQ) * 4, ((K + 3) & ~3) - 4);
#pragma unroll
        for (int h = 0; h < ROWS / GPR; ++h)
            r[h] = *reinterpret_cast<const f32x4*>(base + (size_t)min(row0 + (tid / GKQ) + GPR * h, rows - 1) * ld + kc);
    } else {                 // [k][row]: thread = k (tid % GBK), row-quad 4 (tid / GBK) + GPR h
        const float* kp = base + (size_t)min(k0 + (tid % GBK), K - 1) * ld;
        const int rlast = (rows - 1) & ~3;           // first row of the last (possibly partial) quad
#pragma unroll
        for (int h = 0; h < ROWS / GPR; ++h) {
            const int row = row0 + (tid / GBK) * 4 + GPR * h;
            if (rows % 4 == 0 || row < rlast) {      // (uniform per launch except on the one ragged quad)
                r[h] = *reinterpret_cast<const f32x4*>(kp + min(row, rlast));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[h][e] = kp[min(row + e, rows - 1)];
            }
        }
    }
}
// registers -> LDS; what lies outside the matrix (or past the K slice) is zeroed HERE, a step after the loads were issued: a select at
// fetch time would make the wave wait for its youngest loads
template <bool KM, int ROWS>
DEVI void gemm_f32_stash(const f32x4 (&r)[ROWS / GPR], float* __restrict__ tile, int row0, int rows, int k0, int K, int tid) {
    if constexpr (!KM) {
        const bool kok = k0 + (tid % GKQ) * 4 < K;
#pragma unroll
        for (int h = 0; h < ROWS / GPR; ++h) {
            const int lr = (tid / GKQ) + GPR * h;
            const f32x4 v = (kok && row0 + lr < rows) ? r[h] : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(tile + gemm_f32_lds(lr, (tid % GKQ) * 4)) = v;
        }
    } else {
        const bool kok = k0 + (tid % GBK) < K;
#pragma unroll
        for (int h = 0; h < ROWS / GPR; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lr = (tid / GBK) * 4 + GPR * h + e;
                tile[gemm_f32_lds(lr, tid % GBK)] = (kok && row0 + lr < rows) ? r[h][e] : 0.f;
            }
    }
}

template <int MI, bool AKM, bool BKM>     // MI: 16-row blocks per wave along M / 2 (BM = 64 * MI)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_f32_kernel(const GemmF32 g) {
    constexpr int BM = 64 * MI, MB = 2 * MI;      // MB m-blocks x 4 n-blocks of 16 x 16 per wave
    extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
    float (*As)[BM * GRS] = reinterpret_cast<float (*)[BM * GRS]>(gemm_smem);
    float (*Bs)[GBN * GRS] = reinterpret_cast<float (*)[GBN * GRS]>(gemm_smem + 2 * BM * GRS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q4 = (lane >> 4) * 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * GBN;
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * 64;

    f32x4 acc[4][MB];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = (g.K + GBK - 1) / GBK;
    const int kt0 = g.ksplit ? blockIdx.z * g.ksplit : 0;
    const int kt1 = g.ksplit ? min(nk_all, kt0 + g.ksplit) : nk_all;
    // Two K-steps of operands are in flight in registers besides the two LDS stages: at the top of step kt the loads of step kt + 2 are
    // issued into one register set while the other set — step kt + 1, loaded a whole step ago — is stashed into the free LDS stage at
    // the bottom.  The loop is unrolled by two so that the sets swap roles without register copies (a copy would wait for the loads it
    // copies), and the loads are pinned at the top of the step (sched_barrier): a load has two steps of MFMAs to land.
    const int Kend = min(g.K, kt1 * GBK);          // k past this slice reads as zero (split-K slices; the odd extra step of the unrolled loop)
    f32x4 ra[BM / GPR], rb[GBN / GPR], sa[BM / GPR], sb[GBN / GPR];
    gemm_f32_fetch<AKM, BM>(ra, g.A, m0, g.M, g.lda, kt0 * GBK, Kend, tid);
    gemm_f32_fetch<BKM, GBN>(rb, g.B, n0, g.N, g.ldb, kt0 * GBK, Kend, tid);
    gemm_f32_stash<AKM, BM>(ra, As[0], m0, g.M, kt0 * GBK, Kend, tid);
    gemm_f32_stash<BKM, GBN>(rb, Bs[0], n0, g.N, kt0 * GBK, Kend, tid);
    gemm_f32_fetch<AKM, BM>(ra, g.A, m0, g.M, g.lda, (kt0 + 1) * GBK, Kend, tid);
    gemm_f32_fetch<BKM, GBN>(rb, g.B, n0, g.N, g.ldb, (kt0 + 1) * GBK, Kend, tid);
    __syncthreads();

#define MMF_F32_STEP(KT, CUR, HOLD_A, HOLD_B, LOAD_A, LOAD_B)                                                                          \
    {                                                                                                                                  \
        gemm_f32_fetch<AKM, BM>(LOAD_A, g.A, m0, g.M, g.lda, ((KT) + 2) * GBK, Kend, tid);                                             \
        gemm_f32_fetch<BKM, GBN>(LOAD_B, g.B, n0, g.N, g.ldb, ((KT) + 2) * GBK, Kend, tid);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        _Pragma("unroll") for (int kg = 0; kg < GBK / 16; ++kg) {                                                                      \
            f32x4 bn[4], am[MB];                                                                                                       \
            _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                                           \
                bn[nb] = *reinterpret_cast<const f32x4*>(&Bs[CUR][gemm_f32_lds(wn + 16 * nb + j, 16 * kg + q4)]);                      \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                          \
                am[mb] = *reinterpret_cast<const f32x4*>(&As[CUR][gemm_f32_lds(wm + 16 * mb + j, 16 * kg + q4)]);                      \
            _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                              \
                _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                                       \
                    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                  \
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[nb][c], am[mb][c], acc[nb][mb], 0, 0, 0);               \
        }                                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        /* stage CUR ^ 1 was last read in the previous step, which every wave left through the barrier */                             \
        gemm_f32_stash<AKM, BM>(HOLD_A, As[(CUR) ^ 1], m0, g.M, ((KT) + 1) * GBK, Kend, tid);                                          \
        gemm_f32_stash<BKM, GBN>(HOLD_B, Bs[(CUR) ^ 1], n0, g.N, ((KT) + 1) * GBK, Kend, tid);                                         \
        __syncthreads();                                                                                                               \
    }
    for (int kt = kt0; kt < kt1; kt += 2) {
        MMF_F32_STEP(kt, 0, ra, rb, sa, sb)
        MMF_F32_STEP(kt + 1, 1, sa, sb, ra, rb)        // (kt + 1 == kt1 on an odd step count: a step of zeros)
    }
#undef MMF_F32_STEP

    // epilogue: acc[nb][mb][r] = C[m = wm + 16 mb + j][n = wn + 16 nb + q4 + r]
    const uint32_t dkey = g.drop.thr16 ? drop_key(g.drop) : 0u;
    float* Cz = g.ksplit ? g.C + (size_t)blockIdx.z * g.M * g.N : g.C;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = m0 + wm + 16 * mb + j;
        if (m >= g.M) continue;
        const int orow = g.grp_in > 0 ? m + (m / g.grp_in) * g.grp_pad + g.grp_off : m;
        const float* rt = g.rowtab ? g.rowtab + (size_t)g.rowidx[m] * g.rowtab_ld : nullptr;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int n = n0 + wn + 16 * nb + q4;
            if (n >= g.N) continue;
            f32x4 v = acc[nb][mb];
            if (g.ksplit) {        // raw partial sums: the epilogue runs in splitk_f32_reduce
                float* cp = Cz + (size_t)m * g.N + n;
                if (n + 3 < g.N && (g.N & 3) == 0) *reinterpret_cast<f32x4*>(cp) = v;
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = v[r];
                continue;
            }
            const bool vec = g.vec_ok && n + 3 < g.N;
            const size_t coff = (size_t)orow * g.ldc + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= g.N) { v[r] = 0.f; continue; }
                if (g.bias) v[r] += g.bias[n + r];
                if (g.coladd) v[r] += g.coladd[n + r];
                if (rt) v[r] += rt[n + r];
            }
            if (g.act == 1) {
                if (g.U) {
                    f32x4 u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) u[r] = gelu_exact_grad(v[r]);
                    if (vec) *reinterpret_cast<f32x4*>(g.U + coff) = u;
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < g.N) g.U[coff + r] = u[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_exact(v[r]);
            } else if (g.act == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
            } else if (g.act == 2 || g.act == 4) {
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (vec) x = *reinterpret_cast<const f32x4*>(g.aux + coff);
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) x[r] = g.aux[coff + r];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= (g.act == 2) ? x[r] : 1.f - x[r] * x[r];
            }
            if (g.drop.thr16) {
                const uint32_t idx = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
                if ((idx & 3u) == 0u) {
                    const f32x4 sc = drop_scale4(dkey, idx, g.drop.thr16, g.drop.scale);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= sc[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= drop_scale1(dkey, idx + r, g.drop.thr16, g.drop.scale);
                }
            }
            if (g.resid) {
                const float* rp = g.resid + (size_t)m * g.ldr + n;
                if (vec) { const f32x4 x = *reinterpret_cast<const f32x4*>(rp); v += x; }
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) v[r] += rp[r];
            }
            float* cp = g.C + coff;
            if (vec) {
                if (g.beta != 0.f) v += g.beta * *reinterpret_cast<const f32x4*>(cp);
                *reinterpret_cast<f32x4*>(cp) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < g.N) cp[r] = g.beta != 0.f ? v[r] + g.beta * cp[r] : v[r];
            }
        }
    }
}

// C[m][n] = beta C[m][n] + sum_z slab[z][m][n]   (split-K weight gradients; no other epilogue)
__global__ __launch_bounds__(256) void splitk_f32_reduce_kernel(const float* __restrict__ ws, int splits, long n, int N, float* __restrict__ C, int ldc, float beta) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = ws[i];
    for (int z = 1; z < splits; ++z) a += ws[(long)z * n + i];
    const long m = i / N;
    float* c = C + m * ldc + (i - m * N);
    *c = beta != 0.f ? a + beta * *c : a;
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
    const float* q; const float* k; const float* v; float* out;
    int ldq, ldk, ldv, ldo;
    const float* mask;
    int B, heads, Sq, Sk;
    float scale;
    int cfrom;     // first key of the causal tail (== Sk: none)
    int mhs;       // per-head mask (mmf_attn_desc.mask_head_stride): mask entries between the heads of a sample, 0 = one mask for all heads
    int mqs, mbs;  // per-query mask (mmf_attn_desc.mask_query_stride / mask_batch_stride): mask entries between query rows (0: the key mask [B, Sk]) / samples
    float* lse;    // [B, heads, Sq]: row maximum + log2(row sum) of the scaled scores in log2 units (training: saved for the backward)
    DropoutCfg drop;   // attention-probability dropout, element index ((b heads + head) Sq + q) Sk + key
    // backward only
    const float* o; const float* d_o; float* dq; float* dk; float* dv; float* delta;
};

// additive mask of (query q, key) in log2 units for the per-query form (what BertSelfAttentionJit.forward accepts as a [B, 1, S, S] mask,
// hf_layers.py:187-190); q is clamped by the caller, keys past Sk are padding
DEVI float query_mask(const AttnF32& a, int b, int h, int q, int key) {
    return key < a.Sk ? a.mask[(size_t)b * a.mbs + (size_t)h * a.mhs + (size_t)q * a.mqs + key] * 1.4426950408889634f : -INFINITY;
}
// keep-scale of the probability of (query q, key) under attention dropout (1 when dropout is off)
DEVI float attn_drop(const AttnF32& a, uint32_t dkey, int bh, int q, int key) {
    return drop_scale1(dkey, ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Sk + (uint32_t)key, a.drop.thr16, a.drop.scale);
}

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
            Ms[key] = key < a.Sk ? ((a.mask && !a.mqs) ? a.mask[(size_t)b * a.Sk + key] * LOG2E : 0.f) : -INFINITY;
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
                if (a.mqs) madd = query_mask(a, b, h, min(qme, a.Sq - 1), key);
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
    if (a.lse && g == 0 && qme < a.Sq) a.lse[(size_t)bh * a.Sq + qme] = mx + __builtin_amdgcn_logf(sum);     // v_log_f32 = log2
    if (a.drop.thr16) {
        const uint32_t dkey = drop_key(a.drop);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][r] *= attn_drop(a, dkey, bh, min(qme, a.Sq - 1), min(16 * t + 4 * g + r, a.Sk - 1));
            }
        }
    }

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
            float* op = a.out + ((size_t)b * a.Sq + qr) * a.ldo + h * D + j;
#pragma unroll
            for (int eb = 0; eb < NM; ++eb) op[16 * eb] = oc[eb][r] * inv;
        }
    }
}

// More keys than one LDS image holds (head_dim 64: 257 .. 512 keys, head_dim 128: 129 .. 256 — what the bf16 kernels run, BERT's
// max_position_embeddings): K / V are staged in blocks of 16 * MAXT keys and the score tiles are computed TWICE instead of being held in
// registers — pass 1 finds each query's row maximum over all blocks (K blocks only), pass 2 recomputes a block's scores, exponentiates against
// that maximum, accumulates the row sum and P.V.  The softmax stays the reference's exact two-pass form (max, then sum of exp(x - max)); same
// operand layouts, masks, dropout indices and output as attn_f32_fwd_kernel.
template <int D, int MAXT>
__global__ __launch_bounds__(512) void attn_f32_fwd_long_kernel(const AttnF32 a) {
    constexpr int RS = D + 4, NM = D / 16, KB = 16 * MAXT;
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ float att_smem[];
    float* Ks = att_smem;
    float* Vs = Ks + KB * RS;
    float* Ms = Vs + KB * RS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int q0 = blockIdx.x * 128 + wave * 16;
    const bool active = q0 < a.Sq;                // (idle waves run along to the barriers)
    const int nblk = (a.Sk + KB - 1) / KB;
    const int qme = q0 + j;
    const int qcl = min(qme, a.Sq - 1);
    f32x4 qv[NM];
    {
        const float* qp = a.q + ((size_t)b * a.Sq + qcl) * a.ldq + h * D + 4 * g;
#pragma unroll
        for (int m = 0; m < NM; ++m) qv[m] = *reinterpret_cast<const f32x4*>(qp + 16 * m);
    }
    auto stage = [&](int kb, bool with_v) {
        constexpr int QPR = D / 4;
        const int key0 = kb * KB, nt = min(MAXT, (a.Sk - key0 + 15) >> 4);
        for (int idx = threadIdx.x; idx < nt * 16 * QPR; idx += 512) {
            const int lk = idx / QPR, qd = idx - lk * QPR;
            const int kr = min(key0 + lk, a.Sk - 1);
            *reinterpret_cast<f32x4*>(Ks + lk * RS + 4 * qd) = *reinterpret_cast<const f32x4*>(a.k + ((size_t)b * a.Sk + kr) * a.ldk + h * D + 4 * qd);
            if (with_v) *reinterpret_cast<f32x4*>(Vs + lk * RS + 4 * qd) = *reinterpret_cast<const f32x4*>(a.v + ((size_t)b * a.Sk + kr) * a.ldv + h * D + 4 * qd);
        }
        for (int lk = threadIdx.x; lk < nt * 16; lk += 512)
            Ms[lk] = key0 + lk < a.Sk ? ((a.mask && !a.mqs) ? a.mask[(size_t)b * a.Sk + key0 + lk] * LOG2E : 0.f) : -INFINITY;
        return nt;
    };
    const float sl2 = a.scale * LOG2E;
    // x[r] = masked, scaled score (log2 units) of (query q0 + j, key key0 + 16 t + 4 g + r)
    auto scores = [&](int key0, int t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + (16 * t + j) * RS + 4 * g;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const f32x4 kq = *reinterpret_cast<const f32x4*>(kr + 16 * m);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[c], qv[m][c], acc, 0, 0, 0);
        }
        const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + 16 * t + 4 * g);
        f32x4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + 16 * t + 4 * g + r;
            float madd = mk[r];
            if (a.mqs) madd = query_mask(a, b, h, qcl, key);
            if (key >= a.cfrom && key < a.Sk) madd = (qme >= a.cfrom && key <= qme) ? 0.f : -10000.f * LOG2E;
            x[r] = acc[r] * sl2 + madd;
        }
        return x;
    };
    float mx = -INFINITY;
    for (int kb = 0; kb < nblk; ++kb) {
        const int nt = stage(kb, false);
        __syncthreads();
        if (active) {
#pragma unroll 2
            for (int t = 0; t < nt; ++t) {
                const f32x4 x = scores(kb * KB, t);
                mx = fmaxf(fmaxf(mx, fmaxf(x[0], x[1])), fmaxf(x[2], x[3]));
            }
        }
        __syncthreads();
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    f32x4 oc[NM];
#pragma unroll
    for (int eb = 0; eb < NM; ++eb) oc[eb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t dkey = a.drop.thr16 ? drop_key(a.drop) : 0u;
    for (int kb = 0; kb < nblk; ++kb) {
        const int nt = stage(kb, true);
        __syncthreads();
        if (active) {
#pragma unroll 2
            for (int t = 0; t < nt; ++t) {
                f32x4 pr = scores(kb * KB, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(pr[r] - mx);
                    sum += p;
                    pr[r] = a.drop.thr16 ? p * attn_drop(a, dkey, bh, qcl, min(kb * KB + 16 * t + 4 * g + r, a.Sk - 1)) : p;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* vr = Vs + (16 * t + 4 * g + c) * RS + j;
#pragma unroll
                    for (int eb = 0; eb < NM; ++eb) oc[eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[c], vr[16 * eb], oc[eb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (a.lse && g == 0 && qme < a.Sq) a.lse[(size_t)bh * a.Sq + qme] = mx + __builtin_amdgcn_logf(sum);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float inv = 1.0f / __shfl(sum, 4 * g + r, 64);
        const int qr = q0 + 4 * g + r;
        if (qr < a.Sq) {
            float* op = a.out + ((size_t)b * a.Sq + qr) * a.ldo + h * D + j;
#pragma unroll
            for (int eb = 0; eb < NM; ++eb) op[16 * eb] = oc[eb][r] * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// attention backward, fp32 (fp32 training: the reference's default arithmetic, training_loop.py:199-211), two launches built like the
// forward.  With P = softmax(S), Pd = dropout(P), O = Pd V:   dPd = dO V^T,  delta = rowsum(dO o O),  dS = P o (dropmask o dPd - delta),
// dQ = scale dS K,  dK = scale dS^T Q,  dV = Pd^T dO.  P is recomputed from the saved row statistic lse (one exp2 per element).
//   attn_f32_bwd_dq_kernel:  the forward's roles — K, V rows in LDS, wave = 16 queries: S^T and dPd^T blocks (A = K / V rows, B = the
//     query's Q / dO row in registers), dS formed in the accumulator layout, which IS the A-operand layout of dQ = dS K (k-slot = key
//     quad), per key tile: nothing but the dQ accumulators lives across tiles.  Writes delta for the second launch.
//   attn_f32_bwd_dkv_kernel: the roles swapped — Q, dO rows in LDS, wave = 16 KEYS whose K, V rows sit in registers: S and dPd blocks
//     with A = Q / dO rows, B = K / V; the accumulator layout (rows = queries 4 (lane / 16) + r, column = the lane's key) is the
//     A-operand layout of dV = Pd^T dO and dK = dS^T Q (k-slot = query quad); dK / dV accumulate in registers over the query tiles.
// No atomics, fixed summation order.
// ------------------------------------------------------------------------------------------------
template <int D, int MAXT>
__global__ __launch_bounds__(512) void attn_f32_bwd_dq_kernel(const AttnF32 a) {
    constexpr int RS = D + 4, NM = D / 16;
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ float att_smem[];
    float* Ks = att_smem;
    float* Vs = Ks + 16 * MAXT * RS;
    float* Ms = Vs + 16 * MAXT * RS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int q0 = blockIdx.x * 128 + wave * 16;
    // keys in blocks of 16 * MAXT (one block = one LDS image; more than one beyond 256 / 128 keys: same arithmetic, dQ accumulates across blocks)
    constexpr int KB = 16 * MAXT;
    const int nblk = (a.Sk + KB - 1) / KB;
    auto stage = [&](int kb) {
        constexpr int QPR = D / 4;
        const int key0 = kb * KB, nt = min(MAXT, (a.Sk - key0 + 15) >> 4);
        for (int idx = threadIdx.x; idx < nt * 16 * QPR; idx += 512) {
            const int lk = idx / QPR, qd = idx - lk * QPR;
            const int kr = min(key0 + lk, a.Sk - 1);
            *reinterpret_cast<f32x4*>(Ks + lk * RS + 4 * qd) = *reinterpret_cast<const f32x4*>(a.k + ((size_t)b * a.Sk + kr) * a.ldk + h * D + 4 * qd);
            *reinterpret_cast<f32x4*>(Vs + lk * RS + 4 * qd) = *reinterpret_cast<const f32x4*>(a.v + ((size_t)b * a.Sk + kr) * a.ldv + h * D + 4 * qd);
        }
        for (int lk = threadIdx.x; lk < nt * 16; lk += 512)
            Ms[lk] = key0 + lk < a.Sk ? ((a.mask && !a.mqs) ? a.mask[(size_t)b * a.Sk + key0 + lk] * LOG2E : 0.f) : -INFINITY;
        return nt;
    };
    int nt = stage(0);
    const int qme = min(q0 + j, a.Sq - 1);
    f32x4 qv[NM], dov[NM];
    float dl = 0.f;
    {
        const size_t ro = (size_t)b * a.Sq + qme;
        const float* qp = a.q + ro * a.ldq + h * D + 4 * g;
        const float* dp = a.d_o + ro * a.ldo + h * D + 4 * g;
        const float* op = a.o + ro * a.ldo + h * D + 4 * g;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            qv[m] = *reinterpret_cast<const f32x4*>(qp + 16 * m);
            dov[m] = *reinterpret_cast<const f32x4*>(dp + 16 * m);
            const f32x4 ov = *reinterpret_cast<const f32x4*>(op + 16 * m);
#pragma unroll
            for (int c = 0; c < 4; ++c) dl += dov[m][c] * ov[c];
        }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);                 // delta of query q0 + j
    __syncthreads();
    const bool active = q0 < a.Sq;
    if (!active && nblk == 1) return;           // (one block: no barrier below; more: idle waves run along to the barriers)
    if (active && g == 0 && q0 + j < a.Sq) a.delta[(size_t)bh * a.Sq + q0 + j] = dl;
    const float lse = a.lse[(size_t)bh * a.Sq + qme];
    const float sl2 = a.scale * LOG2E;
    const uint32_t dkey = a.drop.thr16 ? drop_key(a.drop) : 0u;

    f32x4 dq[NM];
#pragma unroll
    for (int eb = 0; eb < NM; ++eb) dq[eb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < nblk; ++kb) {
    if (kb) { __syncthreads(); nt = stage(kb); __syncthreads(); }
    const int key0 = kb * KB;
    if (active)
#pragma unroll 2
    for (int t = 0; t < nt; ++t) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + (16 * t + j) * RS + 4 * g;
        const float* vr = Vs + (16 * t + j) * RS + 4 * g;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const f32x4 kq = *reinterpret_cast<const f32x4*>(kr + 16 * m);
            const f32x4 vq = *reinterpret_cast<const f32x4*>(vr + 16 * m);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[c], qv[m][c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vq[c], dov[m][c], dp, 0, 0, 0);
            }
        }
        const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + 16 * t + 4 * g);
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + 16 * t + 4 * g + r;
            float madd = mk[r];
            if (a.mqs) madd = query_mask(a, b, h, qme, key);
            if (key >= a.cfrom && key < a.Sk) madd = (qme >= a.cfrom && key <= qme) ? 0.f : -10000.f * LOG2E;
            const float p = __builtin_amdgcn_exp2f(s[r] * sl2 + madd - lse);
            const float m = a.drop.thr16 ? attn_drop(a, dkey, bh, qme, min(key, a.Sk - 1)) : 1.f;
            ds[r] = p * (dp[r] * m - dl) * a.scale;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* kc = Ks + (16 * t + 4 * g + c) * RS + j;
#pragma unroll
            for (int eb = 0; eb < NM; ++eb) dq[eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[c], kc[16 * eb], dq[eb], 0, 0, 0);
        }
    }
  }
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = q0 + 4 * g + r;
        if (qr < a.Sq) {
            float* op = a.dq + ((size_t)b * a.Sq + qr) * a.ldq + h * D + j;
#pragma unroll
            for (int eb = 0; eb < NM; ++eb) op[16 * eb] = dq[eb][r];
        }
    }
}

template <int D, int MAXT>     // MAXT: QUERY tiles of 16 (Sq <= 16 * MAXT)
__global__ __launch_bounds__(512) void attn_f32_bwd_dkv_kernel(const AttnF32 a) {
    constexpr int RS = D + 4, NM = D / 16;
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ float att_smem[];
    float* Qs = att_smem;                         // [16 * MAXT][RS]
    float* Os = Qs + 16 * MAXT * RS;              // dO rows
    float* Ls = Os + 16 * MAXT * RS;              // [16 * MAXT] lse (+inf past Sq: probability 0)
    float* Ds = Ls + 16 * MAXT;                   // [16 * MAXT] delta
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int k0 = blockIdx.x * 128 + wave * 16;
    // queries in blocks of 16 * MAXT (one block = one LDS image; more than one beyond 256 / 128 queries: dK / dV accumulate across blocks)
    constexpr int QB = 16 * MAXT;
    const int nblk = (a.Sq + QB - 1) / QB;
    auto stage = [&](int qb) {
        constexpr int QPR = D / 4;
        const int qs0 = qb * QB, nq_ = min(MAXT, (a.Sq - qs0 + 15) >> 4);
        for (int idx = threadIdx.x; idx < nq_ * 16 * QPR; idx += 512) {
            const int lq = idx / QPR, qd = idx - lq * QPR;
            const int qr = min(qs0 + lq, a.Sq - 1);
            *reinterpret_cast<f32x4*>(Qs + lq * RS + 4 * qd) = *reinterpret_cast<const f32x4*>(a.q + ((size_t)b * a.Sq + qr) * a.ldq + h * D + 4 * qd);
            *reinterpret_cast<f32x4*>(Os + lq * RS + 4 * qd) = *reinterpret_cast<const f32x4*>(a.d_o + ((size_t)b * a.Sq + qr) * a.ldo + h * D + 4 * qd);
        }
        for (int lq = threadIdx.x; lq < nq_ * 16; lq += 512) {
            Ls[lq] = qs0 + lq < a.Sq ? a.lse[(size_t)bh * a.Sq + qs0 + lq] : INFINITY;
            Ds[lq] = qs0 + lq < a.Sq ? a.delta[(size_t)bh * a.Sq + qs0 + lq] : 0.f;
        }
        return nq_;
    };
    int nq = stage(0);
    const int kme = k0 + j;                        // this lane's key
    const int kcl = min(kme, a.Sk - 1);
    f32x4 kv[NM], vv[NM];
    {
        const float* kp = a.k + ((size_t)b * a.Sk + kcl) * a.ldk + h * D + 4 * g;
        const float* vp = a.v + ((size_t)b * a.Sk + kcl) * a.ldv + h * D + 4 * g;
#pragma unroll
        for (int m = 0; m < NM; ++m) { kv[m] = *reinterpret_cast<const f32x4*>(kp + 16 * m); vv[m] = *reinterpret_cast<const f32x4*>(vp + 16 * m); }
    }
    __syncthreads();
    const bool active = k0 < a.Sk;
    if (!active && nblk == 1) return;
    const float sl2 = a.scale * LOG2E;
    const float mkey = kme < a.Sk ? ((a.mask && !a.mqs) ? a.mask[(size_t)b * a.Sk + kme] * LOG2E : 0.f) : -INFINITY;
    const bool tail = kme >= a.cfrom && kme < a.Sk;
    const uint32_t dkey = a.drop.thr16 ? drop_key(a.drop) : 0u;

    f32x4 dk[NM], dv[NM];
#pragma unroll
    for (int eb = 0; eb < NM; ++eb) { dk[eb] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[eb] = dk[eb]; }
  for (int qb = 0; qb < nblk; ++qb) {
    if (qb) { __syncthreads(); nq = stage(qb); __syncthreads(); }
    const int qs0 = qb * QB;
    if (active)
#pragma unroll 2
    for (int u = 0; u < nq; ++u) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        const float* qr = Qs + (16 * u + j) * RS + 4 * g;
        const float* orow = Os + (16 * u + j) * RS + 4 * g;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const f32x4 qq = *reinterpret_cast<const f32x4*>(qr + 16 * m);
            const f32x4 oq = *reinterpret_cast<const f32x4*>(orow + 16 * m);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[c], kv[m][c], s, 0, 0, 0);       // s[r] = S[query 16 u + 4 g + r][key k0 + j]
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(oq[c], vv[m][c], dp, 0, 0, 0);
            }
        }
        const f32x4 ls = *reinterpret_cast<const f32x4*>(Ls + 16 * u + 4 * g);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(Ds + 16 * u + 4 * g);
        f32x4 pd, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qs0 + 16 * u + 4 * g + r;
            float madd = mkey;
            if (a.mqs) madd = query_mask(a, b, h, min(q, a.Sq - 1), kme);       // (padded query rows carry lse = +inf: p = 0 whatever is read for them)
            if (tail) madd = (q >= a.cfrom && kme <= q) ? 0.f : -10000.f * LOG2E;
            const float p = __builtin_amdgcn_exp2f(s[r] * sl2 + madd - ls[r]);
            const float m = a.drop.thr16 ? attn_drop(a, dkey, bh, min(q, a.Sq - 1), kcl) : 1.f;
            pd[r] = p * m;
            ds[r] = p * (dp[r] * m - dl[r]) * a.scale;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* oc = Os + (16 * u + 4 * g + c) * RS + j;
            const float* qc = Qs + (16 * u + 4 * g + c) * RS + j;
#pragma unroll
            for (int eb = 0; eb < NM; ++eb) {
                dv[eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pd[c], oc[16 * eb], dv[eb], 0, 0, 0);
                dk[eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[c], qc[16 * eb], dk[eb], 0, 0, 0);
            }
        }
    }
  }
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kr = k0 + 4 * g + r;
        if (kr < a.Sk) {
            float* kp = a.dk + ((size_t)b * a.Sk + kr) * a.ldk + h * D + j;
            float* vp = a.dv + ((size_t)b * a.Sk + kr) * a.ldv + h * D + j;
#pragma unroll
            for (int eb = 0; eb < NM; ++eb) { kp[16 * eb] = dk[eb][r]; vp[16 * eb] = dv[eb][r]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward, fp32 in / fp32 out: one wave per row, the row held in registers (H <= 2048), two-pass statistics
// (mean, then the biased variance of the centred values), eps inside the square root.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_f32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, int rows, int H,
                                                          float eps, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
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
    if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }      // training: saved for mmf_layernorm_f32_bwd
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
__global__ __launch_bounds__(256) void pad_rows_f32_kernel(const float* __restrict__ src, long lds, int K, float* __restrict__ dst, int KP, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long r = i / KP;
    const int c = (int)(i - r * KP);
    dst[i] = c < K ? src[r * lds + c] : 0.f;
}
// op 0: a * b, 1: max(a, 0), 3: a + b   (the op codes of mmf_eltwise); 4: a * (1 - b^2) (backward of tanh, b = the saved output);
// 5: a where b > 0 else 0 (backward of relu, b = the saved output)
__global__ __launch_bounds__(256) void eltwise_f32_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    y[i] = op == 0 ? x * b[i] : (op == 1 ? fmaxf(x, 0.f) : (op == 3 ? x + b[i] : (op == 4 ? x * (1.f - b[i] * b[i]) : (b[i] > 0.f ? x : 0.f))));
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
// ---- M4C (mmf/models/m4c.py) on fp32 rows ----------------------------------------------------------------------------------------------
// F.normalize(x, dim=-1): y[r, :D] = x[r, :D] / max(||x[r, :D]||, eps) at row strides ldx / ldy (y may be a column slice of the wider
// concatenated OCR feature row, m4c.py:235-237).  One wave per row.
__global__ __launch_bounds__(256) void l2norm_rows_f32_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + (size_t)r * ldx;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = xr[c]; s += v * v; }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), eps);
    float* yr = y + (size_t)r * ldy;
    for (int c = lane; c < D; c += 64) yr[c] = xr[c] * inv;
}
// out[r] = idx[r] < rows_a ? a[idx[r]] : b[idx[r] - rows_a]   (PrevPredEmbeddings' two-source gather, m4c.py:526-528)
__global__ __launch_bounds__(256) void gather_rows2_f32_kernel(const float* __restrict__ a, long rows_a, const float* __restrict__ b, long rows_b,
                                                                const int64_t* __restrict__ idx, float* __restrict__ out, int n, int H) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    long ix = idx[r];
    ix = ix < 0 ? 0 : (ix >= rows_a + rows_b ? rows_a + rows_b - 1 : ix);
    const float* src = ix < rows_a ? a + (size_t)ix * H : b + (size_t)(ix - rows_a) * H;
    for (int c = lane * 4; c < H; c += 256) *reinterpret_cast<f32x4*>(out + (size_t)r * H + c) = *reinterpret_cast<const f32x4*>(src + c);
}
// OcrPtrNet.forward (m4c.py:474-493): out[b, t, n] = scale <q[b, t], k[b, n]> + mask_add[b, n]; one workgroup per (b, t), a wave per n
__global__ __launch_bounds__(256) void ptr_scores_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ mask_add,
                                                              float* __restrict__ out, int ldo, int T, int N, int HQ, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bt = blockIdx.x, b = bt / T;
    const float* qr = q + (size_t)bt * HQ;
    for (int n = wave; n < N; n += 4) {
        const float* kr = k + ((size_t)b * N + n) * HQ;
        float s = 0.f;
        for (int c = lane * 4; c < HQ; c += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qr + c), bb = *reinterpret_cast<const f32x4*>(kr + c);
            s += (a[0] * bb[0] + a[1] * bb[1]) + (a[2] * bb[2] + a[3] * bb[3]);
        }
        s = wave_sum(s);
        if (lane == 0) out[(size_t)bt * ldo + n] = s * scale + (mask_add ? mask_add[(size_t)b * N + n] : 0.f);
    }
}

}  // namespace

template <int MI, bool AKM, bool BKM>
static void launch_gemm_f32_one(const GemmF32& g, dim3 grid, hipStream_t s) {
    constexpr int lds = 2 * (64 * MI + GBN) * GRS * (int)sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<MI, AKM, BKM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<MI, AKM, BKM>), grid, dim3(256), lds, s, g);
}
// (the k-major layouts exist with 64-row tiles only: their staging keeps more registers live, and three waves per SIMD — 168 registers —
// measured better than 128-row tiles at two)
static void launch_gemm_f32_64(const mmf_gemm_desc* d, const GemmF32& g, dim3 grid, hipStream_t s) {
    if (d->a_kmajor) launch_gemm_f32_one<1, true, true>(g, grid, s);
    else if (d->b_kmajor) launch_gemm_f32_one<1, false, true>(g, grid, s);
    else launch_gemm_f32_one<1, false, false>(g, grid, s);
}

// K splits of a (weight-gradient shaped) problem given a workspace: enough z-slices to put ~2 tiles on every CU, >= 8 K-steps each
extern "C" int mmf_gemm_f32_splits(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + GBN - 1) / GBN);      // (the weight-gradient layout runs 64-row tiles)
    const int nk = (K + GBK - 1) / GBK;
    if (tiles >= 512 || nk < 32) return 1;
    int sp = (int)((768 + tiles - 1) / tiles);
    if (sp > nk / 8) sp = nk / 8;
    if (sp > 32) sp = 32;
    return sp < 1 ? 1 : sp;
}

extern "C" int mmf_gemm_f32(const mmf_gemm_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->A && d->B && d->C, "gemm_f32: null operand");
    MMF_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm_f32: empty problem");
    MMF_CHECK_ARG(d->a_f32 && d->b_f32 && d->out_f32, "gemm_f32: operands and output are fp32 (a_f32 = b_f32 = out_f32 = 1)");
    MMF_CHECK_ARG(!d->a_kmajor || d->b_kmajor, "gemm_f32: layouts are forward (row, row), dgrad (row, k-major) and weight gradient (k-major, k-major)");
    MMF_CHECK_ARG((d->lda % 4) == 0 && (d->ldb % 4) == 0 && (((uintptr_t)d->A | (uintptr_t)d->B) & 15) == 0,
                  "gemm_f32: lda, ldb must be multiples of 4 and A, B 16-byte aligned");
    // a row operand is read in 4-element quads along K: its leading dimension must cover round_up(K, 4) and the padding must hold zeros
    const int k4 = (d->K + 3) / 4 * 4;
    MMF_CHECK_ARG(d->a_kmajor ? d->lda >= d->M : d->lda >= k4, "gemm_f32: lda must cover round_up(K, 4) for a row operand (k-major: lda >= M)");
    MMF_CHECK_ARG(d->b_kmajor ? d->ldb >= d->N : d->ldb >= k4, "gemm_f32: ldb must cover round_up(K, 4) for a row operand (k-major: ldb >= N)");
    MMF_CHECK_ARG(d->ldc >= d->N, "gemm_f32: ldc < N");
    MMF_CHECK_ARG(d->act >= 0 && d->act <= 4, "gemm_f32: act must be 0 (none), 1 (gelu), 2 (x aux), 3 (tanh) or 4 (x (1 - aux^2))");
    MMF_CHECK_ARG((d->act != 2 && d->act != 4) || d->aux, "gemm_f32: act 2 / 4 need aux");
    MMF_CHECK_ARG(!d->U || d->act == 1, "gemm_f32: U (the saved gelu') goes with act 1");
    MMF_CHECK_ARG(!d->rowsum_out, "gemm_f32: no fused row sums (use mmf_colsum_f32 for the bias gradient)");
    MMF_CHECK_ARG(!d->rowtab || (d->rowidx && d->rowtab_ld >= d->N), "gemm_f32: rowtab needs rowidx and rowtab_ld >= N");
    MMF_CHECK_ARG(!d->resid || d->ldr >= d->N, "gemm_f32: ldr < N");
    MMF_CHECK_ARG(d->grp_in >= 0, "gemm_f32: grp_in < 0");
    hipStream_t s = (hipStream_t)stream;
    GemmF32 g;
    g.A = (const float*)d->A; g.B = (const float*)d->B; g.C = (float*)d->C;
    g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
    g.bias = d->bias; g.coladd = d->coladd; g.rowtab = d->rowtab; g.rowidx = d->rowidx; g.rowtab_ld = d->rowtab_ld;
    g.act = d->act; g.U = (float*)d->U; g.aux = (const float*)d->aux; g.resid = (const float*)d->resid; g.ldr = d->ldr;
    g.drop = DropoutCfg{d->drop_key, d->drop_thr16, d->drop_scale, d->drop_seed};
    g.beta = d->beta;
    g.grp_in = d->grp_in; g.grp_pad = d->grp_pad; g.grp_off = d->grp_off;
    g.ksplit = 0;
    g.vec_ok = (d->ldc % 4) == 0 && (((uintptr_t)d->C | (uintptr_t)d->U | (uintptr_t)d->aux | (uintptr_t)d->resid | (uintptr_t)d->rowtab) & 15) == 0 &&
               (!d->resid || (d->ldr % 4) == 0) && (!d->rowtab || (d->rowtab_ld % 4) == 0);
    const int nt = (d->N + GBN - 1) / GBN;
    int splits = 1;
    if (d->splitk_ws) {
        splits = mmf_gemm_f32_splits(d->M, d->N, d->K);
        MMF_CHECK_ARG(d->splitk_ws_bytes >= (int64_t)splits * d->M * d->N * 4, "gemm_f32: split-K workspace too small (mmf_gemm_f32_splits(M, N, K) * M * N * 4 bytes)");
        MMF_CHECK_ARG(splits == 1 || (!d->bias && !d->coladd && !d->rowtab && d->act == 0 && !d->resid && d->drop_thr16 == 0 && d->grp_in == 0),
                      "gemm_f32: split-K takes no epilogue besides beta");
    }
    // 64-row tiles everywhere: 96-140 registers = three waves per SIMD, and more, smaller tiles round off better on 256 CUs; the 128-row
    // instantiation measured 82.6 / 91.8 TFLOP/s on the QKV / FFN-up shapes against 84.7 / 93.1 (tools/fp32_bench.py) and was dropped
    const int bm = 64;
    dim3 grid(nt, (d->M + bm - 1) / bm, splits);
    MMF_CHECK_ARG(grid.y <= 65535u, "gemm_f32: M too large for one launch");
    if (splits > 1) {
        const int nk = (d->K + GBK - 1) / GBK;
        g.ksplit = (nk + splits - 1) / splits;
        g.C = (float*)d->splitk_ws;
    }
    launch_gemm_f32_64(d, g, grid, s);
    MMF_CHECK_LAUNCH();
    if (splits > 1) {
        const long n = (long)d->M * d->N;
        hipLaunchKernelGGL(splitk_f32_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)d->splitk_ws, splits, n, d->N,
                           (float*)d->C, d->ldc, d->beta);
        MMF_CHECK_LAUNCH();
    }
    return 0;
}

template <int D, int MAXT>
static int launch_attn_f32(const AttnF32& a, hipStream_t s) {
    constexpr int lds = (2 * 16 * MAXT * (D + 4) + 16 * MAXT) * (int)sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_fwd_kernel<D, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once = true;
    }
    if (a.Sk > 16 * MAXT) {     // more keys than one LDS image: the blocked two-pass form
        static bool once_long = false;
        if (!once_long) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_fwd_long_kernel<D, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            once_long = true;
        }
        hipLaunchKernelGGL((attn_f32_fwd_long_kernel<D, MAXT>), dim3((a.Sq + 127) / 128, a.B * a.heads), dim3(512), lds, s, a);
    } else
    hipLaunchKernelGGL((attn_f32_fwd_kernel<D, MAXT>), dim3((a.Sq + 127) / 128, a.B * a.heads), dim3(512), lds, s, a);
    MMF_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmf_attention_f32_fwd(const mmf_attn_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->q && d->k && d->v && d->ctx, "attention_f32_fwd: null operand");
    MMF_CHECK_ARG(d->B > 0 && d->heads > 0 && d->Sq > 0 && d->Sk > 0, "attention_f32_fwd: empty problem");
    const int hd = d->head_dim ? d->head_dim : 64;
    MMF_CHECK_ARG(hd == 64 || hd == 128, "attention_f32_fwd: head_dim must be 64 or 128");
    MMF_CHECK_ARG(d->Sk <= (hd == 64 ? 512 : 256) && d->Sq <= (hd == 64 ? 512 : 256), "attention_f32_fwd: Sq, Sk <= 512 (head_dim 64) / 256 (head_dim 128), like the bf16 kernels");
    MMF_CHECK_ARG(d->mask_query_stride == 0 || (d->mask && d->mask_query_stride >= d->Sk && d->causal_tail == 0),
                  "attention_f32_fwd: mask_query_stride must cover a mask row (>= Sk) and replaces the causal tail");
    MMF_CHECK_ARG(!d->ctx_f32 && d->q_batch_rows == 0 && d->kv_batch_rows == 0 && (d->mask_batch_stride == 0 || d->mask_query_stride != 0),
                  "attention_f32_fwd: no ctx_f32 (ctx IS fp32) and no K|V cache strides");
    MMF_CHECK_ARG(d->causal_tail >= 0 && d->causal_tail <= d->Sk && (d->causal_tail == 0 || d->Sq == d->Sk),
                  "attention_f32_fwd: a causal tail needs self-attention (Sq == Sk)");
    const int HD = d->heads * hd;
    MMF_CHECK_ARG(d->ldq >= HD && d->ldk >= HD && d->ldv >= HD && d->ldo >= HD, "attention_f32_fwd: leading dimension < heads * head_dim");
    MMF_CHECK_ARG((d->ldq % 4) == 0 && (d->ldk % 4) == 0 && (d->ldv % 4) == 0 && (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) == 0,
                  "attention_f32_fwd: q / k / v rows are read as 16-byte quads (pointers 16-byte aligned, leading dimensions multiples of 4)");
    MMF_CHECK_ARG((size_t)d->B * d->heads <= 65535u, "attention_f32_fwd: B * heads too large for one launch");
    AttnF32 a;
    a.q = (const float*)d->q; a.k = (const float*)d->k; a.v = (const float*)d->v; a.out = (float*)d->ctx;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
    a.mask = d->mask; a.B = d->B; a.heads = d->heads; a.Sq = d->Sq; a.Sk = d->Sk; a.scale = d->scale;
    a.cfrom = d->Sk - d->causal_tail;
    a.mqs = d->mask_query_stride;
    a.mhs = d->mask_head_stride;
    MMF_CHECK_ARG(a.mhs == 0 || (a.mqs != 0 && a.mhs >= (d->Sq - 1) * a.mqs + d->Sk), "attention_f32_fwd: mask_head_stride goes with a per-query mask and must cover one head's [Sq, Sk] mask");
    a.mbs = d->mask_batch_stride > 0 ? d->mask_batch_stride : (a.mhs ? d->heads * a.mhs : d->Sq * d->mask_query_stride);
    MMF_CHECK_ARG(a.mqs == 0 || a.mbs >= (a.mhs ? (d->heads - 1) * a.mhs : 0) + (d->Sq - 1) * a.mqs + d->Sk, "attention_f32_fwd: mask_batch_stride must cover the per-query mask of a sample");
    a.lse = d->lse;
    a.drop = DropoutCfg{d->drop_key, d->drop_thr16, d->drop_scale, d->drop_seed};
    a.o = nullptr; a.d_o = nullptr; a.dq = a.dk = a.dv = a.delta = nullptr;
    return hd == 64 ? launch_attn_f32<64, 16>(a, (hipStream_t)stream) : launch_attn_f32<128, 8>(a, (hipStream_t)stream);
}

template <int D, int MAXT>
static int launch_attn_f32_bwd(const AttnF32& a, hipStream_t s) {
    constexpr int lds1 = (2 * 16 * MAXT * (D + 4) + 16 * MAXT) * (int)sizeof(float);
    constexpr int lds2 = (2 * 16 * MAXT * (D + 4) + 2 * 16 * MAXT) * (int)sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_bwd_dq_kernel<D, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_bwd_dkv_kernel<D, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        once = true;
    }
    hipLaunchKernelGGL((attn_f32_bwd_dq_kernel<D, MAXT>), dim3((a.Sq + 127) / 128, a.B * a.heads), dim3(512), lds1, s, a);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_f32_bwd_dkv_kernel<D, MAXT>), dim3((a.Sk + 127) / 128, a.B * a.heads), dim3(512), lds2, s, a);
    MMF_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmf_attention_f32_bwd(const mmf_attn_bwd_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->f.q && d->f.k && d->f.v && d->f.ctx && d->f.lse && d->dctx && d->dq && d->dk && d->dv && d->delta,
                  "attention_f32_bwd: null operand (q, k, v, ctx = O, lse, dctx, dq, dk, dv, delta)");
    const mmf_attn_desc* f = &d->f;
    MMF_CHECK_ARG(f->B > 0 && f->heads > 0 && f->Sq > 0 && f->Sk > 0, "attention_f32_bwd: empty problem");
    const int hd = f->head_dim ? f->head_dim : 64;
    MMF_CHECK_ARG(hd == 64 || hd == 128, "attention_f32_bwd: head_dim must be 64 or 128");
    const int smax = hd == 64 ? 512 : 256;
    MMF_CHECK_ARG(f->Sk <= smax && f->Sq <= smax, "attention_f32_bwd: Sq, Sk <= 512 (head_dim 64) / 256 (head_dim 128), like the bf16 kernels");
    MMF_CHECK_ARG(f->mask_query_stride == 0 || (f->mask && f->mask_query_stride >= f->Sk && f->causal_tail == 0),
                  "attention_f32_bwd: mask_query_stride must cover a mask row (>= Sk) and replaces the causal tail");
    MMF_CHECK_ARG(!f->ctx_f32 && f->q_batch_rows == 0 && f->kv_batch_rows == 0 && (f->mask_batch_stride == 0 || f->mask_query_stride != 0),
                  "attention_f32_bwd: no ctx_f32 / K|V cache strides");
    MMF_CHECK_ARG(f->causal_tail >= 0 && f->causal_tail <= f->Sk && (f->causal_tail == 0 || f->Sq == f->Sk), "attention_f32_bwd: a causal tail needs Sq == Sk");
    const int HD = f->heads * hd;
    MMF_CHECK_ARG(f->ldq >= HD && f->ldk >= HD && f->ldv >= HD && f->ldo >= HD, "attention_f32_bwd: leading dimension < heads * head_dim");
    MMF_CHECK_ARG((f->ldq % 4) == 0 && (f->ldk % 4) == 0 && (f->ldv % 4) == 0 && (f->ldo % 4) == 0 &&
                  (((uintptr_t)f->q | (uintptr_t)f->k | (uintptr_t)f->v | (uintptr_t)f->ctx | (uintptr_t)d->dctx) & 15) == 0,
                  "attention_f32_bwd: rows are read as 16-byte quads (aligned pointers, leading dimensions multiples of 4)");
    MMF_CHECK_ARG((size_t)f->B * f->heads <= 65535u, "attention_f32_bwd: B * heads too large for one launch");
    AttnF32 a;
    a.q = (const float*)f->q; a.k = (const float*)f->k; a.v = (const float*)f->v; a.out = nullptr;
    a.ldq = f->ldq; a.ldk = f->ldk; a.ldv = f->ldv; a.ldo = f->ldo;
    a.mask = f->mask; a.B = f->B; a.heads = f->heads; a.Sq = f->Sq; a.Sk = f->Sk; a.scale = f->scale;
    a.mqs = f->mask_query_stride;
    a.mhs = f->mask_head_stride;
    MMF_CHECK_ARG(a.mhs == 0 || (a.mqs != 0 && a.mhs >= (f->Sq - 1) * a.mqs + f->Sk), "attention_f32_bwd: mask_head_stride goes with a per-query mask and must cover one head's [Sq, Sk] mask");
    a.mbs = f->mask_batch_stride > 0 ? f->mask_batch_stride : (a.mhs ? f->heads * a.mhs : f->Sq * f->mask_query_stride);
    MMF_CHECK_ARG(a.mqs == 0 || a.mbs >= (a.mhs ? (f->heads - 1) * a.mhs : 0) + (f->Sq - 1) * a.mqs + f->Sk, "attention_f32_bwd: mask_batch_stride must cover the per-query mask of a sample");
    a.cfrom = f->Sk - f->causal_tail;
    a.lse = f->lse;
    a.drop = DropoutCfg{f->drop_key, f->drop_thr16, f->drop_scale, f->drop_seed};
    a.o = (const float*)f->ctx; a.d_o = (const float*)d->dctx; a.dq = (float*)d->dq; a.dk = (float*)d->dk; a.dv = (float*)d->dv; a.delta = d->delta;
    return hd == 64 ? launch_attn_f32_bwd<64, 16>(a, (hipStream_t)stream) : launch_attn_f32_bwd<128, 8>(a, (hipStream_t)stream);
}

extern "C" int mmf_layernorm_f32_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int H, float eps,
                                     void* stream) {
    MMF_CHECK_ARG(x && gamma && beta && y, "layernorm_f32_fwd: null operand");
    MMF_CHECK_ARG(rows > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "layernorm_f32_fwd: H % 4 == 0, H <= 2048");
    MMF_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_f32_fwd: 16-byte alignment");
    hipLaunchKernelGGL(ln_f32_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, H, eps, (float*)nullptr, (float*)nullptr);
    MMF_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmf_layernorm_f32_fwd_stats(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int rows, int H,
                                           float eps, void* stream) {
    MMF_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_f32_fwd_stats: null operand");
    MMF_CHECK_ARG(rows > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "layernorm_f32_fwd_stats: H % 4 == 0, H <= 2048");
    MMF_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_f32_fwd_stats: 16-byte alignment");
    hipLaunchKernelGGL(ln_f32_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, H, eps, mean, rstd);
    MMF_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmf_pad_rows_f32(const float* src, int K, float* dst, int KP, int rows, void* stream) {
    MMF_CHECK_ARG(src && dst && rows > 0 && K > 0 && KP >= K, "pad_rows_f32: bad operand");
    const long n = (long)rows * KP;
    hipLaunchKernelGGL(pad_rows_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (long)K, K, dst, KP, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
// the same from a column slice of wider rows: dst[r][0..KP) = src[r * ld_src + 0..K) followed by zeros (the fixed-vocabulary part of M4C's
// [B T, 5000 + 50] score gradient as a 16-byte-row GEMM operand)
extern "C" int mmf_slice_rows_f32(const float* src, int ld_src, int K, float* dst, int KP, int rows, void* stream) {
    MMF_CHECK_ARG(src && dst && rows > 0 && K > 0 && KP >= K && ld_src >= K, "slice_rows_f32: bad operand");
    const long n = (long)rows * KP;
    hipLaunchKernelGGL(pad_rows_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, K, dst, KP, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmf_eltwise_f32(int op, const float* a, const float* b, float* y, long n, void* stream) {
    MMF_CHECK_ARG(a && y && n > 0 && (op == 0 || op == 1 || (op >= 3 && op <= 5)) && (op == 1 || b), "eltwise_f32: bad operand (op 0 mul, 1 relu, 3 add, 4 tanh backward, 5 relu backward)");
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

extern "C" int mmf_l2norm_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int D, float eps, void* stream) {
    MMF_CHECK_ARG(x && y && rows > 0 && D > 0 && ldx >= D && ldy >= D, "l2norm_rows_f32: bad operand");
    hipLaunchKernelGGL(l2norm_rows_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, D, eps);
    MMF_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmf_gather_rows2_f32(const float* a, int64_t rows_a, const float* b, int64_t rows_b, const int64_t* idx, float* out, int n, int H, void* stream) {
    MMF_CHECK_ARG(a && b && idx && out && n > 0 && H > 0 && (H % 4) == 0 && rows_a >= 0 && rows_b >= 0 && rows_a + rows_b > 0, "gather_rows2_f32: bad operand");
    MMF_CHECK_ARG((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "gather_rows2_f32: 16-byte alignment");
    hipLaunchKernelGGL(gather_rows2_f32_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, (long)rows_a, b, (long)rows_b, idx, out, n, H);
    MMF_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmf_ptr_scores_f32(const float* q, const float* k, const float* mask_add, float* out, int ldo, int B, int T, int N, int HQ, float scale,
                                  void* stream) {
    MMF_CHECK_ARG(q && k && out && B > 0 && T > 0 && N > 0 && HQ > 0 && (HQ % 4) == 0 && ldo >= N, "ptr_scores_f32: bad operand");
    MMF_CHECK_ARG((((uintptr_t)q | (uintptr_t)k) & 15) == 0, "ptr_scores_f32: 16-byte alignment");
    hipLaunchKernelGGL(ptr_scores_f32_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, q, k, mask_add, out, ldo, T, N, HQ, scale);
    MMF_CHECK_LAUNCH();
    return 0;
}
