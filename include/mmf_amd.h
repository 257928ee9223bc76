/* mmf_amd.h — C ABI of libmmf_amd.so: the MI355X (gfx950 / CDNA4) kernels behind MMF's cross-modal
 * transformer fusion hot path (VisualBERT first).
 *
 * The reference (facebookresearch/mmf) has no FFI on this path: every operation below is, there, a
 * stock ATen call issued from Python (`nn.Linear`, `torch.matmul`, `softmax`, `nn.LayerNorm`,
 * `nn.Dropout`, `nn.Embedding`, autograd).  Each entry point cites the reference call site it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding and the
 * `torch.autograd.Function` / `nn.Module` mirror that a reference maintainer would register through
 * `mmf.common.registry`.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless the name ends in `_host`;
 *   - `stream` is a `hipStream_t` passed as `void*`; every call only ENQUEUES work on it (no host
 *     sync, no allocation) so the whole step can be captured in a hipGraph;
 *   - "bf16" buffers are IEEE bfloat16 (upper 16 bits of fp32, round-to-nearest-even);
 *   - return 0 on success, non-zero on error; `mmf_amd_last_error()` returns the message;
 *   - matrices are row-major with an explicit leading dimension (`ld*`, in elements).
 */
#ifndef MMF_AMD_H
#define MMF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMF_AMD_ABI_VERSION 1

/* ---- library ------------------------------------------------------------------------------ */
int mmf_amd_abi_version(void);
/* Integer tuning knobs for on-hardware sweeps (0 = built-in heuristic).  Not part of the reference's interface. */
enum { MMF_TUN_GEMM_WIDE = 2,      /* forward-form GEMM tile: 0 model picks, -1 never a wide tile, 1 .. 6 force 256x96 / 192x192 / 256x128 / 128x96 / 128x128 / 192x96 (tests, A/B) */
       MMF_TUN_ALT_FORMS = 3,      /* cross-check hooks (tests compare kernel forms that serve different shapes in production): bit 0 LayerNorm with the one-wave-per-row
                                      kernels even when H % 256 == 0; bit 1 head_dim-64 attention forward with > 128 queries as two 4-wave workgroups per head; bit 2
                                      attention backward as the separate dQ and dK/dV kernels where the one-pass kernel would run (and no keep-bit table); bit 3
                                      LayerNorm backward with one row in flight per half-wave; bit 4 the 256-row wide GEMM tiles also where few token rows would take
                                      the 128x96 tile (A/B of that rule); bit 5 mmf_gemm_bf16_grouped_ln never lets the LayerNorm ride (A/B) */
       MMF_TUN_EPI_NT = 6,         /* GEMM epilogue non-temporal stores: 0 default, else value - 1 = mask (bit 0 bf16 C, bit 1 saved gelu', bit 2 fp32 C) */
       MMF_TUN_NT_SITE_KEEP = 8,   /* bit s set: the bf16 output of GEMM calls tagged MMF_GEMM_SITE(s) is stored TEMPORALLY (stays in L2 / the Infinity Cache for the
                                      kernel that consumes it next) although MMF_TUN_EPI_NT stores outputs non-temporally; 0 (default): no exception (A/B) */
       MMF_TUN_WGRAD_WIDE = 10,    /* grouped weight-gradient launch: 0 the 256x128 wide tile when every problem is a whole number of such tiles and the launch fills
                                      most of a round of the 256 CUs, 1 never (the 128x128 tiles, two workgroups per CU), 2 whenever the shapes allow (tests, A/B) */
       MMF_TUN_SC1_SITE = 14,      /* bit s set: the bf16 output of GEMM calls tagged MMF_GEMM_SITE(s) is stored write-through (`sc1`, not `nt`): the line is dropped from
                                      the writing XCD's L2 like a streaming store but allocates in the Infinity Cache, where the next kernel finds it (an `nt` store
                                      bypasses it: tools/cold_operand_probe.py, profiles/r04_store_policy.txt).  0: the measured default (MMF_SITE_SC1_DEFAULT: the FFN
                                      up-projection's GELU output and the FFN-down dgrad's du, the two A operands of the K = 3072 GEMMs); 1: no site at all (A/B) */
       MMF_TUN_SCATTER_ATOMIC = 17,   /* 1: mmf_rows_scatter_add with an index array always takes the fp32-atomic kernel (the reference the deterministic owner kernel
                                          is tested against; by itself the library takes atomics beyond 16384 rows and for ids on more than 64 rows) */
       MMF_TUN_GEMM_PERSIST = 18,  /* NT-form GEMMs on the persistent kernel (gemm_persist.h: one workgroup per CU walks several tiles, the epilogue of a tile runs
                                      under the K-loop of the next): 0 where the measured rule says so (several tiles per workgroup, short K-loops), -1 never, 1 / 2 / 3 always, on the
                                      256x96 / 192x192 / 256x128 tile, 256 + mask: exactly the calls tagged MMF_GEMM_SITE(s) with bit s of mask set (A/B) */
       MMF_TUN_COUNT = 19 };       /* (seven knobs; the other slots were measurement switches of rounds 1 - 5 whose losing branches are gone) */
/* Call-site tag of a GEMM (bits 20..23 of mmf_gemm_desc::debug_flags; 0 = untagged).  It selects nothing by itself: it only names the call for
 * MMF_TUN_NT_SITE_KEEP.  The encoder layer's calls: */
#define MMF_GEMM_SITE(s) (((s) & 15) << 20)
enum { MMF_SITE_QKV_FWD = 1,        /* Q|K|V projection -> read next by the attention forward */
       MMF_SITE_ATTN_OUT_FWD = 2,   /* attention output projection (+ residual) -> read next by the LayerNorm */
       MMF_SITE_FFN_UP_FWD = 3,     /* intermediate GELU(h W1^T) -> the A operand of the FFN-down GEMM */
       MMF_SITE_FFN_DOWN_FWD = 4,   /* FFN output projection (+ residual) -> LayerNorm */
       MMF_SITE_FFN_DOWN_DGRAD = 5, /* du = (dlin2 W2) * gelu' -> the A operand of the FFN-up dgrad (and of its weight gradient) */
       MMF_SITE_FFN_UP_DGRAD = 6,   /* da = du W1 + dres -> LayerNorm backward */
       MMF_SITE_ATTN_OUT_DGRAD = 7, /* dctx = dlin1 Wo -> attention backward */
       MMF_SITE_QKV_DGRAD = 8 };    /* dx = dqkv Wqkv + dres -> the layer below */
int mmf_amd_set_tunable(int which, int value);
int mmf_amd_get_tunable(int which);
const char* mmf_amd_last_error(void);
/* Name of the gfx target the device code was compiled for ("gfx950"). */
const char* mmf_amd_target(void);

/* ---- GEMM with fused epilogue ----------------------------------------------------------------
 * C[m][n] = epilogue( sum_k A(m,k) * B(n,k) ), bf16 MFMA, fp32 accumulate.
 *   a_kmajor = 0: A(m,k) = A[m*lda + k]      a_kmajor = 1: A(m,k) = A[k*lda + m]
 *   b_kmajor = 0: B(n,k) = B[n*ldb + k]      b_kmajor = 1: B(n,k) = B[k*ldb + n]
 * Replaces: nn.Linear forward  (mmf/modules/hf_layers.py:169,179-180; HF BertSelfOutput /
 * BertIntermediate / BertOutput at hf_layers.py:248,289,290; mmf/modules/embeddings.py:352;
 * mmf/models/visual_bert.py:401), and its autograd dgrad (a=0,b=1) / wgrad (a=1,b=1)
 * (mmf/trainers/core/training_loop.py:211).
 * Epilogue, in this order, on v = acc:
 *   v += bias[n]; v += coladd[n]; v += rowtab[rowidx[m]*rowtab_ld + n];
 *   act==1: U[m][n] = gelu_erf'(v) (if U), v = gelu_erf(v)   (HF BertIntermediate; exact-erf GELU)
 *   act==2: v *= aux[m][n]                                   (backward of the above: aux = the saved U)
 *   act==3: v = tanh(v)                                      (HF BertPooler, mmf/models/mmbt.py:311)
 *   act==4: v *= 1 - aux[m][n]^2                             (backward of tanh: aux = the saved output)
 *   dropout(v) with (drop_key, drop_thr16, drop_scale), element index m*N+n
 *   v += resid[m][n]                                   (HF BertSelfOutput / BertOutput residual)
 *   out_f32 ? C = v + beta*C (float) : C = bf16(v)
 * Output rows can be remapped: row = m + (m / grp_in) * grp_pad + grp_off when grp_in > 0
 * (writes the visual rows of the joint [B, T+R, H] sequence, embeddings.py:447-451).
 * Constraints: lda, ldb multiples of 8; a row operand is read in 8-element chunks along K, so its
 * leading dimension must cover round_up(K, 8) and the padding must be finite (zeros); a k-major
 * operand must be readable up to round_up(rows, 8) columns; fp32 operands
 * (a_f32 / b_f32) are converted to bf16 on the fly (at most one of the two).
 */
typedef struct mmf_gemm_desc {
    const void* A;
    const void* B;
    void* C;
    int M, N, K;
    int lda, ldb, ldc;
    int a_kmajor, b_kmajor;
    int a_f32, b_f32, out_f32;
    float beta;
    const float* bias;
    const float* coladd;
    const float* rowtab;
    const int64_t* rowidx;
    int rowtab_ld;
    int act;
    void* U;
    const void* aux;
    const void* resid;
    int ldr;
    uint32_t drop_key;
    uint32_t drop_thr16;
    float drop_scale;
    const uint32_t* drop_seed; /* optional device word mixed into drop_key at run time (hipGraph replays) */
    int grp_in, grp_pad, grp_off;
    void* splitk_ws;          /* optional fp32 workspace enabling deterministic split-K (fp32 output, no epilogue) */
    int64_t splitk_ws_bytes;  /* >= mmf_gemm_splitk_splits(M,N,K) * M * (N + 1) * 4 to take effect */
    float* rowsum_out;        /* optional fp32 [M], weight-gradient form only (a_kmajor && b_kmajor, bf16 operands):
                                 rowsum_out[m] = sum_k A[k][m] — the bias gradient (column sums of dY; autograd of the `+ bias`
                                 of nn.Linear, hf_layers.py:169-180) computed by the same launch with one extra MFMA per A
                                 fragment against a ones operand; with split-K it travels through the workspace (behind the
                                 slabs) and is summed by the slab reduction, else it is written directly */
    int debug_flags;          /* 0 in production.  bit 8: 4-wave workgroups, bit 9: never use 128x96 tiles, bit 12 / 13: force / forbid the K-split wave layout, bits 4-7: ablation switches, bit 17: never a wide (one workgroup per CU) tile; (bit 18 is the Python binding's: no skinny-path workspace) */
} mmf_gemm_desc;
int mmf_gemm_bf16(const mmf_gemm_desc* d, void* stream);
/* Skinny problems (a row-major A of at most 64 rows and a long reduction: the classification heads of visual_bert.py:349-404 and their input
 * gradients): number of K-slices mmf_gemm_bf16 spreads over the chip when `splitk_ws` holds splits * M * round_up(N, 8) floats — ANY epilogue,
 * it runs on the slab sums in a second kernel.  Returns 1 when the problem is not skinny (no workspace needed). */
int mmf_gemm_skinny_splits(int M, int N, int K, int a_kmajor);
/* `count` (1..8) independent GEMMs of ONE operand layout (a_kmajor, b_kmajor, a_f32, b_f32 equal) in one launch: the tile
 * lists are concatenated, so small problems fill the chip together.  No split-K (splitk_ws ignored); each problem keeps its
 * own epilogue and rowsum_out.  Replaces the four weight-gradient GEMMs autograd issues per transformer layer
 * (dW = dY^T X of the Linear layers at hf_layers.py:169-180 and of HF BertSelfOutput / BertIntermediate / BertOutput,
 * call sites hf_layers.py:248,289,290; mmf/trainers/core/training_loop.py:211). */
int mmf_gemm_bf16_grouped(const mmf_gemm_desc* descs, int count, void* stream);
/* The grouped launch above together with ONE deferred LayerNorm backward that does not depend on it (the training step's order: a layer's four weight
 * gradients, then the first LayerNorm backward of the layer below; reference: autograd of HF BertOutput.LayerNorm, mmf/modules/hf_layers.py:290, beside the
 * weight gradients of nn.Linear at :169-180,248,289-290).  `ln` is mmf_layernorm_bwd's argument list with dgamma = dbeta = dbias = NULL (column-sum partials
 * only: finish with mmf_layernorm_bwd_reduce_multi).  Where the weight gradients run one 256 x 128 tile per CU and leave CUs idle (216 tiles on 256 CUs for a
 * BERT-base layer at 7296 tokens) and the LayerNorm is 768 wide with two rows per half-wave, the LayerNorm runs on those CUs inside the SAME launch; otherwise
 * the two launches follow each other.  Results are bit-identical either way.  MMF_TUN_ALT_FORMS bit 5: never ride (A/B). */
typedef struct mmf_ln_bwd_desc {
    const void* dy; const void* x; const float* mean; const float* rstd; const float* gamma;
    void* dx; void* dlin;                       /* bf16 [rows, H]; dlin (dropout backward of the preceding Linear) may be NULL when drop_thr16 == 0 */
    uint32_t drop_key, drop_thr16; float drop_scale; const uint32_t* drop_seed;
    float* partials;                            /* mmf_layernorm_bwd_ws_floats(H) floats */
    int rows, H;
} mmf_ln_bwd_desc;
int mmf_gemm_bf16_grouped_ln(const mmf_gemm_desc* descs, int count, const mmf_ln_bwd_desc* ln, void* stream);
/* Development aid (profiling, not a reference operation): while `buf` (device memory, (1 + capacity_records) * 64 bytes,
 * zeroed) is set, every workgroup of every GEMM launch appends one 64-byte timeline record of s_memrealtime (100 MHz) stamps
 * {launch << 32 | block, HW_ID | XCC_ID << 32, entry, first stage landed, K loop done, tile staged, stores drained, tile};
 * word 0 of the buffer counts the records.  NULL switches it off. */
int mmf_gemm_set_probe(void* buf, int64_t capacity_records);
/* Host-side log of the launches probed since mmf_gemm_set_probe: 5 int64 per launch {launch id, layout (bit 0 A k-major, bit 1 B k-major,
 * bit 2 grouped), M, N, K} (grouped: problems, tiles, K) copied into out_host; returns the number of launches logged. */
int mmf_gemm_probe_log(int64_t* out_host, int capacity);
/* Development aid, attention kernels: like mmf_gemm_set_probe with one 96-byte record per WAVE {kernel (0 fwd, 1 dQ, 2 dK/dV),
 * batch * heads + head, 4 * blockIdx.y + wave, HW_ID, 8 stamps}.  Only a library built with -DMMF_ATTN_PROBE records (the
 * stamps are compiled out of the regular build, where the call returns 1); tools/attn_timeline.py. */
int mmf_attention_set_probe(void* buf, int64_t capacity_records);
/* Measurement aid: family / tile of the kernel the last GEMM call on this thread launched ("gemm_wide_kernel 256x96", ...). */
const char* mmf_gemm_last_kernel(void);
/* Number of K splits mmf_gemm_bf16 will use for this shape when given a workspace (1 = no split). */
int mmf_gemm_splitk_splits(int M, int N, int K);

/* ---- fused multi-head attention -------------------------------------------------------------
 * Replaces BertSelfAttentionJit.forward, mmf/modules/hf_layers.py:161-213 (scores = QK^T /
 * sqrt(d) + mask; softmax; dropout; PV; head merge) without materialising the [B,A,S,S] tensors,
 * and its autograd backward.  head_dim is 64 (Sq, Sk <= 512 = BERT's max_position_embeddings; the tuned forms — two workgroups per CU forward,
 * one-pass backward — up to 256, the same kernels with more key tiles and the two-kernel backward beyond) or 128 (Sq, Sk <= 256; tuned up to 128: the
 * visual and co-attention streams of mmf/models/vilbert.py:153-247,347-475; q and k/v may come from different sequences, Sq != Sk).  A head's K and V
 * are always staged whole in LDS, so the softmax is the reference's exact two-pass form at every length.
 * q/k/v/ctx are token-major: element (b, s, head, e) at ptr[(b*S + s)*ld + head*head_dim + e]  (for the packed QKV projection output: q = qkv, k = qkv+H,
 * v = qkv+2H, ld = 3H).  `mask` is the ADDITIVE key mask of visual_bert.py:94-106, shape [B, Sk]
 * fp32 (0 or -10000), or NULL.  lse[b][head][s] = log-sum-exp of the masked, scaled scores row.
 * Dropout element index: ((b*heads + head)*Sq + q)*Sk_pad + key with Sk_pad = round_up(Sk, 32).
 */
typedef struct mmf_attn_desc {
    const void* q;
    const void* k;
    const void* v;
    int ldq, ldk, ldv;
    const float* mask;
    void* ctx;
    int ldo;
    float* lse;
    int B, heads, Sq, Sk;
    float scale;
    uint32_t drop_key;
    uint32_t drop_thr16;
    float drop_scale;
    const uint32_t* drop_seed;
    int head_dim;   /* 0 = 64 */
    float* ctx_f32; /* optional fp32 copy of ctx (layout of ctx, ldo): forward writes it, backward then forms
                       delta = rowsum(dO o O) from the unrounded O, which keeps sum_key dS = 0 to fp32 accuracy */
    int causal_tail; /* 0 = none.  n > 0: the prefix-LM mask of M4C's multimodal transformer, MMT.forward,
                        mmf/models/m4c.py:424-440 (a [B,1,L,L] mask there), without materialising it: the last n
                        positions (decoding steps) are visible only to each other, causally (key <= query); all
                        other pairs use `mask`.  Needs Sq == Sk and head_dim 64. */
    int q_batch_rows;      /* forward only, 0 = Sq: rows between the first query of consecutive batches (queries that live inside a
                              longer per-sample buffer) */
    int kv_batch_rows;     /* forward only, 0 = Sk: the same for k / v — a per-sample K|V cache longer than the Sk keys in use.
                              Incremental greedy decoding of M4C (mmf/models/m4c.py:284-304 re-runs the whole multimodal
                              transformer per step): step i attends from ONE new row to the cached encoder rows and the i
                              decoding rows before it, i.e. Sq = 1, Sk = E + i + 1, both strides = E + D */
    int mask_batch_stride; /* 0 = Sk (Sq * mask_query_stride with a per-query mask): mask entries per batch; a custom value is forward only */
    int mask_query_stride; /* 0: `mask` is the additive KEY mask [B, Sk] every model of the path uses (the reference's [B, 1, 1, S]).  n >= Sk: `mask` is a
                              materialised additive mask per (query, key) pair, entry (b, q, key) at mask[b * mask_batch_stride + q * n + key] — what
                              BertSelfAttentionJit.forward accepts as a [B, 1, S, S] attention_mask (`attention_scores + attention_mask`,
                              mmf/modules/hf_layers.py:187-190; MMT.forward builds one, mmf/models/m4c.py:424-440).  head_dim 64, forward and backward;
                              replaces `causal_tail` (which is the cheaper form of M4C's mask). */
    uint32_t* keep_bits;   /* optional, NULL = off.  The probability-dropout decisions of the forward as a bit table: the forward writes it while it draws them
                              (one ballot per MFMA register, no second hash), the backward reads ONE word per lane and key tile instead of hashing every
                              probability again — the counter hash is the largest VALU item of the backward (12.6 of 58 us at the VQA2 shape,
                              profiles/r05_experiments.txt).  Same decisions, bit for bit.  mmf_attention_keep_bits_words(...) 32-bit words, 0 = this
                              shape's kernels do not take a table (pass NULL).  Word ((bh * nqt + qt) * nkt + kt) * 32 + j, bit x = keep(query 32 qt + x,
                              key 32 kt + j); nqt = ceil(Sq / 32), nkt = ceil(Sk / 32). */
    const uint32_t* keep_lanes; /* optional, NULL = off (forward only).  The decisions were drawn AHEAD of this call by mmf_attention_draw_keep_bits with the same
                              (drop_key, drop_seed, drop_thr16, shape): the forward reads them in its own lane order — a 16-bit field per key tile, one 16-byte
                              load per lane — instead of hashing (8 of the forward's 31 us at the VQA2 shape), and does not write `keep_bits` (the same
                              launch drew that table too; hand it to the backward as before).  Same decisions, same outputs, bit for bit.
                              mmf_attention_keep_lanes_words(...) 32-bit words, 16-byte aligned. */
    int mask_head_stride;  /* 0: one mask for all heads.  n > 0 (with mask_query_stride): `mask` holds one [Sq, Sk] mask per (sample, head) — what
                              BertSelfAttentionJit.forward accepts as a [B, heads, S, S] attention_mask (the `+` of mmf/modules/hf_layers.py:187-190
                              broadcasts whatever it is given) — entry (b, head, q, key) at mask[b * mask_batch_stride + head * n + q * mask_query_stride
                              + key]; mask_batch_stride then defaults to heads * n.  Forward and backward, bf16 and fp32 kernels. */
} mmf_attn_desc;
int mmf_attention_fwd(const mmf_attn_desc* d, void* stream);
/* The probability-dropout decisions of up to any number of attention sites in ONE launch, ahead of the kernels that use them (they
 * depend on (key, seed word, element index) only, never on the scores — hf_layers.py:196-200 draws them with nn.Dropout after the softmax).  Writes, per
 * site, the key-major table `keep_bits` (what the forward would have written) and the lane-major `keep_lanes` the forward then reads.  A training step
 * issues it once at its head on a side stream (a parallel branch of the step's hipGraph beside the embedding stage). */
#define MMF_ATTN_DRAW_MAX 48
typedef struct mmf_attn_draw_site {
    uint32_t drop_key, drop_thr16;
    const uint32_t* drop_seed;
    int B, heads, Sq, Sk, head_dim;
    uint32_t* keep_bits;    /* [mmf_attention_keep_bits_words] */
    uint32_t* keep_lanes;   /* [mmf_attention_keep_lanes_words], 16-byte aligned */
} mmf_attn_draw_site;
/* seed_offset is added to every site's seed word before it enters the key: 0 = the decisions a forward launched NOW would draw; 1 = those of the next
 * step of a graphed loop, whose head (mmf_step_advance) adds one to the word (a step drawing its successor's decisions beside its own HBM-bound AdamW
 * launches).  Round 6 measured both placements inside the VisualBERT step (a side branch of the hipGraph beside the embedding stage, or beside AdamW):
 * the forwards gain 57 us, the 89 us draw hides nowhere and the branch costs the replay idle time: +0.11 ms per step, so the training steps of this
 * package keep hashing in the kernels (profiles/r06_experiments.txt section 1; the step-level plumbing of that experiment: commit 8ce4125). */
int mmf_attention_draw_keep_bits(const mmf_attn_draw_site* sites, int n, uint32_t seed_offset, void* stream);
/* Words of keep_lanes for this shape (0 where mmf_attention_keep_bits_words' shape rule gives 0): B * heads * ceil(Sq / 32) * 64 lanes * 4. */
int64_t mmf_attention_keep_lanes_words(int B, int heads, int Sq, int Sk, int head_dim);
/* Words of mmf_attn_desc.keep_bits for this shape, 0 when its backward is the two-kernel form, which hashes (head_dim 64 beyond 256 queries or keys, head_dim 128
 * beyond 128). */
int64_t mmf_attention_keep_bits_words(int B, int heads, int Sq, int Sk, int head_dim);

/* Backward of the same operator (what autograd derives for BertSelfAttentionJit.forward, mmf/modules/hf_layers.py:138-213, in the reference):
 * dQ, dK, dV from dctx with the probabilities recomputed from the saved row log-sum-exp and the dropout decisions replayed from the key. */
typedef struct mmf_attn_bwd_desc {
    mmf_attn_desc f;   /* forward operands (ctx = forward output O, lse = saved) */
    const void* dctx;  /* bf16, same layout as ctx (ldo) */
    void* dq;
    void* dk;
    void* dv;          /* bf16, layouts ldq / ldk / ldv like q / k / v */
    float* delta;      /* workspace [B, heads, Sq] fp32 */
} mmf_attn_bwd_desc;
int mmf_attention_bwd(const mmf_attn_bwd_desc* d, void* stream);

/* ---- LayerNorm ------------------------------------------------------------------------------
 * y = (x - mean) * rstd * gamma + beta over the last dim H (eps inside the sqrt), fp32 statistics.
 * Replaces nn.LayerNorm(H, eps=1e-12) in HF BertSelfOutput / BertOutput / BertEmbeddings
 * (call sites hf_layers.py:248,290; embeddings.py:456) and BertPredictionHeadTransform
 * (visual_bert.py:328).  x, y bf16 [rows, H]; mean, rstd fp32 [rows] (saved for backward).
 * H % 4 == 0, H <= 2048.
 */
int mmf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int rows, int H, float eps, void* stream);
/* Backward.  dy, x bf16 [rows,H] -> dx bf16 (grad w.r.t. the pre-LN sum; this is also the gradient
 * of the residual branch).  If dlin != NULL it receives dropout_backward(dx) with the forward's
 * dropout config, element index row*H + col (the gradient of the Linear output that fed the
 * residual add); with thr16 == 0 pass dlin = NULL and use dx.  dgamma/dbeta/dbias (fp32 [H], any
 * may be NULL) get the column sums of dy*xhat, dy and dlin; `accumulate` != 0 adds to them.
 * partials: fp32 workspace of mmf_layernorm_bwd_ws_floats(H) floats.
 * Deferred column sums: where mmf_layernorm_bwd_deferrable(rows, H) is 1, a call with dgamma = dbeta = dbias = NULL leaves the
 * per-workgroup partial sums of dgamma / dbeta in `partials`; mmf_layernorm_bwd_reduce_multi finishes up to MMF_MT_MAX such
 * calls in ONE launch (a training step has 26 LayerNorm backwards whose parameter gradients nobody reads before the optimizer).
 */
int mmf_layernorm_bwd_ws_floats(int H);
int mmf_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      void* dx, void* dlin, uint32_t drop_key, uint32_t drop_thr16, float drop_scale,
                      const uint32_t* drop_seed, float* dgamma, float* dbeta, float* dbias, int accumulate,
                      float* partials, int rows, int H, void* stream);
/* LayerNorm followed by nn.Dropout as ONE launch each way (BertVisioLinguisticEmbeddings, mmf/modules/embeddings.py:343-345: `self.dropout(self.LayerNorm(
 * embeddings))`): forward y = dropout(LN(x)) — the LayerNorm output rounded to bf16, then the keep scale, element index row*H + col, bit-identical to
 * mmf_layernorm_fwd + mmf_dropout_bf16; backward mmf_layernorm_bwd_din applies the same mask to the incoming gradient while loading it (= mmf_dropout_bf16 +
 * mmf_layernorm_bwd without dlin / dbias).  Widths with mmf_layernorm_dropout_fusable(H) != 0 only (H % 256 == 0, H <= 1024); deferral of the column sums
 * as for mmf_layernorm_bwd. */
int mmf_layernorm_dropout_fusable(int H);
int mmf_layernorm_dropout_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int H, float eps,
                              uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, void* stream);
int mmf_layernorm_bwd_din(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx, uint32_t in_key, uint32_t in_thr16,
                          float in_scale, const uint32_t* in_seed, float* dgamma, float* dbeta, int accumulate, float* partials, int rows, int H, void* stream);

/* ---- embeddings (BertVisioLinguisticEmbeddings, mmf/modules/embeddings.py:329-345) ------------
 * Text rows of the joint pre-LayerNorm sequence: y[b*S + t] = word[ids[b,t]] + pos[t] + type[seg[b,t]].
 * Tables are fp32 [*, H]; y is bf16 [B*S, H]; ids/seg int64 [B, T].  General form (MMBT, mmf/models/mmbt.py:84-129,
 * 245-250): y[b*S + row0 + t] = word[ids[b,t]] + pos[pos0 + t] + type[seg[b,t]].
 */
/* Table sizes V (word rows), P (position rows), NT (type rows) bound the indices like nn.Embedding does: pos0 + T > P is an
 * argument error; an id outside [0, V) or a segment id outside [0, NT) cannot be seen by the host without a sync, so the kernel
 * writes a zero row for it and raises the device-side index-error flag (mmf_amd_take_index_error).  0 = unchecked. */
int mmf_embed_text_fwd(const int64_t* ids, const int64_t* seg, const float* word, const float* pos,
                       const float* type, void* y, int B, int T, int S, int H, int row0, int pos0, int V, int P, int NT,
                       void* stream);
/* `image_text_alignment` of BertVisioLinguisticEmbeddings.get_position_embeddings_visual (mmf/modules/embeddings.py:373-397): out[r] (fp32
 * [rows, H]) = mean over the valid a of pos[align[r][a]] (align int64 [rows, A], -1 = padding; no valid entry: zero) + typ[typ_idx[r]]
 * (optional: the region's visual token-type row) — the per-region addend the visual projection GEMM gathers in its epilogue.  Backward:
 * dpos[align[r][a]] += dvis[r] / count[r] (fp32 atomics; dvis bf16, row (b, i) at (b * bstride + i) * ld).  An index outside [0, P) other
 * than -1 is skipped and raises the index-error flag. */
int mmf_align_pos_fwd(const int64_t* align, const float* pos, const float* typ, const int64_t* typ_idx, float* out, int rows, int A, int H, int P, int NT,
                      void* stream);
int mmf_align_pos_bwd(const void* dvis, int ld, int nb, int rpb, int bstride, const int64_t* align, float* dpos, int A, int H, int P, void* stream);
/* 1 if any index-consuming kernel (embedding gathers, scatter-adds) met an out-of-table index since the last call, else 0;
 * clears the flag.  Synchronises with the device: call it between steps, not inside a captured region. */
int mmf_amd_take_index_error(void);
/* MMF Transformer per-modality embedding sum (mmf/models/transformers/backends/huggingface.py:145-155):
 * y[b*S + row0 + i] = x[b*L + i] + pos[pos0 + i] + type[seg[b,i]]; x, y bf16 rows of H, tables fp32; pos and
 * (seg, type) may be NULL (the reference adds them only when the modality has position / segment ids). */
int mmf_rows_add_embed(const void* x, const int64_t* seg, const float* pos, const float* type, void* y, int B, int L,
                       int S, int H, int row0, int pos0, void* stream);
/* UNITER (mmf/models/uniter.py:74-78): y[r, :] = bf16(x[r, :] + table[idx[r], :]) — region features (fp32 [rows, D]) plus the
 * mask embedding row selected by the image mask; idx / table NULL: plain conversion.  D % 4 == 0. */
int mmf_rows_add_table_f32(const float* x, const int64_t* idx, const float* table, void* y, int rows, int D, void* stream);
/* The `torch.cat(list_embeddings, dim=1)` of huggingface.py:159 and its backward split, one modality block per
 * call: dst[(b*dst_bstride + i), :] = src[(b*src_bstride + i), :], b < nb, i < rpb; strides in rows; H % 8 == 0. */
int mmf_copy_rows_bf16(const void* src, int src_bstride, void* dst, int dst_bstride, int nb, int rpb, int H,
                       void* stream);
/* out[idx[r]] += x[row r] for r in [0, nb*rpb): row r = (b, i) lives at x + (b*bstride + i)*ld.
 * idx == NULL means bucket (i + idx_base) (position ids) when per_pos != 0, else bucket idx_base.
 * `out` [nbuckets, H] fp32 is added to (caller zero-fills when not accumulating).  With an index array, up to 16384 source rows and H a multiple of 16 the sums are formed WITHOUT
 * atomics, in source-row order (one owner workgroup per distinct bucket): deterministic; beyond that, and for idx == NULL with per_pos == 0, fp32 atomics.
 * few_buckets != 0: the table has `nbuckets` rows and (almost) every index is 0 or 1 (token-type tables,
 * position_ids_visual == 0): deterministic two-stage column sums through `ws`
 * (mmf_rows_scatter_add_ws_floats(H) floats) instead of atomics.
 * Indices outside [0, nbuckets) (nbuckets > 0) are skipped and raise the index-error flag instead of writing out of bounds.
 * skip_bucket >= 0: rows that map to that bucket are dropped — nn.Embedding(padding_idx=pad_token_id) keeps the
 * [PAD] row's gradient at zero (HF BertEmbeddings word_embeddings; reached from embeddings.py:309).
 */
int mmf_rows_scatter_add_ws_floats(int H);
/* The four small table gradients of BertVisioLinguisticEmbeddings' backward (embeddings.py:329-345, 411-419) in two launches instead of seven
 * mmf_rows_scatter_add ones: x = gradient of the pre-LayerNorm sum, bf16 [B, S = T + R, ld]; dpos[pos0 + t] += sum_b x[b][t] (text positions),
 * dtyp[seg[b][t]] += x[b][t] (text token types), dtyp_vis[vt[b][r]] += x[b][T + r], dpos_vis[0] += sum over every visual row (position_ids_visual == 0).
 * Any output may be NULL; R may be 0.  Buckets 0 / 1 are summed deterministically (fixed order), larger in-range ones by atomics, ids outside
 * [0, NT) / [0, NTV) are skipped and raise the index-error flag.  ws: mmf_embed_tables_bwd_ws_floats(T + R, H) floats. */
int mmf_embed_tables_bwd_ws_floats(int S, int H);
int mmf_embed_tables_bwd(const void* x, int ld, int B, int T, int R, const int64_t* seg, const int64_t* vt, int pos0, float* dpos, int P, float* dtyp, int NT,
                         float* dtyp_vis, int NTV, float* dpos_vis, int H, float* ws, void* stream);
int mmf_rows_scatter_add(const void* x, int ld, int nb, int rpb, int bstride, const int64_t* idx, int idx_ld,
                         int per_pos, int idx_base, float* out, int H, int few_buckets, int nbuckets, float* ws,
                         int skip_bucket, void* stream);

/* ---- small row utilities ----------------------------------------------------------------------
 * gather: out[b] = dropout(x[b*S + index[b]]), bf16 rows of H (visual_bert.py:389-400, the
 * `pooler_strategy: vqa` token pick + dropout).  scatter: dx[b*S + index[b]] = dropout_bwd(dout[b]).
 */
int mmf_gather_rows(const void* x, const int64_t* index, void* out, int B, int S, int H,
                    uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed,
                    void* stream);
int mmf_scatter_rows(const void* dout, const int64_t* index, void* dx, int B, int S, int H,
                     uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed,
                     void* stream);
/* The same backward as ONE pass that writes the whole [B, S, H] gradient (the selected row, zeros elsewhere): no zero fill before it. */
int mmf_scatter_rows_full(const void* dout, const int64_t* index, void* dx, int B, int S, int H, uint32_t drop_key, uint32_t drop_thr16,
                          float drop_scale, const uint32_t* drop_seed, void* stream);
/* out[n] = beta*out[n] + sum over rows of x (bf16); rows = nb groups of rpb rows, group stride
 * bstride rows, row stride ld.  Bias gradients. */
int mmf_colsum_bf16(const void* x, int ld, int nb, int rpb, int bstride, int N, float* out, float beta,
                    float* partials, void* stream);
int mmf_colsum_ws_floats(int N);
int mmf_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* 2-D casts with leading dimensions (elements); f32->bf16 zero-fills the pad columns [cols, ldd). */
int mmf_cast2d_f32_to_bf16(const float* src, int lds, void* dst, int ldd, int rows, int cols, void* stream);
int mmf_cast2d_bf16_to_f32(const void* src, int lds, float* dst, int ldd, int rows, int cols, void* stream);
/* ViLBERT `dynamic_attention` (mmf/models/vilbert.py:174-176, 199-212): the visual self-attention's queries and keys are
 * multiplied, per sample, by 1 + sigmoid(Linear(masked mean of the text stream)).
 *   mmf_masked_mean_fwd : pool[b][c] = sum_t x[b][t][c] mask[b][t] / sum_t mask[b][t]    (:204-205; x bf16 [B,T,H], mask fp32 [B,T])
 *   mmf_masked_mean_bwd : dx[b][t][c] = dpool[b][c] mask[b][t] / sum_t mask[b][t]         (autograd of the above, bf16 out)
 *   mmf_rowgroup_scale  : x[g * rows_per_group + r][c] *= gate[g][c] for c < C, in place (:211-212; the Q|K columns of the
 *                         packed [rows, ld] bf16 projection, gate fp32 [groups, C])
 *   mmf_rowgroup_scale_bwd : given y = x * gate and dy, writes dx = dy * gate over dy and dgate[g][c] = sum_r dy * x.
 *   mmf_gate_sigmoid_fwd / _bwd : gate[b][col0 + c] = 1 + sigmoid(z[b][c]) written into the packed [B, ldg] gate (the Q half at col0 = 0,
 *                         the K half at col0 = C: no torch.cat), :206-209; backward dz = dgate * s (1 - s), s = gate - 1. */
int mmf_gate_sigmoid_fwd(const float* z, float* gate, int ldg, int col0, int B, int C, void* stream);
int mmf_gate_sigmoid_bwd(const float* dgate, const float* gate, int ldg, int col0, float* dz, int B, int C, void* stream);
int mmf_masked_mean_fwd(const void* x, const float* mask, float* pool, int B, int T, int H, void* stream);
int mmf_masked_mean_bwd(const float* dpool, const float* mask, void* dx, int B, int T, int H, void* stream);
int mmf_rowgroup_scale(void* x, int ld, const float* gate, int groups, int rows_per_group, int C, void* stream);
int mmf_rowgroup_scale_bwd(void* dy, const void* y, int ld, const float* gate, float* dgate, int groups, int rows_per_group, int C,
                           void* stream);
/* y[i] = x[i] * keep_scale(i): nn.Dropout forward AND backward (embeddings.py:458), bf16, n < 2^32. */
int mmf_dropout_bf16(const void* x, void* y, int64_t n, uint32_t drop_key, uint32_t drop_thr16, float drop_scale,
                     const uint32_t* drop_seed, void* stream);
/* Dropout everywhere: keep(i) = hash(key', i) >= thr16 with key' = drop_key + drop_seed[0] * 0x9E3779B1 when a
 * drop_seed pointer is given (else key' = drop_key).  mmf_seed_advance increments that device word; put it at the
 * head of a captured step so every hipGraph replay draws fresh masks while forward and backward still agree. */
int mmf_seed_advance(uint32_t* seed, void* stream);
/* mmf_seed_advance and mmf_optim_state_advance (below) as ONE one-thread launch at the head of a captured training step; either pointer may be NULL. */
int mmf_step_advance(uint32_t* seed, float* state, int schedule, float warmup_steps, float total_steps, void* stream);
/* du = dh * g with g = gelu_erf'(u) as saved by the forward epilogue (act == 1): backward of HF
 * BertIntermediate's activation when it is not fused into a dgrad GEMM epilogue (act == 2). */
int mmf_gelu_bwd_bf16(const void* dh, const void* g, void* du, int64_t n, void* stream);
/* Pointwise bf16 ops of ViLBERT's poolers and stream fusion (mmf/models/vilbert.py:799-826, 1315-1320):
 * op 0: out = a * b; op 1: out = relu(a); op 2: out = a * (b > 0) (ReLU backward, b = forward output); op 3: out = a + b
 * (UNITER's image + position embedding sum, mmf/models/uniter.py:82). */
int mmf_eltwise_bf16(int op, const void* a, const void* b, void* out, int64_t n, void* stream);
/* dx = dy * (1 - y^2): backward of the tanh in HF BertPooler (mmf/models/mmbt.py:311), y = saved output. */
int mmf_tanh_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, void* stream);
int mmf_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* mask_add[b][s] = (1 - mask[b][s]) * -10000  (visual_bert.py:94-106); mask int64 [B,S]. */
int mmf_make_additive_mask(const int64_t* mask, float* out, int64_t n, void* stream);
/* VisualBERT.forward's input massaging in one launch (visual_bert.py:444-467, 525-556; the `vqa` pooling index of :389-392): from the
 * text mask int64 [B, T] and the per-sample region counts image_dim int64 [B] (NULL: all R regions valid) it writes image_mask [B, R]
 * (r < image_dim), attention_mask [B, T + R] (their concatenation), visual_embeddings_type [B, R] (zeros), the additive mask
 * (1 - attention_mask) * -10000 as fp32 [B, T + R] and pool_index[b] = sum_t input_mask[b][t] - 2. */
int mmf_visual_masks(const int64_t* input_mask, const int64_t* image_dim, int B, int T, int R, int64_t* image_mask, int64_t* attention_mask,
                     int64_t* visual_embeddings_type, float* mask_add, int64_t* pool_index, void* stream);

/* ---- loss: LogitBinaryCrossEntropy, mmf/modules/losses.py:225-251 ---------------------------
 * loss = mean(BCEWithLogits(scores, targets)) * N.  scores/targets fp32 [B, N] (row stride N).
 * fwd writes the scalar to loss[0] (ws: fp32 workspace of mmf_bce_logits_ws_floats() floats; the
 * two-stage reduction is deterministic).  bwd writes dscores (bf16, row stride ldd >= N, pad columns
 * zeroed) = gscale * (sigmoid(x) - t) / B, gscale read from the device scalar gloss (or 1 if NULL).
 */
int mmf_bce_logits_ws_floats(void);
int mmf_bce_logits_fwd(const float* scores, const float* targets, float* loss, float* ws, int B, int N, void* stream);
int mmf_bce_logits_bwd(const float* scores, const float* targets, const float* gloss, void* dscores, int ldd,
                       int B, int N, void* stream);

/* ---- loss: nn.CrossEntropyLoss(ignore_index) as used by MMF's `cross_entropy` loss (mmf/modules/losses.py) ---------
 * logits fp32 [B, C] (C <= a few dozen classes), labels int64 [B]; mean over the non-ignored rows.  fwd writes
 * loss[0] and count[0] (number of contributing rows, reused by bwd).  bwd: dlogits fp32 [B, C]. */
int mmf_cross_entropy_fwd(const float* logits, const int64_t* labels, float* loss, float* count, int B, int C,
                          int ignore_index, void* stream);
int mmf_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* count, const float* gloss,
                          float* dlogits, int B, int C, int ignore_index, void* stream);

/* Vocabulary-sized softmax cross-entropy — the masked-LM loss of VisualBERTForPretraining (mmf/models/visual_bert.py:215,270-277:
 * nn.CrossEntropyLoss(ignore_index=-1) over the [B * (T + R), vocab] prediction scores).  logits fp32 [R, C] with row stride ld;
 * forward writes lse[r] (log-sum-exp of row r; 0 for ignored rows), rowloss[r], loss[0] = mean over the counted rows (NaN when
 * none counts, like torch; the reference's own test asserts that, tests/models/test_visual_bert.py:71-98) and count[0];
 * backward writes d = gloss / count * (softmax - onehot) as bf16 [R, ldd] (ldd % 8 == 0, pad columns and ignored rows zero):
 * the ready-made operand of the decoder's input- and weight-gradient GEMMs.  A label outside [0, C) other than ignore_index
 * is skipped and raises the index-error flag (mmf_amd_take_index_error), where torch raises. */
int mmf_vocab_cross_entropy_fwd(const float* logits, int ld, const int64_t* labels, float* lse, float* rowloss, float* loss, float* count,
                                int R, int C, int ignore_index, void* stream);
int mmf_vocab_cross_entropy_bwd(const float* logits, int ld, const int64_t* labels, const float* lse, const float* count, const float* gloss,
                                void* dlogits, int ldd, int R, int C, int ignore_index, void* stream);

/* Masked soft-target KL divergence — ViLBERT's masked-region classification loss (mmf/models/vilbert.py:1070-1071,1150-1157,
 * `visual_target: 0`): sum over the rows with row_label == 1 of KLDivLoss(log_softmax(logits[r]), target[r]) (0 where the target
 * is 0, like torch), divided by the number of such rows.  logits fp32 [R, C] (row stride ld), target fp32 [R, C] (ldt), row_label
 * int64 [R] (image_label).  Forward writes lse / tsum / rowloss per row, loss[0], count[0]; backward writes gloss / count *
 * (softmax * sum_c target - target) as bf16 [R, ldd] (ldd % 8 == 0, pad columns and unlabelled rows zero), the operand of the
 * image-prediction decoder's gradient GEMMs. */
int mmf_soft_target_kl_fwd(const float* logits, int ld, const float* target, int ldt, const int64_t* row_label, float* lse, float* tsum,
                           float* rowloss, float* loss, float* count, int R, int C, void* stream);
int mmf_soft_target_kl_bwd(const float* logits, int ld, const float* target, int ldt, const int64_t* row_label, const float* lse,
                           const float* tsum, const float* count, const float* gloss, void* dlogits, int ldd, int R, int C, void* stream);

/* ---- M4C (mmf/models/m4c.py) -------------------------------------------------------------------
 * F.normalize(x, dim=-1) of the appearance / FastText / PHOC features (m4c.py:195,212,217,223): y[r, :D] = x[r, :D] /
 * max(||x[r, :D]||_2, eps), written as bf16 at row stride ldy — `y` may point at a column offset inside the wider
 * concatenated OCR feature row of m4c.py:235-237, any element alignment.  x is fp32 (x_f32 != 0) or bf16, row stride ldx.
 * inv_norm[r] = 1 / max(||x||, eps) is saved for the backward: dx = (g - y <g, y>) * inv_norm. */
int mmf_l2norm_rows_fwd(const void* x, int x_f32, int ldx, void* y, int ldy, float* inv_norm, int rows, int D, float eps, void* stream);
int mmf_l2norm_rows_bwd(const void* g, int ldg, const void* y, int ldy, const float* inv_norm, void* dx, int lddx, int rows, int D,
                        void* stream);
/* `_batch_gather(cat([ans_emb, ocr_emb]), prev_inds)` of PrevPredEmbeddings.forward (m4c.py:526-528, 566-578) without
 * building the concatenation: out[r] = idx[r] < rows_a ? a[idx[r]] : b[idx[r] - rows_a]; bf16 rows of H (H % 8 == 0); the
 * caller folds the batch offset of the OCR rows into idx.  Backward is mmf_rows_scatter_add into an fp32 [rows_a + rows_b, H]. */
int mmf_gather_rows2(const void* a, int64_t rows_a, const void* b, int64_t rows_b, const int64_t* idx, void* out, int n, int H,
                     void* stream);
/* OcrPtrNet.forward (m4c.py:474-493): out[b, t, n] = scale * <q[b, t, :], k[b, n, :]> + mask_add[b, n]; q bf16 [B*T, HQ],
 * k bf16 [B*N, HQ], mask_add fp32 [B, N] (0 / -10000) or NULL, out fp32 with row stride ldo — pointed at column
 * `num_choices` of the [B*T, num_choices + N] score matrix it realises the torch.cat of m4c.py:282.  Backward: dscores
 * fp32 (row stride ldd, same column offset) -> dq, dk bf16. */
int mmf_ptr_scores_fwd(const void* q, const void* k, const float* mask_add, float* out, int ldo, int B, int T, int N, int HQ,
                       float scale, void* stream);
int mmf_ptr_scores_bwd(const float* dscores, int ldd, const void* q, const void* k, void* dq, void* dk, int B, int T, int N, int HQ,
                       float scale, void* stream);
/* M4CDecodingBCEWithMaskLoss.forward (mmf/modules/losses.py:581-592): loss = sum_r w[r] sum_n BCEWithLogits(x[r,n], t[r,n])
 * / max(sum_r w[r], 1); scores / targets fp32 [rows, N], row_weight = train_loss_mask flattened to [rows].  fwd writes
 * loss[0] and count[0] (the denominator, reused by bwd); deterministic two-stage reduction through ws
 * (mmf_bce_rowmask_ws_floats() floats).  bwd: dscores fp32 [rows, N] = gloss[0] * w[r] * (sigmoid(x) - t) / count. */
int mmf_bce_rowmask_ws_floats(void);
int mmf_bce_rowmask_fwd(const float* scores, const float* targets, const float* row_weight, float* loss, float* count, float* ws,
                        int rows, int N, void* stream);
int mmf_bce_rowmask_bwd(const float* scores, const float* targets, const float* row_weight, const float* count, const float* gloss,
                        float* dscores, int rows, int N, void* stream);

/* ---- UNITER pretraining heads (mmf/models/uniter.py:36-39: tasks mlm, itm, mrc, mrfr, wra) -------------------------------------
 * MRFR, mmf/models/transformers/heads/mrfr.py:85-90: F.mse_loss(prediction_feat, feat_targets, reduction="mean") on the fp32 output
 * of the tied projection GEMM.  fwd writes loss[0] (deterministic two-stage sum through ws, mmf_mse_ws_floats() floats); bwd writes
 * gloss * 2 (pred - target) / (rows * cols) as bf16 [rows, ldd] (ldd % 8 == 0, pad columns zero): the operand of the projection's
 * input- and weight-gradient GEMMs.
 * Masked form (row_label != NULL, int64 [rows]) — ViLBERT's masked region REGRESSION, `visual_target: 1`, mmf/models/vilbert.py:1139-1148:
 * nn.MSELoss(reduction="none") summed over the rows with label == 1 and divided by max(number of their elements, 1); `count` receives that
 * denominator (it must be given) and the backward reads it: rows without the label get a zero gradient. */
int mmf_mse_ws_floats(void);
int mmf_mse_fwd(const float* pred, int ldp, const float* target, int ldt, const int64_t* row_label, float* loss, float* count, float* ws, int rows, int cols,
                void* stream);
int mmf_mse_bwd(const float* pred, int ldp, const float* target, int ldt, const int64_t* row_label, const float* count, const float* gloss, void* dpred, int ldd,
                int rows, int cols, void* stream);
/* WRA, mmf/models/transformers/heads/wra.py:36-83 over mmf/modules/ot.py: per sample b the optimal-transport distance between the text
 * rows [0, M) and the region rows [M, M + N) of the joint sequence seq [B, S, H] (bf16, row stride ld, S >= M + N) under the cosine cost
 * (ot.py:15-25, F.normalize eps), the transport plan by `iterations` IPOT steps (ot.py:38-84, beta, k = 1; a constant of the
 * backward pass) and loss = (sum of dist over label == 1 - sum over label == 0) / (number of such samples).  fp32 arithmetic, one
 * workgroup per sample, M, N <= 128.  Saved for the backward: xinv [B, M], yinv [B, N] (1 / max(|row|, eps)), plan [B, N, M], cost
 * [B, M, N]; dist [B] is also an output.  bwd writes the gradient of the M + N rows of every sample into dseq (bf16, row stride ldd);
 * rows beyond M + N are not touched. */
typedef struct mmf_wra_desc {
    const void* seq; int ld;
    int B, S, H, M, N;
    const float* txt_pad;    /* [B, M] 1.0 = padding */
    const float* img_pad;    /* [B, N] */
    const int64_t* label;    /* [B] is_correct */
    float* xinv; float* yinv; float* plan; float* cost; float* dist;
    float beta;              /* 0 = 0.5 */
    float eps;               /* 0 = 1e-5 */
    int iterations;          /* 0 = 50 */
} mmf_wra_desc;
int mmf_wra_fwd(const mmf_wra_desc* d, float* loss, float* count, void* stream);
int mmf_wra_bwd(const mmf_wra_desc* d, const float* gloss, const float* count, void* dseq, int ldd, void* stream);

/* ---- optimizer: AdamW (mmf/modules/optimizers.py:8-17; transformers.AdamW semantics) ----------
 * One fused pass over a flat fp32 parameter arena: p, g, m, v [n].  Weight decay is
 * given per SEGMENT: seg_end[i] (exclusive prefix ends, int64 [nseg]) and seg_wd[i] (weight decay of
 * segment i; segment starts must be multiples of 4 elements).  Also refreshes the bf16 shadow `p16`
 * (may be NULL).  mode 0 = transformers.AdamW update rule, mode 1 = torch.optim.AdamW.  g is
 * multiplied by grad_scale first (loss-scale / world-size folding).
 */
int mmf_adamw_step(float* p, const float* g, float* m, float* v, void* p16, int64_t n, const int64_t* seg_end,
                   const float* seg_wd, int nseg, float lr, float beta1, float beta2, float eps, int step,
                   int correct_bias, int mode, float grad_scale, void* stream);

/* Multi-tensor form: up to MMF_MT_MAX separately allocated parameter tensors per launch (fp32 p/g/m/v, optional bf16
 * shadow p16 refreshed in the same pass), per-tensor lr and weight decay (the two BERT groups of
 * mmf/utils/modeling.py:18-46 and the finetune LR multiplier).  If norm_sq != NULL the gradients are scaled by
 * min(1, max_norm / (sqrt(norm_sq[0]) + 1e-6)) on the fly: clip_gradients (mmf/utils/general.py:33-50) folded
 * into the update.  mmf_l2norm_sq_multi computes (or accumulates) sum(g^2) over a tensor list deterministically;
 * ws: mmf_l2norm_sq_ws_floats(list) floats. */
#define MMF_MT_MAX 52
typedef struct mmf_adamw_multi_desc {
    int n;
    void* p[MMF_MT_MAX];
    const void* g[MMF_MT_MAX];
    void* m[MMF_MT_MAX];
    void* v[MMF_MT_MAX];
    void* p16[MMF_MT_MAX];    /* optional bf16 mirror (GEMM weight shadow), refreshed in the same pass */
    void* p32[MMF_MT_MAX];    /* optional fp32 mirror (slice of a packed Q|K|V bias), refreshed in the same pass */
    int64_t numel[MMF_MT_MAX];
    float lr[MMF_MT_MAX];
    float wd[MMF_MT_MAX];
    float beta1, beta2, eps;
    int step, correct_bias, mode;
    float grad_scale;
    const float* norm_sq;
    float max_norm;
    const float* dev_state;   /* optional device words {step, lr multiplier}: the update reads the step count (bias
                                 correction) and the schedule factor from HBM, so the launch can be replayed from a
                                 hipGraph; `step` above is then ignored.  Advanced by mmf_optim_state_advance. */
    uint64_t g_bf16_mask;     /* bit i set: g[i] is a bf16 buffer (the data-parallel step's bf16 wire buffer after the all-reduce,
                                 mmf/trainers/core/device.py:104-110 there an fp32 DDP bucket): read directly, no fp32 unpack pass */
} mmf_adamw_multi_desc;
int mmf_adamw_multi(const mmf_adamw_multi_desc* d, void* stream);
/* state[0] += 1 (optimizer step count t); state[1] = LR multiplier for step t: schedule 0 -> 1, schedule 1 ->
 * MMF's `warmup_linear` (mmf/modules/schedulers.py:34-37 = transformers.get_linear_schedule_with_warmup) evaluated
 * at t - 1 (the scheduler is stepped after the optimizer): (t-1)/warmup while t-1 < warmup, else
 * max(0, (total - (t-1)) / max(1, total - warmup)). */
int mmf_optim_state_advance(float* state, int schedule, float warmup_steps, float total_steps, void* stream);
typedef struct mmf_ln_reduce_list {
    int n;
    const float* partials[MMF_MT_MAX];      /* the workspace a deferred mmf_layernorm_bwd filled */
    int rows[MMF_MT_MAX];
    int H[MMF_MT_MAX];
    float* dgamma[MMF_MT_MAX];              /* fp32 [H], overwritten */
    float* dbeta[MMF_MT_MAX];
} mmf_ln_reduce_list;
int mmf_layernorm_bwd_deferrable(int rows, int H);
int mmf_layernorm_bwd_reduce_multi(const mmf_ln_reduce_list* d, void* stream);
typedef struct mmf_tensor_list {
    int n;
    const void* ptr[MMF_MT_MAX];
    int64_t numel[MMF_MT_MAX];
} mmf_tensor_list;
int mmf_l2norm_sq_ws_floats(const mmf_tensor_list* d);
int mmf_l2norm_sq_multi(const mmf_tensor_list* d, float* out, int accumulate, float* ws, void* stream);
/* The gradient bucket of the data-parallel reducer (DistributedDataParallel's bucket copy, mmf/trainers/core/device.py:104-110) packed in ONE
 * launch: dst[off[t] + i] = scale * src_t[i] for up to MMF_MT_MAX fp32 tensors, dst fp32 or (dst_bf16) the bf16 wire type; scale = the mean's
 * 1 / world applied before the rounding. */
typedef struct mmf_offset_list {
    int64_t off[MMF_MT_MAX];                /* element offset of tensor t inside dst */
} mmf_offset_list;
int mmf_pack_f32_multi(const mmf_tensor_list* d, const mmf_offset_list* off, void* dst, int dst_bf16, float scale, void* stream);

/* ---- transposed weight shadows -------------------------------------------------------------------------------------
 * dst[i] (bf16 [cols, rows]) = transpose of src[i] (bf16 [rows, cols], row-major), for up to MMF_MT_MAX matrices per
 * launch; rows and cols multiples of 64.  Keeps W^T twins of the bf16 weight shadows current so that the input-gradient
 * GEMM of nn.Linear (dX = dY W; autograd of mmf/modules/hf_layers.py:169-180,248,289-290) runs with two row operands
 * instead of a k-major W.  Not a reference operation: an internal layout choice. */
typedef struct mmf_transpose_list {
    int n;
    const void* src[MMF_MT_MAX];
    void* dst[MMF_MT_MAX];
    int rows[MMF_MT_MAX];
    int cols[MMF_MT_MAX];
} mmf_transpose_list;
int mmf_transpose_bf16_multi(const mmf_transpose_list* d, void* stream);

/* ---- fp32-accurate forward path (mmf_amd/csrc/fp32_path.hip) -----------------------------------
 * The reference computes in fp32 unless `training.fp16` switches autocast on (mmf/trainers/core/training_loop.py:199); these
 * entry points evaluate the same operations with fp32 activations on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32:
 * exact fp32 products, fp32 accumulation), so that outputs agree with the reference to fp32 round-off (the 1e-3 bound).
 * Forward (inference / evaluation) only.
 *
 * mmf_gemm_f32: mmf_gemm_bf16 with A, B, C, U, aux and resid fp32 (set a_f32 = b_f32 = out_f32 = 1) on v_mfma_f32_16x16x4_f32: the
 * forward (row, row), dgrad (row, k-major) and weight-gradient (k-major, k-major) layouts; the epilogue of mmf_gemm_bf16 (bias,
 * coladd, rowtab[rowidx], act 0-4 with exact-erf GELU and its saved derivative U, dropout, resid, beta, row remap) except
 * rowsum_out (mmf_colsum_f32); deterministic split-K through splitk_ws (no epilogue besides beta then).  lda, ldb multiples of 4;
 * a row operand's leading dimension must cover round_up(K, 4) with zeros in the padding.  Replaces nn.Linear forward at hf_layers.py:169-180, 248, 289-290, embeddings.py:352,
 * visual_bert.py:146, 328-330 and, for fp32 training (training_loop.py:199-211), its autograd dgrad / wgrad.
 * mmf_attention_f32_fwd: mmf_attention_fwd with q / k / v / ctx fp32 (hf_layers.py:161-213 in eval mode; vilbert.py:153-247, 388-475):
 * head_dim 64 with Sk <= 256 or head_dim 128 with Sk <= 128, Sq != Sk allowed, key mask and the prefix-LM causal_tail (m4c.py:424-440);
 * K and V of a (batch, head) are staged once per workgroup in LDS.  Optional lse output and probability dropout (training); ctx_f32 and
 * the K|V-cache strides must be unset.  Round 5: a materialised per-query mask (mask_query_stride / mask_batch_stride, as for mmf_attention_fwd)
 * is read per (query, key) by the forward and both backward kernels.
 * mmf_layernorm_f32_fwd: nn.LayerNorm over fp32 rows (hf_layers.py:248,290; embeddings.py:456; visual_bert.py:328).
 * mmf_embed_text_f32_fwd / mmf_gather_rows_f32: mmf_embed_text_fwd / mmf_gather_rows (no dropout) writing / moving fp32 rows. */
int mmf_gemm_f32(const mmf_gemm_desc* d, void* stream);
int mmf_gemm_f32_splits(int M, int N, int K);   /* K splits mmf_gemm_f32 uses for this shape when given splitk_ws (>= splits * M * N * 4 bytes) */
int mmf_attention_f32_fwd(const mmf_attn_desc* d, void* stream);
int mmf_layernorm_f32_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int H, float eps, void* stream);
int mmf_embed_text_f32_fwd(const int64_t* ids, const int64_t* seg, const float* word, const float* pos, const float* type, float* y,
                           int B, int T, int S, int H, int row0, int pos0, int V, int P, int NT, void* stream);
int mmf_gather_rows_f32(const float* x, const int64_t* index, float* out, int B, int S, int H, void* stream);
/* mmf_rows_add_embed with fp32 rows (MMF Transformer per-modality embedding sum, huggingface.py:145-155). */
int mmf_rows_add_embed_f32(const float* x, const int64_t* seg, const float* pos, const float* type, float* y, int B, int L, int S, int H,
                           int row0, int pos0, void* stream);

/* ---- fp32 training (round 3): the backward kernels of the fp32 path (mmf_amd/csrc/fp32_train.hip, attention in fp32_path.hip) ----
 * The reference trains in fp32 unless `training.fp16` is set (mmf/trainers/core/training_loop.py:199-211); `mmf_amd.fp32_training()`
 * runs the VisualBERT training step on these kernels plus mmf_gemm_f32's dgrad / weight-gradient layouts.
 * mmf_attention_f32_bwd: autograd of mmf_attention_f32_fwd (hf_layers.py:161-213).  All tensors fp32; f.ctx = the forward output O,
 *   f.lse = the row statistic the forward saved (row maximum + log2 of the row sum of the scaled, masked scores in log2 units),
 *   delta = workspace [B, heads, Sq]; dq / dk / dv in the layouts of q / k / v; same dropout key as the forward; two launches
 *   (dQ with K, V staged in LDS; dK, dV with Q, dO staged), no atomics.
 * mmf_layernorm_f32_fwd_stats: mmf_layernorm_f32_fwd that also returns mean and rstd [rows]; mmf_layernorm_f32_bwd: dx, dgamma, dbeta
 *   (partials: workspace of mmf_layernorm_f32_bwd_blocks(rows) * 2 * H floats).
 * mmf_colsum_f32: out[n] (+)= sum_rows x[row][n] (bias gradients), ws >= mmf_colsum_f32_slices(rows) * N floats.
 * mmf_dropout_f32: y = x * keepscale(hash(key, element index)), forward and backward of nn.Dropout on fp32 rows.
 * mmf_scatter_add_rows_f32: out[idx[r] + r * dst_stride] += g[src(r)] with src(r) = (r / grp) * grp_stride + grp_off + r % grp for
 *   grp > 0 (else r): autograd of the embedding gathers (rows with idx == skip, nn.Embedding's padding_idx, are dropped) and of the
 *   pooling gather (dst_stride = S).  fp32 atomics. */
int mmf_attention_f32_bwd(const mmf_attn_bwd_desc* d, void* stream);
int mmf_layernorm_f32_fwd_stats(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int rows, int H,
                                float eps, void* stream);
int mmf_layernorm_f32_bwd_blocks(int rows);
int mmf_layernorm_f32_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma,
                          float* dbeta, float* partials, int rows, int H, void* stream);
int mmf_colsum_f32_slices(int rows);
int mmf_colsum_f32(const float* x, int ld, int rows, int N, float* out, int accumulate, float* ws, void* stream);
int mmf_dropout_f32(const float* x, float* y, long n, uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, void* stream);
int mmf_scatter_add_rows_f32(const float* g, int ld, int rows, int H, int grp, int grp_stride, int grp_off, const int64_t* idx, long dst_stride,
                             long skip, int NT, float* out, int ldo, void* stream);
/* mmf_bce_logits_bwd / mmf_vocab_cross_entropy_bwd with fp32 gradients (fp32 training); ldd: a multiple of 4 covering C, the pad columns are zeroed. */
int mmf_vocab_cross_entropy_f32_bwd(const float* logits, int ld, const int64_t* labels, const float* lse, const float* count, const float* gloss,
                                    float* dlogits, int ldd, int R, int C, int ignore_index, void* stream);
int mmf_bce_logits_f32_bwd(const float* scores, const float* targets, const float* gloss, float* dscores, int B, int N, void* stream);

/* fp32 row operators of the widened models on the fp32 path: zero-padded copy of short rows (the 5-d / 7-d box geometry operands of
 * vilbert.py:906 / uniter.py:81 become 16-byte rows), element-wise a * b (op 0), relu (1), a + b (3), a (1 - b^2) (4: tanh backward), a [b > 0] (5: relu backward) (vilbert.py:803,818,1318;
 * uniter.py:82), and the dynamic_attention pooling / gating of vilbert.py:204-212 (mmf_masked_mean_fwd / mmf_rowgroup_scale on fp32). */
int mmf_pad_rows_f32(const float* src, int K, float* dst, int KP, int rows, void* stream);
int mmf_eltwise_f32(int op, const float* a, const float* b, float* y, long n, void* stream);
int mmf_masked_mean_f32(const float* x, const float* mask, float* pool, int B, int T, int H, void* stream);
int mmf_rowgroup_scale_f32(float* x, int ld, const float* gate, int groups, int rows_per_group, int C, void* stream);

/* fp32 backwards of the operators round 5 added to `mmf_amd.fp32_training()` (the reference trains all of them in fp32 unless `training.fp16` is
 * set, mmf/trainers/core/training_loop.py:199-211) — the bf16 kernels of the same names templated on fp32 rows:
 *   mmf_masked_mean_f32_bwd / mmf_rowgroup_scale_f32_bwd : ViLBERT dynamic_attention pooling and Q | K gating (vilbert.py:204-212), see
 *       mmf_masked_mean_bwd / mmf_rowgroup_scale_bwd;
 *   mmf_align_pos_f32_bwd : the image_text_alignment position term (mmf/modules/embeddings.py:373-397), see mmf_align_pos_bwd;
 *   mmf_soft_target_kl_f32_bwd : ViLBERT's masked-region KL loss (vilbert.py:1150-1157), gradient as fp32 [R, ldd] (ldd % 4 == 0);
 *   mmf_l2norm_rows_f32_bwd : autograd of F.normalize (m4c.py:195,212,217,223): dx = (g - y <g, y>) / max(||x||, eps), the factor recomputed from
 *       x into inv_ws [rows];
 *   mmf_ptr_scores_f32_bwd : autograd of OcrPtrNet's scores (m4c.py:474-493): dq = scale ds k, dk = scale ds^T q per sample. */
/* Touched-row exchange of an embedding-table gradient between data-parallel ranks (the reference all-reduces the dense table,
 * mmf/trainers/core/device.py:104-110): out[id] = sum of rows[perm[j]] over the segment of equal ids in `sorted_ids` (a STABLE sort of the ids gathered
 * from all ranks, `perm` its permutation), added in sorted order without atomics — bit-identical on every rank.  ids < 0 (duplicates removed by the
 * sender) and ids >= V are skipped; rows of `out` that no id names are left untouched. */
int mmf_segment_sum_rows_f32(const int64_t* sorted_ids, const int64_t* perm, const float* rows, float* out, int M, int H, int64_t V, void* stream);
int mmf_slice_rows_f32(const float* src, int ld_src, int K, float* dst, int KP, int rows, void* stream);   /* dst[r][:KP] = src[r * ld_src + :K], zero padded */
int mmf_masked_mean_f32_bwd(const float* dpool, const float* mask, float* dx, int B, int T, int H, void* stream);
int mmf_rowgroup_scale_f32_bwd(float* dy, const float* y, int ld, const float* gate, float* dgate, int groups, int rows_per_group, int C, void* stream);
int mmf_align_pos_f32_bwd(const float* dvis, int ld, int nb, int rpb, int bstride, const int64_t* align, float* dpos, int A, int H, int P, void* stream);
int mmf_soft_target_kl_f32_bwd(const float* logits, int ld, const float* target, int ldt, const int64_t* row_label, const float* lse,
                               const float* tsum, const float* count, const float* gloss, float* dlogits, int ldd, int R, int C, void* stream);
int mmf_mse_f32_bwd(const float* pred, int ldp, const float* target, int ldt, const int64_t* row_label, const float* count, const float* gloss, float* dpred, int ldd,
                    int rows, int cols, void* stream);      /* mmf_mse_bwd with an fp32 gradient (ViLBERT visual_target 1, vilbert.py:1139-1148); ldd % 4 == 0 */
int mmf_nce_f32_bwd(const float* target, const int64_t* neg, const int64_t* label, const float* scores, const float* lse, const float* count, const float* gloss,
                    float* dpred, int ldd, int M, int N, int K, void* stream);      /* mmf_nce_bwd with an fp32 gradient (visual_target 2, vilbert.py:1158-1227) */
int mmf_l2norm_rows_f32_bwd(const float* g, int ldg, const float* y, int ldy, const float* x, int ldx, float* inv_ws, float* dx, int lddx, int rows,
                            int D, float eps, void* stream);
int mmf_ptr_scores_f32_bwd(const float* dscores, int ldd, const float* q, const float* k, float* dq, float* dk, int B, int T, int N, int HQ,
                           float scale, void* stream);

/* M4C's row operators on fp32 rows (the fp32-accurate forward path): mmf_l2norm_rows_fwd, mmf_gather_rows2 and mmf_ptr_scores_fwd with fp32
 * operands (m4c.py:195,212-223; :526-528; :474-493). */
int mmf_l2norm_rows_f32(const float* x, int ldx, float* y, int ldy, int rows, int D, float eps, void* stream);
int mmf_gather_rows2_f32(const float* a, int64_t rows_a, const float* b, int64_t rows_b, const int64_t* idx, float* out, int n, int H, void* stream);
int mmf_ptr_scores_f32(const float* q, const float* k, const float* mask_add, float* out, int ldo, int B, int T, int N, int HQ, float scale,
                       void* stream);

/* ViLBERT `in_batch_pairs` / `fast_mode` batch expansion (mmf/models/vilbert.py:678-725) on bf16 activations [Bs, n] -> [reps * Bs, n]:
 * mode 0: out[i * Bs + j] = x[j] (`unsqueeze(0).expand`, the image side / fast_mode's text), mode 1: out[i * reps + j] = x[i] (`unsqueeze(1).expand`, the
 * text side); mmf_reduce_batch_bf16 is the backward (the sum over the broadcast index, fp32 accumulation).  n % 8 == 0. */
int mmf_expand_batch_bf16(const void* x, void* out, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream);
int mmf_reduce_batch_bf16(const void* g, void* dx, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream);
/* The same on fp32 activations (mmf_amd.fp32_training() / fp32_inference(): `in_batch_pairs` / `fast_mode` in the reference's default arithmetic); n % 4 == 0. */
int mmf_expand_batch_f32(const float* x, float* out, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream);
int mmf_reduce_batch_f32(const float* g, float* dx, int64_t Bs, int64_t reps, int64_t n, int mode, void* stream);

/* ViLBERT's masked-region NCE loss (`visual_target: 2`, mmf/models/vilbert.py:1158-1227): pred fp32 [M, N] = the image-prediction head's output
 * for all M = B * R regions, target fp32 [M, N] the region features, neg int64 [M, K] flat indices (into the M regions) of each region's K
 * negatives, label int64 [M] (1 = masked region).  score[r][j] = <sample_j, pred[r]> with sample_0 = target[r], sample_j = target[neg[r][j-1]];
 * loss = mean over the labelled regions of (logsumexp_j score - score_0) (CrossEntropyLoss against class 0); NaN when nothing is labelled.
 * fwd saves scores [M, K + 1] and lse [M]; bwd writes g / count * (sum_j softmax_j sample_j - sample_0) as the zero-padded bf16 operand
 * (row stride ldd, a multiple of 8) of the decoder's gradient GEMMs, zeros on unlabelled rows.  N % 4 == 0, K <= 1024. */
int mmf_nce_fwd(const float* pred, const float* target, const int64_t* neg, const int64_t* label, float* scores, float* lse, float* rowloss, float* loss,
                float* count, int M, int N, int K, void* stream);
int mmf_nce_bwd(const float* target, const int64_t* neg, const int64_t* label, const float* scores, const float* lse, const float* count, const float* gloss,
                void* dpred, int ldd, int M, int N, int K, void* stream);

/* ---- layout probes (tests only): dump what the hardware does so tests can pin the assumptions -- */
int mmf_probe_mfma16(const void* a, const void* b, float* d, void* stream);   /* 64 lanes x 8 bf16 each, out 64x4 */
int mmf_probe_mfma32(const void* a, const void* b, float* d, void* stream);   /* out 64x16 */
int mmf_probe_tr16(const void* lds_image_2048_bf16, const int* byte_addr_per_lane, void* out_64x4_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMF_AMD_H */
