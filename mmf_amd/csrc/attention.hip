// mmf_amd :: fused multi-head attention forward / backward for gfx950 (head_dim 64 with Sq, Sk <= 512 = BERT's
// max_position_embeddings, or head_dim 128 — ViLBERT's visual and co-attention streams, mmf/models/vilbert.py:153-247,347-475 —
// with Sq, Sk <= 256).  A whole head's K and V always sit in LDS (128 KB at the caps), so the softmax stays the reference's exact
// two-pass form at every length; up to 256 keys (128 at head_dim 128) the tuned forms below run — two workgroups per CU forward, the
// one-pass backward — beyond that the same kernels with more key tiles, one workgroup per CU, and the two-kernel backward.
//
// Replaces BertSelfAttentionJit.forward (mmf/modules/hf_layers.py:161-213):
//     scores = Q K^T / sqrt(d) + mask ; probs = softmax(scores) ; probs = dropout(probs) ;
//     ctx = probs V ; permute + view to [B, S, H]
// and its autograd backward, WITHOUT materialising the [B, A, S, S] scores / probs tensors
// (80 MB fp32 each per layer at the VisualBERT VQA2 shape).
//
// Work decomposition: a 256-thread workgroup owns one (batch, head) and 128 query rows; each of
// its 4 waves owns 32 query rows and ALL keys (Sk <= 256 -> <= 8 key tiles of 32), so the softmax
// is the exact two-pass formulation of the reference (no online rescaling) held entirely in
// registers.  A whole head's K and V (29 KB each at S = 228) are staged once in LDS, row-major; the
// operands that need the transposed view (V in P V, K in dS K, Q / dO in the dK / dV products) are read with the
// gfx950 hardware transpose read ds_read_b64_tr_b16, so no transposed copy is ever built.
//
// MFMA: v_mfma_f32_32x32x16_bf16, D[i][j] (+)= sum_k A[i][k] B[k][j]; lane l supplies row/col
// x = l & 31 and the 8 reduction slots (h = l >> 5, e = 0..7); it receives D[(r&3)+8(r>>2)+4h][x],
// r = 0..15.  The hardware pairs slot (h, e) of A with slot (h, e) of B, so any assignment of
// reduction indices to slots is valid as long as both operands use the same one; the kernels use
//   feature reductions  (over d)   : slot (h, e) of step s  <->  d = 16 s + 8 h + e
//   key / query reductions         : slot (h, e) of step u  <->  index 16 u + 4 h + (e&3) + 8 (e>>2)
// The second assignment is exactly the register order in which a lane holds a score tile, so the
// probabilities feed the next MFMA straight from registers (no LDS round trip, no permutes).
#include "common.h"
#include "mmf_amd.h"

namespace {

// Row-major LDS image of a [rows][D] bf16 matrix: 2D bytes per row, 16-byte chunks XOR-swizzled so that both access
// patterns are bank-conflict free: 16 lanes reading the same logical chunk of 16 different rows (frag_rm), and 32 lanes
// reading a 4-row x 4-chunk block (the transpose read in frag_tr).  With 128-byte rows (D = 64) two consecutive rows
// cover all 64 banks, so the swizzle advances every second row; with 256-byte rows (D = 128) every row starts on bank
// 0: rows differ in the 64-byte window ((row & 3) << 2) and, across groups of four rows, in the chunk within it.
template <int D> DEVI int swz(int row);
template <> DEVI int swz<64>(int row) { return (row >> 1) & 7; }
template <> DEVI int swz<128>(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <int D> DEVI int rm_off(int row, int chunk) { return row * (2 * D) + ((chunk ^ swz<D>(row)) << 4); }

// Stage rows [0, npad) of a token-major [rows][D] bf16 matrix (row stride ld) into the swizzled LDS image by LDS-DMA
// (global_load_lds_dwordx4): one wave-instruction fills 1 KiB lane-linear (8 rows at D = 64, 4 rows at D = 128), so
// the chunk swizzle is applied to each lane's SOURCE address.  Rows >= nvalid read a 16-byte zero buffer.  All DMAs
// of a tile are issued back to back; the caller waits once (s_waitcnt vmcnt(0) + barrier).  npad is a multiple of 32.
static __device__ uint4 g_zero16;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

template <int D, int NW = 4>
DEVI void stage_rows(const bf16* g, int ld, int nvalid, int npad, unsigned char* lds_rm, int tid) {
    constexpr int CPR = D / 8;          // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;       // rows per wave-instruction
    const int wave = tid >> 6, lane = tid & 63;
    for (int r0 = wave * RPI; r0 < npad; r0 += NW * RPI) {
        const int row = r0 + lane / CPR;
        const int logical = (lane % CPR) ^ swz<D>(row);
        const bf16* src = (row < nvalid) ? g + (size_t)row * ld + logical * 8 : reinterpret_cast<const bf16*>(&g_zero16);
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds_rm + r0 * (2 * D)), 16, 0, 0);
    }
}
DEVI void stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// feature-reduction operand from a row-major LDS image: tile row x (absolute row = row0 + x), step s
template <int D>
DEVI bf16x8 frag_rm(const unsigned char* lds_rm, int row0, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds_rm + rm_off<D>(row0 + (lane & 31), 2 * s + (lane >> 5)));
}
// feature-reduction operand straight from global memory (one row per lane, clamped by caller)
DEVI bf16x8 frag_global(const bf16* rowptr, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(rowptr + 16 * s + 8 * (lane >> 5));
}
// index-reduction operand by hardware transpose read (ds_read_b64_tr_b16) from a ROW-MAJOR image: slot (h, e) of
// lane x receives M[idx0 + 16u + 4h + (e&3) + 8(e>>2)][d0 + x].  A 16-lane group reads a 4-row x 16-column block
// (lane p addresses row p>>2, 8 bytes at column 4(p&3)) and lane i of the group gets column i of the 4 rows.
template <int D>
DEVI bf16x8 frag_tr(const unsigned char* lds_rm, int d0, int idx0, int u, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int row = idx0 + 16 * u + 4 * (g >> 1) + (p >> 2);
    const int c = d0 + 16 * (g & 1) + (p & 3) * 4;
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_rm + rm_off<D>(row, c >> 3) + (c & 7) * 2));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds_rm + rm_off<D>(row + 8, c >> 3) + (c & 7) * 2));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}
// index-reduction operand from a score tile held in registers (regs 8u .. 8u+7)
DEVI bf16x8 frag_regs(const f32x16& p, int u) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16)p[8 * u + e];
    return r;
}

// A wave's transposed result tile T^T[d][row] (lane x = row, registers -> d = 32 dt + (r&3) + 8(r>>2) + 4h) -> bf16 rows
// g[row * ld + 0..63] for row < nvalid, through a wave-private swizzled row-major LDS image (4 KB).
DEVI void store_tile_rows(const f32x16 (&acc)[2], float mul, unsigned char* img, bf16* g, int ld, int nvalid, int lane) {
    const int x = lane & 31, h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<bf16x4*>(img + rm_off<64>(x, 4 * dt + c) + 8 * h) =
                pack4(acc[dt][4 * c + 0] * mul, acc[dt][4 * c + 1] * mul, acc[dt][4 * c + 2] * mul, acc[dt][4 * c + 3] * mul);
    asm volatile("" ::: "memory");      // LDS operations of one wave complete in order; keep the compiler from reordering them
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), chunk = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(img + rm_off<64>(row, chunk));
        if (row < nvalid) *reinterpret_cast<uint4*>(g + (size_t)row * ld + 8 * chunk) = v;
    }
}

// head_dim 128: the tile is T^T[d][row] in four accumulators (d = 32 dt + ...), rows of 256 bytes through a wave-private 8 KB image
DEVI void store_tile_rows128(const f32x16 (&acc)[4], float mul, unsigned char* img, bf16* g, int ld, int nvalid, int lane) {
    const int x = lane & 31, h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<bf16x4*>(img + rm_off<128>(x, 4 * dt + c) + 8 * h) =
                pack4(acc[dt][4 * c + 0] * mul, acc[dt][4 * c + 1] * mul, acc[dt][4 * c + 2] * mul, acc[dt][4 * c + 3] * mul);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4), chunk = lane & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(img + rm_off<128>(row, chunk));
        if (row < nvalid) *reinterpret_cast<uint4*>(g + (size_t)row * ld + 8 * chunk) = v;
    }
}
template <int D> DEVI void store_tile_rows_d(const f32x16 (&acc)[D / 32], float mul, unsigned char* img, bf16* g, int ld, int nvalid, int lane);
template <> DEVI void store_tile_rows_d<64>(const f32x16 (&acc)[2], float mul, unsigned char* img, bf16* g, int ld, int nvalid, int lane) {
    store_tile_rows(acc, mul, img, g, ld, nvalid, lane);
}
template <> DEVI void store_tile_rows_d<128>(const f32x16 (&acc)[4], float mul, unsigned char* img, bf16* g, int ld, int nvalid, int lane) {
    store_tile_rows128(acc, mul, img, g, ld, nvalid, lane);
}

// The same for an fp32 copy of the rows (256-byte rows; 16-byte chunks XOR-swizzled by row & 15), 8 KB image.  The copy is read again only by the
// backward pass, a few milliseconds later: it is stored non-temporally (it would only push the next kernels' operands out of the Infinity Cache;
// -0.01 .. -0.08 ms per step on three boxes, profiles/r04_store_policy.txt section 9).
DEVI void store_tile_rows_f32(const f32x16 (&acc)[2], float mul, unsigned char* img, float* g, int ld, int nvalid, int lane) {
    const int x = lane & 31, h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<float4*>(img + x * 256 + (((8 * dt + 2 * c + h) ^ (x & 15)) << 4)) =
                make_float4(acc[dt][4 * c + 0] * mul, acc[dt][4 * c + 1] * mul, acc[dt][4 * c + 2] * mul, acc[dt][4 * c + 3] * mul);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4), chunk = lane & 15;
        const float4 v = *reinterpret_cast<const float4*>(img + row * 256 + ((chunk ^ (row & 15)) << 4));
        if (row < nvalid) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(g + (size_t)row * ld + 4 * chunk));
    }
}

// Prefix-LM mask of M4C's multimodal transformer (mmf/models/m4c.py:424-440), folded into the key mask: keys at or after
// `cfrom` (the decoding steps) are visible only to queries q >= cfrom with key <= q, whatever the key mask says; every
// other (query, key) pair keeps the additive key mask.  Values are in the log2 domain like lds_mask.
DEVI void tail_mask(float (&mkv)[4], int key0, int q, int cfrom, int Sk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = key0 + i;
        if (key >= cfrom && key < Sk) mkv[i] = (q >= cfrom && key <= q) ? 0.f : -10000.f * 1.4426950408889634f;
    }
}

// Mask modes of the kernels: the additive key mask [B, Sk] (every model of the path), the same plus M4C's causal tail, or a materialised
// additive mask per (query, key) pair, [B, Sq, Sk] — what `BertSelfAttentionJit.forward` accepts as a `[B, 1, S, S]` attention_mask
// (mmf/modules/hf_layers.py:161-213: `attention_scores + attention_mask`, any broadcastable shape).
enum { MASK_KEY = 0, MASK_TAIL = 1, MASK_QUERY = 2 };

// Four mask values of ONE query row (log2 domain) for keys key0 .. key0 + 3; keys >= Sk are padding (-inf).
DEVI void query_mask4(float (&mkv)[4], const float* mrow, int key0, int Sk, bool vec) {
    if (vec && key0 + 3 < Sk) {
        const float4 m = *reinterpret_cast<const float4*>(mrow + key0);
        mkv[0] = m.x * 1.4426950408889634f; mkv[1] = m.y * 1.4426950408889634f; mkv[2] = m.z * 1.4426950408889634f; mkv[3] = m.w * 1.4426950408889634f;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) mkv[i] = (key0 + i < Sk) ? mrow[key0 + i] * 1.4426950408889634f : -INFINITY;
    }
}

struct AttnArgs {
    const bf16* q; const bf16* k; const bf16* v;
    int ldq, ldk, ldv;
    const float* mask;
    bf16* ctx; int ldo;
    float* ctx32;        // optional fp32 copy of ctx (same ldo): makes delta = rowsum(dO o O) exact in backward
    float* lse;
    int B, heads, Sq, Sk, skp, hd;
    int cfrom;           // first key of the causal tail (== Sk when there is none), see mmf_attn_desc.causal_tail
    int q_bs, kv_bs, m_bs;   // rows between consecutive batches of q / of k, v / mask entries per batch (defaults Sq, Sk, Sk)
    int m_qs;                // per-query mask (mmf_attn_desc.mask_query_stride): mask entries between consecutive query rows; 0 = one mask row per batch
    int m_hs;                // per-head mask (mmf_attn_desc.mask_head_stride): mask entries between consecutive heads of a sample; 0 = one mask for all heads
    float scale;
    DropoutCfg drop;
    uint32_t* keep;          // optional dropout keep-bit table (mmf_attn_desc.keep_bits): written by the forward kernels, read by attn_bwd_fused_kernel
    const uint32_t* keep_lanes;   // optional (mmf_attn_desc.keep_lanes): the decisions drawn ahead of the forward by attn_keep_draw_kernel, in the forward's own lane order
    // backward only
    const bf16* dctx; bf16* dq; bf16* dk; bf16* dv; float* delta;
};

// Timeline probe (development aid, compiled in only with -DMMF_ATTN_PROBE; tools/attn_timeline.py): every wave appends one
// record of 12 u64 {kernel id, bh, 4 * blockIdx.y + wave, HW_ID, t0 .. t7} of s_memrealtime ticks (100 MHz).
#ifdef MMF_ATTN_PROBE
__device__ unsigned long long* g_attn_probe;
__device__ unsigned g_attn_probe_cap;
struct WaveProbe {
    unsigned long long t[8];
    DEVI void at(int i) {
        __builtin_amdgcn_sched_barrier(0);
        t[i] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
    }
    DEVI void flush(int kernel, int bh, int sub) {
        unsigned long long* buf = g_attn_probe;
        if (!buf || (threadIdx.x & 63)) return;
        const unsigned slot = atomicAdd(reinterpret_cast<unsigned*>(buf), 1u);
        if (slot >= g_attn_probe_cap) return;
        unsigned long long* r = buf + 12 * (1 + (size_t)slot);
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        r[0] = kernel; r[1] = bh; r[2] = sub; r[3] = hw;
        for (int i = 0; i < 8; ++i) r[4 + i] = t[i];
    }
};
#define PROBE_DECL WaveProbe wp = {}
#define PROBE_AT(i) wp.at(i)
#define PROBE_FLUSH(k, bh, sub) wp.flush(k, bh, sub)
#else
#define PROBE_DECL
#define PROBE_AT(i)
#define PROBE_FLUSH(k, bh, sub)
#endif

// Probability dropout of one 32 x 32 score tile held as S^T (lane = query x, half h; register 4c + i = key 8c + 4h + i of the tile), element index
// rowbase + key.  With `keep` != NULL the decisions also go to the keep-bit table the one-pass backward reads (mmf_attn_desc.keep_bits): the compare
// of register (c, i) IS a 64-bit lane mask — low half = key 8c + i against the wave's 32 queries, high half = key 8c + 4 + i — i.e. two finished table
// words; lane j < 32 collects the word of key j (v_writelane) and the half-wave stores 128 contiguous bytes per tile.
extern "C" __device__ int mmf_writelane(int value_sgpr, int lane, int old) __asm("llvm.amdgcn.writelane.i32");       // v_writelane_b32: lane `lane` of the result = value, the others keep `old`
DEVI void drop_tile(f32x16& acc, uint32_t dkey, uint32_t tilebase, int h, int lane, const DropoutCfg& drop, uint32_t* keep) {
    if (!keep) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 ds = drop_scale4(dkey, tilebase + 8 * c + 4 * h, drop.thr16, drop.scale);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * c + i] *= ds[i];
        }
        return;
    }
    int kw = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t idx4 = tilebase + 8 * c + 4 * h;
        const uint32_t h0 = drop_hash(dkey, idx4 >> 1), h1 = drop_hash(dkey, (idx4 >> 1) + 1);
        const uint32_t half[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
        unsigned long long m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool on = half[i] >= drop.thr16;
            m[i] = __builtin_amdgcn_ballot_w64(on);
            acc[4 * c + i] *= on ? drop.scale : 0.f;
        }
        // v_writelane through the LLVM intrinsic (this clang has no __builtin_amdgcn_writelane; see mmf_writelane), so that the backend's hazard
        // recognizer spaces the VALU compare that wrote the lane mask and the v_writelane that reads it as DATA.  Round 5 used inline assembly
        // with a hand-measured s_nop (nine of 32 words came out stale without it): invisible to the recognizer, i.e. one compiler upgrade away
        // from corrupt backward-only masks.  tests/test_kernels_gpu.py (keep-bit tests) compare every kernel form against the hashing path.
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kw = mmf_writelane((int)(uint32_t)m[i], 8 * c + i, kw);
            kw = mmf_writelane((int)(uint32_t)(m[i] >> 32), 8 * c + 4 + i, kw);
        }
    }
    if (lane < 32) keep[lane] = (uint32_t)kw;
}

// The same decisions WITHOUT the multiplication: what attn_keep_draw_kernel runs ahead of the step (mmf_attention_draw_keep_bits).  Returns this lane's
// 16 decisions of the tile (bit 4c + i = register 4c + i of the S^T tile, i.e. query x against key 8c + 4h + i) and writes the key-major table words
// exactly as drop_tile does.
DEVI uint32_t draw_tile(uint32_t dkey, uint32_t tilebase, int h, int lane, uint32_t thr16, uint32_t* keep) {
    int kw = 0;
    uint32_t own = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t idx4 = tilebase + 8 * c + 4 * h;
        const uint32_t h0 = drop_hash(dkey, idx4 >> 1), h1 = drop_hash(dkey, (idx4 >> 1) + 1);
        const uint32_t half[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
        unsigned long long m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool on = half[i] >= thr16;
            m[i] = __builtin_amdgcn_ballot_w64(on);
            own |= on ? (1u << (4 * c + i)) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kw = mmf_writelane((int)(uint32_t)m[i], 8 * c + i, kw);
            kw = mmf_writelane((int)(uint32_t)(m[i] >> 32), 8 * c + 4 + i, kw);
        }
    }
    if (lane < 32) keep[lane] = (uint32_t)kw;
    return own;
}

// Forward side of the pre-drawn decisions (mmf_attn_desc.keep_lanes): a lane holds the 16 decisions of each of its (up to 8) key tiles as 16-bit
// fields of ONE 16-byte word, fetched with the Q fragments; a probability costs a bit-field extract, an AND and the multiply the hashing path also
// ends in (same factor — 0 or scale — hence the same bits out).
DEVI uint32_t lane_bits(const u32x4& kl, int t) {
    uint32_t w = kl[0];
    if (t >= 2) w = kl[1];
    if (t >= 4) w = kl[2];
    if (t >= 6) w = kl[3];
    return (t & 1) ? (w >> 16) : (w & 0xffffu);
}
DEVI void drop_tile_drawn(f32x16& acc, uint32_t bits, float scale) {
    const uint32_t sb = __float_as_uint(scale);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t m = (uint32_t)((int)(bits << (31 - r)) >> 31);      // v_bfe_i32: all ones where the element is kept
        acc[r] *= __uint_as_float(sb & m);
    }
}
DEVI u32x4 load_keep_lanes(const uint32_t* keep_lanes, int bh, int Sq, int q0, int lane) {
    return *reinterpret_cast<const u32x4*>(keep_lanes + (((size_t)bh * ((Sq + 31) >> 5) + (q0 >> 5)) * 64 + lane) * 4);
}

// =================================================================================================
// forward
// =================================================================================================
template <int NKT, int D, int MM, bool DRAWN = false>
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void attn_fwd_kernel(AttnArgs a) {
    constexpr bool CZ = MM == MASK_TAIL, MQ = MM == MASK_QUERY;
    constexpr int HD = D, NS = D / 16, NDT = D / 32, ROWB = 2 * D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_k = smem;                         // row-major K  [NKT*32][D]
    unsigned char* lds_v = smem + NKT * 32 * ROWB;       // row-major V  [NKT*32][D]
    float* lds_mask = reinterpret_cast<float*>(lds_v + NKT * 32 * ROWB);  // [NKT*32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.heads, head = bh - b * a.heads;
    const int q0 = blockIdx.y * 128 + wave * 32;
    constexpr int SKP = NKT * 32;
    PROBE_DECL;
    PROBE_AT(0);

    const bf16* kbase = a.k + (size_t)b * a.kv_bs * a.ldk + head * HD;
    const bf16* vbase = a.v + (size_t)b * a.kv_bs * a.ldv + head * HD;
    stage_rows<D>(kbase, a.ldk, a.Sk, SKP, lds_k, tid);
    stage_rows<D>(vbase, a.ldv, a.Sk, SKP, lds_v, tid);
    if constexpr (!MQ) {
        for (int i = tid; i < SKP; i += 256)   // additive mask, already in the log2 domain
            lds_mask[i] = (i < a.Sk) ? (a.mask ? a.mask[(size_t)b * a.m_bs + i] * 1.4426950408889634f : 0.f) : -INFINITY;
    }
    // Q fragments (B operand: column = query row), fetched while the K / V DMA is still in flight
    const int qrow = min(q0 + x, a.Sq - 1);
    const float* mrow = MQ ? a.mask + (size_t)b * a.m_bs + (size_t)head * a.m_hs + (size_t)qrow * a.m_qs : nullptr;      // this lane's row of the per-query mask
    const bool mvec = MQ && ((a.m_qs | a.m_bs | a.m_hs) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.mask) & 15) == 0;
    const bf16* qptr = a.q + ((size_t)b * a.q_bs + qrow) * a.ldq + head * HD;
    bf16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = frag_global(qptr, s, lane);
    u32x4 kl = {};
    if constexpr (DRAWN) { if (q0 < a.Sq) kl = load_keep_lanes(a.keep_lanes, bh, a.Sq, q0, lane); }
    PROBE_AT(1);
    stage_wait();
    PROBE_AT(2);
    if (D != 64 && q0 >= a.Sq) return;     // (the head_dim-64 build has a barrier ahead of its epilogue: idle waves run along)

    // scores^T tiles: sc[t][r] = S[q = q0+x][key = 32t + (r&3) + 8(r>>2) + 4h]
    f32x16 sc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        f32x16 acc = {};
#pragma unroll
        for (int s = 0; s < NS; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_k, 32 * t, s, lane), qf[s], acc, 0, 0, 0);
        sc[t] = acc;
    }

    PROBE_AT(3);
    // softmax over keys (exact two-pass, fp32), as nn.functional.softmax(scores/sqrt(d) + mask); evaluated in the
    // log2 domain (t = s * log2e) so each probability costs one subtract and one v_exp_f32.
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = a.scale * LOG2E;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float mkv[4];
            if constexpr (MQ) query_mask4(mkv, mrow, 32 * t + 8 * c + 4 * h, a.Sk, mvec);
            else {
                const float4 mk = *reinterpret_cast<const float4*>(lds_mask + 32 * t + 8 * c + 4 * h);
                mkv[0] = mk.x; mkv[1] = mk.y; mkv[2] = mk.z; mkv[3] = mk.w;
            }
            if constexpr (CZ) tail_mask(mkv, 32 * t + 8 * c + 4 * h, q0 + x, a.cfrom, a.Sk);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[t][4 * c + i] = sc[t][4 * c + i] * sc2 + mkv[i];
                mx = fmaxf(mx, sc[t][4 * c + i]);
            }
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(sc[t][r] - mx);     // bare v_exp_f32: arguments <= 0, underflow to 0 is the intent
            sc[t][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (h == 0 && q0 + x < a.Sq) a.lse[((size_t)bh) * a.Sq + q0 + x] = (mx + log2f(sum)) * 0.6931471805599453f;

    PROBE_AT(4);
    // dropout on the probabilities (hf_layers.py:201), index ((bh*Sq + q)*SKP + key)
    if constexpr (DRAWN) {
#pragma unroll
        for (int t = 0; t < NKT; ++t) drop_tile_drawn(sc[t], lane_bits(kl, t), a.drop.scale);
    } else
    if (a.drop.thr16) {
        const uint32_t rowbase = ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)(q0 + x)) * (uint32_t)a.skp;
        const uint32_t dkey = drop_key(a.drop);
        const int nkt_rt = a.skp >> 5;      // key tiles of the keep-bit table (template tiles past it hold padding keys only)
        const bool rec = a.keep && q0 < a.Sq;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
            drop_tile(sc[t], dkey, rowbase + 32 * t, h, lane, a.drop,
                      (rec && t < nkt_rt) ? a.keep + (((size_t)bh * ((a.Sq + 31) >> 5) + (q0 >> 5)) * nkt_rt + t) * 32 : nullptr);
    }

    PROBE_AT(5);
    // ctx^T[d][q] = sum_key V^T[d][key] P^T[key][q]
    f32x16 o[NDT] = {};
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 pf = frag_regs(sc[t], u);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_v, 32 * dt, 32 * t, u, lane), pf, o[dt], 0, 0, 0);
        }

    PROBE_AT(6);
    if constexpr (D == 64) {
        // whole 128-byte (bf16) / 256-byte (fp32) rows to global memory, through a wave-private LDS image laid over the
        // K / V images once every wave of the workgroup is done with them
        __syncthreads();
        unsigned char* mine = smem + wave * (4096 + 8192);
        if (q0 < a.Sq) {
            store_tile_rows(o, inv, mine, a.ctx + ((size_t)b * a.Sq + q0) * a.ldo + head * HD, a.ldo, a.Sq - q0, lane);
            if (a.ctx32)
                store_tile_rows_f32(o, inv, mine + 4096, a.ctx32 + ((size_t)b * a.Sq + q0) * a.ldo + head * HD, a.ldo, a.Sq - q0, lane);
        }
    } else
    if (q0 + x < a.Sq) {
        bf16* optr = a.ctx + ((size_t)b * a.Sq + q0 + x) * a.ldo + head * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<bf16x4*>(optr + 32 * dt + 8 * c + 4 * h) =
                    pack4(o[dt][4 * c + 0] * inv, o[dt][4 * c + 1] * inv, o[dt][4 * c + 2] * inv, o[dt][4 * c + 3] * inv);
        if (a.ctx32) {
            float* o32 = a.ctx32 + ((size_t)b * a.Sq + q0 + x) * a.ldo + head * HD;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    __builtin_nontemporal_store(f32x4{o[dt][4 * c + 0] * inv, o[dt][4 * c + 1] * inv, o[dt][4 * c + 2] * inv, o[dt][4 * c + 3] * inv},
                                                reinterpret_cast<f32x4*>(o32 + 32 * dt + 8 * c + 4 * h));
        }
    }
    PROBE_AT(7);
    PROBE_FLUSH(0, bh, 4 * blockIdx.y + wave);
}

// One-round forward for head_dim 64 and more than 128 queries (VisualBERT VQA2: S = 228): ONE 512-thread workgroup per (batch, head),
// wave w owns query rows [32 w, 32 w + 32), K / V staged ONCE per head (the 128-query form stages them once per query half) and the
// scores are computed TWICE instead of being held in 128 registers: pass 1 finds the row maxima, pass 2 recomputes each score tile,
// exponentiates, accumulates the row sums, applies dropout and feeds P V.  That keeps the kernel under 128 VGPRs, so two workgroups fit
// on a CU: B * heads = 384 workgroups are ONE round of the 512 slots instead of 768 workgroups on 512 slots (1.5 rounds, the second
// half-empty; profiles/r02_attention_timeline.txt), with four waves per SIMD to overlap one wave's softmax VALU with another's MFMAs.
// Same arithmetic in the same order as attn_fwd_kernel (scores, maxima, exponentials, row sums, P V accumulation): bit-identical outputs.
template <int NKT, int MM, bool DRAWN = false>
__global__ __launch_bounds__(512, 4) void attn_fwd8_kernel(AttnArgs a) {
    constexpr bool CZ = MM == MASK_TAIL, MQ = MM == MASK_QUERY;
    constexpr int D = 64, HD = 64, NS = 4, NDT = 2, ROWB = 128, SKP = NKT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_k = smem;
    unsigned char* lds_v = smem + SKP * ROWB;
    float* lds_mask = reinterpret_cast<float*>(lds_v + SKP * ROWB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.heads, head = bh - b * a.heads;
    const int q0 = blockIdx.y * 256 + wave * 32;       // (blockIdx.y > 0 only beyond 256 queries: NKT 12 / 16, one workgroup per CU)

    const bf16* kbase = a.k + (size_t)b * a.kv_bs * a.ldk + head * HD;
    const bf16* vbase = a.v + (size_t)b * a.kv_bs * a.ldv + head * HD;
    stage_rows<D, 8>(kbase, a.ldk, a.Sk, SKP, lds_k, tid);
    stage_rows<D, 8>(vbase, a.ldv, a.Sk, SKP, lds_v, tid);
    if constexpr (!MQ) {
        for (int i = tid; i < SKP; i += 512)
            lds_mask[i] = (i < a.Sk) ? (a.mask ? a.mask[(size_t)b * a.m_bs + i] * 1.4426950408889634f : 0.f) : -INFINITY;
    }
    const int qrow = min(q0 + x, a.Sq - 1);
    const float* mrow = MQ ? a.mask + (size_t)b * a.m_bs + (size_t)head * a.m_hs + (size_t)qrow * a.m_qs : nullptr;
    const bool mvec = MQ && ((a.m_qs | a.m_bs | a.m_hs) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.mask) & 15) == 0;
    const bf16* qptr = a.q + ((size_t)b * a.q_bs + qrow) * a.ldq + head * HD;
    bf16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = frag_global(qptr, s, lane);
    u32x4 kl = {};
    if constexpr (DRAWN) { if (q0 < a.Sq) kl = load_keep_lanes(a.keep_lanes, bh, a.Sq, q0, lane); }
    stage_wait();
    const bool active = q0 < a.Sq;          // (idle waves run along to the barrier ahead of the epilogue)

    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = a.scale * LOG2E;
    f32x16 o[NDT] = {};
    float inv = 0.f;
    if (active) {
        // pass 1: row maxima of the masked, scaled scores (log2 domain)
        float mx = -INFINITY;
#pragma unroll 2
        for (int t = 0; t < NKT; ++t) {
            f32x16 acc = {};
#pragma unroll
            for (int s = 0; s < NS; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_k, 32 * t, s, lane), qf[s], acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float mkv[4];
                if constexpr (MQ) query_mask4(mkv, mrow, 32 * t + 8 * c + 4 * h, a.Sk, mvec);
                else {
                    const float4 mk = *reinterpret_cast<const float4*>(lds_mask + 32 * t + 8 * c + 4 * h);
                    mkv[0] = mk.x; mkv[1] = mk.y; mkv[2] = mk.z; mkv[3] = mk.w;
                }
                if constexpr (CZ) tail_mask(mkv, 32 * t + 8 * c + 4 * h, q0 + x, a.cfrom, a.Sk);
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[4 * c + i] * sc2 + mkv[i]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // pass 2: recompute, exponentiate, sum, dropout, P V
        float sum = 0.f;
        const uint32_t dkey = drop_key(a.drop);
        const uint32_t rowbase = ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)(q0 + x)) * (uint32_t)a.skp;
        const int NKT_RT = a.skp >> 5;      // key tiles of the keep-bit table (tiles past it hold padding keys only: nothing to record)
#pragma unroll 2
        for (int t = 0; t < NKT; ++t) {
            f32x16 acc = {};
#pragma unroll
            for (int s = 0; s < NS; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_k, 32 * t, s, lane), qf[s], acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float mkv[4];
                if constexpr (MQ) query_mask4(mkv, mrow, 32 * t + 8 * c + 4 * h, a.Sk, mvec);
                else {
                    const float4 mk = *reinterpret_cast<const float4*>(lds_mask + 32 * t + 8 * c + 4 * h);
                    mkv[0] = mk.x; mkv[1] = mk.y; mkv[2] = mk.z; mkv[3] = mk.w;
                }
                if constexpr (CZ) tail_mask(mkv, 32 * t + 8 * c + 4 * h, q0 + x, a.cfrom, a.Sk);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f((acc[4 * c + i] * sc2 + mkv[i]) - mx);
                    acc[4 * c + i] = p;
                    sum += p;
                }
            }
            if constexpr (DRAWN) drop_tile_drawn(acc, lane_bits(kl, t), a.drop.scale);
            else if (a.drop.thr16)
                drop_tile(acc, dkey, rowbase + 32 * t, h, lane, a.drop,
                          (a.keep && t < NKT_RT) ? a.keep + (((size_t)bh * ((a.Sq + 31) >> 5) + (q0 >> 5)) * NKT_RT + t) * 32 : nullptr);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 pf = frag_regs(acc, u);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_v, 32 * dt, 32 * t, u, lane), pf, o[dt], 0, 0, 0);
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        inv = 1.f / sum;
        if (h == 0 && q0 + x < a.Sq) a.lse[((size_t)bh) * a.Sq + q0 + x] = (mx + log2f(sum)) * 0.6931471805599453f;
    }
    // whole 128-byte (bf16) / 256-byte (fp32) rows to global memory through a wave-private 8 KB LDS image laid over the K / V images
    // (both staged through the same image, one after the other) once every wave of the workgroup is done with them
    __syncthreads();
    if (active) {
        unsigned char* mine = smem + wave * 8192;
        store_tile_rows(o, inv, mine, a.ctx + ((size_t)b * a.Sq + q0) * a.ldo + head * HD, a.ldo, a.Sq - q0, lane);
        if (a.ctx32) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the bf16 image has been read back before it is overwritten)
            store_tile_rows_f32(o, inv, mine, a.ctx32 + ((size_t)b * a.Sq + q0) * a.ldo + head * HD, a.ldo, a.Sq - q0, lane);
        }
    }
}

// =================================================================================================
// dropout decisions drawn ahead of the forward (mmf_attention_draw_keep_bits)
// =================================================================================================
// The keep decisions of the probability dropout are a pure function of (site key, step seed word, element index) — never of the scores — so ONE launch
// can draw them for every attention site of a training step before the first encoder layer runs (on a side stream beside the embedding stage, where the
// chip is idle: a branch of the step's hipGraph), and both attention kernels then READ them: the forward from `keep_lanes` (its own lane order), the
// one-pass backward from `keep_bits` (one word per key lane, as before).  One wave = one (site, batch * head, 32-query tile), all its key tiles.
constexpr int DRAW_MAX = MMF_ATTN_DRAW_MAX;
struct DrawSite {
    uint32_t key, thr16;
    const uint32_t* seed;
    uint32_t* keep;
    uint32_t* lanes;
    int Sq, skp, nqt, nkt;
    int first;          // first wave of this site in the launch
};
struct DrawArgs {
    int n, total;
    uint32_t seed_off;  // added to the seed word: 1 draws the decisions of the NEXT step (whose head advances the word by one) beside this step's AdamW
    DrawSite s[DRAW_MAX];
};
__global__ __launch_bounds__(256) void attn_keep_draw_kernel(DrawArgs a) {
    const int lane = threadIdx.x & 63, x = lane & 31, h = lane >> 5;
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= a.total) return;
    int si = 0;
    while (si + 1 < a.n && g >= a.s[si + 1].first) ++si;
    const DrawSite& st = a.s[si];
    const int local = g - st.first, bh = local / st.nqt, qt = local - bh * st.nqt;
    const uint32_t dkey = st.seed ? st.key + (st.seed[0] + a.seed_off) * 0x9E3779B1u : st.key;       // == drop_key() once the word has advanced by seed_off
    const uint32_t rowbase = ((uint32_t)bh * (uint32_t)st.Sq + (uint32_t)(32 * qt + x)) * (uint32_t)st.skp;
    uint32_t* keep = st.keep + ((size_t)bh * st.nqt + qt) * st.nkt * 32;
    u32x4 w = {};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < st.nkt) {
            const uint32_t bits = draw_tile(dkey, rowbase + 32 * t, h, lane, st.thr16, keep + 32 * t);
            w[t >> 1] |= bits << (16 * (t & 1));
        }
    }
    *reinterpret_cast<u32x4*>(st.lanes + (((size_t)bh * st.nqt + qt) * 64 + lane) * 4) = w;
}

// =================================================================================================
// backward
//   P = softmax row (recomputed from the saved log-sum-exp), Pd = dropout(P)
//   dV = Pd^T dO ; dPd = dO V^T ; dS = P o (dropscale o dPd - delta), delta = rowsum(dO o O)
//   dQ = dS K * scale ; dK = dS^T Q * scale
// =================================================================================================
// dQ kernel: same decomposition as the forward (wave = 32 query rows, all keys).
template <int NKT, int D, int MM>      // MM: MASK_KEY / MASK_TAIL / MASK_QUERY (bool-compatible: false = key mask, true = causal tail)
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr bool CZ = MM == MASK_TAIL, MQ = MM == MASK_QUERY;
    constexpr int HD = D, NS = D / 16, NDT = D / 32, ROWB = 2 * D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SKP = NKT * 32;
    unsigned char* lds_k = smem;                               // row-major K
    unsigned char* lds_v = lds_k + SKP * ROWB;                 // row-major V
    float* lds_mask = reinterpret_cast<float*>(lds_v + SKP * ROWB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.heads, head = bh - b * a.heads;
    const int q0 = blockIdx.y * 128 + wave * 32;
    PROBE_DECL;
    PROBE_AT(0);

    const bf16* kbase = a.k + (size_t)b * a.Sk * a.ldk + head * HD;
    const bf16* vbase = a.v + (size_t)b * a.Sk * a.ldv + head * HD;
    stage_rows<D>(kbase, a.ldk, a.Sk, SKP, lds_k, tid);
    stage_rows<D>(vbase, a.ldv, a.Sk, SKP, lds_v, tid);
    if constexpr (!MQ) {
        for (int i = tid; i < SKP; i += 256)   // additive mask in the log2 domain
            lds_mask[i] = (i < a.Sk) ? (a.mask ? a.mask[(size_t)b * a.Sk + i] * 1.4426950408889634f : 0.f) : -INFINITY;
    }
    // per-query operands straight from global memory, fetched while the K / V DMA is still in flight
    const int qrow = min(q0 + x, a.Sq - 1);
    const float* mrow = MQ ? a.mask + (size_t)b * a.m_bs + (size_t)head * a.m_hs + (size_t)qrow * a.m_qs : nullptr;      // this lane's row of a per-query mask
    const bool mvec = MQ && ((a.m_qs | a.m_bs | a.m_hs) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.mask) & 15) == 0;
    const bf16* qptr = a.q + ((size_t)b * a.Sq + qrow) * a.ldq + head * HD;
    const bf16* doptr = a.dctx + ((size_t)b * a.Sq + qrow) * a.ldo + head * HD;
    bf16x8 qf[NS], dof[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { qf[s] = frag_global(qptr, s, lane); dof[s] = frag_global(doptr, s, lane); }
    const float L = a.lse[(size_t)bh * a.Sq + qrow] * 1.4426950408889634f;   // log2 domain
    const float sc2 = a.scale * 1.4426950408889634f;
    // delta[q] = sum_d dO[q][d] * O[q][d], formed here (one launch less).  With the fp32 copy of O (exact row sums of dS) the
    // workgroup's 128 rows are read cooperatively — two threads per row, each a contiguous half of the head slice — so the
    // cold fp32 rows come in as full cache lines; the result goes through LDS to the lane that owns the query and out to
    // a.delta for the dK/dV kernel (next launch).
    float dl = 0.f;
    float* lds_delta = lds_mask + SKP;      // [128]
    if (a.ctx32) {
        const int r = tid >> 1, hh = tid & 1;
        const int qr = min((int)blockIdx.y * 128 + r, a.Sq - 1);
        const float* orow = a.ctx32 + ((size_t)b * a.Sq + qr) * a.ldo + head * HD + hh * (HD / 2);
        const bf16* drow = a.dctx + ((size_t)b * a.Sq + qr) * a.ldo + head * HD + hh * (HD / 2);
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 16; ++i) {
            const bf16x8 dv = *reinterpret_cast<const bf16x8*>(drow + 8 * i);
            const float4 o0 = *reinterpret_cast<const float4*>(orow + 8 * i);
            const float4 o1 = *reinterpret_cast<const float4*>(orow + 8 * i + 4);
            part += o0.x * (float)dv[0] + o0.y * (float)dv[1] + o0.z * (float)dv[2] + o0.w * (float)dv[3] +
                    o1.x * (float)dv[4] + o1.y * (float)dv[5] + o1.z * (float)dv[6] + o1.w * (float)dv[7];
        }
        part += __shfl_xor(part, 1, 64);
        if (hh == 0) {
            lds_delta[r] = part;
            if ((int)blockIdx.y * 128 + r < a.Sq) a.delta[(size_t)bh * a.Sq + blockIdx.y * 128 + r] = part;
        }
    } else {
        const bf16* orow = a.ctx + ((size_t)b * a.Sq + qrow) * a.ldo + head * HD;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bf16x8 ov = frag_global(orow, s, lane);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)ov[e] * (float)dof[s][e];
        }
        dl += __shfl_xor(dl, 32, 64);
        if (h == 0 && q0 + x < a.Sq) a.delta[(size_t)bh * a.Sq + q0 + x] = dl;
    }
    PROBE_AT(1);
    stage_wait();
    PROBE_AT(2);
    if (a.ctx32) dl = lds_delta[wave * 32 + x];
    if (q0 >= a.Sq) return;
    const uint32_t rowbase = ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)(q0 + x)) * (uint32_t)a.skp;

    f32x16 dqo[NDT] = {};
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t == 1) PROBE_AT(3);
        if (t == NKT / 2) PROBE_AT(4);
        f32x16 s_acc = {}, dp_acc = {};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_k, 32 * t, s, lane), qf[s], s_acc, 0, 0, 0);
            dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_v, 32 * t, s, lane), dof[s], dp_acc, 0, 0, 0);
        }
        // dS^T[key][q] (scaled by `scale` for the dQ product)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float mkv[4];
            if constexpr (MQ) query_mask4(mkv, mrow, 32 * t + 8 * c + 4 * h, a.Sk, mvec);
            else {
                const float4 mk = *reinterpret_cast<const float4*>(lds_mask + 32 * t + 8 * c + 4 * h);
                mkv[0] = mk.x; mkv[1] = mk.y; mkv[2] = mk.z; mkv[3] = mk.w;
            }
            if constexpr (CZ) tail_mask(mkv, 32 * t + 8 * c + 4 * h, q0 + x, a.cfrom, a.Sk);
            f32x4 ds = {1.f, 1.f, 1.f, 1.f};
            if (a.drop.thr16) ds = drop_scale4(drop_key(a.drop), rowbase + 32 * t + 8 * c + 4 * h, a.drop.thr16, a.drop.scale);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __builtin_amdgcn_exp2f(s_acc[4 * c + i] * sc2 + mkv[i] - L);
                s_acc[4 * c + i] = p * (dp_acc[4 * c + i] * ds[i] - dl) * a.scale;
            }
        }
        // dQ^T[d][q] += sum_key K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 df = frag_regs(s_acc, u);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
                dqo[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_k, 32 * dt, 32 * t, u, lane), df, dqo[dt], 0, 0, 0);
        }
    }
    PROBE_AT(5);
    if (q0 + x < a.Sq) {
        bf16* optr = a.dq + ((size_t)b * a.Sq + q0 + x) * a.ldq + head * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<bf16x4*>(optr + 32 * dt + 8 * c + 4 * h) =
                    pack4(dqo[dt][4 * c + 0], dqo[dt][4 * c + 1], dqo[dt][4 * c + 2], dqo[dt][4 * c + 3]);
    }
    PROBE_AT(6);
    PROBE_FLUSH(1, bh, 4 * blockIdx.y + wave);
}

// dK/dV kernel: wave = 32 key rows, loops over all query tiles.
template <int NQT, int D, int MM>
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr bool CZ = MM == MASK_TAIL, MQ = MM == MASK_QUERY;
    constexpr int HD = D, NS = D / 16, NDT = D / 32, ROWB = 2 * D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SQP = NQT * 32;
    unsigned char* lds_q = smem;                               // row-major Q
    unsigned char* lds_do = lds_q + SQP * ROWB;                // row-major dO
    float* lds_lse = reinterpret_cast<float*>(lds_do + SQP * ROWB);  // [SQP]
    float* lds_delta = lds_lse + SQP;                                 // [SQP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.heads, head = bh - b * a.heads;
    const int k0 = blockIdx.y * 128 + wave * 32;
    const int SKP = a.skp;
    PROBE_DECL;
    PROBE_AT(0);

    const bf16* qbase = a.q + (size_t)b * a.Sq * a.ldq + head * HD;
    const bf16* dobase = a.dctx + (size_t)b * a.Sq * a.ldo + head * HD;
    stage_rows<D>(qbase, a.ldq, a.Sq, SQP, lds_q, tid);
    stage_rows<D>(dobase, a.ldo, a.Sq, SQP, lds_do, tid);
    for (int i = tid; i < SQP; i += 256) {
        // padded query rows: lse = +inf -> p = exp(-inf) = 0, so they contribute nothing
        lds_lse[i] = (i < a.Sq) ? a.lse[(size_t)bh * a.Sq + i] * 1.4426950408889634f : INFINITY;   // log2 domain
        lds_delta[i] = (i < a.Sq) ? a.delta[(size_t)bh * a.Sq + i] : 0.f;
    }
    // per-key operands straight from global memory, fetched while the Q / dO DMA is still in flight
    const int krow = min(k0 + x, a.Sk - 1);
    const bool kvalid = (k0 + x) < a.Sk;
    const bf16* kptr = a.k + ((size_t)b * a.Sk + krow) * a.ldk + head * HD;
    const bf16* vptr = a.v + ((size_t)b * a.Sk + krow) * a.ldv + head * HD;
    bf16x8 kf[NS], vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { kf[s] = frag_global(kptr, s, lane); vf[s] = frag_global(vptr, s, lane); }
    const float mk = kvalid ? ((a.mask && !MQ) ? a.mask[(size_t)b * a.Sk + krow] * 1.4426950408889634f : 0.f) : -INFINITY;
    const float* mcol = MQ ? a.mask + (size_t)b * a.m_bs + (size_t)head * a.m_hs + krow : nullptr;       // this lane's key column of a per-query mask
    PROBE_AT(1);
    stage_wait();
    PROBE_AT(2);
    if (k0 >= a.Sk) return;
    const float sc2 = a.scale * 1.4426950408889634f;

    f32x16 dko[NDT] = {}, dvo[NDT] = {};
#pragma unroll 1
    for (int t = 0; t < NQT; ++t) {
#ifdef MMF_ATTN_PROBE
        if (t == 1) PROBE_AT(3);
        if (t == NQT / 2) PROBE_AT(4);
#endif
        // S[q][key], dPd[q][key]: rows q = 32t + (r&3)+8(r>>2)+4h, column key = k0 + x
        f32x16 s_acc = {}, dp_acc = {};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_q, 32 * t, s, lane), kf[s], s_acc, 0, 0, 0);
            dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_do, 32 * t, s, lane), vf[s], dp_acc, 0, 0, 0);
        }
        f32x16 pd;  // dropout(P) for dV
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 L4 = *reinterpret_cast<const float4*>(lds_lse + 32 * t + 8 * c + 4 * h);
            const float4 D4 = *reinterpret_cast<const float4*>(lds_delta + 32 * t + 8 * c + 4 * h);
            const float Lv[4] = {L4.x, L4.y, L4.z, L4.w};
            const float Dv[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 32 * t + 8 * c + 4 * h + i;
                float dsc = 1.f;
                if (a.drop.thr16)
                    dsc = drop_scale1(drop_key(a.drop), ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)SKP + (uint32_t)(k0 + x),
                                      a.drop.thr16, a.drop.scale);
                float mkq = mk;
                if constexpr (MQ) { if (kvalid) mkq = mcol[(size_t)min(q, a.Sq - 1) * a.m_qs] * 1.4426950408889634f; }      // (padded query rows carry lse = +inf: p = 0)
                if constexpr (CZ) {   // causal tail (see tail_mask): this lane's key against query q
                    if (kvalid && k0 + x >= a.cfrom) mkq = (q >= a.cfrom && k0 + x <= q) ? 0.f : -10000.f * 1.4426950408889634f;
                }
                const float p = __builtin_amdgcn_exp2f(s_acc[4 * c + i] * sc2 + mkq - Lv[i]);
                pd[4 * c + i] = p * dsc;
                s_acc[4 * c + i] = p * (dp_acc[4 * c + i] * dsc - Dv[i]) * a.scale;
            }
        }
        // dV^T[d][key] += sum_q dO^T[d][q] Pd[q][key] ; dK^T[d][key] += sum_q Q^T[d][q] dS[q][key]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 pf = frag_regs(pd, u);
            const bf16x8 df = frag_regs(s_acc, u);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                dvo[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_do, 32 * dt, 32 * t, u, lane), pf, dvo[dt], 0, 0, 0);
                dko[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_q, 32 * dt, 32 * t, u, lane), df, dko[dt], 0, 0, 0);
            }
        }
    }
    PROBE_AT(5);
    if (kvalid) {
        bf16* dkptr = a.dk + ((size_t)b * a.Sk + k0 + x) * a.ldk + head * HD;
        bf16* dvptr = a.dv + ((size_t)b * a.Sk + k0 + x) * a.ldv + head * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                *reinterpret_cast<bf16x4*>(dkptr + 32 * dt + 8 * c + 4 * h) =
                    pack4(dko[dt][4 * c + 0], dko[dt][4 * c + 1], dko[dt][4 * c + 2], dko[dt][4 * c + 3]);
                *reinterpret_cast<bf16x4*>(dvptr + 32 * dt + 8 * c + 4 * h) =
                    pack4(dvo[dt][4 * c + 0], dvo[dt][4 * c + 1], dvo[dt][4 * c + 2], dvo[dt][4 * c + 3]);
            }
    }
    PROBE_AT(6);
    PROBE_FLUSH(2, bh, 4 * blockIdx.y + wave);
}


// -------------------------------------------------------------------------------------------------
// One-pass backward (head_dim 64, Sq, Sk <= 256): ONE workgroup of 8 waves per (batch, head); every probability is
// recomputed, dropped out and turned into dS exactly once (the two-kernel form above pays the exp / dropout-hash VALU work
// twice, and it is VALU-bound: profiles/r02_attention_timeline.txt).
//   As a PRODUCER wave w owns key tile w (32 keys): its K and V fragments stay in registers, dK^T / dV^T accumulate in
//   registers over all query tiles (the orientation of attn_bwd_dkv_kernel: S[q][key], lane = key, registers = q).
//   dQ needs the other reduction (over keys = over the producer's lanes, and over all producers): the producer drops its
//   32 x 32 dS tile (bf16, 2 KB) into a patch in LDS; as a CONSUMER wave w owns query tile w, reads the patch back with the
//   hardware transpose read as the B operand of dQ^T[d][q] += K^T[d][key] dS^T[key][q] and accumulates dQ in registers.
//   Step i of 8: producer w works on query tile (w + i) mod 8, so the eight patches of a step belong to eight different
//   consumers; one barrier per step, patches double-buffered.  (LDS float atomics for the dQ reduction were measured at
//   ~150 cycles per wave instruction — 130 us of a 147 us loop — hence the patch hand-over.)
// LDS: Q, dO, K 32 KB each (row-major, swizzled: feature reductions by frag_rm, query / key reductions by frag_tr)
// + 2 x 8 patches of 2 KB + lse / delta = 130 KB.
// -------------------------------------------------------------------------------------------------
// a dS patch: [32 keys][32 queries] bf16, 64-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3
DEVI int patch_off(int row, int byte) { return row * 64 + ((((byte >> 4) ^ (row >> 2)) & 3) << 4) + (byte & 15); }
// B operand of the dQ product from a patch: slot (h, e) of lane x receives dS^T[key = 16u + 4h + (e&3) + 8(e>>2)][q = x]
DEVI bf16x8 frag_tr_patch(const unsigned char* patch, int u, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int row = 16 * u + 4 * (g >> 1) + (p >> 2);
    const int byte = 32 * (g & 1) + 8 * (p & 3);
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(patch + patch_off(row, byte)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(patch + patch_off(row + 8, byte)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}

// Round 5: templated on the head width and the number of waves (= key tiles = query tiles): <64, 8> is the kernel above for up to 256 positions;
// <128, 4> serves ViLBERT's visual stream and co-attention (head_dim 128, up to 128 queries / keys: 101 regions, 128 tokens) — the same 96 KB of
// operand images, four waves with the whole register file each (dK, dV, dQ accumulators of a wave: 192 registers), B x 8 heads = 256 workgroups =
// one per CU — instead of the two-kernel backward that pays the exp / dropout-hash work twice (30 + 40 us per launch at the VQA2 shape).
template <int D, int NW, int MM>
__global__ __launch_bounds__(NW * 64, 1) void attn_bwd_fused_kernel(AttnArgs a) {
    constexpr bool CZ = MM == MASK_TAIL, MQ = MM == MASK_QUERY;
    constexpr int NS = D / 16, NDT = D / 32, ROWB = 2 * D, SP = NW * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_q = smem;                                   // row-major Q   [256][64]
    unsigned char* lds_do = lds_q + SP * ROWB;                     // row-major dO  [256][64]
    unsigned char* lds_k = lds_do + SP * ROWB;                     // row-major K   [256][64]
    unsigned char* patches = lds_k + SP * ROWB;                    // [2][NW] dS patches
    float* lds_lse = reinterpret_cast<float*>(patches + 2 * NW * 2048);
    float* lds_delta = lds_lse + SP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.heads, head = bh - b * a.heads;
    const int k0 = wave * 32;
    const int nqt = (a.Sq + 31) >> 5;
    PROBE_DECL;
    PROBE_AT(0);

    const bf16* qbase = a.q + (size_t)b * a.Sq * a.ldq + head * D;
    const bf16* dobase = a.dctx + (size_t)b * a.Sq * a.ldo + head * D;
    const bf16* kbase = a.k + (size_t)b * a.Sk * a.ldk + head * D;
    stage_rows<D, NW>(kbase, a.ldk, a.Sk, (a.Sk + 31) & ~31, lds_k, tid);
    stage_rows<D, NW>(qbase, a.ldq, a.Sq, nqt * 32, lds_q, tid);
    stage_rows<D, NW>(dobase, a.ldo, a.Sq, nqt * 32, lds_do, tid);
    // V fragments of this wave's keys straight from global memory (B operand of dP = dO V^T)
    const int krow = min(k0 + x, a.Sk - 1);
    const bool kvalid = (k0 + x) < a.Sk;
    const bf16* vptr = a.v + ((size_t)b * a.Sk + krow) * a.ldv + head * D;
    bf16x8 vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) vf[s] = frag_global(vptr, s, lane);
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);      // the later-dispatched half loses VALU arbitration otherwise (MI355X_MICROARCH.md)
    const float mk = kvalid ? ((a.mask && !MQ) ? a.mask[(size_t)b * a.Sk + krow] * 1.4426950408889634f : 0.f) : -INFINITY;
    const float* mcol = MQ ? a.mask + (size_t)b * a.m_bs + (size_t)head * a.m_hs + krow : nullptr;      // per-query mask: this lane's key column, one entry per query row
    // delta[q] = sum_d dO[q][d] O[q][d] (two threads per query row, a contiguous half of the head slice each; from the fp32
    // copy of O when the forward kept one) and the log-sum-exp in the log2 domain; padded rows: lse = +inf -> p = 0
    {
        const int r = tid >> 1, hh = tid & 1;
        const int qr = min(r, a.Sq - 1);
        const bf16* drow = a.dctx + ((size_t)b * a.Sq + qr) * a.ldo + head * D + hh * (D / 2);
        float part = 0.f;
        if (a.ctx32) {
            const float* orow = a.ctx32 + ((size_t)b * a.Sq + qr) * a.ldo + head * D + hh * (D / 2);
#pragma unroll
            for (int i = 0; i < D / 16; ++i) {
                const bf16x8 dv = *reinterpret_cast<const bf16x8*>(drow + 8 * i);
                const float4 o0 = *reinterpret_cast<const float4*>(orow + 8 * i);
                const float4 o1 = *reinterpret_cast<const float4*>(orow + 8 * i + 4);
                part += o0.x * (float)dv[0] + o0.y * (float)dv[1] + o0.z * (float)dv[2] + o0.w * (float)dv[3] +
                        o1.x * (float)dv[4] + o1.y * (float)dv[5] + o1.z * (float)dv[6] + o1.w * (float)dv[7];
            }
        } else {
            const bf16* orow = a.ctx + ((size_t)b * a.Sq + qr) * a.ldo + head * D + hh * (D / 2);
#pragma unroll
            for (int i = 0; i < D / 16; ++i) {
                const bf16x8 dv = *reinterpret_cast<const bf16x8*>(drow + 8 * i);
                const bf16x8 ov = *reinterpret_cast<const bf16x8*>(orow + 8 * i);
#pragma unroll
                for (int e = 0; e < 8; ++e) part += (float)ov[e] * (float)dv[e];
            }
        }
        part += __shfl_xor(part, 1, 64);
        if (hh == 0) {
            lds_delta[r] = (r < a.Sq) ? part : 0.f;
            lds_lse[r] = (r < a.Sq) ? a.lse[(size_t)bh * a.Sq + r] * 1.4426950408889634f : INFINITY;
        }
    }
    PROBE_AT(1);
    stage_wait();
    PROBE_AT(2);
    const bool produce = k0 < a.Sk, consume = wave < nqt;
    bf16x8 kf[NS];      // B operand of S = Q K^T for this wave's keys
    if (produce) {
#pragma unroll
        for (int s = 0; s < NS; ++s) kf[s] = frag_rm<D>(lds_k, k0, s, lane);
    }
    PROBE_AT(3);

    const float sc2 = a.scale * 1.4426950408889634f;
    const uint32_t dkey = drop_key(a.drop);
    f32x16 dko[NDT] = {}, dvo[NDT] = {}, dqo[NDT] = {};
    // keep-bit table of the forward (mmf_attn_desc.keep_bits): this lane's key against the 32 queries of tile t is ONE word; the word of the next step is
    // fetched a step ahead
    const int nkt_rt = a.skp >> 5;
    const uint32_t* kbits = a.keep ? a.keep + ((size_t)bh * nqt * nkt_rt + wave) * 32 + x : nullptr;
    uint32_t kw_next = 0;
    if (kbits && produce && wave < nqt) kw_next = kbits[(size_t)wave * nkt_rt * 32];      // (step 0: query tile = wave)
#pragma unroll 1
    for (int it = 0; it < NW; ++it) {
        const int t = (wave + it) & (NW - 1);          // query tile this wave produces dS for in this step
        const uint32_t kw = kw_next >> (4 * h);        // bit 8c + i = keep(query 32t + 8c + 4h + i, this key)
        if (kbits && produce) {
            const int tn = (wave + it + 1) & (NW - 1);
            if (it + 1 < NW && tn < nqt) kw_next = kbits[(size_t)tn * nkt_rt * 32];
        }
        if (produce && t < nqt) {
            unsigned char* patch = patches + ((it & 1) * NW + wave) * 2048;
            // S[q][key], dPd[q][key]: rows q = 32t + (r&3) + 8(r>>2) + 4h, column key = k0 + x
            f32x16 s_acc = {}, dp_acc = {};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_q, 32 * t, s, lane), kf[s], s_acc, 0, 0, 0);
                dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm<D>(lds_do, 32 * t, s, lane), vf[s], dp_acc, 0, 0, 0);
            }
            f32x16 pd;  // dropout(P) for dV
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 L4 = *reinterpret_cast<const float4*>(lds_lse + 32 * t + 8 * c + 4 * h);
                const float4 D4 = *reinterpret_cast<const float4*>(lds_delta + 32 * t + 8 * c + 4 * h);
                const float Lv[4] = {L4.x, L4.y, L4.z, L4.w};
                const float Dv[4] = {D4.x, D4.y, D4.z, D4.w};
                float dsc[4] = {1.f, 1.f, 1.f, 1.f};
                if (kbits) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dsc[i] = (kw & (1u << (8 * c + i))) ? a.drop.scale : 0.f;
                } else if (a.drop.thr16) {
                    // one hash serves the two keys of a pair (neighbouring lanes): even lanes hash queries i = 0, 2, odd lanes
                    // i = 1, 3, and the lanes swap (DPP quad_perm 1,0,3,2) — half the hashing of a per-element draw
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int iq = 2 * j + (x & 1);
                        const uint32_t q = 32 * t + 8 * c + 4 * h + iq;
                        const uint32_t idx = ((uint32_t)bh * (uint32_t)a.Sq + q) * (uint32_t)a.skp + (uint32_t)(k0 + x);
                        const uint32_t mine = drop_hash(dkey, idx >> 1);
                        const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);
                        const uint32_t he = (x & 1) ? other : mine, ho = (x & 1) ? mine : other;   // hashes of queries 2j, 2j+1
                        const uint32_t ve = (x & 1) ? (he >> 16) : (he & 0xffffu), vo = (x & 1) ? (ho >> 16) : (ho & 0xffffu);
                        dsc[2 * j] = (ve >= a.drop.thr16) ? a.drop.scale : 0.f;
                        dsc[2 * j + 1] = (vo >= a.drop.thr16) ? a.drop.scale : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float mkq = mk;
                    if constexpr (CZ) {   // causal tail (see tail_mask): this lane's key against query q
                        const int q = 32 * t + 8 * c + 4 * h + i;
                        if (kvalid && k0 + x >= a.cfrom) mkq = (q >= a.cfrom && k0 + x <= q) ? 0.f : -10000.f * 1.4426950408889634f;
                    }
                    if constexpr (MQ) {   // (padded query rows carry lse = +inf: p = 0 whatever is read for them)
                        const int q = min(32 * t + 8 * c + 4 * h + i, a.Sq - 1);
                        if (kvalid) mkq = mcol[(size_t)q * a.m_qs] * 1.4426950408889634f;
                    }
                    const float p = __builtin_amdgcn_exp2f(s_acc[4 * c + i] * sc2 + mkq - Lv[i]);
                    pd[4 * c + i] = p * dsc[i];
                    s_acc[4 * c + i] = p * (dp_acc[4 * c + i] * dsc[i] - Dv[i]);      // dS without the 1/sqrt(d): applied to dK, dQ at the end
                }
                // dS tile -> patch, row = key (this lane), 4 consecutive queries 8c + 4h .. + 3
                *reinterpret_cast<bf16x4*>(patch + patch_off(x, 16 * c + 8 * h)) =
                    pack4(s_acc[4 * c + 0], s_acc[4 * c + 1], s_acc[4 * c + 2], s_acc[4 * c + 3]);
            }
            // dV^T[d][key] += sum_q dO^T[d][q] Pd[q][key] ; dK^T[d][key] += sum_q Q^T[d][q] dS[q][key]
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 pf = frag_regs(pd, u);
                const bf16x8 df = frag_regs(s_acc, u);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    dvo[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_do, 32 * dt, 32 * t, u, lane), pf, dvo[dt], 0, 0, 0);
                    dko[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_q, 32 * dt, 32 * t, u, lane), df, dko[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        // consumer: query tile `wave`, the patch of key tile kt (produced in this step by wave kt)
        const int kt = (wave - it) & (NW - 1);
        if (consume && kt * 32 < a.Sk) {
            const unsigned char* src = patches + ((it & 1) * NW + kt) * 2048;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 dst = frag_tr_patch(src, u, lane);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    dqo[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(lds_k, 32 * dt, 32 * kt, u, lane), dst, dqo[dt], 0, 0, 0);
            }
        }
    }
    PROBE_AT(4);
    // Epilogue: the three 32 x 64 result tiles of a wave (dK, dV of its keys, dQ of its queries) go through a private 4 KB LDS
    // image each (the operand images are dead after the barrier) so that global memory sees whole 128-byte rows, 16 bytes per
    // lane, instead of 8-byte pieces at a 4.6 KB stride.
    __syncthreads();
    constexpr int IMG = 32 * ROWB;              // one 32-row result tile as whole rows: 4 KB (head_dim 64) / 8 KB (128)
    unsigned char* mine = smem + wave * (3 * IMG);
    if (produce) {
        store_tile_rows_d<D>(dko, a.scale, mine, a.dk + ((size_t)b * a.Sk + k0) * a.ldk + head * D, a.ldk, a.Sk - k0, lane);
        store_tile_rows_d<D>(dvo, 1.f, mine + IMG, a.dv + ((size_t)b * a.Sk + k0) * a.ldv + head * D, a.ldv, a.Sk - k0, lane);
    }
    PROBE_AT(5);
    if (consume)
        store_tile_rows_d<D>(dqo, a.scale, mine + 2 * IMG, a.dq + ((size_t)b * a.Sq + k0) * a.ldq + head * D, a.ldq, a.Sq - k0, lane);
    PROBE_AT(6);
    PROBE_FLUSH(3, bh, wave);
}

// shapes whose backward is the one-pass kernel (attn_bwd_fused_kernel<64, 8, *> / <128, 4, *>); every forward form of those shapes writes the table
bool keep_bits_shape(int hd, int Sq, int Sk) { return hd == 64 ? (Sq <= 256 && Sk <= 256) : (Sq <= 128 && Sk <= 128); }

int fill_args(const mmf_attn_desc* d, AttnArgs& a) {
    MMF_CHECK_ARG(d && d->q && d->k && d->v, "attention: null operand");
    MMF_CHECK_ARG(d->B > 0 && d->heads > 0 && d->Sq > 0 && d->Sk > 0, "attention: empty shape");
    const int hd = d->head_dim ? d->head_dim : 64;
    MMF_CHECK_ARG(hd == 64 || hd == 128, "attention: head_dim must be 64 or 128");
    MMF_CHECK_ARG(d->Sk <= 512 && d->Sq <= 512, "attention: Sq and Sk must be <= 512 (a head's K and V are staged whole in LDS)");
    MMF_CHECK_ARG(hd == 64 || (d->Sk <= 256 && d->Sq <= 256), "attention: head_dim 128 is built for Sq, Sk <= 256");
    MMF_CHECK_ARG((d->ldq % 8) == 0 && (d->ldk % 8) == 0 && (d->ldv % 8) == 0 && (d->ldo % 8) == 0,
                  "attention: leading dimensions must be multiples of 8 elements");
    a.q = (const bf16*)d->q; a.k = (const bf16*)d->k; a.v = (const bf16*)d->v;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv;
    a.mask = d->mask; a.ctx = (bf16*)d->ctx; a.ldo = d->ldo; a.lse = d->lse; a.ctx32 = d->ctx_f32;
    a.B = d->B; a.heads = d->heads; a.Sq = d->Sq; a.Sk = d->Sk; a.hd = hd;
    a.skp = (d->Sk + 31) / 32 * 32;
    MMF_CHECK_ARG(d->causal_tail >= 0 && d->causal_tail <= d->Sk, "attention: causal_tail out of range");
    MMF_CHECK_ARG(d->causal_tail == 0 || (d->Sq == d->Sk && hd == 64), "attention: a causal tail needs self-attention (Sq == Sk) with head_dim 64");
    a.cfrom = d->Sk - d->causal_tail;
    a.q_bs = d->q_batch_rows > 0 ? d->q_batch_rows : d->Sq;
    a.kv_bs = d->kv_batch_rows > 0 ? d->kv_batch_rows : d->Sk;
    a.m_qs = d->mask_query_stride;
    MMF_CHECK_ARG(a.m_qs == 0 || (d->mask && a.m_qs >= d->Sk), "attention: mask_query_stride must cover a mask row (>= Sk)");
    MMF_CHECK_ARG(a.m_qs == 0 || (hd == 64 && d->causal_tail == 0), "attention: a per-query mask is built for head_dim 64 (and replaces the causal tail)");
    a.m_hs = d->mask_head_stride;
    MMF_CHECK_ARG(a.m_hs == 0 || (a.m_qs != 0 && a.m_hs >= (d->Sq - 1) * a.m_qs + d->Sk),
                  "attention: mask_head_stride goes with a per-query mask (mask_query_stride) and must cover one head's [Sq, Sk] mask");
    a.m_bs = d->mask_batch_stride > 0 ? d->mask_batch_stride : (a.m_hs ? d->heads * a.m_hs : (a.m_qs ? d->Sq * a.m_qs : d->Sk));
    MMF_CHECK_ARG(a.q_bs >= d->Sq && a.kv_bs >= d->Sk && a.m_bs >= d->Sk, "attention: batch strides must cover the sequence");
    MMF_CHECK_ARG(a.m_qs == 0 || a.m_bs >= (a.m_hs ? (d->heads - 1) * a.m_hs : 0) + (d->Sq - 1) * a.m_qs + d->Sk,
                  "attention: mask_batch_stride must cover the per-query mask of a sample");
    a.scale = d->scale;
    a.drop.key = d->drop_key; a.drop.thr16 = d->drop_thr16; a.drop.scale = d->drop_scale; a.drop.seed = d->drop_seed;
    a.dctx = nullptr; a.dq = a.dk = a.dv = nullptr; a.delta = nullptr;
    a.keep = d->keep_bits;
    a.keep_lanes = d->keep_lanes;
    MMF_CHECK_ARG(!a.keep_lanes || (d->drop_thr16 != 0 && keep_bits_shape(hd, d->Sq, d->Sk) && d->q_batch_rows == 0 && d->kv_batch_rows == 0 &&
                                    (reinterpret_cast<uintptr_t>(a.keep_lanes) & 15) == 0),
                  "attention: keep_lanes (decisions drawn by mmf_attention_draw_keep_bits) is taken for the shapes that take keep_bits, with dropout on, 16-byte aligned");
    MMF_CHECK_ARG(!a.keep || (d->drop_thr16 != 0 && keep_bits_shape(hd, d->Sq, d->Sk) && d->q_batch_rows == 0 && d->kv_batch_rows == 0),
                  "attention: keep_bits is taken where the backward is the one-pass kernel (head_dim 64: <= 256 positions, 128: <= 128), with dropout on (mmf_attention_keep_bits_words)");
    return 0;
}

template <typename K>
int set_lds(K kern, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { mmf_amd_set_error(hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

extern "C" int64_t mmf_attention_keep_bits_words(int B, int heads, int Sq, int Sk, int head_dim) {
    const int hd = head_dim ? head_dim : 64;
    if (B <= 0 || heads <= 0 || !keep_bits_shape(hd, Sq, Sk)) return 0;
    if (mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 4) return 0;      // (the two-kernel backward hashes)
    return (int64_t)B * heads * ((Sq + 31) / 32) * ((Sk + 31) / 32) * 32;
}

extern "C" int64_t mmf_attention_keep_lanes_words(int B, int heads, int Sq, int Sk, int head_dim) {
    const int hd = head_dim ? head_dim : 64;
    if (B <= 0 || heads <= 0 || !keep_bits_shape(hd, Sq, Sk)) return 0;
    return (int64_t)B * heads * ((Sq + 31) / 32) * 64 * 4;
}

extern "C" int mmf_attention_draw_keep_bits(const mmf_attn_draw_site* sites, int n, uint32_t seed_offset, void* stream) {
    MMF_CHECK_ARG(n >= 0 && (n == 0 || sites), "attention_draw_keep_bits: null sites");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += DRAW_MAX) {
        DrawArgs a;
        a.n = n - base < DRAW_MAX ? n - base : DRAW_MAX;
        a.seed_off = seed_offset;
        int total = 0;
        for (int i = 0; i < a.n; ++i) {
            const mmf_attn_draw_site& d = sites[base + i];
            const int hd = d.head_dim ? d.head_dim : 64;
            MMF_CHECK_ARG(d.B > 0 && d.heads > 0 && (hd == 64 || hd == 128) && keep_bits_shape(hd, d.Sq, d.Sk) && d.Sq > 0 && d.Sk > 0,
                          "attention_draw_keep_bits: a site must be a shape that takes keep_bits (mmf_attention_keep_bits_words != 0)");
            MMF_CHECK_ARG(d.drop_thr16 != 0 && d.keep_bits && d.keep_lanes && (reinterpret_cast<uintptr_t>(d.keep_lanes) & 15) == 0,
                          "attention_draw_keep_bits: dropout off, or a null / misaligned table");
            DrawSite& t = a.s[i];
            t.key = d.drop_key; t.thr16 = d.drop_thr16; t.seed = d.drop_seed; t.keep = d.keep_bits; t.lanes = d.keep_lanes;
            t.Sq = d.Sq; t.skp = (d.Sk + 31) / 32 * 32; t.nqt = (d.Sq + 31) / 32; t.nkt = t.skp / 32; t.first = total;
            total += d.B * d.heads * t.nqt;
        }
        a.total = total;
        if (total == 0) continue;
        hipLaunchKernelGGL(attn_keep_draw_kernel, dim3((total + 3) / 4), dim3(256), 0, s, a);
        MMF_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int mmf_attention_fwd(const mmf_attn_desc* d, void* stream) {
    AttnArgs a;
    if (int rc = fill_args(d, a)) return rc;
    MMF_CHECK_ARG(d->ctx && d->lse, "attention_fwd: null output");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nkt = a.skp / 32;
    const dim3 grid(a.B * a.heads, (a.Sq + 127) / 128);
#define LAUNCH_FWD_(N, DD, CZ, DR)                                                               \
    {                                                                                            \
        int lds = 2 * N * 32 * (2 * DD) + N * 32 * 4;                                            \
        if (DD == 64 && lds < 4 * 12288) lds = 4 * 12288;   /* the epilogue's row images */      \
        if (int rc = set_lds(attn_fwd_kernel<N, DD, CZ, DR>, lds)) return rc;                    \
        hipLaunchKernelGGL((attn_fwd_kernel<N, DD, CZ, DR>), grid, dim3(256), lds, s, a);        \
    }
    /* (decisions drawn ahead, mmf_attn_desc.keep_lanes: shapes with <= 8 key tiles only, checked in fill_args) */
#define LAUNCH_FWD(N, DD, CZ) { if (a.keep_lanes) LAUNCH_FWD_(N, DD, CZ, true) else LAUNCH_FWD_(N, DD, CZ, false) }
#define LAUNCH_FWD8(MMODE)                                                                       \
    {                                                                                            \
        const int lds = 2 * 8 * 32 * 128 + 8 * 32 * 4;                                           \
        const dim3 grid8(a.B * a.heads, (a.Sq + 255) / 256);                                     \
        if (a.keep_lanes) {                                                                      \
            if (int rc = set_lds(attn_fwd8_kernel<8, MMODE, true>, lds)) return rc;              \
            hipLaunchKernelGGL((attn_fwd8_kernel<8, MMODE, true>), grid8, dim3(512), lds, s, a); \
        } else {                                                                                 \
            if (int rc = set_lds(attn_fwd8_kernel<8, MMODE>, lds)) return rc;                    \
            hipLaunchKernelGGL((attn_fwd8_kernel<8, MMODE>), grid8, dim3(512), lds, s, a);       \
        }                                                                                        \
    }
    const bool cz = a.cfrom < a.Sk;
    if (a.hd == 64 && nkt > 8) {
        // more than 256 keys (head_dim 64): the one-round kernel with 12 or 16 key tiles — K and V of the head whole in LDS (96 / 128 KB), one
        // workgroup per CU and per 256 queries, scores computed twice, same arithmetic in the same order as the shorter forms
        const dim3 gridL(a.B * a.heads, (a.Sq + 255) / 256);
#define LAUNCH_FWD_LONG(N, MMODE)                                                                 \
    {                                                                                            \
        const int lds = 2 * N * 32 * 128 + N * 32 * 4;                                           \
        if (int rc = set_lds(attn_fwd8_kernel<N, MMODE>, lds)) return rc;                        \
        hipLaunchKernelGGL((attn_fwd8_kernel<N, MMODE>), gridL, dim3(512), lds, s, a);           \
    }
        if (a.m_qs) { if (nkt <= 12) LAUNCH_FWD_LONG(12, MASK_QUERY) else LAUNCH_FWD_LONG(16, MASK_QUERY) }
        else if (nkt <= 12) { if (cz) LAUNCH_FWD_LONG(12, MASK_TAIL) else LAUNCH_FWD_LONG(12, MASK_KEY) }
        else { if (cz) LAUNCH_FWD_LONG(16, MASK_TAIL) else LAUNCH_FWD_LONG(16, MASK_KEY) }
#undef LAUNCH_FWD_LONG
        MMF_CHECK_LAUNCH();
        return 0;
    }
    if (a.m_qs) {      // per-query mask [B, Sq, Sk] (head_dim 64): the same two kernel forms, mask read per (query, key) from global memory
        if (nkt > 4 && a.Sq > 128 && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 2)) LAUNCH_FWD8(MASK_QUERY)
        else if (nkt <= 4) LAUNCH_FWD(4, 64, MASK_QUERY)
        else LAUNCH_FWD(8, 64, MASK_QUERY)
        MMF_CHECK_LAUNCH();
        return 0;
    }
    // head_dim 64 with more than 128 queries: the one-round form (one 8-wave workgroup per (batch, head), two per CU; attn_fwd8_kernel).
    // MMF_TUN_ALT_FORMS bit 1 keeps the two-workgroups-per-head form (A/B measurements, bit-equality test).
    if (a.hd == 64 && nkt > 4 && a.Sq > 128 && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 2)) {
        if (cz) LAUNCH_FWD8(MASK_TAIL) else LAUNCH_FWD8(MASK_KEY)
        MMF_CHECK_LAUNCH();
        return 0;
    }
    if (a.hd == 128) { if (nkt <= 4) LAUNCH_FWD(4, 128, MASK_KEY) else LAUNCH_FWD_(8, 128, MASK_KEY, false) }
    else if (nkt <= 4) { if (cz) LAUNCH_FWD(4, 64, MASK_TAIL) else LAUNCH_FWD(4, 64, MASK_KEY) }
    else { if (cz) LAUNCH_FWD(8, 64, MASK_TAIL) else LAUNCH_FWD(8, 64, MASK_KEY) }
#undef LAUNCH_FWD
#undef LAUNCH_FWD_
#undef LAUNCH_FWD8
    MMF_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmf_attention_bwd(const mmf_attn_bwd_desc* d, void* stream) {
    AttnArgs a;
    MMF_CHECK_ARG(d, "attention_bwd: null desc");
    if (int rc = fill_args(&d->f, a)) return rc;
    MMF_CHECK_ARG(d->f.ctx && d->f.lse && d->dctx && d->dq && d->dk && d->dv && d->delta, "attention_bwd: null operand");
    MMF_CHECK_ARG(d->f.q_batch_rows == 0 && d->f.kv_batch_rows == 0 && d->f.mask_batch_stride == 0,
                  "attention_bwd: custom batch strides are a forward-only (decoding) feature");
    a.dctx = (const bf16*)d->dctx; a.dq = (bf16*)d->dq; a.dk = (bf16*)d->dk; a.dv = (bf16*)d->dv; a.delta = d->delta;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);

    // delta = rowsum(dO o O) is formed inside the dQ kernel and handed to the dK/dV kernel through d->delta

    const int nkt = a.skp / 32;
    const int nqt = (a.Sq + 31) / 32;
    const bool cz = a.cfrom < a.Sk;
    if (a.hd == 64 && nkt <= 8 && nqt <= 8 && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 4)) {
        const int lds = 3 * 256 * 128 + 16 * 2048 + 2 * 256 * 4;
        if (a.m_qs) {      // per-query mask: the one-pass kernel only
            if (int rc = set_lds(attn_bwd_fused_kernel<64, 8, MASK_QUERY>, lds)) return rc;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<64, 8, MASK_QUERY>), dim3(a.B * a.heads), dim3(512), lds, s, a);
        } else if (cz) {
            if (int rc = set_lds(attn_bwd_fused_kernel<64, 8, MASK_TAIL>, lds)) return rc;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<64, 8, MASK_TAIL>), dim3(a.B * a.heads), dim3(512), lds, s, a);
        } else {
            if (int rc = set_lds(attn_bwd_fused_kernel<64, 8, MASK_KEY>, lds)) return rc;
            hipLaunchKernelGGL((attn_bwd_fused_kernel<64, 8, MASK_KEY>), dim3(a.B * a.heads), dim3(512), lds, s, a);
        }
        MMF_CHECK_LAUNCH();
        return 0;
    }
    if (a.hd == 128 && nkt <= 4 && nqt <= 4 && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 4)) {
        // head_dim 128 (ViLBERT's visual stream and co-attention), up to 128 queries / keys: the one-pass kernel with four waves per (batch, head)
        const int lds = 3 * 128 * 256 + 8 * 2048 + 2 * 128 * 4;
        if (int rc = set_lds(attn_bwd_fused_kernel<128, 4, MASK_KEY>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, MASK_KEY>), dim3(a.B * a.heads), dim3(256), lds, s, a);
        MMF_CHECK_LAUNCH();
        return 0;
    }
    {
        const dim3 grid(a.B * a.heads, (a.Sq + 127) / 128);
#define LAUNCH_DQ(N, DD, CZ)                                                                     \
    {                                                                                            \
        const int lds = 2 * N * 32 * (2 * DD) + N * 32 * 4 + 128 * 4;                            \
        if (int rc = set_lds(attn_bwd_dq_kernel<N, DD, CZ>, lds)) return rc;                     \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<N, DD, CZ>), grid, dim3(256), lds, s, a);         \
    }
        if (a.m_qs) LAUNCH_DQ(16, 64, MASK_QUERY)      // a per-query mask where the one-pass kernel does not run (beyond 256 positions): one 16-tile form
        else if (a.hd == 128) { if (nkt <= 4) LAUNCH_DQ(4, 128, false) else LAUNCH_DQ(8, 128, false) }
        else if (nkt <= 4) { if (cz) LAUNCH_DQ(4, 64, true) else LAUNCH_DQ(4, 64, false) }
        else if (nkt <= 8) { if (cz) LAUNCH_DQ(8, 64, true) else LAUNCH_DQ(8, 64, false) }
        else { if (cz) LAUNCH_DQ(16, 64, true) else LAUNCH_DQ(16, 64, false) }      // 257 .. 512 keys: K, V whole in LDS (128 KB), one workgroup per CU
#undef LAUNCH_DQ
        MMF_CHECK_LAUNCH();
    }
    {
        const dim3 grid(a.B * a.heads, (a.Sk + 127) / 128);
#define LAUNCH_DKV(N, DD, CZ)                                                                    \
    {                                                                                            \
        const int lds = 2 * N * 32 * (2 * DD) + 2 * N * 32 * 4;                                  \
        if (int rc = set_lds(attn_bwd_dkv_kernel<N, DD, CZ>, lds)) return rc;                    \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<N, DD, CZ>), grid, dim3(256), lds, s, a);        \
    }
        if (a.m_qs) LAUNCH_DKV(16, 64, MASK_QUERY)
        else if (a.hd == 128) { if (nqt <= 4) LAUNCH_DKV(4, 128, false) else LAUNCH_DKV(8, 128, false) }
        else if (nqt <= 4) { if (cz) LAUNCH_DKV(4, 64, true) else LAUNCH_DKV(4, 64, false) }
        else if (nqt <= 8) { if (cz) LAUNCH_DKV(8, 64, true) else LAUNCH_DKV(8, 64, false) }
        else { if (cz) LAUNCH_DKV(16, 64, true) else LAUNCH_DKV(16, 64, false) }    // 257 .. 512 queries: Q, dO whole in LDS
#undef LAUNCH_DKV
        MMF_CHECK_LAUNCH();
    }
    return 0;
}

// Development aid (see WaveProbe): buf = (1 + capacity) * 96 bytes of zeroed device memory, NULL switches the probe off.  Only a
// library built with -DMMF_ATTN_PROBE records anything; the regular build returns an error.
extern "C" int mmf_attention_set_probe(void* buf, int64_t capacity_records) {
#ifdef MMF_ATTN_PROBE
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    unsigned cap = (unsigned)(capacity_records > 0 ? capacity_records : 0);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_probe), &p, sizeof(p)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_attn_probe_cap), &cap, sizeof(cap)) != hipSuccess) {
        mmf_amd_set_error("attention_set_probe: hipMemcpyToSymbol failed");
        return 2;
    }
    return 0;
#else
    (void)buf; (void)capacity_records;
    mmf_amd_set_error("attention_set_probe: this library was built without -DMMF_ATTN_PROBE");
    return 1;
#endif
}
