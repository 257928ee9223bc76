// mmf_amd :: bf16 MFMA GEMM for gfx950 with fused epilogues.
//
// Replaces the stock ATen `nn.Linear` / `torch.matmul` calls on the reference hot path:
//   Q/K/V projections        mmf/modules/hf_layers.py:169,179-180
//   attention output dense   HF BertSelfOutput   (call site hf_layers.py:248)
//   FFN up + GELU            HF BertIntermediate (call site hf_layers.py:289)
//   FFN down                 HF BertOutput       (call site hf_layers.py:290)
//   visual projection        mmf/modules/embeddings.py:319,352
//   classifier head          mmf/models/visual_bert.py:327-330
// and their autograd backward (dgrad / wgrad), mmf/trainers/core/training_loop.py:211.
//
// C[m][n] = sum_k A(m,k) * B(n,k)          (fp32 accumulate on v_mfma_f32_16x16x32_bf16)
//   A "row"    : A(m,k) = A[m*lda + k]        A "k-major": A(m,k) = A[k*lda + m]
//   B "row"    : B(n,k) = B[n*ldb + k]        B "k-major": B(n,k) = B[k*ldb + n]
//   forward  Y = X W^T      : A row,     B row      (W is [out,in] like nn.Linear)
//   dgrad    dX = dY W      : A row,     B k-major  (reduction index = out = row index of W)
//   wgrad    dW = dY^T X    : A k-major, B k-major  (reduction index = token = row index of both)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), wave tile 64x64 = 4x4 MFMA 16x16 fragments.
// Register-staged double-buffered LDS (global -> VGPR -> LDS), one barrier per K-tile.
// LDS images (16 KiB per operand per stage, 64 KiB per workgroup -> 2 workgroups / CU):
//   row operand     : [128 rows][8 chunks of 16 B], chunk ^= (row & 7)   -> conflict-free ds_read_b128
//   k-major operand : [64 k-rows][256 B], bytes rotated by 32*((k&3) + 4*((k>>3)&1)) within the row
//                     -> conflict-free ds_read_b64_tr_b16 (hardware 4x16 transpose read)
// The MFMA is issued with operands swapped (A-operand = B tile fragment) so each lane ends up holding 4 CONSECUTIVE n
// for one m; the fp32 tile is then staged through LDS and written row-wise (16 bytes per lane, full cache lines).
#include "gemm_common.h"
#include "gemm_wide.h"
#include "gemm_persist.h"
#include "ln_bwd_dev.h"

using namespace gemm;

#ifndef MMF_EPI_NT_DEFAULT
#define MMF_EPI_NT_DEFAULT 7
#endif
// Call sites (MMF_SITE_*) whose bf16 output is stored write-through (`sc1`) instead of non-temporally: a non-temporal store bypasses the Infinity
// Cache, and the GEMM that reads the 45 MB activation next as its A operand then streams it from HBM with a K-loop ring that covers ~1.5 us of
// latency: 49 instead of 39 us for the FFN-down forward (tools/cold_operand_probe.py).  In the step the producer pays part of it back; per
// site, same process, interleaved (profiles/r04_store_policy.txt): FFN-up forward -0.02 .. -0.03 ms, FFN-down dgrad -0.06 ms, every other site
// +0.00 .. +0.04 ms with `sc1` and +0.01 .. +0.13 ms with plain stores.
#ifndef MMF_SITE_SC1_DEFAULT
#define MMF_SITE_SC1_DEFAULT ((1 << MMF_SITE_FFN_UP_FWD) | (1 << MMF_SITE_FFN_DOWN_DGRAD))
#endif


namespace {

DEVI unsigned long long probe_now() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz, one counter for the whole device

// ---- one output tile --------------------------------------------------------------------------------
template <typename AT, typename BT, bool A_KMAJOR, bool B_KMAJOR, bool RAGGED, int NWN, int BN_, int KS, bool RS>
DEVI void gemm_tile(const AT* __restrict__ A, const BT* __restrict__ B, int M, int N, int K, int lda, int ldb,
                    int tile_m, int tile_n, int split, int splits, int dbg, const EpiArgs& epi, unsigned char* smem, const Probe& pr) {
    constexpr bool A_DMA = is_bf16<AT>::value, B_DMA = is_bf16<BT>::value;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    unsigned long long pt[5] = {0, 0, 0, 0, 0};
    const bool probing = pr.buf != nullptr;
    if (probing) pt[0] = probe_now();
    // Wave grid over the 128 x BN_ tile.  BN_ = 128: 2 x NWN waves of 64 x (128/NWN).  BN_ = 96 (8 waves only): 4 x 2 waves of 32 x 48,
    // used where 128-wide tiles would leave a third of the CUs idle in the last round (N = 768 / 2304 at M = 7296).
    // KS = 2 (8 waves): 2 x 2 waves of 64 x (BN_/2), times two K-halves — each wave multiplies ONE of the two 32-deep
    // halves of a stage.  Same MFMA count per wave and step, a third fewer LDS operand reads (a 64x64 wave tile reuses
    // each fragment twice as often as a 64x32 one); the two halves are summed once, in the epilogue's LDS stage.
    constexpr int NTH = 128 * NWN;
    constexpr int WGN = (KS == 2) ? 2 : ((BN_ == 96) ? 2 : NWN), WGM = (2 * NWN / KS) / WGN;
    constexpr int WTM = 128 / WGM, WTN = BN_ / WGN, NFM = WTM / 16, NFN = WTN / 16;
    const int wk = (KS == 2) ? wave / (WGM * WGN) : 0;
    const int wm = (wave % (WGM * WGN)) / WGN, wn = wave % WGN;
    const int m0 = tile_m * BM, n0 = tile_n * BN_;

    f32x4 acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Bias gradient riding on the weight-gradient GEMM: rowsum[m] = sum_k A[k][m] = (A^T . 1)[m].  The first column of
    // workgroups multiplies every A fragment once more, against an all-ones operand; every column of that 16x16 result
    // holds the row sums.  (Only the wn == 0 waves do it, so each row is produced once per K-half.)
    // (RS is a compile-time switch: the extra accumulators would push the K-split layout over its register budget.)
    const bool do_rowsum = RS && epi.rowsum_col >= 0 && tile_n == 0 && wn == 0;
    f32x4 accr[RS ? NFM : 1];
#pragma unroll
    for (int i = 0; i < (RS ? NFM : 1); ++i) accr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

    const int NB = (BN_ == 128) ? N : min(N, n0 + BN_);   // B rows / columns beyond the tile are never fetched
    Stage<AT, 1024 / NTH> sa;
    Stage<BT, 1024 / NTH> sb;
    const int nk_all = (K + BK - 1) / BK;
    const int kt0 = (int)((long)nk_all * split / splits), kt1 = (int)((long)nk_all * (split + 1) / splits);
    const int nk = kt1 - kt0;

    // prologue: tile kt0 -> stage 0
    if (A_DMA) stage_dma<A_KMAJOR, RAGGED, NTH>(reinterpret_cast<const bf16*>(A), lda, m0, kt0 * BK, M, K, smem, tid);
    else { stage_load<AT, A_KMAJOR, RAGGED, NTH>(sa, A, lda, m0, kt0 * BK, M, K, tid); stage_store<AT, A_KMAJOR, NTH>(sa, smem, tid); }
    if (B_DMA) stage_dma<B_KMAJOR, RAGGED || BN_ != 128, NTH>(reinterpret_cast<const bf16*>(B), ldb, n0, kt0 * BK, NB, K, smem + OPER_BYTES, tid);
    else { stage_load<BT, B_KMAJOR, RAGGED, NTH>(sb, B, ldb, n0, kt0 * BK, N, K, tid); stage_store<BT, B_KMAJOR, NTH>(sb, smem + OPER_BYTES, tid); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (probing) pt[1] = probe_now();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const unsigned char* la = smem + cur * 2 * OPER_BYTES;
        const unsigned char* lb = la + OPER_BYTES;
        unsigned char* na = smem + (cur ^ 1) * 2 * OPER_BYTES;
        // Prefetch the next K-tile while this one is multiplied (the last iteration re-fetches its own
        // tile into the idle buffer so the loop body stays branch-free).  bf16 operands go straight to
        // LDS by DMA; fp32 operands are converted in registers and written after the MFMAs.
        const int kn = kt0 + ((kt + 1 < nk) ? kt + 1 : kt);
        if (!(dbg & 1)) {
        if (A_DMA) stage_dma<A_KMAJOR, RAGGED, NTH>(reinterpret_cast<const bf16*>(A), lda, m0, kn * BK, M, K, na, tid);
        else stage_load<AT, A_KMAJOR, RAGGED, NTH>(sa, A, lda, m0, kn * BK, M, K, tid);
        if (B_DMA) stage_dma<B_KMAJOR, RAGGED || BN_ != 128, NTH>(reinterpret_cast<const bf16*>(B), ldb, n0, kn * BK, NB, K, na + OPER_BYTES, tid);
        else stage_load<BT, B_KMAJOR, RAGGED, NTH>(sb, B, ldb, n0, kn * BK, N, K, tid);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch issue ahead of the MFMAs
        if (!(dbg & 2))
#pragma unroll
        for (int k2 = 0; k2 < 2 / KS; ++k2) {
            const int kk = (KS == 2) ? wk : k2;
            bf16x8 fa[NFM], fb[NFN];
#pragma unroll
            for (int f = 0; f < NFM; ++f) fa[f] = read_frag<A_KMAJOR>(la, wm * WTM, f, kk, lane);
#pragma unroll
            for (int f = 0; f < NFN; ++f) fb[f] = read_frag<B_KMAJOR>(lb, wn * WTN, f, kk, lane);
#pragma unroll
            for (int i = 0; i < NFM; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            if (RS && do_rowsum) {
#pragma unroll
                for (int i = 0; i < NFM; ++i) accr[RS ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], accr[RS ? i : 0], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!A_DMA) stage_store<AT, A_KMAJOR, NTH>(sa, na, tid);
        if (!B_DMA) stage_store<BT, B_KMAJOR, NTH>(sb, na + OPER_BYTES, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(dbg & 4)) __syncthreads();
    }

    if (probing) pt[2] = probe_now();
    if (dbg & 8) {   // debugging: keep the accumulators alive but skip the epilogue
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int j = 0; j < NFN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // Epilogue.  The MFMA D fragment gives a lane 4 consecutive n of one m; dumped as-is the global accesses would be
    // 32-byte segments scattered over 16 rows.  Stage the fp32 tile through the (now idle) LDS stages instead and walk
    // it row-wise: every thread then owns 8 consecutive columns of one row, so bias / residual / saved-gelu' loads and
    // the output stores are full-line, 16-byte-per-lane transactions.
    constexpr int CLD = BN_ + 4;   // +4 floats: 16 rows of a fragment column land on distinct banks
    float* cs = reinterpret_cast<float*>(smem);
    float* rs = cs + BM * CLD;     // [BM] row sums (bias-gradient partials), behind the C stage
    if (RS && do_rowsum && wk == 0 && lane < 16) {
#pragma unroll
        for (int i = 0; i < NFM; ++i) rs[wm * WTM + i * 16 + lane] = accr[RS ? i : 0][0];
    }
    if (wk == 0) {
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int j = 0; j < NFN; ++j) {
                const int row = wm * WTM + i * 16 + (lane & 15), col = wn * WTN + j * 16 + (lane >> 4) * 4;
                *reinterpret_cast<float4*>(cs + row * CLD + col) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
    }
    __syncthreads();
    if (KS == 2) {
        if (RS && do_rowsum && wk == 1 && lane < 16) {
#pragma unroll
            for (int i = 0; i < NFM; ++i) rs[wm * WTM + i * 16 + lane] += accr[RS ? i : 0][0];
        }
        if (wk == 1) {   // add the second K-half in place (each (wm, wn) region has exactly one writer per phase)
#pragma unroll
            for (int i = 0; i < NFM; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j) {
                    const int row = wm * WTM + i * 16 + (lane & 15), col = wn * WTN + j * 16 + (lane >> 4) * 4;
                    float4* pc = reinterpret_cast<float4*>(cs + row * CLD + col);
                    const float4 v = *pc;
                    *pc = make_float4(v.x + acc[i][j][0], v.y + acc[i][j][1], v.z + acc[i][j][2], v.w + acc[i][j][3]);
                }
        }
        __syncthreads();
    }
    if (probing) pt[3] = probe_now();
    constexpr int SEG = BN_ / 8;
    if (epilogue_fast_ok(epi) && splits <= 1) {
        // the lean epilogue of the common cases (bias, GELU + saved derivative / multiplier / dropout / residual, bf16 output): no per-column
        // range flags, one 16-byte side-input load
        const uint32_t dkey = drop_key(epi.drop);
        const bf16* sidep = epi.resid ? epi.resid : (epi.act == 2 ? epi.aux : nullptr);
        const int side_ld = epi.resid ? epi.ldr : epi.ldc;
        for (int idx = tid; idx < BM * SEG; idx += NTH) {
            const int row = idx / SEG, seg = idx - row * SEG;
            const int m = m0 + row, n = n0 + seg * 8;
            if (m >= M || n >= N) continue;
            uint4 side = make_uint4(0, 0, 0, 0);
            if (sidep) side = *reinterpret_cast<const uint4*>(sidep + (size_t)m * side_ld + n);
            epilogue8_fast(epi, m, n, load_f8(cs + row * CLD + seg * 8), side, dkey);
        }
    } else {
        for (int idx = tid; idx < BM * SEG; idx += NTH) {
            const int row = idx / SEG, seg = idx - row * SEG;
            epilogue8(epi, m0 + row, n0 + seg * 8, load_f8(cs + row * CLD + seg * 8), split);
        }
    }
    if (RS && epi.rowsum_col >= 0 && tile_n == 0 && tid < BM && m0 + tid < M) {
        if (epi.rowsum_direct) epi.rowsum_direct[m0 + tid] = rs[tid];     // no split-K: this tile saw the whole reduction
        else reinterpret_cast<float*>(epi.C)[(size_t)splits * epi.slab_stride + (size_t)split * M + m0 + tid] = rs[tid];
    }
    if (probing) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the record's last stamp includes the drain of this wave's stores
        pt[4] = probe_now();
        if (tid == 0) {
            const unsigned slot = atomicAdd(reinterpret_cast<unsigned*>(pr.buf), 1u);
            if (slot < pr.cap) {
                unsigned long long* r = pr.buf + 8 * (size_t)(slot + 1);
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                r[0] = ((unsigned long long)pr.launch << 32) | blockIdx.x;
                r[1] = ((unsigned long long)xcc << 32) | hw;
                r[2] = pt[0]; r[3] = pt[1]; r[4] = pt[2]; r[5] = pt[3]; r[6] = pt[4];
                r[7] = ((unsigned long long)(unsigned)tile_m << 32) | (unsigned)tile_n | ((unsigned long long)(unsigned)split << 56);
            }
        }
    }
}

// XCD-aware tile order shared by both entry points: block b runs on XCD b % 8; every XCD gets a contiguous run of the tile list.
DEVI int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
// 8-row super-rows, column-major inside: the ~64 tiles resident on one XCD share 8 A row-panels and a few B
// panels that fit its 4 MiB L2 (the weight panel is then re-read once per super-row, not once per row).
DEVI void super_row_tile(int bid, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    const int per_sr = 8 * tiles_n;
    const int sr = bid / per_sr, rem = bid - sr * per_sr;
    const int h = min(8, tiles_m - sr * 8);
    tile_n = rem / h;
    tile_m = sr * 8 + (rem - tile_n * h);
}

template <typename AT, typename BT, bool A_KMAJOR, bool B_KMAJOR, bool RAGGED, int NWN, int BN_, int KS = 1, bool RS = false>
__global__ __launch_bounds__(128 * NWN, NWN) void gemm_bf16_kernel(const AT* __restrict__ A, const BT* __restrict__ B,
                                                             int M, int N, int K, int lda, int ldb,
                                                             int tiles_m, int tiles_n, int splits, int dbg, EpiArgs epi, Probe pr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ntile = tiles_m * tiles_n;
    int bid = xcd_remap(blockIdx.x, ntile * splits);
    const int split = bid / ntile;
    bid -= split * ntile;
    int tile_m, tile_n;
    super_row_tile(bid, tiles_m, tiles_n, tile_m, tile_n);
    gemm_tile<AT, BT, A_KMAJOR, B_KMAJOR, RAGGED, NWN, BN_, KS, RS>(A, B, M, N, K, lda, ldb, tile_m, tile_n, split, splits, dbg, epi, smem, pr);
}

// ---- grouped launch: several independent problems of one operand layout in ONE grid --------------------------------------
// The tile lists of the problems are concatenated (each in its own super-row order) and the XCD remap runs over the whole
// list.  Used for the weight gradients of a transformer layer (four GEMMs with 36 - 144 output tiles each and a 7296-long
// reduction): together they fill the 512 workgroup slots of the chip in one round WITHOUT split-K, so the fp32 slabs, their
// reduction kernels and three launch boundaries per layer disappear.
constexpr int MAXG = 8;
struct GroupProblem {
    const void* A; const void* B;
    int M, N, K, lda, ldb, tiles_m, tiles_n;
    EpiArgs epi;
};
struct GroupArgs {
    int count, total;
    int start[MAXG + 1];
    GroupProblem p[MAXG];
};

template <typename AT, typename BT, bool A_KMAJOR, bool B_KMAJOR, bool RAGGED, int NWN, int BN_, int KS = 1, bool RS = false>
__global__ __launch_bounds__(128 * NWN, NWN) void gemm_bf16_grouped_kernel(GroupArgs g, Probe pr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid = xcd_remap(blockIdx.x, g.total);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MAXG; ++i) gi += (i < g.count && bid >= g.start[i]) ? 1 : 0;
    const GroupProblem& P = g.p[gi];
    bid -= g.start[gi];
    int tile_m, tile_n;
    super_row_tile(bid, P.tiles_m, P.tiles_n, tile_m, tile_n);
    gemm_tile<AT, BT, A_KMAJOR, B_KMAJOR, RAGGED, NWN, BN_, KS, RS>(reinterpret_cast<const AT*>(P.A), reinterpret_cast<const BT*>(P.B), P.M, P.N, P.K,
                                                                  P.lda, P.ldb, tile_m, tile_n, 0, 1, 0, P.epi, smem, pr);
}

// The same grouping on the wide tile (gemm_wide.h, one workgroup per CU, ping-pong K-loop on a three-stage LDS-DMA ring): the weight
// gradients of a layer are 114 K-steps long, so the K-loop rate is all that matters, and the 128-row kernel's two workgroups per CU
// stage 64 FLOP per byte against 85 for 256 x 128.  216 tiles for the VisualBERT layer (72 + 72 + 54 + 18): one round on 256 CUs.
template <int BM_, int BN_, int WGM, int WGN, bool AKM, bool BKM, bool RS>
__global__ __launch_bounds__(512, 2) void gemm_wide_grouped_kernel(GroupArgs g, Probe pr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid = wide_xcd_remap(blockIdx.x, g.total);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MAXG; ++i) gi += (i < g.count && bid >= g.start[i]) ? 1 : 0;
    const GroupProblem& P = g.p[gi];
    bid -= g.start[gi];
    int tile_m, tile_n;
    wide_super_row(bid, P.tiles_m, P.tiles_n, tile_m, tile_n);
    wide_tile<BM_, BN_, WGM, WGN, 3, false, 0, AKM, BKM, RS>(reinterpret_cast<const bf16*>(P.A), reinterpret_cast<const bf16*>(P.B), P.M, P.N, P.K, P.lda,
                                                               P.ldb, tile_m, tile_n, P.epi, pr, smem);
}

// The grouped weight gradients with a LayerNorm-backward RIDER (round 6): a VisualBERT layer's 216 tiles run one per CU (147 KB of LDS each) and leave 40 of the 256 CUs
// idle for the whole launch; the kernel that follows in the step — the first LayerNorm backward of the layer BELOW, HBM-bound, 45 MB — depends on nothing
// the tiles compute.  Workgroups g.total .. g.total + nrider - 1 of this launch are that LayerNorm backward: a rider workgroup is two independent 256-thread
// halves, each walking the blocks (blk = 2 * rider + half + 2 * nrider * k) of the SAME block decomposition ln_bwd_h_kernel uses (lnk::ln_bwd_h_block: same
// rows per half-wave, same partials layout, bit-identical results); both halves run the same number of iterations so that the workgroup-wide barriers match.
// Riders carry the highest block indices: they are dispatched after the tiles, onto the CUs still free (a tile's LDS leaves no room for one beside it).
struct LnRider {
    const bf16* dy; const bf16* x; const float* mean; const float* rstd; const float* gamma;
    bf16* dx; bf16* dlin; DropoutCfg drop; float* partials; int rows, nblk, nrider;
};
template <int BM_, int BN_, int WGM, int WGN, bool AKM, bool BKM, bool RS>
__global__ __launch_bounds__(512) void gemm_wide_grouped_ln_kernel(GroupArgs g, LnRider r, Probe pr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= g.total) {
        const int rid = (int)blockIdx.x - g.total, half = (int)threadIdx.x >> 8;
        float* red = reinterpret_cast<float*>(smem) + half * LN_BWD_RED_FLOATS(3);
        const int per = 2 * r.nrider, iters = (r.nblk + per - 1) / per;
        for (int k = 0; k < iters; ++k) {
            const int blk = 2 * rid + half + k * per;
            // (streaming loads of dy / x: the rows pass the L2 in which the tiles share their operand panels - same process 6.988 -> 6.965 ms per step; streaming
            // stores as well 6.976: dlin is the next kernel's operand)
            lnk::ln_bwd_h_block<3, false, 2, false, 1>(r.dy, r.x, r.mean, r.rstd, r.gamma, r.dx, r.dlin, r.drop, r.partials, r.rows, DropoutCfg{0u, 0u, 1.f, nullptr},
                                                       (int)threadIdx.x & 255, blk, r.nblk, blk < r.nblk, red);
        }
        return;
    }
    int bid = wide_xcd_remap(blockIdx.x, g.total);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MAXG; ++i) gi += (i < g.count && bid >= g.start[i]) ? 1 : 0;
    const GroupProblem& P = g.p[gi];
    bid -= g.start[gi];
    int tile_m, tile_n;
    wide_super_row(bid, P.tiles_m, P.tiles_n, tile_m, tile_n);
    wide_tile<BM_, BN_, WGM, WGN, 3, false, 0, AKM, BKM, RS>(reinterpret_cast<const bf16*>(P.A), reinterpret_cast<const bf16*>(P.B), P.M, P.N, P.K, P.lda,
                                                               P.ldb, tile_m, tile_n, P.epi, pr, smem);
}

// C[m][n] = beta * C[m][n] + sum_s slab[s][m][n]   (N % 4 == 0).  Behind the `splits` slabs the workspace holds
// [splits][M] row-sum partials (bias gradient); the threads past the last float4 group sum those into rowsum[m].
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long n, int N, int M,
                                                             float* __restrict__ C, int ldc, float beta, float* __restrict__ rowsum) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = t * 4;
    if (i >= n) {
        const long m = t - n / 4;
        if (rowsum != nullptr && m < M) {
            const float* rp = ws + (long)splits * n;
            float a = rp[m];
            for (int s = 1; s < splits; ++s) a += rp[(long)s * M + m];
            rowsum[m] = a;
        }
        return;
    }
    f32x4 acc = load_f4(ws + i);
    for (int s = 1; s < splits; ++s) acc += load_f4(ws + (long)s * n + i);
    const long m = i / N;
    float* c = C + m * ldc + (i - m * N);
    if (beta != 0.f) acc += beta * load_f4(c);
    *reinterpret_cast<float4*>(c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// Skinny problems (M <= 64 rows: the classification heads, mmf/models/visual_bert.py:349-404, and their input gradients): a handful of
// 128-row tiles each walking the whole reduction leaves the chip idle (6 workgroups x 49 K-steps for the classifier's input gradient),
// so the K range is split over ~128 workgroups whatever the epilogue: the GEMM pass writes raw fp32 partial tiles (slab s = K-slice s,
// rows of round_up(N, 8) floats), this kernel adds the slabs in a fixed order and runs the ORIGINAL fused epilogue on the sums.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ ws, int splits, long slab, int n8, EpiArgs e) {
    const int seg = n8 >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= e.M * seg) return;
    const int m = idx / seg, n = (idx - m * seg) * 8;
    const float* p = ws + (size_t)m * n8 + n;
    f32x8 v = load_f8(p);
    for (int s = 1; s < splits; ++s) v += load_f8(p + (size_t)s * slab);
    epilogue8(e, m, n, v, 0);
}

// name of the kernel family the last mmf_gemm_bf16 / mmf_gemm_bf16_grouped call on this thread launched (measurement aid)
static thread_local const char* g_last_kernel = "";

// timeline probe state (host): set by mmf_gemm_set_probe, consumed by every GEMM launch while set
static Probe g_probe = {nullptr, 0u, 0u};
// host-side log of the probed launches: {launch id, layout (bit 0 A k-major, bit 1 B k-major, 4 = grouped), M, N, K} per launch
static int64_t g_probe_log[4096][5];
static int g_probe_nlog = 0;
static int g_note[4] = {0, 0, 0, 0};        // layout, M, N, K of the call being launched (set by the entry points while probing)
static Probe next_probe() {
    Probe p = g_probe;
    if (p.buf) {
        if (g_probe_nlog < 4096) {
            int64_t* r = g_probe_log[g_probe_nlog++];
            r[0] = p.launch; r[1] = g_note[0]; r[2] = g_note[1]; r[3] = g_note[2]; r[4] = g_note[3];
        }
        g_probe.launch++;
    }
    return p;
}

template <typename AT, typename BT, bool AK, bool BK_, bool RG, int NWN, int BN_, int KS = 1, bool RS = false>
int launch_n(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    const int tm = (d->M + BM - 1) / BM, tn = (d->N + BN_ - 1) / BN_;
    const int splits = e.splits > 1 ? e.splits : 1;
    const EpiArgs& e2 = e;
    constexpr int cstage = BM * (BN_ + 4) * (int)sizeof(float) + BM * (int)sizeof(float);   // C stage + row sums
    constexpr int lds_bytes = cstage > 4 * OPER_BYTES ? cstage : 4 * OPER_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<AT, BT, AK, BK_, RG, NWN, BN_, KS, RS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (ae != hipSuccess) { mmf_amd_set_error(hipGetErrorString(ae)); return 2; }
        attr_set = true;
    }
    g_last_kernel = BN_ == 96 ? (KS == 2 ? "gemm_bf16_kernel 128x96 ksplit" : "gemm_bf16_kernel 128x96") : (KS == 2 ? "gemm_bf16_kernel 128x128 ksplit" : "gemm_bf16_kernel 128x128");
    hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, AK, BK_, RG, NWN, BN_, KS, RS>), dim3(tm * tn * splits), dim3(128 * NWN), lds_bytes, s,
                       reinterpret_cast<const AT*>(d->A), reinterpret_cast<const BT*>(d->B), d->M, d->N, d->K,
                       d->lda, d->ldb, tm, tn, splits, (d->debug_flags >> 4) & 15, e2, next_probe());
    MMF_CHECK_LAUNCH();
    return 0;
}

template <typename AT, typename BT, bool AK, bool BK_, bool RG, int NWN, int BN_, int KS = 1, bool RS = false>
int launch_grouped_n(const GroupArgs& g, hipStream_t s) {
    constexpr int cstage = BM * (BN_ + 4) * (int)sizeof(float) + BM * (int)sizeof(float);
    constexpr int lds_bytes = cstage > 4 * OPER_BYTES ? cstage : 4 * OPER_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_grouped_kernel<AT, BT, AK, BK_, RG, NWN, BN_, KS, RS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (ae != hipSuccess) { mmf_amd_set_error(hipGetErrorString(ae)); return 2; }
        attr_set = true;
    }
    g_last_kernel = "gemm_bf16_grouped_kernel 128x128";
    hipLaunchKernelGGL((gemm_bf16_grouped_kernel<AT, BT, AK, BK_, RG, NWN, BN_, KS, RS>), dim3(g.total), dim3(128 * NWN), lds_bytes, s, g, next_probe());
    MMF_CHECK_LAUNCH();
    return 0;
}


template <int BM_, int BN_, int WGM, int WGN, bool AKM, bool BKM, bool RS>
int launch_wide_grouped(const GroupArgs& g, hipStream_t s) {
    constexpr int lds_bytes = 3 * (BM_ + BN_) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_grouped_kernel<BM_, BN_, WGM, WGN, AKM, BKM, RS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (ae != hipSuccess) { mmf_amd_set_error(hipGetErrorString(ae)); return 2; }
        attr_set = true;
    }
    g_last_kernel = "gemm_wide_grouped_kernel 256x128";
    hipLaunchKernelGGL((gemm_wide_grouped_kernel<BM_, BN_, WGM, WGN, AKM, BKM, RS>), dim3(g.total), dim3(512), lds_bytes, s, g, next_probe());
    MMF_CHECK_LAUNCH();
    return 0;
}

template <bool RS>
int launch_wide_grouped_ln(const GroupArgs& g, const LnRider& r, hipStream_t s) {
    constexpr int lds_bytes = 3 * (256 + 128) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_grouped_ln_kernel<256, 128, 4, 2, true, true, RS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (ae != hipSuccess) { mmf_amd_set_error(hipGetErrorString(ae)); return 2; }
        attr_set = true;
    }
    g_last_kernel = "gemm_wide_grouped_ln_kernel 256x128";
    hipLaunchKernelGGL((gemm_wide_grouped_ln_kernel<256, 128, 4, 2, true, true, RS>), dim3(g.total + r.nrider), dim3(512), lds_bytes, s, g, r, next_probe());
    MMF_CHECK_LAUNCH();
    return 0;
}

// ---- wide tiles (gemm_wide.h): one workgroup per CU ---------------------------------------------------------------------
template <int BM_, int BN_, int WGM, int WGN, int NS>
int launch_wide(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    const int tm = (d->M + BM_ - 1) / BM_, tn = d->N / BN_;
    constexpr int lds_bytes = NS * (BM_ + BN_) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t a0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<BM_, BN_, WGM, WGN, NS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<BM_, BN_, WGM, WGN, NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (a0 != hipSuccess || a1 != hipSuccess) { mmf_amd_set_error(hipGetErrorString(a0 != hipSuccess ? a0 : a1)); return 2; }
        attr_set = true;
    }
    const bf16* A = reinterpret_cast<const bf16*>(d->A);
    const bf16* B = reinterpret_cast<const bf16*>(d->B);
    g_last_kernel = BM_ == 192 ? (BN_ == 96 ? "gemm_wide_kernel 192x96" : "gemm_wide_kernel 192x192") : (BM_ == 128 ? (BN_ == 96 ? "gemm_wide_kernel 128x96" : "gemm_wide_kernel 128x128") : (BN_ == 96 ? "gemm_wide_kernel 256x96" : "gemm_wide_kernel 256x128"));
#ifdef MMF_WIDE_ABLATE
    const int abl = (d->debug_flags >> 4) & 7;
#define MMF_WIDE_ABL_CASE(V)                                                                                                          \
    if (abl == V) {                                                                                                                   \
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<BM_, BN_, WGM, WGN, NS, true, V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
        hipLaunchKernelGGL((gemm_wide_kernel<BM_, BN_, WGM, WGN, NS, true, V>), dim3(tm * tn), dim3(512), lds_bytes, s, A, B, d->M, d->N, d->K, d->lda, d->ldb, tm, tn, e, next_probe()); \
        MMF_CHECK_LAUNCH();                                                                                                           \
        return 0;                                                                                                                     \
    }
    MMF_WIDE_ABL_CASE(1) MMF_WIDE_ABL_CASE(2) MMF_WIDE_ABL_CASE(3) MMF_WIDE_ABL_CASE(4) MMF_WIDE_ABL_CASE(5) MMF_WIDE_ABL_CASE(6) MMF_WIDE_ABL_CASE(7)
#undef MMF_WIDE_ABL_CASE
#endif
    if (d->M % BM_)
        hipLaunchKernelGGL((gemm_wide_kernel<BM_, BN_, WGM, WGN, NS, true>), dim3(tm * tn), dim3(512), lds_bytes, s, A, B, d->M, d->N, d->K, d->lda, d->ldb, tm, tn, e, next_probe());
    else
        hipLaunchKernelGGL((gemm_wide_kernel<BM_, BN_, WGM, WGN, NS, false>), dim3(tm * tn), dim3(512), lds_bytes, s, A, B, d->M, d->N, d->K, d->lda, d->ldb, tm, tn, e, next_probe());
    MMF_CHECK_LAUNCH();
    return 0;
}

// ---- persistent tiles (gemm_persist.h): one workgroup per CU walks its tiles, a tile's epilogue runs under the next tile's K-loop -------------
template <int BM_, int BN_, int WGM, int WGN, int EPI>
int launch_persist_e(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    const int tm = (d->M + BM_ - 1) / BM_, tn = d->N / BN_;
    const int lds_bytes = 3 * (BM_ + BN_) * 128 + d->N * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t a0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_persist_kernel<BM_, BN_, WGM, WGN, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (a0 != hipSuccess) { mmf_amd_set_error(hipGetErrorString(a0)); return 2; }
        attr_set = true;
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        if (cus <= 0) cus = 256;
    }
    const int ntile = tm * tn;
    int grid = (ntile + 7) / 8 * 8;
    if (grid > cus / 8 * 8) grid = cus / 8 * 8;
    g_last_kernel = BM_ == 192 ? "gemm_persist_kernel 192x192" : (BN_ == 96 ? "gemm_persist_kernel 256x96" : "gemm_persist_kernel 256x128");
    // The bf16 output is stored PLAIN (write-back) here unless a tunable asks otherwise: this kernel's stores trickle out under the K-loops instead of
    // arriving as one burst per round, and leaving them to the L2 / Infinity Cache hands the next kernel its input on chip.  In the step (same process,
    // profiles/r06_persist_experiments.txt): QKV forward 7.49 (nt) / 7.46 (sc1) / 7.35 (plain) ms per step; the one-burst kernels of gemm_wide.h lose
    // with plain stores (7.66 against 7.48).  The saved gelu' keeps its non-temporal stores (read a whole backward pass later).
    EpiArgs e2 = e;
    if (mmf_amd_get_tunable(MMF_TUN_EPI_NT) == 0 && mmf_amd_get_tunable(MMF_TUN_SC1_SITE) == 0) { e2.nt &= ~1; e2.sc1 &= ~1; }
    hipLaunchKernelGGL((gemm_persist_kernel<BM_, BN_, WGM, WGN, EPI>), dim3(grid), dim3(512), lds_bytes, s, reinterpret_cast<const bf16*>(d->A),
                       reinterpret_cast<const bf16*>(d->B), d->M, d->N, d->K, d->lda, d->ldb, tm, tn, e2);
    MMF_CHECK_LAUNCH();
    return 0;
}
// epilogue class of the persistent kernel, or -1 when it has none for this call
static int persist_epi(const EpiArgs& e) {
    if (!epilogue_fast_ok(e)) return -1;
    if (e.act == 1) return PEPI_GELU;
    if (e.act == 2 || e.resid) return PEPI_SIDE;
    if (e.drop.thr16) return -1;
    return PEPI_PLAIN;
}
template <int BM_, int BN_, int WGM, int WGN>
int launch_persist(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    switch (persist_epi(e)) {
        case PEPI_PLAIN: return launch_persist_e<BM_, BN_, WGM, WGN, PEPI_PLAIN>(d, e, s);
        case PEPI_GELU: return launch_persist_e<BM_, BN_, WGM, WGN, PEPI_GELU>(d, e, s);
        default:
            if constexpr (BM_ == 192) return 1;          // (192 x 192 with a side input does not fit the register file: never chosen, never built)
            else return launch_persist_e<BM_, BN_, WGM, WGN, PEPI_SIDE>(d, e, s);
    }
}
// 0: not on the persistent kernel; 1 / 2 / 3: on its 256 x 96 / 192 x 192 / 256 x 128 tile
static int persist_choice(const mmf_gemm_desc* d, const EpiArgs& e) {
    int t = mmf_amd_get_tunable(MMF_TUN_GEMM_PERSIST);
    if (t < 0 || (d->debug_flags & 131072) || persist_epi(e) < 0) return 0;
    bool by_rule = t == 0;
    if (by_rule && mmf_amd_get_tunable(MMF_TUN_GEMM_WIDE) != 0) return 0;       // (a forced one-tile kernel is an A/B of THAT kernel)
    if (t >= 256) {       // 256 + mask: exactly the calls tagged with one of these sites, on the model's tile (per-site A/B inside the step)
        const int site = (d->debug_flags >> 20) & 15;
        if (site == 0 || !(((t - 256) >> site) & 1)) return 0;
        t = 0;
    }
    static const int BNs[4] = {0, 96, 192, 128};
    const long maxld = (long)(d->lda > d->ldc ? d->lda : d->ldc) > (long)d->ldr ? (long)(d->lda > d->ldc ? d->lda : d->ldc) : (long)d->ldr;
    if ((d->K % 64) != 0 || (d->M % 64) != 0 || d->M < 512 || (long)d->M * maxld * 2 >= (1L << 31) || (long)d->N * d->ldb * 2 >= (1L << 31)) return 0;
    const int pe = persist_epi(e);
    auto fits = [&](int c) { return !(c == 2 && pe == PEPI_SIDE) && (d->N % BNs[c]) == 0 && d->K / 64 >= 12 && 3 * ((c == 2 ? 192 : 256) + BNs[c]) * 128 + d->N * 4 <= 163840; };
    if (t >= 1 && t <= 3) return fits(t) ? t : 0;
    // rounds of the 32 workgroups of an XCD x K-steps x the measured step time of the tile's K-loop (profiles/r02_wide_gemm_ablation.txt)
    static const double step_us[4] = {0, 0.68, 0.81, 0.79};
    static const int BMs[4] = {0, 256, 192, 256};
    int pick = 0;
    double best = 1e30;
    long pick_rounds = 0;
    for (int c = 1; c <= 3; ++c) {
        if (!fits(c)) continue;
        const long tiles = (long)((d->M + BMs[c] - 1) / BMs[c]) * (d->N / BNs[c]);
        const long rounds = ((tiles + 7) / 8 + 31) / 32;
        const double us = rounds * (d->K / 64) * step_us[c];
        if (us < best) { best = us; pick = c; pick_rounds = rounds; }
    }
    // Where it pays (profiles/r06_persist_experiments.txt): only where a workgroup walks SEVERAL tiles with SHORT K-loops - there the ring that never
    // drains and the epilogue under the next K-loop replace a prologue and an epilogue per tile.  One tile per workgroup has nothing to overlap with
    // (its deferred units then cost more than the LDS-staged row-wise epilogue of gemm_wide.h: 47 vs 39 us at N = 768, K = 3072).
    if (by_rule && (pick_rounds < 2 || d->K / 64 > 24)) return 0;
    return pick;
}

// Modelled launch time (us) of a tile shape: rounds x (K-steps x max(staging, MFMA) + fixed), with the measured per-CU staging rate
// (85 GB/s, profiles/r02_lds_dma_ceiling.txt) and 90 % of the per-CU MFMA peak; `slots` = workgroups resident per CU.
// Round 3: the K-step is max(staging, MFMA) + half the smaller one — what the ablations of the shipped kernels show
// (profiles/r02_wide_gemm_ablation.txt: DMA alone 0.46, MFMA alone 0.38 / 0.58 / 0.52, together 0.63 / 0.81 / 0.79 us per step for
// 256x96 / 192x192 / 256x128) — with the measured rates 100 GB/s per CU and 8.1 TFLOP/s per CU (the sustained clock, not 2.4 GHz).
static double tile_cost(long M, long N, long K, int bm, int bn, int slots) {
    const long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const double rounds = (double)((tiles + 256L * slots - 1) / (256L * slots));
    const double stage_us = (double)(bm + bn) * 128.0 * slots / 100e3;
    const double mfma_us = 2.0 * bm * bn * 64.0 * slots / 8.1e6;
    const double hi = stage_us > mfma_us ? stage_us : mfma_us, lo = stage_us > mfma_us ? mfma_us : stage_us;
    return rounds * ((double)((K + 63) / 64) * (hi + 0.5 * lo) + 5.0);
}
// 0: keep the 128-row kernel; 1: 256 x 96, 2: 192 x 192, 3: 256 x 128.
static int wide_choice(const mmf_gemm_desc* d) {
    const int force = mmf_amd_get_tunable(MMF_TUN_GEMM_WIDE);
    if (force < 0 || (d->debug_flags & 131072) || d->M < 512 || (d->K % 64) != 0 || d->K < 128) return 0;
    static const int BMs[7] = {0, 256, 192, 256, 128, 128, 192}, BNs[7] = {0, 96, 192, 128, 96, 128, 96};
    if (force >= 1 && force <= 6) return (d->N % BNs[force]) == 0 ? force : 0;
    // Measured INSIDE the step (round 4, tools/step_ab.py, profiles/r04_in_step_choices.txt): the FFN-down dgrad (N = 3072, K = 768, times the saved gelu')
    // on the 256 x 96 tile, 928 tiles in 3.6 rounds, instead of the 256 x 128 tile the isolated measurements and the cost model pick (44.5 us isolated,
    // 54 - 60 us in the step): 7.63 against 7.89, 7.53 against 7.85 and 7.64 against 7.86 ms per step on three boxes. 
    if (d->act == 2 && (d->N % 96) == 0 && d->N >= 2304 && d->K <= 1024) return 1;
    // One workgroup per CU cannot hide a heavy epilogue behind a co-resident workgroup's K loop: the GELU up-projection (two
    // bf16 outputs, erf + exp per element) measured 70.7 us with wide tiles against 60.5 us inside the training step.
    if (d->act == 1) return 0;
    // Few token rows (a batch whose text padding was trimmed, mmf_amd/common/prefetch.py: 32 x 124 = 3968 rows): the 256-row tiles of an N = 768 site
    // fill half the chip (16 x 8 = 128 tiles).  The same ping-pong kernel on a 128 x 96 tile puts a workgroup on (nearly) every CU: out-proj 14.8 ->
    // 11.7 us, FFN-down 34.9 -> 29.2, FFN-up dgrad 30.8 -> 25.8, QKV dgrad 23.9 -> 20.2 at M = 3968 (tools/gemm_ab.py --M 3968 --tun 2:0,1,4); it
    // loses as soon as its tiles need a second round (M = 5248: 41 x 8 tiles, 52.6 against 35.9 us) and on the wide outputs (N >= 2304).
    if ((d->N % 96) == 0 && d->N <= 1024 && d->act != 1 && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 16)) {
        const long t128 = (long)((d->M + 127) / 128) * (d->N / 96), t256 = (long)((d->M + 255) / 256) * (d->N / 96);
        if (t128 <= 256 && t256 <= 160) return 4;
        // between the buckets (4224 .. 6144 rows: a 128-row tiling needs a second round) the 192 x 96 tile still fits one round and beats 256 x 96 by 9 - 10 % at every
        // site (M = 4224 / 5248 / 6144: FFN-down 35.1 -> 31.9 / 36.4 -> 33.0 / 39.4 -> 35.2 us, tools/gemm_ab.py --M .. --tun 2:1,6); 7296 rows: 38 x 8 tiles, a second round.
        const long t192 = (long)((d->M + 191) / 192) * (d->N / 96);
        if (t192 <= 256 && d->M >= 1024) return 6;
    }
    // ... and on a 128 x 128 tile where N is a multiple of 128 only (ViLBERT's 1024-wide visual stream and connection layers at 3200 / 4096 rows: 104 / 128
    // tiles of 256 x 128 on 256 CUs): out-proj 19.0 -> 14.5 us, FFN-down 18.6 -> 14.1, dgrads 15.8 -> 12.6 and 40.3 -> 33.5 (K = 4096) at M = 3200.
    if ((d->N % 128) == 0 && (d->N % 96) != 0 && d->N <= 1024 && d->act != 1 && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 16)) {
        const long t128 = (long)((d->M + 127) / 128) * (d->N / 128), t256 = (long)((d->M + 255) / 256) * (d->N / 128);
        if (t128 <= 256 && t256 <= 160) return 5;
    }
    double best = tile_cost(d->M, d->N, d->K, 128, 128, 2);
    if ((d->N % 96) == 0) { const double c = tile_cost(d->M, d->N, d->K, 128, 96, 2); if (c < best) best = c; }
    int pick = 0;
    for (int c = 1; c <= 3; ++c) {
        if (d->N % BNs[c]) continue;
        const double t = tile_cost(d->M, d->N, d->K, BMs[c], BNs[c], 1);
        if (t < best * 0.97) { best = t; pick = c; }
    }
    return pick;
}

template <typename AT, typename BT, bool AK, bool BK_, bool RG>
int launch(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    if constexpr (!AK && !BK_ && is_bf16<AT>::value && is_bf16<BT>::value) {
        if (e.splits <= 1) {
            switch (persist_choice(d, e)) {
                case 1: return launch_persist<256, 96, 4, 2>(d, e, s);
                case 2: return launch_persist<192, 192, 2, 4>(d, e, s);
                case 3: return launch_persist<256, 128, 4, 2>(d, e, s);
                default: break;
            }
            const int wc = wide_choice(d);
            switch (wc) {
                case 1: return launch_wide<256, 96, 4, 2, 3>(d, e, s);
                case 2: return launch_wide<192, 192, 2, 4, 3>(d, e, s);
                case 3: return launch_wide<256, 128, 4, 2, 3>(d, e, s);
                case 4: return launch_wide<128, 96, 4, 2, 3>(d, e, s);
                case 5: return launch_wide<128, 128, 4, 2, 3>(d, e, s);
                case 6: return launch_wide<192, 96, 4, 2, 3>(d, e, s);
                default: break;
            }
        }
    }
    // 8 waves (4 per SIMD with two workgroups per CU) hide LDS / MFMA-issue latency better than 4 waves of 64x64;
    // bit 8 of debug_flags selects the 4-wave form for A/B measurements.
    if (AK && BK_ && is_bf16<AT>::value && is_bf16<BT>::value && e.rowsum_col >= 0)   // weight gradient carrying the bias gradient
        return launch_n<AT, BT, AK, BK_, RG, 4, 128, 1, AK && BK_>(d, e, s);
    if (d->debug_flags & 256) return launch_n<AT, BT, AK, BK_, RG, 2, 128>(d, e, s);
    // Wave layout: 2x4 waves of 64x32 over both K-halves of a stage (KS = 1), or 2x2 waves of 64x64 times the two
    // K-halves (KS = 2: a third fewer LDS operand reads, one extra pass over the LDS C stage at the end).  Measured
    // (tools/micro_sweep.py gemm): KS = 2 wins 1-5 % once a workgroup runs >= 24 K-steps (FFN-down forward, the long
    // dgrads, the split weight gradients) and loses 3-14 % on the 12-step K = 768 GEMMs.  Bit 12 forces it, bit 13 forbids.
    const int splits = e.splits > 1 ? e.splits : 1;
    const int ksteps = (d->K + BK - 1) / BK / splits;
    const bool ks2 = (d->debug_flags & 4096) || (ksteps >= 24 && !(d->debug_flags & 8192));
    // 96-wide tiles when they need fewer rounds of the 512 workgroup slots (256 CUs x 2) than 128-wide ones
    if (!RG && !AK && is_bf16<AT>::value && is_bf16<BT>::value && (d->N % 96) == 0 && !(d->debug_flags & 512)) {
        const long tm = d->M / BM;
        const long r128 = (tm * ((d->N + 127) / 128) + 511) / 512, r96 = (tm * (d->N / 96) + 511) / 512;
        if (r96 * 96 < r128 * 128)
            return ks2 ? launch_n<AT, BT, AK, BK_, false, 4, 96, 2>(d, e, s) : launch_n<AT, BT, AK, BK_, false, 4, 96>(d, e, s);
    }
    return ks2 ? launch_n<AT, BT, AK, BK_, RG, 4, 128, 2>(d, e, s) : launch_n<AT, BT, AK, BK_, RG, 4, 128>(d, e, s);
}

}  // namespace

static int check_and_fill(const mmf_gemm_desc* d, EpiArgs& e) {
    MMF_CHECK_ARG(d && d->A && d->B && d->C, "mmf_gemm_bf16: null operand");
    MMF_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "mmf_gemm_bf16: empty shape");
    MMF_CHECK_ARG((d->lda % 8) == 0 && (d->ldb % 8) == 0, "mmf_gemm_bf16: lda/ldb must be multiples of 8 elements");
    // row operands are fetched in 8-element chunks along K: the last chunk may run into the row padding
    // (which must hold finite values, normally zeros), so the leading dimension has to cover it.
    const int k8 = (d->K + 7) / 8 * 8;
    MMF_CHECK_ARG(d->a_kmajor || d->lda >= k8, "mmf_gemm_bf16: lda must cover round_up(K, 8) for a row operand");
    MMF_CHECK_ARG(d->b_kmajor || d->ldb >= k8, "mmf_gemm_bf16: ldb must cover round_up(K, 8) for a row operand");
    MMF_CHECK_ARG(!(d->a_f32 && d->b_f32), "mmf_gemm_bf16: at most one fp32 operand");
    MMF_CHECK_ARG((d->act != 2 && d->act != 4) || d->aux, "mmf_gemm_bf16: act=2/4 needs aux");
    MMF_CHECK_ARG(d->act >= 0 && d->act <= 4, "mmf_gemm_bf16: unknown act");
    MMF_CHECK_ARG(!d->rowtab || d->rowidx, "mmf_gemm_bf16: rowtab needs rowidx");
    e.C = d->C; e.ldc = d->ldc; e.out_f32 = d->out_f32; e.beta = d->beta;
    e.bias = d->bias; e.coladd = d->coladd; e.rowtab = d->rowtab; e.rowidx = d->rowidx; e.rowtab_ld = d->rowtab_ld;
    e.act = d->act; e.U = reinterpret_cast<bf16*>(d->U); e.aux = reinterpret_cast<const bf16*>(d->aux);
    e.resid = reinterpret_cast<const bf16*>(d->resid); e.ldr = d->ldr;
    e.drop.key = d->drop_key; e.drop.thr16 = d->drop_thr16; e.drop.scale = d->drop_scale; e.drop.seed = d->drop_seed;
    e.grp_in = d->grp_in; e.grp_pad = d->grp_pad; e.grp_off = d->grp_off;
    e.M = d->M; e.N = d->N;
    e.slab_stride = 0; e.splits = 1; e.rowsum_col = -1; e.rowsum_direct = nullptr;
    // MMF_TUN_EPI_NT: 0 = the default mask below, else (value - 1) is the mask (1 = no non-temporal stores at all)
    const int ntt = mmf_amd_get_tunable(MMF_TUN_EPI_NT);
    e.nt = ntt > 0 ? ntt - 1 : MMF_EPI_NT_DEFAULT;
    e.sc1 = 0;
    // call-site exception (MMF_TUN_NT_SITE_KEEP, default 0 = none): the tagged call's bf16 output is read by the very next kernel
    const int site = (d->debug_flags >> 20) & 15;
    if (site != 0 && ((mmf_amd_get_tunable(MMF_TUN_NT_SITE_KEEP) >> site) & 1)) e.nt &= ~1;
    // write-through instead of non-temporal for the outputs whose consumer is a long-K GEMM (MMF_TUN_SC1_SITE; measured in the step, round 4)
    int sc1_sites = mmf_amd_get_tunable(MMF_TUN_SC1_SITE);
    if (sc1_sites == 0) sc1_sites = MMF_SITE_SC1_DEFAULT;
    if (site != 0 && ((sc1_sites >> site) & 1)) { e.nt &= ~1; e.sc1 |= 1; }
    return 0;
}

extern "C" int mmf_gemm_bf16(const mmf_gemm_desc* d, void* stream) {
    EpiArgs e;
    if (int rc = check_and_fill(d, e)) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_probe.buf) { g_note[0] = (d->a_kmajor ? 1 : 0) | (d->b_kmajor ? 2 : 0); g_note[1] = d->M; g_note[2] = d->N; g_note[3] = d->K; }
    // Split-K: a weight-gradient GEMM has few output tiles (768x768 -> 36) and a long reduction (K = tokens).
    // With a workspace, the K range is spread over `splits` workgroups per tile; each writes an fp32 partial slab
    // and a second kernel sums the slabs in a fixed order (deterministic, no atomics).
    float* final_c = nullptr; int final_ldc = 0; float final_beta = 0.f;
    {
        const int sp = mmf_gemm_splitk_splits(d->M, d->N, d->K);
        const bool plain = d->out_f32 && !d->bias && !d->coladd && !d->rowtab && d->act == 0 && !d->resid && !d->drop_thr16 &&
                           d->grp_in == 0;
        if (plain && sp > 1 && d->splitk_ws && d->splitk_ws_bytes >= (long)sp * d->M * (d->N + 1) * (long)sizeof(float)) {
            e.splits = sp;
            e.slab_stride = (long)d->M * d->N;
            final_c = reinterpret_cast<float*>(d->C); final_ldc = d->ldc; final_beta = d->beta;
            e.C = d->splitk_ws; e.ldc = d->N; e.beta = 0.f;
        }
    }
    // skinny path: any epilogue, applied by splitk_epilogue_kernel after the slab sum
    bool skinny = false;
    EpiArgs e_final;
    int n8 = 0;
    if (!final_c && !d->rowsum_out) {
        const int sp = mmf_gemm_skinny_splits(d->M, d->N, d->K, d->a_kmajor);
        n8 = (d->N + 7) / 8 * 8;
        if (sp > 1 && d->splitk_ws && d->splitk_ws_bytes >= (long)sp * d->M * n8 * (long)sizeof(float)) {
            skinny = true;
            e_final = e;
            e = EpiArgs{};
            e.C = d->splitk_ws; e.ldc = n8; e.out_f32 = 1; e.M = d->M; e.N = d->N;
            e.splits = sp; e.slab_stride = (long)d->M * n8; e.rowsum_col = -1;
        }
    }
    if (d->rowsum_out) {
        MMF_CHECK_ARG(d->a_kmajor && d->b_kmajor && !d->a_f32 && !d->b_f32,
                      "mmf_gemm_bf16: rowsum_out needs the weight-gradient form (both operands k-major, bf16)");
        e.rowsum_col = 1;   // on; with split-K the partials live behind the slabs: ws[splits * M * N + split * M + m]
        if (!final_c) e.rowsum_direct = d->rowsum_out;   // one workgroup per tile sees the whole reduction: written directly
    }

    // ragged unless every tile is full and every chunk in range
    const bool ragged = (d->M % BM) || (d->N % BN) || (d->K % BK);
    const int key = (d->a_kmajor ? 1 : 0) | (d->b_kmajor ? 2 : 0) | (d->a_f32 ? 4 : 0) | (d->b_f32 ? 8 : 0);
    int rc = -1;
    if (rc < 0) switch (key) {
        case 0: rc = ragged ? launch<bf16, bf16, false, false, true>(d, e, s) : launch<bf16, bf16, false, false, false>(d, e, s); break;
        case 4: rc = ragged ? launch<float, bf16, false, false, true>(d, e, s) : launch<float, bf16, false, false, false>(d, e, s); break;
        case 2: rc = ragged ? launch<bf16, bf16, false, true, true>(d, e, s) : launch<bf16, bf16, false, true, false>(d, e, s); break;
        case 3: rc = ragged ? launch<bf16, bf16, true, true, true>(d, e, s) : launch<bf16, bf16, true, true, false>(d, e, s); break;
        case 11: rc = ragged ? launch<bf16, float, true, true, true>(d, e, s) : launch<bf16, float, true, true, false>(d, e, s); break;
        default:
            mmf_amd_set_error("mmf_gemm_bf16: unsupported operand layout combination");
            return 1;
    }
    if (rc != 0) return rc;
    if (skinny) {
        const int threads = d->M * (n8 / 8);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(d->splitk_ws), e.splits, e.slab_stride, n8, e_final);
        MMF_CHECK_LAUNCH();
        g_last_kernel = "gemm_bf16_kernel 128x128 skinny split-K";
        return 0;
    }
    if (final_c) {
        const long n = (long)d->M * d->N;
        const long threads = n / 4 + (d->rowsum_out ? d->M : 0);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(d->splitk_ws), e.splits, n, d->N, d->M, final_c, final_ldc, final_beta,
                           d->rowsum_out);
        MMF_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int mmf_gemm_splitk_splits(int M, int N, int K) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int nk_all = (K + BK - 1) / BK;
    if (tiles >= 512 || nk_all < 16 || (N % 4) != 0) return 1;
    // Two workgroups are resident per CU (512 slots).  Cost in k-tile units: rounds x k-tiles per workgroup, plus ~1.5
    // k-tiles per split for writing and re-reading one more fp32 slab (fitted to tools/micro_sweep.py on MI355X:
    // qkv 2304x768 -> 4, out 768x768 -> 8, ffn 3072x768 -> 3).
    int best = 1; double bc = 1e30;
    for (int sp = 1; sp <= 16 && sp * 8 <= nk_all; ++sp) {
        const double c = (double)((tiles * sp + 511) / 512) * ((nk_all + sp - 1) / sp) + 1.5 * sp;
        if (c < bc - 1e-9) { bc = c; best = sp; }
    }
    return best;
}

// K-slices of the skinny path (splitk_epilogue_kernel): 0 / 1 = not a skinny problem.  Row operand A with at most 64 rows, at least 24
// K-steps; slices of >= 2 K-steps, about 128 workgroups in all.  Workspace: splits * M * round_up(N, 8) floats (mmf_gemm_desc::splitk_ws).
extern "C" int mmf_gemm_skinny_splits(int M, int N, int K, int a_kmajor) {
    if (a_kmajor || M > 64) return 1;
    const int tiles = (N + BN - 1) / BN, nk = (K + BK - 1) / BK;
    // (a short reduction gains nothing that the second launch does not cost again: 15 -> 10 us on the device for K = 768, one more kernel to
    // enqueue in a host-bound decoding loop; the path is for the LONG reductions — 49 K-steps through 6 workgroups for the classifier's
    // input gradient, 47 us)
    if (nk < 24 || tiles >= 96) return 1;
    int sp = 128 / tiles;
    if (sp > nk / 2) sp = nk / 2;
    if (sp > 32) sp = 32;
    return sp < 2 ? 1 : sp;
}

// Several GEMMs of ONE operand layout in one grid (see gemm_bf16_grouped_kernel).  Every problem runs without split-K and with its
// own epilogue; `rowsum_out` is honoured (written directly).  debug_flags, splitk_ws are ignored.
static int grouped_impl(const mmf_gemm_desc* descs, int count, const LnRider* rider, bool* rode, void* stream) {
    MMF_CHECK_ARG(descs && count >= 1 && count <= MAXG, "mmf_gemm_bf16_grouped: 1 .. 8 problems");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GroupArgs g;
    g.count = count;
    int total = 0;
    bool ragged = false, rowsum = false;
    const int key = (descs[0].a_kmajor ? 1 : 0) | (descs[0].b_kmajor ? 2 : 0) | (descs[0].a_f32 ? 4 : 0) | (descs[0].b_f32 ? 8 : 0);
    for (int i = 0; i < count; ++i) {
        const mmf_gemm_desc* d = descs + i;
        GroupProblem& P = g.p[i];
        if (int rc = check_and_fill(d, P.epi)) return rc;
        const int k = (d->a_kmajor ? 1 : 0) | (d->b_kmajor ? 2 : 0) | (d->a_f32 ? 4 : 0) | (d->b_f32 ? 8 : 0);
        MMF_CHECK_ARG(k == key, "mmf_gemm_bf16_grouped: all problems must share the operand layout and dtypes");
        if (d->rowsum_out) {
            MMF_CHECK_ARG(key == 3, "mmf_gemm_bf16_grouped: rowsum_out needs the weight-gradient form (both operands k-major, bf16)");
            P.epi.rowsum_col = 1; P.epi.rowsum_direct = d->rowsum_out; rowsum = true;
        }
        P.A = d->A; P.B = d->B; P.M = d->M; P.N = d->N; P.K = d->K; P.lda = d->lda; P.ldb = d->ldb;
        P.tiles_m = (d->M + BM - 1) / BM; P.tiles_n = (d->N + BN - 1) / BN;
        g.start[i] = total;
        total += P.tiles_m * P.tiles_n;
        ragged = ragged || (d->M % BM) || (d->N % BN) || (d->K % BK);
    }
    for (int i = count; i <= MAXG; ++i) g.start[i] = total;
    g.total = total;
    if (g_probe.buf) { g_note[0] = 4 | key; g_note[1] = count; g_note[2] = total; g_note[3] = descs[0].K; }
    // weight-gradient form on the wide tile: every problem a whole number of 256 x 128 tiles and 64-deep K-steps, plain fp32 outputs
    if (key == 3 && mmf_amd_get_tunable(MMF_TUN_WGRAD_WIDE) != 1) {
        bool ok = true;
        int wtotal = 0;
        for (int i = 0; i < count; ++i) {
            const mmf_gemm_desc* d = descs + i;
            ok = ok && (d->M % 256) == 0 && (d->N % 128) == 0 && (d->K % 64) == 0 && d->K >= 192 && d->out_f32 && !d->bias && !d->coladd && !d->rowtab &&
                 d->act == 0 && !d->resid && !d->drop_thr16 && d->grp_in == 0 && (d->ldc % 4) == 0;
            wtotal += (d->M / 256) * (d->N / 128);
        }
        // (a launch far below one round of the 256 CUs is better off on the 128-row tiles: twice the workgroups, two per CU)
        if (ok && (wtotal >= 160 || mmf_amd_get_tunable(MMF_TUN_WGRAD_WIDE) == 2)) {
            int t = 0;
            for (int i = 0; i < count; ++i) {
                g.p[i].tiles_m = descs[i].M / 256; g.p[i].tiles_n = descs[i].N / 128;
                g.start[i] = t;
                t += g.p[i].tiles_m * g.p[i].tiles_n;
            }
            for (int i = count; i <= MAXG; ++i) g.start[i] = t;
            g.total = t;
            if (rider && t + 8 <= 256) {      // CUs left over by the tiles: the LayerNorm backward rides on them (gemm_wide_grouped_ln_kernel)
                LnRider r = *rider;
                r.nrider = 256 - t < (r.nblk + 1) / 2 ? 256 - t : (r.nblk + 1) / 2;
                if (rode) *rode = true;
                return rowsum ? launch_wide_grouped_ln<true>(g, r, s) : launch_wide_grouped_ln<false>(g, r, s);
            }
            return rowsum ? launch_wide_grouped<256, 128, 4, 2, true, true, true>(g, s) : launch_wide_grouped<256, 128, 4, 2, true, true, false>(g, s);
        }
    }
    switch (key) {
        case 0: return ragged ? launch_grouped_n<bf16, bf16, false, false, true, 4, 128>(g, s) : launch_grouped_n<bf16, bf16, false, false, false, 4, 128>(g, s);
        case 2: return ragged ? launch_grouped_n<bf16, bf16, false, true, true, 4, 128>(g, s) : launch_grouped_n<bf16, bf16, false, true, false, 4, 128>(g, s);
        case 3:
            if (rowsum) return ragged ? launch_grouped_n<bf16, bf16, true, true, true, 4, 128, 1, true>(g, s) : launch_grouped_n<bf16, bf16, true, true, false, 4, 128, 1, true>(g, s);
            return ragged ? launch_grouped_n<bf16, bf16, true, true, true, 4, 128, 2>(g, s) : launch_grouped_n<bf16, bf16, true, true, false, 4, 128, 2>(g, s);
        default:
            mmf_amd_set_error("mmf_gemm_bf16_grouped: unsupported operand layout combination");
            return 1;
    }
}

extern "C" int mmf_gemm_bf16_grouped(const mmf_gemm_desc* descs, int count, void* stream) { return grouped_impl(descs, count, nullptr, nullptr, stream); }

extern "C" __attribute__((visibility("hidden"))) int mmf_lnb_rider_blocks(int rows, int H);       // rowops.hip (library-internal)
// The grouped launch above AND one deferred LayerNorm backward that does not depend on it (mmf_layernorm_bwd with dgamma = dbeta = dbias = NULL: column-sum
// partials only).  Where the problems run on the wide tile and leave CUs idle (a VisualBERT layer's weight gradients: 216 tiles on 256 CUs) and the LayerNorm is
// the H = 768 two-row form, it rides on the idle CUs of the SAME launch (gemm_wide_grouped_ln_kernel); otherwise the two are launched one after the other.
// Same results either way, bit for bit.
extern "C" int mmf_gemm_bf16_grouped_ln(const mmf_gemm_desc* descs, int count, const mmf_ln_bwd_desc* ln, void* stream) {
    MMF_CHECK_ARG(ln && ln->dy && ln->x && ln->mean && ln->rstd && ln->gamma && ln->dx && ln->partials && ln->rows > 0, "mmf_gemm_bf16_grouped_ln: null LayerNorm operand");
    MMF_CHECK_ARG(ln->drop_thr16 == 0 || ln->dlin, "mmf_gemm_bf16_grouped_ln: dropout needs dlin");
    LnRider r;
    const int nblk = mmf_lnb_rider_blocks(ln->rows, ln->H);
    if (nblk > 0) {
        r.dy = reinterpret_cast<const bf16*>(ln->dy); r.x = reinterpret_cast<const bf16*>(ln->x); r.mean = ln->mean; r.rstd = ln->rstd; r.gamma = ln->gamma;
        r.dx = reinterpret_cast<bf16*>(ln->dx); r.dlin = reinterpret_cast<bf16*>(ln->dlin);
        r.drop = DropoutCfg{ln->drop_key, ln->drop_thr16, ln->drop_scale, ln->drop_seed};
        r.partials = ln->partials; r.rows = ln->rows; r.nblk = nblk; r.nrider = 0;
    }
    bool rode = false;
    if (int rc = grouped_impl(descs, count, nblk > 0 ? &r : nullptr, &rode, stream)) return rc;
    if (rode) return 0;
    return mmf_layernorm_bwd(ln->dy, ln->x, ln->mean, ln->rstd, ln->gamma, ln->dx, ln->dlin, ln->drop_key, ln->drop_thr16, ln->drop_scale, ln->drop_seed,
                             nullptr, nullptr, nullptr, 0, ln->partials, ln->rows, ln->H, stream);
}

// Development aid: while a probe buffer is set, every workgroup of every GEMM launch appends one 64-byte timeline record (see Probe).
// buf: device memory, (1 + capacity) * 64 bytes, zeroed by the caller; NULL switches the probe off.
extern "C" int mmf_gemm_set_probe(void* buf, int64_t capacity_records) {
    g_probe.buf = reinterpret_cast<unsigned long long*>(buf);
    g_probe.cap = (unsigned)(capacity_records > 0 ? capacity_records : 0);
    g_probe.launch = 0;
    if (buf) g_probe_nlog = 0;       // (the log of the last probed span stays readable after the probe is switched off)
    return 0;
}
// Copies the host-side log of the launches probed since mmf_gemm_set_probe (5 int64 per launch: launch id, layout — bit 0 A k-major, bit 1
// B k-major, bit 2 grouped —, M, N, K; grouped: problems, 128-row tiles, K) into `out`; returns the number of launches logged.
extern "C" int mmf_gemm_probe_log(int64_t* out_host, int capacity) {
    const int n = g_probe_nlog < capacity ? g_probe_nlog : capacity;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 5; ++j) out_host[i * 5 + j] = g_probe_log[i][j];
    return g_probe_nlog;
}

extern "C" const char* mmf_gemm_last_kernel(void) { return g_last_kernel; }
