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

using namespace gemm;


namespace {

// ---- the kernel ---------------------------------------------------------------------------------
template <typename AT, typename BT, bool A_KMAJOR, bool B_KMAJOR, bool RAGGED, int NWN, int BN_, int KS = 1, bool RS = false>
__global__ __launch_bounds__(128 * NWN, NWN) void gemm_bf16_kernel(const AT* __restrict__ A, const BT* __restrict__ B,
                                                             int M, int N, int K, int lda, int ldb,
                                                             int tiles_m, int tiles_n, int splits, int dbg, EpiArgs epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool A_DMA = is_bf16<AT>::value, B_DMA = is_bf16<BT>::value;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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

    // XCD-aware tile order: block b runs on XCD b % 8; give every XCD a contiguous run of the
    // (m-major, n-fastest) tile list so the A row panel and the weight panel stay in its L2.
    const int ntile = tiles_m * tiles_n;
    const int nblk = ntile * splits;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int split = bid / ntile;
    bid -= split * ntile;
    // 8-row super-rows, column-major inside: the ~64 tiles resident on one XCD share 8 A row-panels and a few B
    // panels that fit its 4 MiB L2 (the weight panel is then re-read once per super-row, not once per row).
    int tile_m, tile_n;
    {
        const int per_sr = 8 * tiles_n;
        const int sr = bid / per_sr, rem = bid - sr * per_sr;
        const int h = min(8, tiles_m - sr * 8);
        tile_n = rem / h;
        tile_m = sr * 8 + (rem - tile_n * h);
    }
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
    constexpr int SEG = BN_ / 8;
    for (int idx = tid; idx < BM * SEG; idx += NTH) {
        const int row = idx / SEG, seg = idx - row * SEG;
        epilogue8(epi, m0 + row, n0 + seg * 8, load_f8(cs + row * CLD + seg * 8), split);
    }
    if (RS && epi.rowsum_col >= 0 && tile_n == 0 && tid < BM && m0 + tid < M)
        reinterpret_cast<float*>(epi.C)[(size_t)splits * epi.slab_stride + (size_t)split * M + m0 + tid] = rs[tid];
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
    hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, AK, BK_, RG, NWN, BN_, KS, RS>), dim3(tm * tn * splits), dim3(128 * NWN), lds_bytes, s,
                       reinterpret_cast<const AT*>(d->A), reinterpret_cast<const BT*>(d->B), d->M, d->N, d->K,
                       d->lda, d->ldb, tm, tn, splits, (d->debug_flags >> 4) & 15, e2);
    MMF_CHECK_LAUNCH();
    return 0;
}


// ---- 256 x 128 tile, three-stage LDS-DMA ring (EXPERIMENTAL, debug_flags bit 14; not used by default) -----------------
// Status: correct (tests/test_gemm256_gpu.py, incl. a 30-launch race screen) and 20-40 % SLOWER than the 128-row kernel on every
// forward / dgrad shape of the VisualBERT layer (572 us per layer, 549 with all fragment reads of a step issued first - bit 15 -
// against 477; tools/micro_sweep.py tile256, profiles/r01_gemm256_experiment.txt): with one
// workgroup of 8 waves per CU (2 waves per SIMD) the compiler-scheduled read-wait-MFMA sequence of a step leaves the MFMA pipe
// idle while fragments are in flight, and the deeper ring does not buy that back.  Kept as the tested skeleton (ring, counted
// waits, masked ragged-M epilogue) for the phase-interleaved schedule of cdna_hip_programming.md section 5.
// One workgroup per CU (144 KiB of LDS), 8 waves as 4 x 2 wave tiles of 64 x 64 (8 fragment reads per 16 MFMAs), K-step 64.
// Per step a CU stages 48 KiB for 4.2 MFLOP (85 FLOP per staged byte against 64 for two 128 x 128 workgroups), and the ring
// keeps TWO stages in flight: iteration kt waits with a counted `s_waitcnt vmcnt(6)` (the six LDS-DMA instructions of stage
// kt + 1 may still be outstanding), crosses one raw `s_barrier`, issues stage kt + 2 into the buffer everybody finished
// reading in iteration kt - 1, and multiplies stage kt.  A is a row operand (two 128-row images per stage), B a row or a
// k-major operand (the same LDS images and fragment reads as the 128 x 128 kernel); M may be ragged (row clamp + masked
// epilogue), N % 128 == 0 and K % 64 == 0.  The fp32 tile is staged through the idle ring for the row-wise epilogue.
constexpr int BM2 = 256, NSTAGE2 = 3, STAGE2_BYTES = 3 * OPER_BYTES;

template <bool B_KMAJOR, bool RAGGED, int SCHED>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K,
                                                          int lda, int ldb, int tiles_m, int tiles_n, EpiArgs epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware order, as in gemm_bf16_kernel
        const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int tile_m, tile_n;
    {
        const int per_sr = 4 * tiles_n;          // 4-row super-rows (1024 rows of A), column-major inside
        const int sr = bid / per_sr, rem = bid - sr * per_sr;
        const int h = min(4, tiles_m - sr * 4);
        tile_n = rem / h;
        tile_m = sr * 4 + (rem - tile_n * h);
    }
    const int m0 = tile_m * BM2, n0 = tile_n * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    auto issue = [&](int kt, int buf) {
        unsigned char* st = smem + buf * STAGE2_BYTES;
        stage_dma<false, RAGGED, 512>(A, lda, m0, kt * BK, M, K, st, tid);
        stage_dma<false, RAGGED, 512>(A, lda, m0 + 128, kt * BK, M, K, st + OPER_BYTES, tid);
        stage_dma<B_KMAJOR, false, 512>(B, ldb, n0, kt * BK, N, K, st + 2 * OPER_BYTES, tid);
    };
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int buf = 0;
    if constexpr (SCHED == 2) {
        // Ping-pong schedule (bit 16; UNTESTED ON HARDWARE as of round 1 - written against the ISA, to be race-screened with
        // tests/test_gemm256_gpu.py before any use).  The 8 waves form two groups of four, one wave of each group per SIMD; the
        // second group runs ONE barrier interval behind the first, so on every SIMD one wave issues its 16 MFMAs (COMP) while
        // the other fetches its next 8 fragments and issues its share of the LDS-DMA for the stage two K-tiles ahead (LOAD).
        // A K-tile is LOAD0 | COMP0 | LOAD1 | COMP1 with a raw s_barrier after each; global barrier numbering below counts from
        // the first barrier after the prologue, interval I(n) lies between barriers n and n + 1:
        //   group 0:  LOAD0[t] in I(4t), COMP0[t] in I(4t+1), LOAD1[t] in I(4t+2), COMP1[t] in I(4t+3)
        //   group 1:  the same, one interval later.
        // RAW (stage t+1 is first read in I(4t+4)): every wave retires its own stage-(t+1) pieces with the counted vmcnt at the
        //   end of LOAD1[t] (I(4t+2) / I(4t+3)), i.e. before barrier 4t+4.
        // WAR (stage t+2 overwrites the buffer of stage t-1, whose last reads are group 1's LOAD1[t-1] in I(4t-1)): every LOAD
        //   ends with lgkmcnt(0) before its barrier, and the earliest re-staging is group 0's LOAD0[t] in I(4t), after barrier 4t.
        const int grp = wave >> 2;
        if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned char* la = smem + buf * STAGE2_BYTES + (wm >> 1) * OPER_BYTES;
            const unsigned char* lb = smem + buf * STAGE2_BYTES + 2 * OPER_BYTES;
            unsigned char* nst = smem + (buf >= 1 ? buf - 1 : 2) * STAGE2_BYTES;
            const bool more = kt + 2 < nk;
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ---- LOAD kk
#pragma unroll
                for (int f = 0; f < 4; ++f) fa[f] = read_frag<false>(la, (wm & 1) * 64, f, kk, lane);
#pragma unroll
                for (int f = 0; f < 4; ++f) fb[f] = read_frag<B_KMAJOR>(lb, wn * 64, f, kk, lane);
                if (more) {
                    if (kk == 0) {
                        stage_dma<false, RAGGED, 512, 0, 2>(A, lda, m0, (kt + 2) * BK, M, K, nst, tid);
                        stage_dma<false, RAGGED, 512, 0, 1>(A, lda, m0 + 128, (kt + 2) * BK, M, K, nst + OPER_BYTES, tid);
                    } else {
                        stage_dma<false, RAGGED, 512, 1, 2>(A, lda, m0 + 128, (kt + 2) * BK, M, K, nst + OPER_BYTES, tid);
                        stage_dma<B_KMAJOR, false, 512, 0, 2>(B, ldb, n0, (kt + 2) * BK, N, K, nst + 2 * OPER_BYTES, tid);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kk == 1) {
                    if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- COMP kk
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) issue(kt + 2, buf >= 1 ? buf - 1 : 2);      // (kt + 2) % 3 == (buf + 2) % 3
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* la = smem + buf * STAGE2_BYTES + (wm >> 1) * OPER_BYTES;
        const unsigned char* lb = smem + buf * STAGE2_BYTES + 2 * OPER_BYTES;
        if constexpr (SCHED == 1) {     // bit 15: all 16 fragment reads of the step first, so the second half's land behind the first half's MFMAs
            bf16x8 fa[2][4], fb[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int f = 0; f < 4; ++f) fa[kk][f] = read_frag<false>(la, (wm & 1) * 64, f, kk, lane);
#pragma unroll
                for (int f = 0; f < 4; ++f) fb[kk][f] = read_frag<B_KMAJOR>(lb, wn * 64, f, kk, lane);
            }
            __builtin_amdgcn_sched_barrier(0);     // without it the scheduler sinks the reads back between the MFMAs
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) fa[f] = read_frag<false>(la, (wm & 1) * 64, f, kk, lane);
#pragma unroll
            for (int f = 0; f < 4; ++f) fb[f] = read_frag<B_KMAJOR>(lb, wn * 64, f, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
    __syncthreads();       // every wave is done with the ring: it becomes the fp32 C stage
    constexpr int CLD = BN + 4;
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wm * 64 + i * 16 + (lane & 15), col = wn * 64 + j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<float4*>(cs + row * CLD + col) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    __syncthreads();
    constexpr int SEG = BN / 8;
    for (int idx = tid; idx < BM2 * SEG; idx += 512) {
        const int row = idx / SEG, seg = idx - row * SEG;
        epilogue8(epi, m0 + row, n0 + seg * 8, load_f8(cs + row * CLD + seg * 8), 0);
    }
}

template <bool BK_, bool RG, int SCHED>
int launch256(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    const int tm = (d->M + BM2 - 1) / BM2, tn = d->N / BN;
    constexpr int cstage = BM2 * (BN + 4) * (int)sizeof(float);
    constexpr int lds_bytes = cstage > NSTAGE2 * STAGE2_BYTES ? cstage : NSTAGE2 * STAGE2_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<BK_, RG, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (ae != hipSuccess) { mmf_amd_set_error(hipGetErrorString(ae)); return 2; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<BK_, RG, SCHED>), dim3(tm * tn), dim3(512), lds_bytes, s, reinterpret_cast<const bf16*>(d->A),
                       reinterpret_cast<const bf16*>(d->B), d->M, d->N, d->K, d->lda, d->ldb, tm, tn, e);
    MMF_CHECK_LAUNCH();
    return 0;
}

template <typename AT, typename BT, bool AK, bool BK_, bool RG>
int launch(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    // 8 waves (4 per SIMD with two workgroups per CU) hide LDS / MFMA-issue latency better than 4 waves of 64x64;
    // bit 8 of debug_flags selects the 4-wave form for A/B measurements.
    if (AK && BK_ && is_bf16<AT>::value && is_bf16<BT>::value && e.rowsum_col >= 0)   // weight gradient carrying the bias gradient
        return launch_n<AT, BT, AK, BK_, RG, 4, 128, 1, AK && BK_>(d, e, s);
    // 256 x 128 tiles with the three-stage ring (experimental: debug_flags bit 14 selects it)
    if constexpr (!AK && is_bf16<AT>::value && is_bf16<BT>::value) {
        if ((d->debug_flags & 16384) && e.splits <= 1 && (d->N % BN) == 0 && (d->K % BK) == 0 && d->M >= BM2)
        {
            const int sched = (d->debug_flags & 65536) ? 2 : ((d->debug_flags & 32768) ? 1 : 0);   // bit 16 ping-pong, bit 15 reads-first
            if (d->M % BM2) return sched == 2 ? launch256<BK_, true, 2>(d, e, s) : sched == 1 ? launch256<BK_, true, 1>(d, e, s) : launch256<BK_, true, 0>(d, e, s);
            return sched == 2 ? launch256<BK_, false, 2>(d, e, s) : sched == 1 ? launch256<BK_, false, 1>(d, e, s) : launch256<BK_, false, 0>(d, e, s);
        }
    }
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

extern "C" int mmf_gemm_bf16(const mmf_gemm_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->A && d->B && d->C, "mmf_gemm_bf16: null operand");
    MMF_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "mmf_gemm_bf16: empty shape");
    MMF_CHECK_ARG((d->lda % 8) == 0 && (d->ldb % 8) == 0, "mmf_gemm_bf16: lda/ldb must be multiples of 8 elements");
    // row operands are fetched in 8-element chunks along K: the last chunk may run into the row padding
    // (which must hold finite values, normally zeros), so the leading dimension has to cover it.
    const int k8 = (d->K + 7) / 8 * 8;
    MMF_CHECK_ARG(d->a_kmajor || d->lda >= k8, "mmf_gemm_bf16: lda must cover round_up(K, 8) for a row operand");
    MMF_CHECK_ARG(d->b_kmajor || d->ldb >= k8, "mmf_gemm_bf16: ldb must cover round_up(K, 8) for a row operand");
    MMF_CHECK_ARG(!(d->a_f32 && d->b_f32), "mmf_gemm_bf16: at most one fp32 operand");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    EpiArgs e;
    e.C = d->C; e.ldc = d->ldc; e.out_f32 = d->out_f32; e.beta = d->beta;
    e.bias = d->bias; e.coladd = d->coladd; e.rowtab = d->rowtab; e.rowidx = d->rowidx; e.rowtab_ld = d->rowtab_ld;
    e.act = d->act; e.U = reinterpret_cast<bf16*>(d->U); e.aux = reinterpret_cast<const bf16*>(d->aux);
    e.resid = reinterpret_cast<const bf16*>(d->resid); e.ldr = d->ldr;
    e.drop.key = d->drop_key; e.drop.thr16 = d->drop_thr16; e.drop.scale = d->drop_scale; e.drop.seed = d->drop_seed;
    e.grp_in = d->grp_in; e.grp_pad = d->grp_pad; e.grp_off = d->grp_off;
    e.M = d->M; e.N = d->N;
    // Split-K: a weight-gradient GEMM has few output tiles (768x768 -> 36) and a long reduction (K = tokens).
    // With a workspace, the K range is spread over `splits` workgroups per tile; each writes an fp32 partial slab
    // and a second kernel sums the slabs in a fixed order (deterministic, no atomics).
    e.slab_stride = 0; e.splits = 1; e.rowsum_col = -1;
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
    if (d->rowsum_out) {
        MMF_CHECK_ARG(d->a_kmajor && d->b_kmajor && !d->a_f32 && !d->b_f32 && final_c,
                      "mmf_gemm_bf16: rowsum_out needs the weight-gradient form (both operands k-major, bf16 A) with split-K active");
        e.rowsum_col = 1;   // on; the partials live behind the slabs: ws[splits * M * N + split * M + m]
    }
    MMF_CHECK_ARG((d->act != 2 && d->act != 4) || d->aux, "mmf_gemm_bf16: act=2/4 needs aux");
    MMF_CHECK_ARG(d->act >= 0 && d->act <= 4, "mmf_gemm_bf16: unknown act");
    MMF_CHECK_ARG(!d->rowtab || d->rowidx, "mmf_gemm_bf16: rowtab needs rowidx");

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
    if (const int f = mmf_amd_get_tunable(MMF_TUN_SPLITK_FORCE)) return f < 1 ? 1 : (f > nk_all / 2 ? nk_all / 2 : f);
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
