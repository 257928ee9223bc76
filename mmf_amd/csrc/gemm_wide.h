// mmf_amd :: wide-tile bf16 MFMA GEMM (forward / NT form: both operands row-major, K contiguous), one workgroup per CU.
//
// Why: measured on MI355X (tools/ubench/dma_bench.hip, profiles/r02_lds_dma_ceiling.txt) the L2 -> LDS staging path delivers
// about 20 TB/s chip-wide = 75 - 100 GB/s per CU for the GEMM access pattern, whatever the DMA depth or workgroup shape, and the
// in-step timeline probe (tools/gemm_timeline.py) shows the K-loop of the 128 x 128 kernel sitting right on that ceiling
// (0.82 - 0.92 us per K-step for the two resident workgroups = 64 KB staged per 4.2 MFLOP).  A 128 x 128 tile stages 64 FLOP per
// byte: at 85 GB/s per CU that caps the MFMA pipe near 55 %.  The lever is FLOP per staged byte, i.e. ONE large tile per CU:
//     256 x  96 : 70 FLOP/B  (N = 768 outputs: 232 tiles on 256 CUs)
//     192 x 192 : 96 FLOP/B  (N = 2304:        456 tiles = 2 rounds at 89 %)
//     192 x 256 : 110 FLOP/B (N = 3072:        456 tiles)
//
// Schedule ("ping-pong"): 512 threads = 8 waves = two groups of four, one wave of each group on every SIMD.  A K-step (BK = 64)
// is two barrier intervals; in every interval one group runs its MFMAs for a whole K-step (COMP) while the other fetches ALL
// fragments of its next K-step from LDS into registers (LOAD); the groups swap roles at each raw s_barrier, group 1 running one
// interval behind group 0.  So on every SIMD the matrix pipe always has one wave issuing MFMAs while its partner waits for LDS:
//     interval   I(2t)              I(2t+1)
//     group 0    LOAD(t)            COMP(t)
//     group 1    COMP(t-1)          LOAD(t)
// Operand stages stream L2 -> LDS by LDS-DMA into a ring of NS stages.  Stage t + NS - 1 is issued (by all waves) in I(2t) into
// the slot whose last reader was group 1's LOAD(t - 1) in I(2t - 1); every wave retires its own pieces of stage t + 1 with a
// counted s_waitcnt vmcnt at the end of I(2t + 1), ahead of the barrier that opens I(2t + 2), where group 0 first reads it.
//   RAW: DMA pieces of stage s are waited for by their issuing wave before a barrier that every reader passes afterwards.
//   WAR: every LOAD ends with s_waitcnt lgkmcnt(0) before its barrier; a slot is re-issued only after the barrier that follows
//        its last LOAD.
// The accumulators go through LDS (fp32, one wave-row group per pass) to the same row-wise fused epilogue as the 128-row kernel.
#pragma once
#include "gemm_common.h"

namespace gemm {

template <int N> DEVI void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BM x BN tile, WGM x WGN wave grid, NS ring stages.
//   WGM * WGN == 8: every wave multiplies both 32-deep halves of a 64-deep K-step over its (BM / WGM) x (BN / WGN) tile.  (Rounds 3 - 5 also had
//   a K-split layout - four wave positions, the two ping-pong groups multiplying one K-half each, 14 -> 10 fragment reads per step on 256 x 96: 3 - 5 %
//   faster in isolation at long K, 0.1 - 0.3 ms per step slower inside the step on every box: profiles/r03_gemm_ab_ks.txt, r04_in_step_choices.txt.
//   Removed in round 6.)
// AKM / BKM: the operand is k-major (A(m,k) = A[k * lda + m]): the weight-gradient form dW = dY^T X, both operands token-major with the
//   reduction running over their rows.  Its LDS image is the 128-row kernel's ([64 k-rows][256 B] per 128 tile rows, bytes rotated per
//   k-row, read with ds_read_b64_tr_b16: gemm_common.h), one 16 KiB sub-image per 128 tile rows; a DMA piece is 32 k-rows of one
//   sub-image, so pieces, ring slots and the whole schedule are those of the row-major form.
// RS: the bias gradient rides on the launch (EpiArgs::rowsum_direct): the tile_n == 0 workgroups multiply every A fragment once more
//   against an all-ones operand (see gemm_tile in gemm.hip).
template <int BM_, int BN_, int WGM, int WGN, int NS, bool RAGGED_M, int ABL, bool AKM, bool BKM, bool RS>
DEVI void wide_tile(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb, int tile_m, int tile_n,
                    const EpiArgs& epi, const Probe& pr, unsigned char* smem) {
    // ABL (ablation builds only, -DMMF_WIDE_ABLATE): bit 0 no DMA issue in the loop, bit 1 no MFMA, bit 2 no fragment reads, bit 3 no epilogue
    constexpr int dbg = ABL;
    static_assert(WGM * WGN == 8, "eight waves");
    constexpr int KK = 2;                        // 32-deep sub-steps a wave multiplies per 64-deep K-step
    static_assert(NS == 3, "the schedule below is written for a three-stage ring");
    static_assert(BM_ % 64 == 0 && BN_ % 32 == 0, "tile shape");
    static_assert(!AKM || (BM_ % 128 == 0 && !RAGGED_M), "k-major A: whole 128-row sub-images");
    static_assert(!BKM || BN_ % 128 == 0, "k-major B: whole 128-row sub-images");
    static_assert(!RS || AKM, "row sums ride on the weight-gradient form");
    constexpr int WTM = BM_ / WGM, WTN = BN_ / WGN, NFM = WTM / 16, NFN = WTN / 16;
    constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = BM_ / 64;                 // LDS-DMA wave-instructions per wave for the A image (64 rows each, all 8 waves)
    constexpr int PB_FULL = BN_ / 64;            // same for B; a trailing 32-row half piece is issued by waves 0..3 only
    constexpr bool B_HALF = (BN_ % 64) != 0;
    constexpr int P_LO = PA + PB_FULL + (B_HALF ? 1 : 0), P_HI = PA + PB_FULL;   // pieces per stage: waves 0..3 / waves 4..7

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int gpos = wave;                       // position in the wave grid
    const int wm = gpos / WGN, wn = gpos % WGN;
    const bool probing = pr.buf != nullptr;
    unsigned long long pt[5] = {0, 0, 0, 0, 0};
    if (probing) pt[0] = __builtin_amdgcn_s_memrealtime();

    const int m0 = tile_m * BM_, n0 = tile_n * BN_;
    const int nk = K / BK;

    // per-thread DMA source pointers (they advance by one K-step per stage); the chunk swizzle of the row-major LDS image is
    // applied to the source address: lane-linear destination, chunk c of row r lands at r * 128 + ((c ^ (r & 7)) << 4)
    const int sw = (tid >> 3) & 7;
    const int kchunk = ((tid & 7) ^ sw) * 8;
    const bf16* pa[PA];
    const bf16* pb[PB_FULL + 1];          // (the last entry is the trailing half piece; unused unless B_HALF)
    // k-major piece i: k-rows (tid >> 4) + 32 * (i & 1) of sub-image i >> 1; the row's rotation is applied to the source column
    const int km_kr = tid >> 4;
    const int km_col = (((tid & 15) - (rot_kmajor(km_kr) >> 4)) & 15) * 8;      // (the rotation repeats every 16 k-rows: same for both halves)
    const size_t a_step = AKM ? (size_t)BK * lda : (size_t)BK, b_step = BKM ? (size_t)BK * ldb : (size_t)BK;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        if constexpr (AKM) {
            pa[i] = A + (size_t)(km_kr + 32 * (i & 1)) * lda + m0 + (i >> 1) * 128 + km_col;
        } else {
            int row = m0 + (tid >> 3) + 64 * i;
            if (RAGGED_M) row = row < M ? row : M - 1;
            pa[i] = A + (size_t)row * lda + kchunk;
        }
    }
#pragma unroll
    for (int i = 0; i < PB_FULL + 1; ++i) {
        if constexpr (BKM) {
            pb[i] = B + (size_t)(km_kr + 32 * (i & 1)) * ldb + n0 + (i >> 1) * 128 + km_col;
        } else {
            int row = n0 + (tid >> 3) + 64 * i;
            row = row < N ? row : N - 1;            // (only the never-issued upper half of a trailing half piece can be out of range)
            pb[i] = B + (size_t)row * ldb + kchunk;
        }
    }
    auto issue = [&](int slot) {
        unsigned char* st = smem + slot * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)pa[i], (lds_vp)(st + i * 8192), 16, 0, 0);
            pa[i] += a_step;
        }
#pragma unroll
        for (int i = 0; i < PB_FULL; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)pb[i], (lds_vp)(st + A_BYTES + i * 8192), 16, 0, 0);
            pb[i] += b_step;
        }
        if (B_HALF) {
            if (wave < 4) __builtin_amdgcn_global_load_lds((glb_vp)pb[PB_FULL], (lds_vp)(st + A_BYTES + PB_FULL * 8192), 16, 0, 0);
            pb[PB_FULL] += b_step;
        }
    };
    f32x4 acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[KK][NFM], fb[KK][NFN];
    // bias gradient riding on the weight-gradient form: rowsum[m] = sum_k A(m, k), one extra MFMA per A fragment against all-ones
    // (every row of that 16 x 16 result holds the sums); only the first column of tiles and of waves does it
    const bool do_rowsum = RS && epi.rowsum_direct != nullptr && tile_n == 0 && wn == 0;
    f32x4 accr[RS ? NFM : 1];
#pragma unroll
    for (int i = 0; i < (RS ? NFM : 1); ++i) accr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

    // fragment addresses: row operand image, fragment f (16 rows), sub-step kk: lane l -> row (l & 15), chunk kk*4 + (l >> 4)
    const int frow = lane & 15, fswz = lane & 7;
    const int a_off = (wm * WTM + frow) * 128, b_off = A_BYTES + (wn * WTN + frow) * 128;
    const int c0_ = (((lane >> 4)) ^ fswz) << 4, c1_ = ((4 + (lane >> 4)) ^ fswz) << 4;
    const int c0 = c0_, c1 = c1_;
    // k-major image (read_frag<true> of gemm_common.h, addresses hoisted): lane (g = l >> 4, p = l & 15) reads 8 bytes at k-row
    // 32 kk + 8 g + (p >> 2) (+ 4 for the second half) of its sub-image, byte column ((col * 2 + rot) & 255), col = the fragment's
    // first tile row + 4 (p & 3), rot = 32 ((p >> 2) + 4 (g & 1)); sub-step kk adds 32 k-rows = 8192 bytes.
    int ka_off[AKM ? NFM : 1], kb_off[BKM ? NFN : 1];
    if constexpr (AKM || BKM) {
        const int g = lane >> 4, p = lane & 15;
        const int rot = 32 * ((p >> 2) + 4 * (g & 1)), krow = (8 * g + (p >> 2)) * 256;
        const int kk0 = 0;
        if constexpr (AKM) {
#pragma unroll
            for (int f = 0; f < NFM; ++f) {
                const int r = wm * WTM + f * 16;          // tile row of the fragment
                ka_off[f] = (r >> 7) * 16384 + krow + ((((r & 127) + (p & 3) * 4) * 2 + rot) & 255) + kk0;
            }
        }
        if constexpr (BKM) {
#pragma unroll
            for (int f = 0; f < NFN; ++f) {
                const int r = wn * WTN + f * 16;
                kb_off[f] = A_BYTES + (r >> 7) * 16384 + krow + ((((r & 127) + (p & 3) * 4) * 2 + rot) & 255) + kk0;
            }
        }
    }
    auto read_km = [&](const unsigned char* q) {
        typedef s16x4 __attribute__((address_space(3))) * lds_p;
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(q + 1024));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    };
    // LOAD: all 2 * (NFM + NFN) fragments of a K-step, with the P DMA pieces of the stage issued in between (one piece after
    // every few reads, order pinned): the LDS read port and the address path of the LDS-DMA are different units.
    constexpr int NREAD = KK * (NFM + NFN);
    auto load_frags = [&](int slot, int islot, bool with_issue) {
        if constexpr ((dbg & 4) != 0) { if (with_issue && !(dbg & 1)) issue(islot); return; }
        const unsigned char* st = smem + slot * STAGE;
        unsigned char* ist = smem + islot * STAGE + wave * 1024;
        constexpr int PT = P_LO;                         // pieces to place (the last one is the waves-0..3-only half piece, if any)
        int placed = 0;
#pragma unroll
        for (int r = 0; r < NREAD; ++r) {
            const int kk = r / (NFM + NFN), f = r % (NFM + NFN);
            const int coff = kk ? c1 : c0;
            if (f < NFN) {
                if constexpr (BKM) fb[kk][f] = read_km(st + kb_off[f] + kk * 8192);
                else fb[kk][f] = *reinterpret_cast<const bf16x8*>(st + b_off + f * 2048 + coff);
            } else {
                if constexpr (AKM) fa[kk][f - NFN] = read_km(st + ka_off[f - NFN] + kk * 8192);
                else fa[kk][f - NFN] = *reinterpret_cast<const bf16x8*>(st + a_off + (f - NFN) * 2048 + coff);
            }
            // after read r, issue the pieces that are due: piece q goes after read floor((q + 1) * NREAD / (PT + 1)) - 1
#pragma unroll
            for (int q = 0; q < PT; ++q) {
                if (((q + 1) * NREAD) / (PT + 1) - 1 == r) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (with_issue && !(dbg & 1)) {
                        if (q < PA) {
                            __builtin_amdgcn_global_load_lds((glb_vp)pa[q], (lds_vp)(ist + q * 8192), 16, 0, 0);
                            pa[q] += a_step;
                        } else if (q < PA + PB_FULL) {
                            __builtin_amdgcn_global_load_lds((glb_vp)pb[q - PA], (lds_vp)(ist + A_BYTES + (q - PA) * 8192), 16, 0, 0);
                            pb[q - PA] += b_step;
                        } else {
                            if (wave < 4) __builtin_amdgcn_global_load_lds((glb_vp)pb[PB_FULL], (lds_vp)(ist + A_BYTES + PB_FULL * 8192), 16, 0, 0);
                            pb[PB_FULL] += b_step;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ++placed;
                }
            }
        }
        (void)placed;
    };
    auto compute = [&]() {
        if constexpr ((dbg & 2) != 0) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int i = 0; i < NFM; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
        if constexpr (RS) {
            if (do_rowsum) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int i = 0; i < NFM; ++i) accr[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[kk][i], accr[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_one_stage_left = [&]() { if (wave < 4) wait_vm<P_LO>(); else wait_vm<P_HI>(); };

    // ---- side-input prefetch of the epilogue (see epilogue8_fast) -------------------------------------------------------------------
    // Every thread owns fixed (row, 8-column segment) slots of the row-wise epilogue pass; their residual / multiplier rows are fetched
    // into registers when the K-loop has two steps left.  The loads are issued unconditionally per wave (clamped addresses) so that
    // the counted wait of the penultimate K-step can step over exactly NSIDE of them.
    constexpr int CLD = BN_ + 4;
    constexpr int RING = NS * STAGE;
    constexpr int WPP_MAX = RING / (WTM * CLD * 4);
    constexpr int WPP = WPP_MAX >= WGM ? WGM : (WPP_MAX >= 1 ? WPP_MAX : 1);
    static_assert(WTM * CLD * 4 <= RING, "one wave-row of the fp32 tile must fit in the ring");
    static_assert(WGM % WPP == 0, "passes");
    constexpr int SEG = BN_ / 8;
    constexpr int NPASS = WGM / WPP, PER_PASS = WPP * WTM * SEG, NIT = (PER_PASS + 511) / 512, NSIDE = NPASS * NIT;
    const bool fast = epilogue_fast_ok(epi) && !(dbg & 8);
    const bf16* sidep = fast ? (epi.resid ? epi.resid : (epi.act == 2 ? epi.aux : nullptr)) : nullptr;
    const int side_ld = epi.resid ? epi.ldr : epi.ldc;
    uint4 side[NSIDE];
#pragma unroll
    for (int i = 0; i < NSIDE; ++i) side[i] = make_uint4(0, 0, 0, 0);
    auto prefetch_side = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                int idx = tid + 512 * i;
                idx = idx < PER_PASS ? idx : PER_PASS - 1;
                const int row = idx / SEG, seg = idx - row * SEG;
                int m = m0 + p * WPP * WTM + row;
                m = m < M ? m : M - 1;
                side[p * NIT + i] = *reinterpret_cast<const uint4*>(sidep + (size_t)m * side_ld + n0 + seg * 8);
            }
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: two stages in flight, wait for stage 0
    issue(0);
    if (nk > 1) { issue(1); wait_one_stage_left(); } else wait_vm<0>();
    bar();
    if (probing) pt[1] = __builtin_amdgcn_s_memrealtime();

    // Both groups run the SAME instruction stream (one register allocation); group 1 enters it one barrier later.  With I(n) the
    // interval between barriers n and n + 1 (counted from the prologue's), group 0 runs LOAD(t) in I(2t) and COMP(t) in
    // I(2t + 1), group 1 LOAD(t) in I(2t + 1) and COMP(t) in I(2t + 2).  Every wave issues its pieces of stage t + 2 inside its
    // OWN LOAD(t) - the interval in which its SIMD partner runs MFMAs - into the slot of stage t - 1 (last read: group 1's
    // LOAD(t - 1) in I(2t - 1); earliest re-issue: group 0's LOAD(t) in I(2t)), and ends LOAD(t) with a counted wait that
    // retires its pieces of stage t + 1 (issued one K-step earlier; stage t + 2 stays in flight): group 1's wait falls in
    // I(2t + 1), group 0's in I(2t), both ahead of the barrier that opens I(2t + 2), where stage t + 1 is first read.
    // The main loop is branch-free; the last two K-steps (nothing left to issue) are peeled.
    if (grp == 1) bar();
    int slot = 0;                       // ring slot of stage t
    for (int t = 0; t < nk - 2; ++t) {
        const int islot = slot == 0 ? NS - 1 : slot - 1;
        load_frags(slot, islot, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_one_stage_left();
        bar();
        compute();
        bar();
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    if (nk >= 2) {
        if (sidep) prefetch_side();
        load_frags(slot, 0, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sidep) wait_vm<NSIDE>(); else wait_vm<0>();      // (in-order retirement: the last stage's DMA precedes the side loads)
        bar();
        compute();
        bar();
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    if (nk < 2 && sidep) prefetch_side();
    load_frags(slot, 0, false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    compute();
    bar();
    if (grp == 0) bar();
    if (probing) pt[2] = __builtin_amdgcn_s_memrealtime();
    if (dbg & 8) {
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int j = 0; j < NFN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }

    if constexpr (RS) {
        if (do_rowsum && lane < 16) {
#pragma unroll
            for (int i = 0; i < NFM; ++i) {
                const int m = m0 + wm * WTM + i * 16 + lane;
                if (m < M) epi.rowsum_direct[m] = accr[i][0];
            }
        }
    }
    // epilogue: fp32 tile through the (idle) ring, WPP wave-rows per pass, then the row-wise fused epilogue
    float* cs = reinterpret_cast<float*>(smem);
    const uint32_t dkey = drop_key(epi.drop);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (wm / WPP == p) {
            const int lr0 = (wm % WPP) * WTM;
#pragma unroll
            for (int i = 0; i < NFM; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j) {
                    const int row = lr0 + i * 16 + (lane & 15), col = wn * WTN + j * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<float4*>(cs + row * CLD + col) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
        }
        __syncthreads();
        if (probing && p == 0) pt[3] = __builtin_amdgcn_s_memrealtime();
        if (fast) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int idx = tid + 512 * i;
                if (idx < PER_PASS) {
                    const int row = idx / SEG, seg = idx - row * SEG;
                    epilogue8_fast(epi, m0 + p * WPP * WTM + row, n0 + seg * 8, load_f8(cs + row * CLD + seg * 8), side[p * NIT + i], dkey);
                }
            }
        } else {
#pragma unroll 1
            for (int idx = tid; idx < PER_PASS; idx += 512) {
                const int row = idx / SEG, seg = idx - row * SEG;
                epilogue8(epi, m0 + p * WPP * WTM + row, n0 + seg * 8, load_f8(cs + row * CLD + seg * 8), 0);
            }
        }
        if (p + 1 < NPASS) __syncthreads();
    }
    if (probing) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pt[4] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            const unsigned s = atomicAdd(reinterpret_cast<unsigned*>(pr.buf), 1u);
            if (s < pr.cap) {
                unsigned long long* r = pr.buf + 8 * (size_t)(s + 1);
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                r[0] = ((unsigned long long)pr.launch << 32) | blockIdx.x;
                r[1] = ((unsigned long long)xcc << 32) | hw;
                r[2] = pt[0]; r[3] = pt[1]; r[4] = pt[2]; r[5] = pt[3]; r[6] = pt[4];
                r[7] = ((unsigned long long)(unsigned)tile_m << 32) | (unsigned)tile_n;
            }
        }
    }
}

// tile order: XCD-aware contiguous runs, 4-row super-rows (column-major inside) so an XCD's resident tiles share panels
DEVI int wide_xcd_remap(int bid, int ntile) {
    const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
DEVI void wide_super_row(int bid, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    const int per_sr = 4 * tiles_n;
    const int sr = bid / per_sr, rem = bid - sr * per_sr;
    const int h = min(4, tiles_m - sr * 4);
    tile_n = rem / h;
    tile_m = sr * 4 + (rem - tile_n * h);
}

template <int BM_, int BN_, int WGM, int WGN, int NS, bool RAGGED_M, int ABL = 0, bool AKM = false, bool BKM = false, bool RS = false>
__global__ __launch_bounds__(512, 2) void gemm_wide_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K,
                                                            int lda, int ldb, int tiles_m, int tiles_n, EpiArgs epi, Probe pr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile_m, tile_n;
    wide_super_row(wide_xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tile_m, tile_n);
    wide_tile<BM_, BN_, WGM, WGN, NS, RAGGED_M, ABL, AKM, BKM, RS>(A, B, M, N, K, lda, ldb, tile_m, tile_n, epi, pr, smem);
}

}  // namespace gemm
