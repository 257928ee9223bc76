// mmf_amd :: persistent wide-tile bf16 MFMA GEMM (NT form) whose epilogue runs UNDER the next tile's K-loop.
//
// Why (profiles/r02_gemm_timeline_wide.txt, r02_wide_gemm_ablation.txt, r05_in_graph_vs_isolated.txt): the K-loop of the one-workgroup-per-CU
// kernel of gemm_wide.h already runs at 60 - 72 % of the MFMA-only rate, but at K = 768 it is only 12 steps long and a tile then spends as long
// again OUTSIDE it - 1.5 - 2.2 us filling the LDS ring, 3 - 9 us in the epilogue (fp32 tile through LDS, row-wise pass, stores that find HBM idle
// during every K-loop and saturated during every epilogue because the 256 workgroups move in lock step), and a second round of tiles pays all of
// it again: QKV forward 35.5 us for 2 x 9.7 us of K-loop, FFN-up + GELU 57 us, FFN-down dgrad 53 us.
//
// This kernel keeps the ping-pong K-loop of gemm_wide.h (two groups of four waves, one LOADs fragments while its SIMD partner runs MFMAs, a
// three-stage LDS-DMA ring with counted vmcnt waits) and changes what surrounds it:
//   * persistent: the grid is one workgroup per CU; workgroup (xcd, j) walks tiles j, j + 32, j + 64, ... of its XCD's contiguous run of the
//     tile list (super-row order: an XCD's 32 resident tiles share 4 A panels and 8 B panels in its L2);
//   * the ring never drains: the stages of a workgroup's tiles form ONE sequence, the last two K-steps of a tile issue the first two stages of
//     the next (the DMA goes through buffer descriptors: per-lane offset in one VGPR per operand, tile / K position in an SGPR offset);
//   * the epilogue is register-resident and deferred: at the end of a tile the accumulators move to a second register set and the next tile's
//     first NFM K-steps each finish one 16-row fragment row of it inside the wave's LOAD interval - bias (read from an LDS copy), GELU + saved
//     derivative / multiplier / hash dropout / residual, bf16 packing, a v_permlane16_swap that gives every lane 8 consecutive columns, and
//     16-byte buffer stores (rows past M fall outside the descriptor - the row is part of the VGPR offset, the only part the hardware checks -
//     and are dropped, so every store ISSUES unconditionally and the counted waits of the ring stay exact: loads, LDS-DMA and stores retire in
//     order on gfx9-family vmcnt);
//   * the unit of deferred work is ONE store instruction's worth (a fragment pair = 16 rows x 32 columns per wave, or a lone 16-column fragment):
//     a store costs the CU about as much issue time as three or four LDS-DMA pieces (measured: ~60 - 70 cycles per wave-instruction whatever
//     its width, profiles/r06_persist_experiments.txt), and stores share the vector-memory pipe with the ring's DMA, so the units are dealt
//     evenly over the tile's first nine K-steps instead of riding in bursts;
//   * side inputs (residual or the saved gelu') are fetched one K-step ahead with ordinary buffer loads: hipcc counts LDS-DMA, loads and
//     stores in one in-order vmcnt and places exact counted waits at their first use (checked in the ISA).
// The last tile of a workgroup issues no further stages, fetches all its side inputs during its last two K-steps and runs its units back to
// back after the loop.
#pragma once
#include "gemm_common.h"
#include "gemm_wide.h"
#include <type_traits>

#ifndef MMF_PERSIST_ABL
#define MMF_PERSIST_ABL 0      // ablation builds only (tools/persist_ablate.sh): bit 0 no epilogue stores, bit 1 no epilogue at all (K-loops only), bit 2 all stores into 16 rows
#endif

namespace gemm {

enum { PEPI_PLAIN = 0,   // bias (optional) -> bf16
       PEPI_GELU = 1,    // bias, exact-erf GELU, saved derivative in U (HF BertIntermediate)
       PEPI_SIDE = 2 };  // bias (optional), then ONE bf16 side input: act == 2 multiplier (aux) or [hash dropout +] residual

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int N_, int I_ = 0, typename F>
DEVI void static_for_steps(F&& f) {
    if constexpr (I_ < N_) {
        f(std::integral_constant<int, I_>{});
        static_for_steps<N_, I_ + 1>(f);
    }
}

DEVI __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    // (descriptor inputs made provably wave-uniform: cdna_hip_programming.md T20)
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// 16 / 8 bytes to a buffer with a compile-time cache policy (0 plain, 1 nt, 2 sc1, 3 both; instruction aux bits: 2 = nt, 16 = sc1)
template <int POL>
DEVI void bstore16(u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, ((POL & 1) ? 2 : 0) | ((POL & 2) ? 16 : 0));
}
template <int POL>
DEVI void bstore8(u32x2 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, ((POL & 1) ? 2 : 0) | ((POL & 2) ? 16 : 0));
}

// lanes of 16-lane row g hold columns 4g .. 4g + 3 of fragment j (x) and of fragment j + 1 (y), two packed bf16 pairs each; afterwards every lane
// holds 8 consecutive columns: row 0 -> fragment j columns 0..7, row 1 -> fragment j + 1 columns 0..7, row 2 -> j 8..15, row 3 -> j + 1 8..15
// (v_permlane16_swap: rows 1 / 3 of the first operand swap with rows 0 / 2 of the second).  An involution: applied to data loaded in the
// 8-column layout it returns the fragment layout.
DEVI void pair_swap(u32x2& x, u32x2& y) {
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
    x = u32x2{s0[0], s1[0]};
    y = u32x2{s0[1], s1[1]};
}
DEVI u32x2 pack_bf4(f32x4 v) {
    bf16x4 t;
    t[0] = (bf16)v[0]; t[1] = (bf16)v[1]; t[2] = (bf16)v[2]; t[3] = (bf16)v[3];
    return __builtin_bit_cast(u32x2, t);
}
DEVI f32x4 unpack_bf4(u32x2 u) {
    const bf16x4 t = __builtin_bit_cast(bf16x4, u);
    return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
}

template <int BM_, int BN_, int WGM, int WGN, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_persist_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb,
                                                               int tiles_m, int tiles_n, EpiArgs epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = 3;
    static_assert(WGM * WGN == 8, "eight waves");
    static_assert(BM_ % 64 == 0 && BN_ % 32 == 0, "tile shape");
    constexpr int WTM = BM_ / WGM, WTN = BN_ / WGN, NFM = WTM / 16, NFN = WTN / 16;
    constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE = A_BYTES + B_BYTES, RING = NS * STAGE;
    constexpr int PA = BM_ / 64, PB_FULL = BN_ / 64;
    constexpr bool B_HALF = (BN_ % 64) != 0;
    constexpr int P_LO = PA + PB_FULL + (B_HALF ? 1 : 0), P_HI = PA + PB_FULL;   // DMA wave-instructions per stage: waves 0..3 / 4..7
    constexpr int NPAIR = NFN / 2, NODD = NFN & 1, NQ = NPAIR + NODD;            // units per fragment row
    constexpr int NU = NFM * NQ;                                                  // units per tile (unit u: fragment row u / NQ, column group u % NQ)
    constexpr int NOUT = EPI == PEPI_GELU ? 2 : 1;
    constexpr int ABL = MMF_PERSIST_ABL;
    constexpr int ST_U = (ABL & 3) ? 0 : NOUT;                                    // store instructions per unit (all unconditional)
    constexpr int LD_U = (EPI == PEPI_SIDE && !(ABL & 2)) ? 1 : 0;               // side-input load instructions per unit
    constexpr int NSCH = 9;                                                       // K-steps of a tile that carry units of the previous tile (K-step NSCH lets the last stores fly, two more switch tiles)
    static_assert(NU <= 2 * NSCH, "at most two units per K-step");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nk = K / BK;                     // >= NSCH + 3 (host)

    // ---- this workgroup's tiles: XCD x owns a contiguous run of the tile list, its workgroups stride through it ---------------------------
    const int ntile = tiles_m * tiles_n;
    const int G8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = ntile >> 3, tr = ntile & 7;
    const int xstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int xcount = tq + (xcd < tr ? 1 : 0);
    if (jx >= xcount) return;
    const int n_mine = (xcount - jx + G8 - 1) / G8;

    // ---- bias -> LDS (behind the ring), once; zeros when the call has none ------------------------------------------------------------------
    float* bias_lds = reinterpret_cast<float*>(smem + RING);
    const bool has_bias = epi.bias != nullptr;
    for (int n = tid; n < N; n += 512) bias_lds[n] = has_bias ? epi.bias[n] : 0.f;
    __syncthreads();

    // ---- LDS-DMA through buffer descriptors ------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (unsigned)M * (unsigned)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (unsigned)N * (unsigned)ldb * 2u);
    // chunk swizzle of the row-major LDS image on the SOURCE address: chunk c of row r lands at r * 128 + ((c ^ (r & 7)) << 4)
    const int kch = ((tid & 7) ^ ((tid >> 3) & 7)) * 16;
    const int voA = (tid >> 3) * lda * 2 + kch, voB = (tid >> 3) * ldb * 2 + kch;
    const int rowsB64 = ldb * 128;                                      // bytes of 64 operand rows
    // The hardware checks only voffset (+ immediate) against the descriptor's size, never the SGPR offset, so nothing here relies on the check: a 64-row
    // piece of A that lies past M (M % 64 == 0, the last row of tiles) re-reads the operand's last 64 rows instead (its products are never stored).
    int sB = 0, kA = 0, mA = 0;                                         // B: byte offset of the NEXT stage to issue (tile origin + k); A: k bytes, tile's first row
    auto dma_piece = [&](int q, unsigned char* ist) {
        if (q < PA) {
            int r0 = mA + q * 64;
            r0 = r0 < M ? r0 : M - 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vp)(ist + q * 8192), 16, voA, r0 * lda * 2 + kA, 0, 0);
        }
        else if (q < PA + PB_FULL) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vp)(ist + A_BYTES + (q - PA) * 8192), 16, voB, sB + (q - PA) * rowsB64, 0, 0);
        else if (wave < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vp)(ist + A_BYTES + PB_FULL * 8192), 16, voB, sB + PB_FULL * rowsB64, 0, 0);
    };
    auto issue = [&](int slot) {
        unsigned char* ist = smem + slot * STAGE + wave * 1024;
#pragma unroll
        for (int q = 0; q < P_LO; ++q) dma_piece(q, ist);
        kA += 128; sB += 128;
    };
    auto tile_of = [&](int it, int& tm_, int& tn_) { wide_super_row(xstart + jx + it * G8, tiles_m, tiles_n, tm_, tn_); };

    f32x4 acc[NFM][NFN], prev[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bf16x8 fa[2][NFM], fb[2][NFN];
    int tm, tn;         // the tile in the accumulators
    f32x4 bq[NFN];      // the tile's bias columns (or zeros): the C operand of the tile's first MFMAs - the bias add costs nothing

    const int frow = lane & 15, fswz = lane & 7, g4 = lane >> 4;
    const int a_off = (wm * WTM + frow) * 128, b_off = A_BYTES + (wn * WTN + frow) * 128;
    const int c0 = ((g4) ^ fswz) << 4, c1 = ((4 + g4) ^ fswz) << 4;

    // LOAD: all 2 * (NFM + NFN) fragments of a K-step with the next stage's DMA pieces issued in between (order pinned)
    constexpr int NREAD = 2 * (NFM + NFN);
    auto load_frags = [&](int slot, int islot, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        const unsigned char* st = smem + slot * STAGE;
        unsigned char* ist = smem + islot * STAGE + wave * 1024;
#pragma unroll
        for (int r = 0; r < NREAD; ++r) {
            const int kk = r / (NFM + NFN), f = r % (NFM + NFN);
            const int coff = kk ? c1 : c0;
            if (f < NFN) fb[kk][f] = *reinterpret_cast<const bf16x8*>(st + b_off + f * 2048 + coff);
            else fa[kk][f - NFN] = *reinterpret_cast<const bf16x8*>(st + a_off + (f - NFN) * 2048 + coff);
            if constexpr (ISSUE) {
#pragma unroll
                for (int q = 0; q < P_LO; ++q) {
                    if (((q + 1) * NREAD) / (P_LO + 1) - 1 == r) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma_piece(q, ist);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ISSUE) { kA += 128; sB += 128; }
    };
    auto compute = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < NFM; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], (FIRST && kk == 0) ? bq[j] : acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- deferred epilogue state ----------------------------------------------------------------------------------------------------------------
    // outputs / side input through buffer descriptors: byte offset of (row, column inside the wave's columns) in the VGPR offset - the part the hardware
    // checks against the descriptor's M * ld * 2 bytes, so rows >= M are dropped (stores) or read as zero (loads) -, the wave's first column in the SGPR offset
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(epi.C, (unsigned)M * (unsigned)epi.ldc * 2u);
    const __amdgpu_buffer_rsrc_t rsU = make_rsrc(EPI == PEPI_GELU ? (const void*)epi.U : (const void*)epi.C, (unsigned)M * (unsigned)epi.ldc * 2u);
    const bf16* sidep = epi.resid ? epi.resid : epi.aux;
    const int side_ld = epi.resid ? epi.ldr : epi.ldc;
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(EPI == PEPI_SIDE ? (const void*)sidep : (const void*)epi.C, (unsigned)M * (unsigned)side_ld * 2u);
    const int pol_c = MMF_EPI_POLICY(epi, 0), pol_u = MMF_EPI_POLICY(epi, 1);
    const uint32_t dkey = (EPI == PEPI_SIDE && epi.drop.thr16) ? drop_key(epi.drop) : 0u;
    const bool mul_side = EPI == PEPI_SIDE && epi.act == 2, add_side = EPI == PEPI_SIDE && epi.resid != nullptr;
    // per-lane column byte offsets inside the wave's WTN columns: 8-column layout of a fragment pair, 4-column layout of a single fragment
    const int col_pair = ((g4 & 1) * 16 + (g4 >> 1) * 8) * 2, col_frag = g4 * 4 * 2;
    const int row_c = (wm * WTM + frow) * epi.ldc * 2, row_s = (wm * WTM + frow) * side_ld * 2;     // lane's row inside the tile (fragment row 0)
    int pm0 = M, pn0 = 0;                        // origin of the tile whose accumulators sit in `prev` (M: nothing yet - every store falls out of range)
    u32x4 side[EPI == PEPI_SIDE ? NU : 1];       // side input of unit u in the STORE layout (a lone fragment uses .xy)

    auto tile_bias = [&]() {       // LDS reads of the CURRENT tile's bias columns (the same for every fragment row), ahead of its first K-step's fragment reads
#pragma unroll
        for (int j = 0; j < NFN; ++j) bq[j] = *reinterpret_cast<const f32x4*>(bias_lds + tn * BN_ + wn * WTN + j * 16 + g4 * 4);
    };
    // side input of unit u of the tile at (m0, n0): LD_U buffer load, issued from every lane
    auto side_issue = [&](auto u_tag, int m0, int n0) {
        (void)side;
        if constexpr (LD_U > 0) {
            constexpr int u = decltype(u_tag)::value, i = u / NQ, q = u % NQ;
            const int so = (n0 + wn * WTN) * 2, vr = row_s + (m0 + i * 16) * side_ld * 2;
            if constexpr (q < NPAIR) side[u] = __builtin_amdgcn_raw_buffer_load_b128(rsS, vr + col_pair + q * 64, so, 0);
            else {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsS, vr + col_frag + (NFN - 1) * 32, so, 0);
                side[u] = u32x4{t[0], t[1], 0u, 0u};
            }
        }
    };
    // one fragment of `prev` through the epilogue arithmetic: fp32 in, packed bf16 out (o), the saved gelu' in o2
    auto frag_math = [&](f32x4 v, u32x2 sv, int i, int j, u32x2& o, u32x2& o2) {
        if constexpr (EPI == PEPI_GELU) {
            f32x4 h, gd;
#pragma unroll
            for (int r = 0; r < 4; ++r) { float hh, gg; gelu_erf_both(v[r], hh, gg); h[r] = hh; gd[r] = gg; }
            o = pack_bf4(h); o2 = pack_bf4(gd);
        } else if constexpr (EPI == PEPI_SIDE) {
            const f32x4 s = unpack_bf4(sv);
            if (mul_side) v *= s;
            if (epi.drop.thr16) {
                const uint32_t idx = (uint32_t)(pm0 + wm * WTM + i * 16 + frow) * (uint32_t)N + (uint32_t)(pn0 + wn * WTN + j * 16 + g4 * 4);
                v *= drop_scale4(dkey, idx, epi.drop.thr16, epi.drop.scale);
            }
            if (add_side) v += s;
            o = pack_bf4(v);
        } else {
            o = pack_bf4(v);
        }
    };
    // unit u of `prev`: arithmetic, layout change and its ST_U stores (issued from every lane; the cache policy is an instruction field: one uniform
    // branch per unit picks among the compiled forms - C: plain / nt / sc1 / both; the saved gelu': plain / nt)
    auto unit = [&](auto u_tag) {
        (void)side;
        constexpr int u = decltype(u_tag)::value, i = u / NQ, q = u % NQ;
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
            for (int j = 0; j < NFN; ++j) asm volatile("" ::"v"(prev[i][j]));
            return;
        }
        const int so_c = (pn0 + wn * WTN) * 2;
        const int vr_c = (ABL & 4) ? (pm0 < M ? frow * epi.ldc * 2 : M * epi.ldc * 2) : row_c + (pm0 + i * 16) * epi.ldc * 2;   // (ABL 4: every store of the kernel lands in 16 rows)
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if constexpr (q < NPAIR) {
            u32x2 sx = u32x2{0u, 0u}, sy = u32x2{0u, 0u};
            if constexpr (EPI == PEPI_SIDE) {          // side input back to the fragment layout
                sx = u32x2{side[u][0], side[u][1]}; sy = u32x2{side[u][2], side[u][3]};
                pair_swap(sx, sy);
            }
            u32x2 ox, oy, ux, uy;
            frag_math(prev[i][2 * q], sx, i, 2 * q, ox, ux);
            frag_math(prev[i][2 * q + 1], sy, i, 2 * q + 1, oy, uy);
            pair_swap(ox, oy);
            if constexpr (EPI == PEPI_GELU) pair_swap(ux, uy);
            const u32x4 oc = u32x4{ox[0], ox[1], oy[0], oy[1]}, ou = u32x4{ux[0], ux[1], uy[0], uy[1]};
            const int vo = vr_c + col_pair + q * 64;
            if constexpr ((ABL & 1) != 0) { asm volatile("" ::"v"(oc)); if constexpr (EPI == PEPI_GELU) asm volatile("" ::"v"(ou)); }
            else {
                auto emit = [&](auto pc_tag, auto pu_tag) {
                    bstore16<decltype(pc_tag)::value>(oc, rsC, vo, so_c);
                    if constexpr (EPI == PEPI_GELU) bstore16<decltype(pu_tag)::value>(ou, rsU, vo, so_c);
                };
                if (pol_u) { if (pol_c == 1) emit(I1{}, I1{}); else if (pol_c == 2) emit(I2{}, I1{}); else if (pol_c == 3) emit(I3{}, I1{}); else emit(I0{}, I1{}); }
                else { if (pol_c == 1) emit(I1{}, I0{}); else if (pol_c == 2) emit(I2{}, I0{}); else if (pol_c == 3) emit(I3{}, I0{}); else emit(I0{}, I0{}); }
            }
        } else {
            u32x2 sx = u32x2{0u, 0u};
            if constexpr (EPI == PEPI_SIDE) sx = u32x2{side[u][0], side[u][1]};
            u32x2 ox, ux;
            frag_math(prev[i][NFN - 1], sx, i, NFN - 1, ox, ux);
            const int vo = vr_c + col_frag + (NFN - 1) * 32;
            if constexpr ((ABL & 1) != 0) { asm volatile("" ::"v"(ox)); if constexpr (EPI == PEPI_GELU) asm volatile("" ::"v"(ux)); }
            else {
                auto emit = [&](auto pc_tag, auto pu_tag) {
                    bstore8<decltype(pc_tag)::value>(ox, rsC, vo, so_c);
                    if constexpr (EPI == PEPI_GELU) bstore8<decltype(pu_tag)::value>(ux, rsU, vo, so_c);
                };
                if (pol_u) { if (pol_c == 1) emit(I1{}, I1{}); else if (pol_c == 2) emit(I2{}, I1{}); else if (pol_c == 3) emit(I3{}, I1{}); else emit(I0{}, I1{}); }
                else { if (pol_c == 1) emit(I1{}, I0{}); else if (pol_c == 2) emit(I2{}, I0{}); else if (pol_c == 3) emit(I3{}, I0{}); else emit(I0{}, I0{}); }
            }
        }
    };

    // ---- schedule of the deferred units: unit u rides in K-step u * NSCH / NU of the next tile (one per step, two where NU > NSCH) ---------------
    // first unit of step t, units of step t (t in 0 .. NSCH; NSCH itself and later: none)
    auto u_lo = [](int t) constexpr { return t >= NSCH ? NU : (t * NU + NSCH - 1) / NSCH; };      // smallest u with u * NSCH / NU >= t
    auto u_cnt = [u_lo](int t) constexpr { return t < 0 ? 0 : u_lo(t + 1) - u_lo(t); };

    // ---- one K-step -----------------------------------------------------------------------------------------------------------------------------
    // T in 0 .. NSCH - 1: the LOAD interval also finishes the units of step T of the previous tile.  Vector-memory operations of such an interval, in
    // program order:   [stage s + 2: P pieces] { [side loads of the units of step T + 1] [stores of the units of step T] in the compiler's order }.
    // Stage s + 1 (read next) was issued one interval earlier, ahead of that interval's side loads and stores: before the barrier at most
    //     P + (loads + stores of this interval) + (loads + stores of the previous interval)
    // operations may be outstanding (the side loads' own waits are the compiler's: exact counts at the first use).
    // PRE: this is the workgroup's LAST tile - no further stage is issued (T = -2: K-step nk - 2, T = -3: K-step nk - 1), and K-step nk - 2 fetches the
    // side inputs of ALL units of THIS tile (its epilogue has no K-loop to hide under).
    // WU = false: the workgroup's FIRST tile - there is no previous tile, the K-step carries nothing.
    int slot = 0;
    auto step = [&](auto t_tag, auto first_tag, auto wu_tag) {
        constexpr int T = (decltype(wu_tag)::value || decltype(t_tag)::value < 0) ? decltype(t_tag)::value : -1;
        constexpr bool ISSUE = T > -2;
        const int islot = slot == 0 ? NS - 1 : slot - 1;
        if constexpr (decltype(first_tag)::value) tile_bias();
        load_frags(slot, islot, std::integral_constant<bool, ISSUE>{});
        if constexpr (T >= 0 && T < NSCH) {
            constexpr int U0 = u_lo(T), UN = u_cnt(T), VN = u_cnt(T + 1), VP = u_cnt(T - 1);
            static_for_steps<VN>([&](auto k) { side_issue(std::integral_constant<int, u_lo(T + 1) + decltype(k)::value>{}, pm0, pn0); });
            static_for_steps<UN>([&](auto k) { unit(std::integral_constant<int, U0 + decltype(k)::value>{}); });
            constexpr int OPS = VN * LD_U + UN * ST_U + (T == 0 ? u_cnt(0) * LD_U : UN * LD_U + VP * ST_U);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wave < 4) wait_vm<P_LO + OPS>(); else wait_vm<P_HI + OPS>();
        } else if constexpr (T == NSCH) {           // (the last units' stores keep flying through this one)
            constexpr int OPS = u_cnt(NSCH - 1) * ST_U;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wave < 4) wait_vm<P_LO + OPS>(); else wait_vm<P_HI + OPS>();
        } else if constexpr (T == -2) {             // last tile, K-step nk - 2: stage nk - 1 must land; behind it only the side prefetch
            static_for_steps<NU>([&](auto k) { side_issue(k, tm * BM_, tn * BN_); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vm<NU * LD_U>();
        } else if constexpr (T == -3) {             // last tile, K-step nk - 1: everything it reads has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wave < 4) wait_vm<P_LO>(); else wait_vm<P_HI>();
        }
        bar();
        compute(first_tag);
        bar();
        slot = slot + 1 == NS ? 0 : slot + 1;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using PLAINSTEP = std::integral_constant<int, -1>;

    // ---- prologue: the first tile's stages 0 and 1 ---------------------------------------------------------------------------------------------
    tile_of(0, tm, tn);
    mA = tm * BM_; kA = 0; sB = tn * BN_ * ldb * 2;
    issue(0);
    issue(1);
    if (wave < 4) wait_vm<P_LO>(); else wait_vm<P_HI>();
    bar();
    if (grp == 1) bar();

    for (int it = 0; it < n_mine; ++it) {
        const bool last = it + 1 == n_mine;
        // K-steps 0 .. NSCH - 1: the previous tile's units ride along
        // (with a side input the first tile runs the same code on out-of-range rows - pm0 == M - instead: a second copy of the steps costs the
        //  256 x 128 form its last free registers)
        if (EPI != PEPI_SIDE && it == 0) {
            static_for_steps<NSCH + 1>([&](auto c) {
                constexpr int C = decltype(c)::value;
                if constexpr (C == 0) step(c, T_{}, F_{}); else step(c, F_{}, F_{});
            });
        } else {
            static_for_steps<u_cnt(0)>([&](auto k) { side_issue(k, pm0, pn0); });          // (ahead of step 0's pieces)
            static_for_steps<NSCH + 1>([&](auto c) {
                constexpr int C = decltype(c)::value;
                if constexpr (C == 0) step(c, T_{}, T_{}); else step(c, F_{}, T_{});
            });
        }
        for (int t = NSCH + 1; t < nk - 2; ++t) step(PLAINSTEP{}, F_{}, T_{});
        // the last two K-steps issue the first two stages of the next tile
        int nm = tm, nn = tn;
        if (!last) {
            tile_of(it + 1, nm, nn);
            mA = nm * BM_; kA = 0; sB = nn * BN_ * ldb * 2;
            step(PLAINSTEP{}, F_{}, T_{});
            step(PLAINSTEP{}, F_{}, T_{});
        } else {
            step(std::integral_constant<int, -2>{}, F_{}, T_{});
            step(std::integral_constant<int, -3>{}, F_{}, T_{});
        }
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int j = 0; j < NFN; ++j) prev[i][j] = acc[i][j];
        pm0 = tm * BM_; pn0 = tn * BN_;
        tm = nm; tn = nn;
    }
    if (grp == 0) bar();
    // ---- the last tile's epilogue, back to back (its side inputs were fetched during its last two K-steps) -----------------------------------------
    static_for_steps<NU>([&](auto u) { unit(u); });
}

}  // namespace gemm
