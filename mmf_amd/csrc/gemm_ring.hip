// mmf_amd :: the fast path of mmf_gemm_bf16 — bf16 x bf16 GEMM for gfx950 built around LDS-DMA.
//
// One 256-thread workgroup per CU (4 waves, one per SIMD) owns a 128 x BN output tile (BN = 96 / 128 /
// 192, picked per shape so that the tile count fills the 256 CUs in whole rounds) and walks K in steps
// of 64 through a RING of LDS stages (5 x 32 KiB at BN = 128: all 160 KiB of the CU).  `global_load_lds`
// DMAs stream the operand tiles straight from L2/HBM into the ring up to NST-1 stages ahead; the only
// synchronisation per K-step is one counted `s_waitcnt vmcnt(n)` (never 0 in steady state) + one raw
// `s_barrier`, so ~128 KiB of loads are in flight per CU while the MFMAs run — the loop is paced by
// the matrix pipe, not by memory latency (the 2-stage kernel in gemm.hip was latency-bound at ~40 %).
//
// Tile order is L2-aware: block b runs on XCD b % 8, every XCD walks a contiguous run of a list that
// is ordered in 8-row super-rows, column-major inside a super-row, so the 32 tiles resident on an XCD
// share 8 A row-panels and 4 B panels that fit its 4 MiB L2.
//
// Same LDS images, fragment reads, MFMA and epilogue as gemm.hip (gemm_common.h).
#include "gemm_common.h"

using namespace gemm;

namespace {

template <int N> DEVI void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool A_KMAJOR, bool B_KMAJOR, int BN_>
struct RingCfg {
    static constexpr int A_BYTES = 16384;
    // k-major B images always use 256-byte rows (128 columns); row-major B images have BN_ rows of 128 B
    static constexpr int B_BYTES = B_KMAJOR ? 16384 : BN_ * 128;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int NST = (163840 / STAGE) > 6 ? 6 : (163840 / STAGE);
    static constexpr int A_DMA = 4;                             // DMA instructions per wave per stage
    static constexpr int B_DMA = B_KMAJOR ? 4 : BN_ / 32;
    static constexpr int N_DMA = A_DMA + B_DMA;
    static constexpr int NFN = BN_ / 32;                        // 16-column fragments per wave along N
};

// Issue the DMAs of one ring stage.  Row-major B with BN_ rows uses BN_/32 instructions per wave.
template <bool A_KMAJOR, bool B_KMAJOR, int BN_>
DEVI void issue_stage(const bf16* A, const bf16* B, int lda, int ldb, int m0, int n0, int k0, int N, unsigned char* st, int tid) {
    using C = RingCfg<A_KMAJOR, B_KMAJOR, BN_>;
    const int wave = tid >> 6;
    // ---- A (128 rows x 64 k, or 64 k-rows x 128 cols)
    if (!A_KMAJOR) {
        const int sw = (tid >> 3) & 7;
        const bf16* src = A + (size_t)(m0 + (tid >> 3)) * lda + k0 + ((tid & 7) ^ sw) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(src + (size_t)32 * i * lda), (lds_vp)(st + i * 4096 + wave * 1024), 16, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = (tid >> 4) + 16 * i;
            const int logical = ((tid & 15) - (rot_kmajor(kr) >> 4)) & 15;
            __builtin_amdgcn_global_load_lds((glb_vp)(A + (size_t)(k0 + kr) * lda + m0 + logical * 8),
                                             (lds_vp)(st + i * 4096 + wave * 1024), 16, 0, 0);
        }
    }
    // ---- B
    unsigned char* sb = st + C::A_BYTES;
    if (!B_KMAJOR) {
        const int sw = (tid >> 3) & 7;
        const bf16* src = B + (size_t)(n0 + (tid >> 3)) * ldb + k0 + ((tid & 7) ^ sw) * 8;
#pragma unroll
        for (int i = 0; i < C::B_DMA; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(src + (size_t)32 * i * ldb), (lds_vp)(sb + i * 4096 + wave * 1024), 16, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = (tid >> 4) + 16 * i;
            const int logical = ((tid & 15) - (rot_kmajor(kr) >> 4)) & 15;
            const bf16* src = B + (size_t)(k0 + kr) * ldb + n0 + logical * 8;
            if (BN_ < 128 && logical * 8 >= BN_) src = reinterpret_cast<const bf16*>(&g_zero16);  // columns beyond the tile
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(sb + i * 4096 + wave * 1024), 16, 0, 0);
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, int BN_>
__global__ __launch_bounds__(256, 1) void gemm_ring_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N,
                                                            int K, int lda, int ldb, int tiles_m, int tiles_n, int splits,
                                                            EpiArgs epi) {
    using C = RingCfg<A_KMAJOR, B_KMAJOR, BN_>;
    constexpr int NST = C::NST, ND = C::N_DMA, NFN = C::NFN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- L2-aware tile order -------------------------------------------------------------------
    const int ntile = tiles_m * tiles_n;
    const int nblk = ntile * splits;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int split = bid / ntile;
    bid -= split * ntile;
    int tile_m, tile_n;
    {
        const int per_sr = 8 * tiles_n;
        const int sr = bid / per_sr, rem = bid - sr * per_sr;
        const int h = min(8, tiles_m - sr * 8);
        tile_n = rem / h;
        tile_m = sr * 8 + (rem - tile_n * h);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN_;

    f32x4 acc[4][NFN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = K / BK;
    const int kt0 = (int)((long)nk_all * split / splits), kt1 = (int)((long)nk_all * (split + 1) / splits);
    const int nk = kt1 - kt0;

    // ---- prologue: fill NST-1 stages --------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue_stage<A_KMAJOR, B_KMAJOR, BN_>(A, B, lda, ldb, m0, n0, (kt0 + s) * BK, N, smem + s * C::STAGE, tid);

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stages issued after kt and still allowed in flight while we wait for stage kt
        const int ahead = min(NST - 2, nk - 1 - kt);
        if (ahead >= 4) wait_vmcnt<4 * ND>();
        else if (ahead == 3) wait_vmcnt<3 * ND>();
        else if (ahead == 2) wait_vmcnt<2 * ND>();
        else if (ahead == 1) wait_vmcnt<1 * ND>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();        // every wave's share of stage kt has landed; stage kt-1 is no longer read
        __builtin_amdgcn_sched_barrier(0);
        {
            const int kf = kt + NST - 1;     // refill the slot that stage kt-1 occupied
            if (kf < nk) {
                int fs = slot + NST - 1; if (fs >= NST) fs -= NST;
                issue_stage<A_KMAJOR, B_KMAJOR, BN_>(A, B, lda, ldb, m0, n0, (kt0 + kf) * BK, N, smem + fs * C::STAGE, tid);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* la = smem + slot * C::STAGE;
        const unsigned char* lb = la + C::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[4], fb[NFN];
#pragma unroll
            for (int f = 0; f < 4; ++f) fa[f] = read_frag<A_KMAJOR>(la, wm * 64, f, kk, lane);
#pragma unroll
            for (int f = 0; f < NFN; ++f) fb[f] = read_frag<B_KMAJOR>(lb, wn * (BN_ / 2), f, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        slot = (slot + 1 == NST) ? 0 : slot + 1;
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NFN; ++j) {
            const int n = n0 + wn * (BN_ / 2) + j * 16 + (lane >> 4) * 4;
            epilogue4(epi, m, n, acc[i][j], split);
        }
    }
}

template <bool AK, bool BKM, int BN_>
int launch_ring(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    using C = RingCfg<AK, BKM, BN_>;
    const int tm = d->M / BM, tn = d->N / BN_;
    const int splits = e.splits > 1 ? e.splits : 1;
    const int lds = C::NST * C::STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<AK, BKM, BN_>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (err != hipSuccess) { mmf_amd_set_error(hipGetErrorString(err)); return 2; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<AK, BKM, BN_>), dim3(tm * tn * splits), dim3(256), lds, s,
                       reinterpret_cast<const bf16*>(d->A), reinterpret_cast<const bf16*>(d->B), d->M, d->N, d->K, d->lda,
                       d->ldb, tm, tn, splits, e);
    MMF_CHECK_LAUNCH();
    return 0;
}

// rounds of 256 CUs needed by `tiles` tiles, as a cost: rounds * tile_area (smaller = better)
inline double cost(int M, int N, int bn, int splits) {
    const long tiles = (long)(M / BM) * (N / bn) * splits;
    const long rounds = (tiles + 255) / 256;
    // bigger tiles amortise operand traffic better: mild preference expressed as a per-tile overhead term
    return (double)rounds * (bn + 24.0) / splits;
}

}  // namespace

// Returns -1 when the shape / layout is not handled by the ring kernel (caller falls back to gemm.hip).
int mmf_gemm_ring_dispatch(const mmf_gemm_desc* d, const gemm::EpiArgs& e, hipStream_t s) {
    if (d->a_f32 || d->b_f32) return -1;
    if ((d->M % BM) || (d->K % BK) || d->K < 2 * BK) return -1;
    if (d->a_kmajor && !d->b_kmajor) return -1;
    const int splits = e.splits > 1 ? e.splits : 1;
    int best = 0; double bc = 1e30;
    const int cands[3] = {128, 96, 192};
    for (int c = 0; c < 3; ++c) {
        const int bn = cands[c];
        if (d->N % bn) continue;
        if (bn == 192 && d->b_kmajor) continue;
        if (bn != 128 && d->a_kmajor) continue;
        const double cc = cost(d->M, d->N, bn, splits);
        if (cc < bc) { bc = cc; best = bn; }
    }
    if (!best) return -1;
    if (!d->a_kmajor && !d->b_kmajor) {
        if (best == 96) return launch_ring<false, false, 96>(d, e, s);
        if (best == 192) return launch_ring<false, false, 192>(d, e, s);
        return launch_ring<false, false, 128>(d, e, s);
    }
    if (!d->a_kmajor && d->b_kmajor) {
        if (best == 96) return launch_ring<false, true, 96>(d, e, s);
        return launch_ring<false, true, 128>(d, e, s);
    }
    return launch_ring<true, true, 128>(d, e, s);
}
