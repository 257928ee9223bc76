// mmf_amd :: bf16 x bf16 GEMM, BK = 32 form (fast path for full tiles).
//
// Same math, LDS-DMA staging, MFMA and epilogue as gemm.hip, re-balanced after profiling showed that kernel
// LDS-bound (per CU and K = 64: 768 cycles of fragment reads + ~512 cycles of DMA writes against 1024 MFMA
// cycles per SIMD, profiles/r01_*).  Here a workgroup is 4 waves with 64 x 64 (or 64 x 48) wave tiles — 1/3 fewer
// fragment bytes per FLOP than the 64 x 32 tiles of the 8-wave form — and the K step is 32, so a workgroup
// needs only 33 KiB of LDS and FOUR workgroups share a CU: 16 waves per CU (4 per SIMD) keep hiding LDS / DMA
// latency, and their phases (DMA issue, MFMA, epilogue) interleave freely.
//
// LDS images (8 KiB per operand per stage):
//   row operand     : [128 rows][4 chunks of 16 B], chunk ^= PI[(row >> 2) & 3], PI = {0,3,2,1}: the 16 lanes of
//                     every ds_read_b128 service group hit 16 distinct 16-byte slots
//   k-major operand : [32 k-rows][256 B], same 32-byte rotation as gemm.hip, ds_read_b64_tr_b16
// Epilogue: the fp32 tile goes through LDS in two 64-row halves (33 KiB) and is written row-wise (epilogue8).
#include "gemm_common.h"

using namespace gemm;

namespace {

constexpr int BK32 = 32;
constexpr int OPER32 = 8192;

DEVI int pi4(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }   // {0,3,2,1}[(row>>2)&3]

template <bool KMAJOR>
DEVI void dma32(const bf16* base, int ld, int r0, int k0, int rlimit, unsigned char* lds, int tid) {
    const int wave = tid >> 6;
    if (!KMAJOR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 2) + 64 * i;
            const int logical = (tid & 3) ^ pi4(row);
            const bf16* src = (row < rlimit) ? base + (size_t)(r0 + row) * ld + k0 + logical * 8 : reinterpret_cast<const bf16*>(&g_zero16);
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + i * 4096 + wave * 1024), 16, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kr = (tid >> 4) + 16 * i;
            const int logical = ((tid & 15) - (rot_kmajor(kr) >> 4)) & 15;
            const bf16* src = (logical * 8 < rlimit) ? base + (size_t)(k0 + kr) * ld + r0 + logical * 8 : reinterpret_cast<const bf16*>(&g_zero16);
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + i * 4096 + wave * 1024), 16, 0, 0);
        }
    }
}

template <bool KMAJOR>
DEVI bf16x8 frag32(const unsigned char* lds, int wrow0, int f, int lane) {
    if (!KMAJOR) {
        const int row = wrow0 + f * 16 + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(lds + row * 64 + (((lane >> 4) ^ pi4(row)) << 4));
    } else {
        return read_frag<true>(lds, wrow0, f, 0, lane);
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, int BN_>
__global__ __launch_bounds__(256, 4) void gemm_k32_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K,
                                                           int lda, int ldb, int tiles_m, int tiles_n, int splits, EpiArgs epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WTN = BN_ / 2, NFN = WTN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

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

    const int nk_all = K / BK32;
    const int kt0 = (int)((long)nk_all * split / splits), kt1 = (int)((long)nk_all * (split + 1) / splits);
    const int nk = kt1 - kt0;

    dma32<A_KMAJOR>(A, lda, m0, kt0 * BK32, 128, smem, tid);
    dma32<B_KMAJOR>(B, ldb, n0, kt0 * BK32, BN_, smem + OPER32, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const unsigned char* la = smem + cur * 2 * OPER32;
        const unsigned char* lb = la + OPER32;
        unsigned char* na = smem + (cur ^ 1) * 2 * OPER32;
        const int kn = kt0 + ((kt + 1 < nk) ? kt + 1 : kt);
        dma32<A_KMAJOR>(A, lda, m0, kn * BK32, 128, na, tid);
        dma32<B_KMAJOR>(B, ldb, n0, kn * BK32, BN_, na + OPER32, tid);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 fa[4], fb[NFN];
#pragma unroll
        for (int f = 0; f < 4; ++f) fa[f] = frag32<A_KMAJOR>(la, wm * 64, f, lane);
#pragma unroll
        for (int f = 0; f < NFN; ++f) fb[f] = frag32<B_KMAJOR>(lb, wn * WTN, f, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NFN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue in two 64-row halves through LDS (see gemm.hip)
    constexpr int CLD = BN_ + 4;
    constexpr int SEG = BN_ / 8;
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NFN; ++j) {
                    const int row = i * 16 + (lane & 15), col = wn * WTN + j * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<float4*>(cs + row * CLD + col) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * SEG; idx += 256) {
            const int row = idx / SEG, seg = idx - row * SEG;
            epilogue8(epi, m0 + half * 64 + row, n0 + seg * 8, load_f8(cs + row * CLD + seg * 8), split);
        }
        __syncthreads();
    }
}

template <bool AK, bool BKM, int BN_>
int launch32(const mmf_gemm_desc* d, const EpiArgs& e, hipStream_t s) {
    const int tm = d->M / BM, tn = d->N / BN_;
    const int splits = e.splits > 1 ? e.splits : 1;
    constexpr int cstage = 64 * (BN_ + 4) * (int)sizeof(float);
    constexpr int lds_bytes = cstage > 4 * OPER32 ? cstage : 4 * OPER32;
    hipLaunchKernelGGL((gemm_k32_kernel<AK, BKM, BN_>), dim3(tm * tn * splits), dim3(256), lds_bytes, s, reinterpret_cast<const bf16*>(d->A),
                       reinterpret_cast<const bf16*>(d->B), d->M, d->N, d->K, d->lda, d->ldb, tm, tn, splits, e);
    MMF_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// -1: shape not handled here
int mmf_gemm_k32_dispatch(const mmf_gemm_desc* d, const gemm::EpiArgs& e, hipStream_t s) {
    if (d->a_f32 || d->b_f32) return -1;
    if ((d->M % BM) || (d->K % BK32) || d->K < 2 * BK32) return -1;
    if (d->a_kmajor && !d->b_kmajor) return -1;
    const bool n128 = (d->N % 128) == 0, n96 = (d->N % 96) == 0;
    if (!n128 && !n96) return -1;
    bool use96 = false;
    if (n96 && !d->a_kmajor) {
        // 1024 workgroup slots (256 CUs x 4): fewer rounds wins; ties go to the larger tile
        const long tm = d->M / BM;
        const long r128 = n128 ? (tm * (d->N / 128) + 1023) / 1024 : (1L << 40), r96 = (tm * (d->N / 96) + 1023) / 1024;
        use96 = !n128 || (r96 * 96 < r128 * 128);
    } else if (!n128) return -1;
    if (!d->a_kmajor && !d->b_kmajor) return use96 ? launch32<false, false, 96>(d, e, s) : launch32<false, false, 128>(d, e, s);
    if (!d->a_kmajor && d->b_kmajor) return use96 ? launch32<false, true, 96>(d, e, s) : launch32<false, true, 128>(d, e, s);
    return launch32<true, true, 128>(d, e, s);
}
