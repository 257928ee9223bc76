// mmf_amd :: device code shared by the GEMM kernels (LDS images, LDS-DMA staging, MFMA fragment reads,
// fused epilogue).  See gemm.hip for the layout description.
#pragma once
#include "common.h"
#include "mmf_amd.h"

namespace gemm {


constexpr int BM = 128, BN = 128, BK = 64;
constexpr int OPER_BYTES = 16384;  // one operand tile in LDS

struct EpiArgs {
    void* C;            // output
    int ldc;
    int out_f32;        // 1: C is float, else bf16
    float beta;         // C = acc + beta*C   (fp32 output only; gradient accumulation)
    const float* bias;  // [N] or null
    const float* coladd;  // [N] extra per-column addend or null
    const float* rowtab;  // optional table gathered per output row: rowtab[rowidx[m]*rowtab_ld + n]
    const int64_t* rowidx;
    int rowtab_ld;
    int act;            // 0 none | 1 gelu(acc) (U := gelu'(acc) if U != null) | 2 acc * aux | 3 tanh(acc) | 4 acc * (1 - aux^2)
    bf16* U;
    const bf16* aux;    // same indexing as C (ldc)
    const bf16* resid;  // added after activation/dropout, ldr
    int ldr;
    DropoutCfg drop;    // applied to (acc + bias) before the residual add; index = m*N + n
    int grp_in, grp_pad, grp_off;  // output row remap: row = m + (m / grp_in) * grp_pad + grp_off
    int M, N;
    long slab_stride;   // split-K: split s writes its partial tile to C + s * slab_stride (fp32 slabs)
    int splits;
    int rowsum_col;     // >= 0: also write per-row sums of A (bias-gradient partials) behind the split-K slabs; -1: off
    float* rowsum_direct;  // no split-K: the row sums go straight here ([M] fp32) instead of behind the slabs
    int nt;             // non-temporal stores: bit 0 the bf16 output C, bit 1 the saved gelu' (U), bit 2 fp32 outputs (MMF_TUN_EPI_NT)
    int sc1;            // write-through stores, same bits (the per-site rule: MMF_TUN_SC1_SITE)
};

// Timeline probe (development aid; off unless mmf_gemm_set_probe was called).  One record of 8 u64 per workgroup:
// {launch id << 32 | block id, HW_ID | XCC_ID << 32, t_entry, t_first_stage_landed, t_kloop_done, t_staged, t_stores_done, tile}.
// Slot 0 of the buffer is the allocation counter.  Timestamps are s_memrealtime ticks (100 MHz, one counter per device).
struct Probe { unsigned long long* buf; unsigned cap; unsigned launch; };

DEVI int rot_kmajor(int krow) { return 32 * ((krow & 3) + 4 * ((krow >> 3) & 1)); }

// ---- global -> register staging -------------------------------------------------------------
template <typename T, int NP = 4>
struct Stage;  // NP = 1024 / threads chunks of 8 elements per thread per operand tile

template <int NP>
struct Stage<bf16, NP> {
    uint4 v[NP];
    template <bool RAGGED>
    DEVI void load(int i, const bf16* p, bool ok) {
        if (RAGGED && !ok) v[i] = make_uint4(0, 0, 0, 0);
        else v[i] = *reinterpret_cast<const uint4*>(p);
    }
};
template <int NP>
struct Stage<float, NP> {
    uint4 v[NP];
    template <bool RAGGED>
    DEVI void load(int i, const float* p, bool ok) {
        if (RAGGED && !ok) { v[i] = make_uint4(0, 0, 0, 0); return; }
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        bf16x8 r;
        r[0] = (bf16)a.x; r[1] = (bf16)a.y; r[2] = (bf16)a.z; r[3] = (bf16)a.w;
        r[4] = (bf16)b.x; r[5] = (bf16)b.y; r[6] = (bf16)b.z; r[7] = (bf16)b.w;
        v[i] = __builtin_bit_cast(uint4, r);
    }
};

// Issue the global loads of one operand tile (rows [r0, r0+128) x k [k0, k0+64)).
template <typename T, bool KMAJOR, bool RAGGED, int NTH = 256>
DEVI void stage_load(Stage<T, 1024 / NTH>& st, const T* base, int ld, int r0, int k0, int R, int K, int tid) {
    if (!KMAJOR) {
        const int kc = tid & 7;
        const int k = k0 + kc * 8;
        const bool kok = k < K;
#pragma unroll
        for (int i = 0; i < 1024 / NTH; ++i) {
            int row = r0 + (tid >> 3) + (NTH / 8) * i;
            if (RAGGED) row = row < R ? row : R - 1;
            st.template load<RAGGED>(i, base + (size_t)row * ld + k, kok);
        }
    } else {
        const int nc = tid & 15;
        const int col = r0 + nc * 8;
        const bool cok = col < R;
#pragma unroll
        for (int i = 0; i < 1024 / NTH; ++i) {
            const int krow = k0 + (tid >> 4) + (NTH / 16) * i;
            const bool ok = cok && (krow < K);
            st.template load<RAGGED>(i, base + (size_t)(RAGGED ? (ok ? krow : 0) : krow) * ld + (RAGGED ? (ok ? col : 0) : col), ok);
        }
    }
}

template <typename T, bool KMAJOR, int NTH = 256>
DEVI void stage_store(const Stage<T, 1024 / NTH>& st, unsigned char* lds, int tid) {
    if (!KMAJOR) {
        const int kc = tid & 7;
        const int sw = (tid >> 3) & 7;
#pragma unroll
        for (int i = 0; i < 1024 / NTH; ++i) {
            const int row = (tid >> 3) + (NTH / 8) * i;
            *reinterpret_cast<uint4*>(lds + row * 128 + ((kc ^ sw) << 4)) = st.v[i];
        }
    } else {
        const int nc = tid & 15;
#pragma unroll
        for (int i = 0; i < 1024 / NTH; ++i) {
            const int krow = (tid >> 4) + (NTH / 16) * i;
            *reinterpret_cast<uint4*>(lds + krow * 256 + ((nc * 16 + rot_kmajor(krow)) & 255)) = st.v[i];
        }
    }
}

// ---- global -> LDS direct (LDS-DMA, global_load_lds_dwordx4) -------------------------------------
// One wave-instruction moves 64 x 16 B = 1 KiB to LDS at (wave-uniform base) + lane * 16, so the LDS
// image is lane-linear and the swizzle / rotation is applied to each lane's SOURCE address instead
// (the same involution the fragment reads apply).  Out-of-range chunks read a 16-byte zero buffer.
static __device__ uint4 g_zero16;

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// I0 / I1 select a sub-range of the 1024 / NTH wave-instructions of a tile (a schedule may spread them over several phases).
template <bool KMAJOR, bool RAGGED, int NTH = 256, int I0 = 0, int I1 = 1024 / NTH>
DEVI void stage_dma(const bf16* base, int ld, int r0, int k0, int R, int K, unsigned char* lds, int tid) {
    const int wave = tid >> 6;
    if (!KMAJOR) {
        const int sw = (tid >> 3) & 7;
        const int k = k0 + ((tid & 7) ^ sw) * 8;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            int row = r0 + (tid >> 3) + (NTH / 8) * i;
            if (RAGGED) row = row < R ? row : R - 1;
            const bf16* src = base + (size_t)row * ld + k;
            if (RAGGED && k >= K) src = reinterpret_cast<const bf16*>(&g_zero16);
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + i * (NTH * 16) + wave * 1024), 16, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            const int kr = (tid >> 4) + (NTH / 16) * i;
            const int logical = ((tid & 15) - (rot_kmajor(kr) >> 4)) & 15;
            const int col = r0 + logical * 8;
            const int krow = k0 + kr;
            const bf16* src = base + (size_t)krow * ld + col;
            if (RAGGED && (col >= R || krow >= K)) src = reinterpret_cast<const bf16*>(&g_zero16);
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + i * (NTH * 16) + wave * 1024), 16, 0, 0);
        }
    }
}

template <typename T> struct is_bf16 { static constexpr bool value = false; };
template <> struct is_bf16<bf16> { static constexpr bool value = true; };

// ---- LDS -> MFMA fragment ---------------------------------------------------------------------
// Fragment f (16 rows starting at wrow0 + 16 f), k sub-step kk (32 k each). Lane l holds tile row
// (l & 15) and the 8 reduction slots of lane group g = l >> 4; both layouts use reduction rows
// kk*32 + 8g + [0,8) for group g, so a row operand and a k-major operand pair up correctly.
template <bool KMAJOR>
DEVI bf16x8 read_frag(const unsigned char* lds, int wrow0, int f, int kk, int lane) {
    if (!KMAJOR) {
        const int row = wrow0 + f * 16 + (lane & 15);
        const int chunk = kk * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
    } else {
        const int g = lane >> 4, p = lane & 15;
        const int col = wrow0 + f * 16 + (p & 3) * 4;
        const int rot = 32 * ((p >> 2) + 4 * (g & 1));
        const int krow0 = kk * 32 + 8 * g + (p >> 2);
        const int cb = (col * 2 + rot) & 255;
        typedef s16x4 __attribute__((address_space(3))) * lds_p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + krow0 * 256 + cb));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + (krow0 + 4) * 256 + cb));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

// ---- epilogue -----------------------------------------------------------------------------------
DEVI f32x4 load_f4(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    return f32x4{t.x, t.y, t.z, t.w};
}
typedef float f32x8 __attribute__((ext_vector_type(8)));

DEVI f32x8 load_f8(const float* p) {
    const f32x4 a = load_f4(p), b = load_f4(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
DEVI f32x8 load_bf8(const bf16* p) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
    f32x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (float)t[i];
    return r;
}
// Non-temporal (`nt`): an epilogue output is never re-read by the kernel that writes it, so it can stream past the operand panels an
// XCD keeps in its L2 instead of evicting them.  Measured (profiles/r03_nt_stores_ab.txt, same box, twice) with every output stored
// that way: step 8.62 -> 8.41 ms, FETCH_SIZE of the 128 x 128 family 85.4 -> 75.4 MB per launch - but the NEXT kernel then finds its
// input further away (attention forward 28 -> 39 us behind a non-temporally stored Q|K|V), so which outputs get it is a per-output
// choice (EpiArgs::nt, MMF_TUN_EPI_NT).
// A 16-byte store in one of the cache policies: 0 plain, 1 non-temporal (`nt`), 2 write-through (`sc1`: the line leaves the XCD's L2 as it is
// written, so the end-of-kernel release has nothing of it left to write back), 3 both.
DEVI void store16_policy(void* p, u32x4 v, int policy) {
    if (policy == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else if (policy == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if (policy == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(p) = v;
}
DEVI void store_bf8(bf16* p, f32x8 v, int policy) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    store16_policy(p, __builtin_bit_cast(u32x4, t), policy);
}
// policy of output `bit` (0 the bf16 C, 1 the saved gelu' U, 2 fp32 C) from EpiArgs::nt (bits 0-2: nt) and EpiArgs::sc1 (bits 0-2: sc1)
#define MMF_EPI_POLICY(e, bit) ((((e).nt >> (bit)) & 1) | ((((e).sc1 >> (bit)) & 1) << 1))

// Epilogue of one output row segment: 8 consecutive columns n..n+7 of row m (fp32 accumulators staged through LDS
// so that every global access of the epilogue is a full 16/32-byte-per-lane, row-contiguous transaction).
DEVI void epilogue8(const EpiArgs& e, int m, int n, f32x8 v, int split) {
    if (m >= e.M || n >= e.N) return;
    const bool full = (n + 8 <= e.N);
    bool ok[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) ok[r] = (n + r) < e.N;
    if (e.bias) {
        if (full) v += load_f8(e.bias + n);
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) v[r] += e.bias[n + r];
        }
    }
    if (e.coladd) {
        if (full) v += load_f8(e.coladd + n);
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) v[r] += e.coladd[n + r];
        }
    }
    if (e.rowtab) {
        const float* t = e.rowtab + (size_t)e.rowidx[m] * e.rowtab_ld + n;
        if (full && ((e.rowtab_ld & 3) == 0)) v += load_f8(t);
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) v[r] += t[r];
        }
    }
    int orow = m;
    if (e.grp_in > 0) orow = m + (m / e.grp_in) * e.grp_pad + e.grp_off;
    const size_t off = (size_t)orow * e.ldc + n + (size_t)split * e.slab_stride;
    const bool vec = full && ((e.ldc & 7) == 0);
    if (e.act == 1) {
        // HF BertIntermediate: h = gelu(v).  U receives gelu'(v) (NOT v): the backward epilogue (act == 2) then
        // only multiplies, and the pre-activation itself is never needed again.
        f32x8 gd;
#pragma unroll
        for (int r = 0; r < 8; ++r) { float hh, gg; gelu_erf_both(v[r], hh, gg); v[r] = hh; gd[r] = gg; }
        if (e.U) {
            if (vec) store_bf8(e.U + off, gd, MMF_EPI_POLICY(e, 1));
            else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (ok[r]) e.U[off + r] = (bf16)gd[r];
            }
        }
    } else if (e.act == 2) {
        if (vec) v *= load_bf8(e.aux + off);
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) v[r] *= (float)e.aux[off + r];
        }
    } else if (e.act == 3) {   // HF BertPooler: tanh
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = tanhf(v[r]);
    } else if (e.act == 4) {   // backward of tanh: v *= 1 - y^2, y = saved output (aux)
        f32x8 y = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (vec) y = load_bf8(e.aux + off);
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) y[r] = (float)e.aux[off + r];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= 1.f - y[r] * y[r];
    }
    if (e.drop.thr16) {
        const uint32_t idx = (uint32_t)m * (uint32_t)e.N + (uint32_t)n;
        const uint32_t dkey = drop_key(e.drop);
        if ((idx & 3u) == 0) {
            const f32x4 s0 = drop_scale4(dkey, idx, e.drop.thr16, e.drop.scale);
            const f32x4 s1 = drop_scale4(dkey, idx + 4, e.drop.thr16, e.drop.scale);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] *= s0[r]; v[r + 4] *= s1[r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= drop_scale1(dkey, idx + r, e.drop.thr16, e.drop.scale);
        }
    }
    if (e.resid) {
        const size_t roff = (size_t)orow * e.ldr + n;
        if (full && ((e.ldr & 7) == 0)) v += load_bf8(e.resid + roff);
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) v[r] += (float)e.resid[roff + r];
        }
    }
    if (e.out_f32) {
        float* C = reinterpret_cast<float*>(e.C) + off;
        if (full && ((e.ldc & 3) == 0)) {
            if (e.beta != 0.f) v += e.beta * load_f8(C);
            store16_policy(C, __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), MMF_EPI_POLICY(e, 2));
            store16_policy(C + 4, __builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), MMF_EPI_POLICY(e, 2));
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) C[r] = v[r] + (e.beta != 0.f ? e.beta * C[r] : 0.f);
        }
    } else {
        bf16* C = reinterpret_cast<bf16*>(e.C) + off;
        if (vec) store_bf8(C, v, MMF_EPI_POLICY(e, 0));
        else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (ok[r]) C[r] = (bf16)v[r];
        }
    }
}

// Lean epilogue of the wide-tile kernels' common cases (gemm_wide.h): bias, act 0 / 2, hash dropout, residual, bf16 output, full
// 8-column segments, no row remap.  The ONE row-wise bf16 side input (the residual, else the act-2 multiplier) arrives in `side`:
// the kernel fetched it while its last K-steps were still running (one workgroup per CU has nothing else to hide that latency).
__host__ __device__ __forceinline__ bool epilogue_fast_ok(const EpiArgs& e) {
    const bool gelu = e.act == 1 && e.U != nullptr && !e.resid && !e.drop.thr16;       // HF BertIntermediate: bias, GELU, saved derivative
    return !e.coladd && !e.rowtab && e.grp_in == 0 && !e.out_f32 && (e.act == 0 || e.act == 2 || gelu) && (e.ldc & 7) == 0 && (e.N & 7) == 0 &&
           (!e.resid || (e.ldr & 7) == 0) && e.splits <= 1 && !(e.resid && e.act == 2);
}
DEVI void epilogue8_fast(const EpiArgs& e, int m, int n, f32x8 v, uint4 side, uint32_t dkey) {
    if (m >= e.M) return;
    if (e.bias) v += load_f8(e.bias + n);
    f32x8 sv;
    {
        const bf16x8 t = __builtin_bit_cast(bf16x8, side);
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = (float)t[i];
    }
    if (e.act == 1) {          // (no side input: `side` is all zeros)
        f32x8 gd;
#pragma unroll
        for (int r = 0; r < 8; ++r) { float hh, gg; gelu_erf_both(v[r], hh, gg); v[r] = hh; gd[r] = gg; }
        store_bf8(e.U + (size_t)m * e.ldc + n, gd, MMF_EPI_POLICY(e, 1));
        store_bf8(reinterpret_cast<bf16*>(e.C) + (size_t)m * e.ldc + n, v, MMF_EPI_POLICY(e, 0));
        return;
    }
    if (e.act == 2) v *= sv;
    if (e.drop.thr16) {
        const uint32_t idx = (uint32_t)m * (uint32_t)e.N + (uint32_t)n;
        const f32x4 s0 = drop_scale4(dkey, idx, e.drop.thr16, e.drop.scale);
        const f32x4 s1 = drop_scale4(dkey, idx + 4, e.drop.thr16, e.drop.scale);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] *= s0[r]; v[r + 4] *= s1[r]; }
    }
    if (e.resid) v += sv;
    store_bf8(reinterpret_cast<bf16*>(e.C) + (size_t)m * e.ldc + n, v, MMF_EPI_POLICY(e, 0));
}

}  // namespace gemm
