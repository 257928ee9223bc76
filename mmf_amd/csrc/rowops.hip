// mmf_amd :: HBM-bound row kernels for gfx950: LayerNorm fwd/bwd, embedding gather / scatter-add,
// gather/scatter of pooled rows, column sums (bias grads), dtype casts, additive mask, BCE-with-logits
// loss, fused AdamW.  One wave (64 lanes) per row, 8-byte (bf16x4) / 16-byte (fp32x4) accesses per
// lane, fp32 arithmetic, wave-shuffle reductions.  Reference call sites are cited at each C entry
// point in include/mmf_amd.h.
#include "common.h"
#include "mmf_amd.h"
#include "ln_bwd_dev.h"

namespace {

// lane `lane` of a wave owns columns (lane + 64*c)*4 .. +3, c = 0..NCH-1 (NCH = ceil(H / 256))
#define COL_OF(c) (((lane) + 64 * (c)) * 4)

DEVI f32x4 load4(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
DEVI f32x4 load4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}
DEVI void store4(bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = pack4(v[0], v[1], v[2], v[3]); }
DEVI void store4(float* p, f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

// ------------------------------------------------------------------------------------------------
// LayerNorm forward
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int rows, int H,
                                                      float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16* xr = x + (size_t)row * H;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (COL_OF(c) < H) { v[c] = load4(xr + COL_OF(c)); s += v[c][0] + v[c][1] + v[c][2] + v[c][3]; }
    }
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (COL_OF(c) < H) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[c][i] - mu; q += d * d; }
        }
    const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
    bf16* yr = y + (size_t)row * H;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (COL_OF(c) < H) {
            const f32x4 g = load4(gamma + COL_OF(c)), b = load4(beta + COL_OF(c));
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (v[c][i] - mu) * rs * g[i] + b[i];
            store4(yr + COL_OF(c), o);
        }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward (+ dropout backward of the producing Linear, + column-sum partials)
// partials layout: [gridDim.x][3][H] : 0 = dgamma, 1 = dbeta, 2 = dbias
// ------------------------------------------------------------------------------------------------
constexpr int LNB_GRID = 128;       // default workgroups (16 waves each); one partial row per workgroup
constexpr int LNB_MAX_GRID = 512;   // the workspace is sized for this many (the largest grid the launches use)
constexpr int LNB_WAVES = 16;      // waves per workgroup for H <= 768; H = 1024 (four column chunks per lane) runs 8 waves so that
                                   // its register budget doubles and nothing spills
template <int NCH> struct LnbWaves { static constexpr int value = (NCH >= 5) ? 4 : (NCH >= 4) ? 8 : LNB_WAVES; };   // 5..8 chunks (H <= 2048): 4 waves

template <int NCH>
__global__ __launch_bounds__(64 * LnbWaves<NCH>::value) void ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, bf16* __restrict__ dx,
                                                      bf16* __restrict__ dlin, DropoutCfg drop, float* __restrict__ partials,
                                                      int rows, int H) {
    constexpr int NWV = LnbWaves<NCH>::value;
    __shared__ float red[NWV][NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[NCH], ab[NCH], al[NCH], gm[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        ag[c] = ab[c] = al[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        gm[c] = (COL_OF(c) < H) ? load4(gamma + COL_OF(c)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // software-pipelined over this wave's rows: the next row's loads are in flight during the reductions of the current one
    const int rstep = gridDim.x * NWV;
    int row = blockIdx.x * NWV + wave;
    bf16x4 xr[NCH], dr_[NCH];
    float mu = 0.f, rs = 0.f;
    auto fetch = [&](int r) {
        if (r < rows) {
            mu = mean[r]; rs = rstd[r];
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (COL_OF(c) < H) {
                    xr[c] = *reinterpret_cast<const bf16x4*>(x + (size_t)r * H + COL_OF(c));
                    dr_[c] = *reinterpret_cast<const bf16x4*>(dy + (size_t)r * H + COL_OF(c));
                }
        }
    };
    fetch(row);
    for (; row < rows; row += rstep) {
        f32x4 xh[NCH], g[NCH];
        float s1 = 0.f, s2 = 0.f;
        const float rs_c = rs;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            xh[c] = g[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (COL_OF(c) < H) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = (float)xr[c][i], dv = (float)dr_[c][i];
                    xh[c][i] = (xv - mu) * rs_c;
                    g[c][i] = dv * gm[c][i];
                    s1 += g[c][i];
                    s2 += g[c][i] * xh[c][i];
                    ag[c][i] += dv * xh[c][i];
                    ab[c][i] += dv;
                }
            }
        }
        fetch(row + rstep);
        const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (COL_OF(c) < H) {
                f32x4 d;
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = rs_c * (g[c][i] - c1 - xh[c][i] * c2);
                store4(dx + (size_t)row * H + COL_OF(c), d);
                if (dlin) {
                    const f32x4 sc = drop_scale4(drop_key(drop), (uint32_t)row * (uint32_t)H + (uint32_t)COL_OF(c), drop.thr16, drop.scale);
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] *= sc[i];
                    store4(dlin + (size_t)row * H + COL_OF(c), d);
                }
                // bias gradient uses the same rounding the wgrad GEMM will see
                const bf16x4 dr = pack4(d[0], d[1], d[2], d[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) al[c][i] += (float)dr[i];
            }
    }
    // combine the waves of the workgroup, one quantity at a time
#pragma unroll 1
    for (int qn = 0; qn < 3; ++qn) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f32x4 v = (qn == 0) ? ag[c] : (qn == 1) ? ab[c] : al[c];
            *reinterpret_cast<float4*>(&red[wave][COL_OF(c)]) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        for (int col = threadIdx.x; col < H; col += 64 * NWV) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t += red[w][col];
            partials[((size_t)blockIdx.x * 3 + qn) * H + col] = t;
        }
    }
}

// out_q[col] (+)= sum_blk partials[blk][q][col].  Workgroup = 8 row-groups x 32 columns.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partials, int nblk, int H, float* o0, float* o1,
                                                             float* o2, int accumulate) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    const int qn = blockIdx.y;
    float* out = (qn == 0) ? o0 : (qn == 1) ? o1 : o2;
    if (out == nullptr) return;
    float s = 0.f;
    if (col < H) {
#pragma unroll 16
        for (int b = rg; b < nblk; b += 8) s += partials[((size_t)b * 3 + qn) * H + col];
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && col < H) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][c];
        out[col] = accumulate ? out[col] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm, H a multiple of 256 (768 / 1024 / 512 ...): HALF a wave per row.  Lane (l & 31) owns the 8 consecutive columns
// (l & 31) * 8 + 256 * c .. + 7 of its row, c = 0 .. H/256 - 1, so every access is a 16-byte-per-lane, 512-byte-per-half-wave
// contiguous transaction (the 8-byte form above reaches 2.4 - 2.9 TB/s, a third of HBM bandwidth), two rows are in flight per
// wave, and the row reductions are 5 cross-lane steps inside the half.
// ------------------------------------------------------------------------------------------------
using lnk::f32x8r; using lnk::load8; using lnk::store8; using lnk::half_sum;

template <int NC>   // NC = H / 256
__global__ __launch_bounds__(256) void ln_fwd_h_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int rows, float eps, DropoutCfg drop) {
    constexpr int H = NC * 256;
    const int hl = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const bf16* xr = x + (size_t)row * H + hl * 8;
    f32x8r v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        v[c] = load8(xr + 256 * c);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[c][i];
    }
    const float mu = half_sum(s) * (1.f / (float)H);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mu; q += d * d; }
    const float rs = rsqrtf(half_sum(q) * (1.f / (float)H) + eps);
    bf16* yr = y + (size_t)row * H + hl * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x8r g = load8(gamma + hl * 8 + 256 * c), b = load8(beta + hl * 8 + 256 * c);
        f32x8r o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (v[c][i] - mu) * rs * g[i] + b[i];
        if (drop.thr16) {      // nn.Dropout on the LayerNorm output (embeddings.py:345), element index row * H + col like mmf_dropout_bf16: the value is
                               // rounded to bf16 first, as the two-launch form stores it between its kernels
            const uint32_t idx = (uint32_t)row * (uint32_t)H + (uint32_t)(hl * 8 + 256 * c);
            const f32x4 s0 = drop_scale4(drop_key(drop), idx, drop.thr16, drop.scale), s1 = drop_scale4(drop_key(drop), idx + 4, drop.thr16, drop.scale);
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[i] = (float)(bf16)o[i] * s0[i]; o[i + 4] = (float)(bf16)o[i + 4] * s1[i]; }
        }
        store8(yr + 256 * c, o);
    }
    if (hl == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// Backward: see ln_bwd_dev.h (ln_bwd_h_block): one 256-thread workgroup per block of 8 half-waves.
template <int NC, bool DBIAS, int NR, bool DIN = false>
__global__ __launch_bounds__(256) void ln_bwd_h_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, bf16* __restrict__ dx,
                                                         bf16* __restrict__ dlin, DropoutCfg drop, float* __restrict__ partials, int rows, DropoutCfg din) {
    __shared__ float red[LN_BWD_RED_FLOATS(NC)];
    lnk::ln_bwd_h_block<NC, DBIAS, NR, DIN>(dy, x, mean, rstd, gamma, dx, dlin, drop, partials, rows, din, (int)threadIdx.x, (int)blockIdx.x, (int)gridDim.x, true, red);
}

// out_q[col] (+)= sum_blk partials[blk][q][col]: 64 columns x 16 block-groups per workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_h_kernel(const float* __restrict__ partials, int nblk, int H, float* o0, float* o1,
                                                                float* o2, int accumulate) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int qn = blockIdx.y;
    float* out = (qn == 0) ? o0 : (qn == 1) ? o1 : o2;
    if (out == nullptr) return;
    float s = 0.f;
    if (col < H) {
#pragma unroll 4
        for (int b = rg; b < nblk; b += 16) s += partials[((size_t)b * 3 + qn) * H + col];
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && col < H) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][c];
        out[col] = accumulate ? out[col] + t : t;
    }
}

// The same reduction for up to MMF_MT_MAX LayerNorm backwards in one launch (blockIdx.z = which one): a training step has 26 of
// them whose dgamma / dbeta nobody reads before the optimizer, so their reductions can run once, at the end of backward.
struct LnReduceArgs {
    const float* partials[MMF_MT_MAX];
    float* o0[MMF_MT_MAX];
    float* o1[MMF_MT_MAX];
    int nblk[MMF_MT_MAX];
    int H[MMF_MT_MAX];
};
__global__ __launch_bounds__(1024) void ln_bwd_reduce_multi_kernel(LnReduceArgs a) {
    __shared__ float red[16][64];
    const int t = blockIdx.z, H = a.H[t], nblk = a.nblk[t];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int qn = blockIdx.y;
    if (blockIdx.x * 64 >= H) return;
    const float* partials = a.partials[t];
    float s = 0.f;
    if (col < H) {
#pragma unroll 4
        for (int b = rg; b < nblk; b += 16) s += partials[((size_t)b * 3 + qn) * H + col];
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && col < H) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += red[i][c];
        (qn == 0 ? a.o0[t] : a.o1[t])[col] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// embeddings
// ------------------------------------------------------------------------------------------------
extern __device__ int g_index_error;
template <typename OutT>   // bf16 (throughput path) or float (fp32-accurate path, fp32_path.hip)
__global__ __launch_bounds__(256) void embed_text_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ seg,
                                                          const float* __restrict__ word, const float* __restrict__ pos,
                                                          const float* __restrict__ type, OutT* __restrict__ y, int B, int T,
                                                          int S, int H, int row0, int pos0, int V, int NT) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * T) return;
    const int b = r / T, t = r - b * T;
    const int64_t id = ids[r];
    const int64_t sg = seg ? seg[r] : 0;
    OutT* yr0 = y + ((size_t)b * S + row0 + t) * H;
    if ((V > 0 && (id < 0 || id >= V)) || (NT > 0 && (sg < 0 || sg >= NT))) {      // out of the table: zero row + error flag
        if (lane == 0) atomicOr(&g_index_error, 1);
        for (int col = lane * 4; col < H; col += 256) store4(yr0 + col, f32x4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const float* w = word + (size_t)id * H;
    const float* p = pos + (size_t)(t + pos0) * H;
    const float* ty = type + (size_t)sg * H;
    OutT* yr = y + ((size_t)b * S + row0 + t) * H;
    for (int col = lane * 4; col < H; col += 256) {
        const f32x4 a = load4(w + col), c = load4(p + col), d = load4(ty + col);
        // same association order as embeddings.py:344 (words + position) + token_type
        store4(yr + col, f32x4{(a[0] + c[0]) + d[0], (a[1] + c[1]) + d[1], (a[2] + c[2]) + d[2], (a[3] + c[3]) + d[3]});
    }
}

// image_text_alignment of BertVisioLinguisticEmbeddings.get_position_embeddings_visual (mmf/modules/embeddings.py:373-397): the position
// embedding of a region is the MEAN of the text position rows of the words aligned with it (align[r][a], -1 = padding; a region with
// no aligned word gets zero), here with the region's visual token-type row added, as ONE fp32 addend row per region that the visual
// projection GEMM gathers in its epilogue (rowtab).  One wave per region row.
__global__ __launch_bounds__(256) void align_pos_fwd_kernel(const int64_t* __restrict__ align, const float* __restrict__ pos, const float* __restrict__ typ,
                                                             const int64_t* __restrict__ typ_idx, float* __restrict__ out, int rows, int A, int H, int P, int NT) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    int cnt = 0;
    for (int a = 0; a < A; ++a) {
        const int64_t ix = align[(size_t)r * A + a];
        if (ix == -1) continue;
        if (ix < 0 || ix >= P) { if (lane == 0) atomicOr(&g_index_error, 1); continue; }
        ++cnt;
    }
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    int64_t ti = typ ? typ_idx[r] : 0;
    if (typ && (ti < 0 || ti >= NT)) { if (lane == 0) atomicOr(&g_index_error, 1); ti = 0; }
    for (int col = lane * 4; col < H; col += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < A; ++a) {
            const int64_t ix = align[(size_t)r * A + a];
            if (ix < 0 || ix >= P) continue;
            acc += load4(pos + (size_t)ix * H + col);
        }
        acc *= inv;
        if (typ) acc += load4(typ + (size_t)ti * H + col);
        store4(out + (size_t)r * H + col, acc);
    }
}
// backward of the mean: dpos[align[r][a]] += dvis[r] / count[r] for every valid a (fp32 atomics, like the word-embedding scatter);
// row r = (b, i) of the visual block lives at dvis + (b * bstride + i) * ld.
template <typename T>   // bf16 gradient rows (throughput path) or fp32 rows (mmf_amd.fp32_training())
__global__ __launch_bounds__(256) void align_pos_bwd_kernel(const T* __restrict__ dvis, int ld, int rpb, int bstride, const int64_t* __restrict__ align,
                                                             float* __restrict__ dpos, int rows, int A, int H, int P) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    int cnt = 0;
    for (int a = 0; a < A; ++a) { const int64_t ix = align[(size_t)r * A + a]; if (ix >= 0 && ix < P) ++cnt; }
    if (cnt == 0) return;
    const float inv = 1.f / (float)cnt;
    const int b = r / rpb, i = r - b * rpb;
    const T* src = dvis + ((size_t)b * bstride + i) * ld;
    for (int col = lane; col < H; col += 64) {
        const float g = (float)src[col] * inv;
        for (int a = 0; a < A; ++a) {
            const int64_t ix = align[(size_t)r * A + a];
            if (ix >= 0 && ix < P) atomicAdd(dpos + (size_t)ix * H + col, g);
        }
    }
}

// y[b*S + row0 + i] = x[b*L + i] + pos[pos0 + i] + type[seg[b,i]]   (pos / seg may be null)
template <typename T>   // bf16 rows (throughput path) or fp32 rows (fp32-accurate path)
__global__ __launch_bounds__(256) void rows_add_embed_kernel(const T* __restrict__ x, const int64_t* __restrict__ seg,
                                                              const float* __restrict__ pos, const float* __restrict__ type,
                                                              T* __restrict__ y, int B, int L, int S, int H, int row0, int pos0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * L) return;
    const int b = r / L, i = r - b * L;
    const T* xr = x + (size_t)r * H;
    const float* p = pos ? pos + (size_t)(i + pos0) * H : nullptr;
    const float* ty = (type && seg) ? type + (size_t)seg[r] * H : nullptr;
    T* yr = y + ((size_t)b * S + row0 + i) * H;
    for (int col = lane * 4; col < H; col += 256) {
        f32x4 v = load4(xr + col);
        if (p) { const f32x4 c = load4(p + col); v += c; }
        if (ty) { const f32x4 d = load4(ty + col); v += d; }
        store4(yr + col, v);
    }
}

// y[r, :] = bf16(x[r, :] + table[idx[r], :])   (x fp32 region features, table fp32 [*, D]; idx may be null: plain cast)
__global__ __launch_bounds__(256) void rows_add_table_f32_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                                  const float* __restrict__ table, bf16* __restrict__ y, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + (size_t)r * D;
    const float* t = (idx && table) ? table + (size_t)idx[r] * D : nullptr;
    bf16* yr = y + (size_t)r * D;
    for (int col = lane * 4; col < D; col += 256) {
        f32x4 v = load4(xr + col);
        if (t) { const f32x4 c = load4(t + col); v += c; }
        store4(yr + col, v);
    }
}

// dst[(b*dst_bs + i), :] = src[(b*src_bs + i), :]  for b < nb, i < rpb: one modality's block of a [B, S, H] sequence
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16* __restrict__ src, int src_bs, bf16* __restrict__ dst, int dst_bs,
                                                         int nb, int rpb, int H) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nb * rpb) return;
    const int b = r / rpb, i = r - b * rpb;
    const bf16* s = src + ((size_t)b * src_bs + i) * H;
    bf16* d = dst + ((size_t)b * dst_bs + i) * H;
    for (int col = lane * 8; col < H; col += 512) *reinterpret_cast<uint4*>(d + col) = *reinterpret_cast<const uint4*>(s + col);
}

// Set by the index-consuming kernels when they meet an id outside its table (nn.Embedding raises IndexError there,
// mmf/modules/embeddings.py:329-345): the offending row is skipped (forward: written as zeros) instead of reading or, in the
// backward, atomically WRITING out of bounds; mmf_amd_take_index_error() reports and clears the flag.
__device__ int g_index_error = 0;
DEVI void flag_index_error() { atomicOr(&g_index_error, 1); }

DEVI int bucket_of(const int64_t* idx, int idx_ld, int per_pos, int idx_base, int b, int i) {
    if (idx) return (int)idx[(size_t)b * idx_ld + i];
    return per_pos ? i + idx_base : idx_base;
}

__global__ __launch_bounds__(256) void scatter_add_direct_kernel(const bf16* __restrict__ x, int ld, int nb, int rpb, int bstride,
                                                                  const int64_t* __restrict__ idx, int idx_ld, int per_pos,
                                                                  int idx_base, float* __restrict__ out, int H, int skip, int nbuckets) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nb * rpb) return;
    const int b = r / rpb, i = r - b * rpb;
    const int bk = bucket_of(idx, idx_ld, per_pos, idx_base, b, i);
    if (bk == skip) return;
    if (nbuckets > 0 && (bk < 0 || bk >= nbuckets)) { if (lane == 0) flag_index_error(); return; }
    const bf16* xr = x + ((size_t)b * bstride + i) * ld;
    float* o = out + (size_t)bk * H;
    for (int col = lane * 4; col < H; col += 256) {
        const f32x4 v = load4(xr + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(o + col + j, v[j]);
    }
}

// The same without atomics on the hot path, deterministic (round 5): one workgroup per source row r = (b, i).  Its waves count the EARLIER rows that share
// its bucket (ballot + popcount over [0, r), a quarter each); the first row of a bucket ("owner") goes on through the rows behind it, adds the rows of the
// bucket in increasing row order and adds the sum to out[bucket] once.  The order of the additions is fixed: the word-embedding gradient (the one
// atomically accumulated tensor of the VisualBERT step) is bit-reproducible, and 4096 rows x 768 columns take ~12 us instead of 46 us of fp32 atomics.
// O(rows^2 / 64) index comparisons in total (262144 wave steps at 4096 rows): used up to SCATTER_UNIQUE_MAX rows, the atomic form beyond.
// Round 6 - bounded runs: an owner walks its run serially (four rows per dependent step), so ONE id on a quarter of the rows made the launch 270 us
// against 162 us of atomics (M4C's previous-prediction gather: ~1000 of 1536 rows carry index 0; MLM batches with a frequent token).  The owner now
// stops after SCATTER_RUN_MAX rows of its bucket; every later row of the bucket (it knows: it counted SCATTER_RUN_MAX earlier ones) adds itself with
// fp32 atomics, in parallel.  An owner that stopped at the limit adds its sum atomically as well (tail rows may be writing); one that walked its whole
// bucket writes it plainly: buckets of fewer than SCATTER_RUN_MAX rows stay bit-reproducible and atomic-free.
constexpr int SCATTER_UNIQUE_MAX = 16384;
constexpr int SCATTER_RUN_MAX = 64;
__global__ __launch_bounds__(256) void scatter_add_unique_kernel(const bf16* __restrict__ x, int ld, int nb, int rpb, int bstride,
                                                                  const int64_t* __restrict__ idx, int idx_ld, float* __restrict__ out, int H, int skip,
                                                                  int nbuckets) {
    extern __shared__ int sb[];      // the bucket of every source row behind r (rows are numbered b * rpb + i): staged once per owner workgroup — scanned from
                                     // global memory the dependent steps of a wave cost a cache round trip each (48 us per launch, no better than the atomics)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = nb * rpb;
    const int r = blockIdx.x;                 // one WORKGROUP per source row: its four waves share the scan and take a quarter of the columns each
    const int bk = [&]() { const int qb = r / rpb; const int64_t v = idx[(size_t)qb * idx_ld + (r - qb * rpb)]; return (v < 0 || v > 0x7ffffffe) ? -2 : (int)v; }();
    if (bk == skip) return;
    if (bk < 0 || (nbuckets > 0 && bk >= nbuckets)) { if (threadIdx.x == 0) flag_index_error(); return; }
    // earlier rows with this bucket (every wave checks the rows [0, r) a quarter each; the count is shared through LDS and saturates at SCATTER_RUN_MAX)
    __shared__ int n_before_s;
    if (threadIdx.x == 0) n_before_s = 0;
    __syncthreads();
    {
        int mine = 0;
        for (int q0 = wave * 64; q0 < r; q0 += 256) {
            const int q = q0 + lane;
            bool hit = false;
            if (q < r) { const int qb = q / rpb; hit = idx[(size_t)qb * idx_ld + (q - qb * rpb)] == (int64_t)bk; }
            mine += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
            if (mine >= SCATTER_RUN_MAX) break;
        }
        if (lane == 0 && mine) atomicAdd(&n_before_s, mine);
    }
    __syncthreads();
    const int n_before = n_before_s;
    const int wq = H >> 2;                    // columns per wave (a multiple of 4)
    const int b = r / rpb, i = r - b * rpb;
    if (n_before >= SCATTER_RUN_MAX) {        // the tail of a long run: this row adds itself
        for (int c0 = 0; c0 < wq; c0 += 256) {
            const int col = wave * wq + c0 + 4 * lane;
            if (c0 + 4 * lane < wq) {
                const f32x4 v = load4(x + ((size_t)b * bstride + i) * ld + col);
                float* o = out + (size_t)bk * H + col;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(o + j, v[j]);
            }
        }
        return;
    }
    if (n_before) return;                     // an earlier row owns the bucket
    // owner: stage the buckets of the rows behind this one, then every wave adds its quarter of the columns of this row and of the next (up to)
    // SCATTER_RUN_MAX - 1 rows of the bucket, in row order
    const int rest = total - (r + 1);
    for (int q = threadIdx.x; q < rest; q += 256) {
        const int qq = r + 1 + q, qb = qq / rpb;
        const int64_t v = idx[(size_t)qb * idx_ld + (qq - qb * rpb)];
        sb[q] = (v < 0 || v > 0x7ffffffe) ? -2 : (int)v;
    }
    __syncthreads();
    for (int c0 = 0; c0 < wq; c0 += 256) {
        const int col = wave * wq + c0 + 4 * lane;
        const bool on = c0 + 4 * lane < wq;
        f32x4 acc = on ? load4(x + ((size_t)b * bstride + i) * ld + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        int taken = 1;                        // rows of the bucket added so far (this one included)
        for (int q0 = 0; q0 < rest && taken < SCATTER_RUN_MAX; q0 += 64) {
            const int q = q0 + lane;
            unsigned long long m = __builtin_amdgcn_ballot_w64(q < rest && sb[q] == bk);
            constexpr int RF = 4;
            while (m && taken < SCATTER_RUN_MAX) {      // up to four rows of the run in flight, added in row order (more in flight measured no faster)
                const bf16* src[RF];
                int n = 0;
#pragma unroll
                for (int u = 0; u < RF; ++u) {
                    src[u] = x;
                    if (m && taken + u < SCATTER_RUN_MAX) {
                        const int j = __builtin_ctzll(m);
                        m &= m - 1;
                        const int qq = r + 1 + q0 + j, qb = qq / rpb, qi = qq - qb * rpb;
                        src[u] = x + ((size_t)qb * bstride + qi) * ld;
                        n = u + 1;
                    }
                }
                taken += n;
                f32x4 v[RF];
#pragma unroll
                for (int u = 0; u < RF; ++u) v[u] = (u < n && on) ? load4(src[u] + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < RF; ++u)
                    if (u < n) acc += v[u];
            }
        }
        if (on) {
            float* o = out + (size_t)bk * H + col;
            if (taken < SCATTER_RUN_MAX) {      // the whole bucket was walked: nobody else writes this row
                const f32x4 old = load4(o);
                *reinterpret_cast<float4*>(o) = make_float4(old[0] + acc[0], old[1] + acc[1], old[2] + acc[2], old[3] + acc[3]);
            } else {                            // a longer run: its tail rows add themselves concurrently
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(o + j, acc[j]);
            }
        }
    }
}

// Position-table gradient: bucket = idx_base + (row index inside the sample), no index array, i.e. out[idx_base + i][c] +=
// sum_b x[b][i][c] — a plain strided sum over the batch (deterministic; the atomic form has every sample hit the same rows).
__global__ __launch_bounds__(256) void scatter_add_pos_kernel(const bf16* __restrict__ x, int ld, int nb, int rpb, int bstride, int idx_base,
                                                               float* __restrict__ out, int H) {
    const int i = blockIdx.x, col = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (col >= H) return;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nb; ++b) a += load4(x + ((size_t)b * bstride + i) * ld + col);
    float* o = out + (size_t)(idx_base + i) * H + col;
    const f32x4 old = load4(o);
    *reinterpret_cast<float4*>(o) = make_float4(old[0] + a[0], old[1] + a[1], old[2] + a[2], old[3] + a[3]);
}

// <= 2 buckets (token-type tables, the single visual position row): deterministic two-stage column sums per bucket.
// stage 1: grid (FEW_GROUPS, ceil(H/256)); partials [FEW_GROUPS][2][H].  Rows whose bucket is >= 2 fall back to atomics.
constexpr int FEW_GROUPS = 32;
__global__ __launch_bounds__(256) void scatter_add_few_kernel(const bf16* __restrict__ x, int ld, int nb, int rpb, int bstride,
                                                               const int64_t* __restrict__ idx, int idx_ld, int per_pos,
                                                               int idx_base, float* __restrict__ out, int H, float* __restrict__ partials, int nbuckets) {
    __shared__ float red[4][2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = nb * rpb;
    const int col = blockIdx.y * 256 + lane * 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (col < H) {
        for (int r = blockIdx.x * 4 + wave; r < total; r += gridDim.x * 4) {
            const int b = r / rpb, i = r - b * rpb;
            const int bk = bucket_of(idx, idx_ld, per_pos, idx_base, b, i);
            const f32x4 v = load4(x + ((size_t)b * bstride + i) * ld + col);
            if (bk == 0) a0 += v;
            else if (bk == 1) a1 += v;
            else if (bk < 0 || bk >= nbuckets) { if (lane == 0) flag_index_error(); }
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(out + (size_t)bk * H + col + j, v[j]);
            }
        }
    }
    *reinterpret_cast<float4*>(&red[wave][0][lane * 4]) = make_float4(a0[0], a0[1], a0[2], a0[3]);
    *reinterpret_cast<float4*>(&red[wave][1][lane * 4]) = make_float4(a1[0], a1[1], a1[2], a1[3]);
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c < H) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            partials[((size_t)blockIdx.x * 2 + k) * H + c] = red[0][k][threadIdx.x] + red[1][k][threadIdx.x] + red[2][k][threadIdx.x] + red[3][k][threadIdx.x];
    }
}
__global__ __launch_bounds__(256) void scatter_add_few_reduce_kernel(const float* __restrict__ partials, int ngroups, int H, int nbuckets,
                                                                      float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= H || k >= nbuckets) return;
    float s = 0.f;
    for (int g = 0; g < ngroups; ++g) s += partials[((size_t)g * 2 + k) * H + c];
    out[(size_t)k * H + c] += s;
}

// The four SMALL table gradients of BertVisioLinguisticEmbeddings' backward (embeddings.py:329-345, 411-419: text positions, text token types,
// visual token types, the single visual position row) in ONE pass over the pre-LayerNorm gradient x [B, S = T + R, H] instead of
// scatter_add_pos + 3 x (scatter_add_few + its reduction): seven short launches, ~80 us of a training step, each a serial walk of a few waves.
// Workgroup (i, column block) owns sequence position i of EVERY sample: it adds the B rows x[b][i] in sample order, which is already the final
// position row (text) and one partial of the bucket sums (token types 0 / 1; every visual row for position_ids_visual == 0).  partials
// [S][3][H]: slots 0 / 1 = the rows with bucket 0 / 1, slot 2 = all rows; embed_tables_reduce_kernel adds them over i in a fixed order
// (deterministic, no atomics for buckets 0 and 1; a bucket >= 2 — never produced by the path's tokenizers — falls back to atomics like scatter_add_few).
__global__ __launch_bounds__(256) void embed_tables_bwd_kernel(const bf16* __restrict__ x, int ld, int B, int T, int R, const int64_t* __restrict__ seg,
                                                                const int64_t* __restrict__ vt, int pos0, float* __restrict__ dpos,
                                                                float* __restrict__ dtyp, int NT, float* __restrict__ dtyp_vis, int NTV,
                                                                float* __restrict__ partials, int H) {
    const int i = blockIdx.x, S = T + R;
    const int col = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (col >= H) return;
    const bool text = i < T;
    const int64_t* ids = text ? seg : vt;
    const int ild = text ? T : R, ii = text ? i : i - T;
    float* tab = text ? dtyp : dtyp_vis;
    const int nb = ids ? (text ? NT : NTV) : 1;
    f32x4 all = {0.f, 0.f, 0.f, 0.f}, a0 = all, a1 = all;
    constexpr int UN = 8;
    for (int b0 = 0; b0 < B; b0 += UN) {
        f32x4 v[UN];
        int bk[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int b = b0 + u < B ? b0 + u : B - 1;
            v[u] = load4(x + ((size_t)b * S + i) * ld + col);
            if (ids) { const int64_t t = ids[(size_t)b * ild + ii]; bk[u] = (t < 0 || t >= (int64_t)nb) ? -1 : (int)t; }     // (range check in 64 bits: 1 << 32 is not bucket 0)
            else bk[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (b0 + u < B) {
                all += v[u];
                if (bk[u] < 0 || bk[u] >= nb) { if (threadIdx.x == 0) flag_index_error(); }
                else if (bk[u] == 0) a0 += v[u];
                else if (bk[u] == 1) a1 += v[u];
                else if (tab) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(tab + (size_t)bk[u] * H + col + j, v[u][j]);
                }
            }
        }
    }
    if (text && dpos) {
        float* o = dpos + (size_t)(pos0 + i) * H + col;
        const f32x4 old = load4(o);
        *reinterpret_cast<float4*>(o) = make_float4(old[0] + all[0], old[1] + all[1], old[2] + all[2], old[3] + all[3]);
    }
    float* p = partials + (size_t)i * 3 * H + col;
    *reinterpret_cast<float4*>(p) = make_float4(a0[0], a0[1], a0[2], a0[3]);
    *reinterpret_cast<float4*>(p + H) = make_float4(a1[0], a1[1], a1[2], a1[3]);
    *reinterpret_cast<float4*>(p + 2 * H) = make_float4(all[0], all[1], all[2], all[3]);
}
// blockIdx.y = q: 0, 1 -> dtyp[q] += sum_{i < T} partials[i][q]; 2, 3 -> dtyp_vis[q - 2] += sum_{i >= T} partials[i][q - 2];
// 4 -> dpos_vis[0] += sum_{i >= T} partials[i][2].  64 columns per workgroup; the positions are dealt to its four waves (i = i0 + wave, step 4) and the four
// partial sums are added in a fixed order: deterministic, 32 independent loads per thread at the VQA2 shape instead of 128.
__global__ __launch_bounds__(256) void embed_tables_reduce_kernel(const float* __restrict__ partials, int T, int R, int H, float* __restrict__ dtyp, int NT,
                                                                   float* __restrict__ dtyp_vis, int NTV, float* __restrict__ dpos_vis) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, q = blockIdx.y;
    const bool text = q < 2;
    const int slot = q == 4 ? 2 : (q & 1);
    float* out = q == 4 ? dpos_vis : (text ? dtyp : dtyp_vis);
    if (!out || (q < 4 && slot >= (text ? NT : NTV))) return;      // (uniform over the workgroup)
    const int i0 = text ? 0 : T, i1 = text ? T : T + R;
    float s0 = 0.f, s1 = 0.f;
    if (c < H) {
        int i = i0 + wave;
        for (; i + 4 < i1; i += 8) {
            s0 += partials[((size_t)i * 3 + slot) * H + c];
            s1 += partials[((size_t)(i + 4) * 3 + slot) * H + c];
        }
        if (i < i1) s0 += partials[((size_t)i * 3 + slot) * H + c];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && c < H) out[(size_t)(q == 4 ? 0 : slot) * H + c] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ x, const int64_t* __restrict__ index,
                                                           T* __restrict__ out, int B, int S, int H, DropoutCfg drop,
                                                           int scatter) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int64_t ix = index[b];
    ix = ix < 0 ? 0 : (ix >= S ? S - 1 : ix);
    for (int col = lane * 4; col < H; col += 256) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f};
        if (drop.thr16) sc = drop_scale4(drop_key(drop), (uint32_t)b * (uint32_t)H + (uint32_t)col, drop.thr16, drop.scale);
        if (!scatter) {
            const f32x4 v = load4(x + ((size_t)b * S + ix) * H + col);
            store4(out + (size_t)b * H + col, v * sc);
        } else {
            const f32x4 v = load4(x + (size_t)b * H + col);
            store4(out + ((size_t)b * S + ix) * H + col, v * sc);
        }
    }
}

// backward of the pooling gather as ONE pass over the whole [B, S, H] gradient: row (b, index[b]) = dropout(dout[b]), every other row zero
// (the zero fill + row scatter of the two-launch form in one launch; one wave per row, 16-byte stores)
__global__ __launch_bounds__(256) void scatter_rows_full_kernel(const bf16* __restrict__ dout, const int64_t* __restrict__ index, bf16* __restrict__ dx,
                                                                 int B, int S, int H, DropoutCfg drop) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long)B * S) return;
    const int b = (int)(r / S);
    int64_t ix = index[b];
    ix = ix < 0 ? 0 : (ix >= S ? S - 1 : ix);
    const bool hit = r - (long)b * S == ix;
    for (int col = lane * 8; col < H; col += 512) {
        f32x8r v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (hit) {
            v = load8(dout + (size_t)b * H + col);
            if (drop.thr16) {
                const uint32_t idx = (uint32_t)b * (uint32_t)H + (uint32_t)col;
                const f32x4 s0 = drop_scale4(drop_key(drop), idx, drop.thr16, drop.scale), s1 = drop_scale4(drop_key(drop), idx + 4, drop.thr16, drop.scale);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] *= s0[i]; v[i + 4] *= s1[i]; }
            }
        }
        store8(dx + (size_t)r * H + col, v);
    }
}

// ------------------------------------------------------------------------------------------------
// column sums: grid (CS_GROUPS, ceil(N/256)); partials [CS_GROUPS][N]
// ------------------------------------------------------------------------------------------------
constexpr int CS_GROUPS = 64;
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, int ld, int nb, int rpb, int bstride, int N,
                                                      float* __restrict__ partials) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * 256 + lane * 4;
    const int total = nb * rpb;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (int r = blockIdx.x * 4 + wave; r < total; r += gridDim.x * 4) {
            const int b = r / rpb, i = r - b * rpb;
            const bf16* p = x + ((size_t)b * bstride + i) * ld + col;
            if (col + 4 <= N) acc += load4(p);
            else for (int j = 0; j < 4; ++j) if (col + j < N) acc[j] += (float)p[j];
        }
    }
    *reinterpret_cast<float4*>(&red[wave][lane * 4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c < N) partials[(size_t)blockIdx.x * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partials, int ngroups, int N, float* __restrict__ out,
                                                             float beta) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < N) {
#pragma unroll 8
        for (int g = rg; g < ngroups; g += 8) s += partials[(size_t)g * N + col];
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && col < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][c];
        out[col] = (beta != 0.f) ? beta * out[col] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------
// casts / mask
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
            bf16x8 r;
            r[0] = (bf16)a.x; r[1] = (bf16)a.y; r[2] = (bf16)a.z; r[3] = (bf16)a.w;
            r[4] = (bf16)b.x; r[5] = (bf16)b.y; r[6] = (bf16)b.z; r[7] = (bf16)b.w;
            *reinterpret_cast<bf16x8*>(dst + i) = r;
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = (bf16)src[j];
        }
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) store4(dst + i, load4(src + i));
        else for (int64_t j = i; j < n; ++j) dst[j] = (float)src[j];
    }
}
__global__ void additive_mask_kernel(const int64_t* __restrict__ m, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (1.0f - (float)m[i]) * -10000.0f;
}

// VisualBERT's input massaging (visual_bert.py:444-467, 525-556, 389-392) in ONE launch, one workgroup per sample:
//   image_mask[b][r] = r < image_dim[b] (all ones without image_dim);  attention_mask[b] = input_mask[b] | image_mask[b];
//   visual_embeddings_type[b][r] = 0;  mask_add = (1 - attention_mask) * -10000;  pool_index[b] = sum_t input_mask[b][t] - 2
// (arange / compare / cast / zeros_like / cat / sum / subtract / additive-mask launches of the unfused form).
__global__ __launch_bounds__(256) void visual_masks_kernel(const int64_t* __restrict__ input_mask, const int64_t* __restrict__ image_dim, int T, int R,
                                                            int64_t* __restrict__ image_mask, int64_t* __restrict__ attention_mask,
                                                            int64_t* __restrict__ vtype, float* __restrict__ mask_add, int64_t* __restrict__ pool_index) {
    __shared__ long red[4];
    const int b = blockIdx.x, S = T + R;
    const long dim = image_dim ? image_dim[b] : (long)R;
    long cnt = 0;
    for (int i = threadIdx.x; i < S; i += 256) {
        long m;
        if (i < T) { m = input_mask[(size_t)b * T + i]; cnt += m; }
        else { m = (i - T) < dim ? 1 : 0; image_mask[(size_t)b * R + (i - T)] = m; vtype[(size_t)b * R + (i - T)] = 0; }
        attention_mask[(size_t)b * S + i] = m;
        mask_add[(size_t)b * S + i] = (1.0f - (float)m) * -10000.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) pool_index[b] = (red[0] + red[1]) + (red[2] + red[3]) - 2;
}

// y = x * keep_scale(index)  (forward and backward of nn.Dropout are the same map)
__global__ __launch_bounds__(256) void dropout_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n, DropoutCfg d) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) store4(y + i, load4(x + i) * drop_scale4(drop_key(d), (uint32_t)i, d.thr16, d.scale));
        else for (int64_t j = i; j < n; ++j) y[j] = (bf16)((float)x[j] * drop_scale1(drop_key(d), (uint32_t)j, d.thr16, d.scale));
    }
}
// du = dh * g, g = gelu'(u) as saved by the forward GEMM epilogue (act == 1)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ dh, const bf16* __restrict__ g, bf16* __restrict__ du, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) store4(du + i, load4(dh + i) * load4(g + i));
        else for (int64_t j = i; j < n; ++j) du[j] = (bf16)((float)dh[j] * (float)g[j]);
    }
}
// small pointwise ops on bf16 vectors: 0: a*b   1: relu(a)   2: a * (b > 0)   3: a + b   (ViLBERT poolers / fusion,
// vilbert.py:799-826,1315-1320; UNITER image + position embedding sum, uniter.py:82)
__global__ __launch_bounds__(256) void eltwise_kernel(int op, const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ out,
                                                       int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = (float)a[i];
    float r;
    if (op == 0) r = x * (float)b[i];
    else if (op == 1) r = x > 0.f ? x : 0.f;
    else if (op == 3) r = x + (float)b[i];
    else r = ((float)b[i] > 0.f) ? x : 0.f;
    out[i] = (bf16)r;
}
// dx = dy * (1 - y^2): backward of tanh (HF BertPooler)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y, bf16* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float yy = (float)y[i]; dx[i] = (bf16)((float)dy[i] * (1.f - yy * yy)); }
}
// 2-D casts with leading dimensions; destination pad columns [cols, ldd) are zero-filled
__global__ __launch_bounds__(256) void cast2d_f32_bf16_kernel(const float* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd, int rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * ldd) return;
    const int r = (int)(i / ldd), c = (int)(i - (int64_t)r * ldd);
    dst[i] = (c < cols) ? (bf16)src[(size_t)r * lds_ + c] : (bf16)0.f;
}
__global__ __launch_bounds__(256) void cast2d_bf16_f32_kernel(const bf16* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[(size_t)r * ldd + c] = (float)src[(size_t)r * lds_ + c];
}

// ------------------------------------------------------------------------------------------------
// BCE with logits (losses.py:246-251)
// ------------------------------------------------------------------------------------------------
constexpr int BCE_BLOCKS = 128;
__global__ __launch_bounds__(256) void bce_fwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                               float* __restrict__ partial, int64_t n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float xv = x[i], tv = t[i];
        // numerically stable form used by ATen: max(x,0) - x*t + log1p(exp(-|x|))
        s += fmaxf(xv, 0.f) - xv * tv + log1pf(__expf(-fabsf(xv)));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void bce_fwd_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ loss, int B) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += partial[i];
    s = wave_sum(s);
    // mean over B*N, times N  ==  sum / B
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                       const float* __restrict__ gloss, bf16* __restrict__ d, int ldd, int B, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * ldd) return;
    const int b = (int)(i / ldd), n = (int)(i - (int64_t)b * ldd);
    float v = 0.f;
    if (n < N) {
        const float g = gloss ? gloss[0] : 1.f;
        const float xv = x[(size_t)b * N + n];
        v = g * (1.f / (1.f + __expf(-xv)) - t[(size_t)b * N + n]) / (float)B;
    }
    d[i] = (bf16)v;
}

// ------------------------------------------------------------------------------------------------
// softmax cross-entropy (nn.CrossEntropyLoss, mean over the rows whose label != ignore_index): C <= 64 classes
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ce_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab, float* __restrict__ loss,
                                                    float* __restrict__ count, int B, int C, int ignore_index) {
    float s = 0.f, n = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        const int64_t y = lab[b];
        if (y == ignore_index || y < 0 || y >= C) continue;
        const float* r = x + (size_t)b * C;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, r[c]);
        float z = 0.f;
        for (int c = 0; c < C; ++c) z += __expf(r[c] - mx);
        s += mx + __logf(z) - r[y];
        n += 1.f;
    }
    s = wave_sum(s); n = wave_sum(n);
    if (threadIdx.x == 0) { loss[0] = s / n; count[0] = n; }   // 0/0 = NaN when every label is ignored, like torch
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab, const float* __restrict__ count,
                                                     const float* __restrict__ gloss, float* __restrict__ dx, int B, int C, int ignore_index) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t y = lab[b];
    float* d = dx + (size_t)b * C;
    if (y == ignore_index || y < 0 || y >= C) { for (int c = 0; c < C; ++c) d[c] = 0.f; return; }
    const float* r = x + (size_t)b * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, r[c]);
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += __expf(r[c] - mx);
    const float g = (gloss ? gloss[0] : 1.f) / count[0];
    for (int c = 0; c < C; ++c) d[c] = g * (__expf(r[c] - mx) / z - (c == y ? 1.f : 0.f));
}

// ------------------------------------------------------------------------------------------------
// vocabulary-sized softmax cross-entropy: the masked-LM loss of VisualBERTForPretraining (mmf/models/visual_bert.py:215,
// 270-277: nn.CrossEntropyLoss(ignore_index=-1) over [B * S, vocab] logits).  One workgroup per row; rows whose label is
// ignored (the visual positions and ~85 % of the text positions) cost one label read.  Forward keeps the row's log-sum-exp so
// that backward writes the gradient in one pass, directly as the zero-padded bf16 operand of the decoder's dgrad / wgrad GEMMs.
// ------------------------------------------------------------------------------------------------
DEVI float block_max256(float v, float* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return r;
}
DEVI float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}
DEVI bool ce_label_counts(int64_t y, int C, int ignore_index) {
    if (y == ignore_index) return false;
    if (y < 0 || y >= C) { flag_index_error(); return false; }   // torch raises for a target outside [0, C): flag it
    return true;
}
__global__ __launch_bounds__(256) void vocab_ce_fwd_kernel(const float* __restrict__ x, int ld, const int64_t* __restrict__ lab,
                                                            float* __restrict__ lse, float* __restrict__ rowloss, int C,
                                                            int ignore_index) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const int64_t y = lab[r];
    if (y == ignore_index || y < 0 || y >= C) {      // uniform over the workgroup
        if (threadIdx.x == 0) { ce_label_counts(y, C, ignore_index); lse[r] = 0.f; rowloss[r] = 0.f; }
        return;
    }
    const float* xr = x + (size_t)r * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, xr[c]);
    mx = block_max256(mx, sh);
    float z = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) z += expf(xr[c] - mx);
    z = block_sum256(z, sh);
    if (threadIdx.x == 0) {
        const float l = mx + logf(z);
        lse[r] = l;
        rowloss[r] = l - xr[y];
    }
}
// loss = sum of the counted rows' losses / their number, in a fixed order (deterministic); 0 / 0 = NaN when every label is
// ignored, like torch (and as tests/models/test_visual_bert.py:71-98 of the reference asserts)
__global__ __launch_bounds__(256) void vocab_ce_finalize_kernel(const float* __restrict__ rowloss, const int64_t* __restrict__ lab,
                                                                 float* __restrict__ loss, float* __restrict__ count, int R, int C,
                                                                 int ignore_index) {
    __shared__ float sh[4];
    float s = 0.f, n = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
        const int64_t y = lab[r];
        if (y != ignore_index && y >= 0 && y < C) { s += rowloss[r]; n += 1.f; }
    }
    s = block_sum256(s, sh);
    n = block_sum256(n, sh);
    if (threadIdx.x == 0) { loss[0] = s / n; count[0] = n; }
}
// d[r][c] = gloss / count * (softmax(x[r])[c] - [c == y_r]) for counted rows, 0 elsewhere (ignored rows, pad columns C..ldd)
template <typename T>     // bf16: the zero-padded GEMM operand of the throughput path; float: the fp32 training path
__global__ __launch_bounds__(256) void vocab_ce_bwd_kernel(const float* __restrict__ x, int ld, const int64_t* __restrict__ lab,
                                                            const float* __restrict__ lse, const float* __restrict__ count,
                                                            const float* __restrict__ gloss, T* __restrict__ d, int ldd, int C,
                                                            int ignore_index) {
    const int r = blockIdx.y;
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c0 >= ldd) return;
    const int64_t y = lab[r];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y != ignore_index && y >= 0 && y < C) {
        const float g = (gloss ? gloss[0] : 1.f) / count[0];
        const float l = lse[r];
        const float* xr = x + (size_t)r * ld;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            if (c < C) v[e] = g * (expf(xr[c] - l) - (c == (int)y ? 1.f : 0.f));
        }
    }
    store4(d + (size_t)r * ldd + c0, v);
}

// ------------------------------------------------------------------------------------------------
// masked soft-target KL divergence: ViLBERT's masked-region classification loss (mmf/models/vilbert.py:1070-1071, 1150-1157,
// visual_target == 0): sum over the rows with image_label == 1 of KLDivLoss(log_softmax(x_r), t_r) = sum_c t (log t - log p),
// divided by the number of such rows.  Same shape as the vocabulary cross-entropy above: forward keeps each row's
// log-sum-exp and target mass, backward writes g / count * (softmax * sum_c t - t) as the bf16 GEMM operand.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void soft_kl_fwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ tgt, int ldt,
                                                           const int64_t* __restrict__ lab, float* __restrict__ lse,
                                                           float* __restrict__ tsum, float* __restrict__ rowloss, int C) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    if (lab[r] != 1) {
        if (threadIdx.x == 0) { lse[r] = 0.f; tsum[r] = 0.f; rowloss[r] = 0.f; }
        return;
    }
    const float* xr = x + (size_t)r * ld;
    const float* tr = tgt + (size_t)r * ldt;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, xr[c]);
    mx = block_max256(mx, sh);
    float z = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) z += expf(xr[c] - mx);
    z = block_sum256(z, sh);
    const float l = mx + logf(z);
    float acc = 0.f, ts = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float t = tr[c];
        ts += t;
        if (t > 0.f) acc += t * (logf(t) - (xr[c] - l));     // torch's kl_div: 0 where the target is 0 (xlogy)
        else acc -= t * (xr[c] - l);                          // (a non-positive target contributes -t * log p, as in torch: t = 0 -> 0)
    }
    acc = block_sum256(acc, sh);
    ts = block_sum256(ts, sh);
    if (threadIdx.x == 0) { lse[r] = l; tsum[r] = ts; rowloss[r] = acc; }
}
__global__ __launch_bounds__(256) void soft_kl_finalize_kernel(const float* __restrict__ rowloss, const int64_t* __restrict__ lab,
                                                                float* __restrict__ loss, float* __restrict__ count, int R) {
    __shared__ float sh[4];
    float s = 0.f, n = 0.f;
    for (int r = threadIdx.x; r < R; r += 256)
        if (lab[r] == 1) { s += rowloss[r]; n += 1.f; }
    s = block_sum256(s, sh);
    n = block_sum256(n, sh);
    if (threadIdx.x == 0) { loss[0] = s / n; count[0] = n; }     // the reference divides by max(n, 0) = n (vilbert.py:1157)
}
template <typename T>     // bf16: the zero-padded GEMM operand of the throughput path; float: the fp32 training path
__global__ __launch_bounds__(256) void soft_kl_bwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ tgt, int ldt,
                                                           const int64_t* __restrict__ lab, const float* __restrict__ lse,
                                                           const float* __restrict__ tsum, const float* __restrict__ count,
                                                           const float* __restrict__ gloss, T* __restrict__ d, int ldd, int C) {
    const int r = blockIdx.y;
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c0 >= ldd) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (lab[r] == 1) {
        const float g = (gloss ? gloss[0] : 1.f) / count[0];
        const float l = lse[r], ts = tsum[r];
        const float* xr = x + (size_t)r * ld;
        const float* tr = tgt + (size_t)r * ldt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            if (c < C) v[e] = g * (expf(xr[c] - l) * ts - tr[c]);
        }
    }
    store4(d + (size_t)r * ldd + c0, v);
}

// ------------------------------------------------------------------------------------------------
// fused AdamW over a flat arena
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, bf16* __restrict__ p16, int64_t n,
                                                     const int64_t* __restrict__ seg_end, const float* __restrict__ seg_wd, int nseg,
                                                     float lr, float beta1, float beta2, float eps, float bc1, float bc2,
                                                     int mode, float grad_scale) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        // segment lookup (binary search over exclusive ends); 4-element groups never straddle a segment
        // boundary because every segment start is 4-aligned in the arena.
        int lo = 0, hi = nseg - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (i < seg_end[mid]) hi = mid; else lo = mid + 1; }
        const float wd = seg_wd[lo];
        const int cnt = (i + 4 <= n) ? 4 : (int)(n - i);
        for (int j = 0; j < cnt; ++j) {
            const float gj = g[i + j] * grad_scale;
            float pj = p[i + j];
            const float mj = beta1 * m[i + j] + (1.f - beta1) * gj;
            const float vj = beta2 * v[i + j] + (1.f - beta2) * gj * gj;
            m[i + j] = mj; v[i + j] = vj;
            if (mode == 0) {
                // transformers.AdamW (optimizers.py:8-17 imports it): step_size = lr*sqrt(bc2)/bc1, denom = sqrt(v)+eps,
                // decoupled decay applied AFTER the update with plain lr.
                const float denom = sqrtf(vj) + eps;
                pj -= (lr * sqrtf(bc2) / bc1) * (mj / denom);
                if (wd > 0.f) pj -= lr * wd * pj;
            } else {
                // torch.optim.AdamW: decay first, denom = sqrt(v)/sqrt(bc2) + eps
                pj *= (1.f - lr * wd);
                const float denom = sqrtf(vj) / sqrtf(bc2) + eps;
                pj -= (lr / bc1) * (mj / denom);
            }
            p[i + j] = pj;
            if (p16) p16[i + j] = (bf16)pj;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// multi-tensor AdamW / gradient L2 norm: one launch covers up to MMF_MT_MAX tensors (grid.y = tensor, grid.x = chunk)
// ------------------------------------------------------------------------------------------------
constexpr int MT_CHUNK = 16384;   // elements per workgroup
__global__ void optim_state_advance_kernel(float* state, int schedule, float warmup, float total) {
    const float t = state[0] + 1.f;
    state[0] = t;
    float f = 1.f;
    if (schedule == 1) {
        const float s = t - 1.f;
        f = (s < warmup) ? s / fmaxf(1.f, warmup) : fmaxf(0.f, (total - s) / fmaxf(1.f, total - warmup));
    }
    state[1] = f;
}

// One-dimensional grid over the (tensor, chunk) pairs of the launch: cstart[t] = first block of tensor t (no empty workgroups
// beside a large tensor).  A workgroup owns ADAM_CHUNK = 4096 elements = ONE pass of 4 x 16 bytes per thread and stream with all
// loads issued before anything is computed; g, m, v and (round 4: -0.04 ms per step, tools/step_ab.py) the fp32 master p are touched once per step, so
// they move with non-temporal hints and leave the caches to the bf16 shadow the next forward reads (tools/ubench/adam_bench.hip: 4.4 -> 5.8 TB/s on the word-embedding table).
constexpr int ADAM_CHUNK = 4096;
struct AdamLaunch {
    mmf_adamw_multi_desc d;
    int cstart[MMF_MT_MAX + 1];
};
// (`total` chunks over gridDim.x workgroups: one chunk each by default; a capped grid would stride over
// the chunk list — the form that runs beside a GEMM launch on the CUs it leaves idle.)
__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamLaunch a, float bc1_in, float bc2_in, int total) {
    const mmf_adamw_multi_desc& d = a.d;
  for (int blk = blockIdx.x; blk < total; blk += gridDim.x) {
    float bc1 = bc1_in, bc2 = bc2_in;
    int t = 0;
    for (int i = 1; i < d.n; ++i) t += (blk >= a.cstart[i]) ? 1 : 0;
    const int64_t n = d.numel[t];
    const int64_t base = (int64_t)(blk - a.cstart[t]) * ADAM_CHUNK;
    if (base >= n) continue;
    float* __restrict__ p = reinterpret_cast<float*>(d.p[t]);
    const float* __restrict__ g = reinterpret_cast<const float*>(d.g[t]);
    const bf16* __restrict__ g16 = reinterpret_cast<const bf16*>(d.g[t]);
    const bool gb = (d.g_bf16_mask >> t) & 1ull;      // the gradient is a bf16 wire buffer
    float* __restrict__ m = reinterpret_cast<float*>(d.m[t]);
    float* __restrict__ v = reinterpret_cast<float*>(d.v[t]);
    bf16* __restrict__ p16 = reinterpret_cast<bf16*>(d.p16[t]);
    float* __restrict__ p32 = reinterpret_cast<float*>(d.p32[t]);
    float lr = d.lr[t];
    const float wd = d.wd[t], b1 = d.beta1, b2 = d.beta2, eps = d.eps;
    if (d.dev_state) {   // step count and schedule factor live in HBM (hipGraph replay)
        const float step = d.dev_state[0];
        lr *= d.dev_state[1];
        if (d.correct_bias) { bc1 = 1.f - powf(b1, step); bc2 = 1.f - powf(b2, step); }
    }
    float gs = d.grad_scale;
    if (d.norm_sq) {   // gradient clipping folded into the update: coef = min(1, max_norm / (||g|| + 1e-6))
        const float coef = d.max_norm / (sqrtf(d.norm_sq[0]) * d.grad_scale + 1e-6f);    // the norm of the SCALED gradients
        gs *= coef < 1.f ? coef : 1.f;
    }
    const int64_t end = (base + ADAM_CHUNK < n) ? base + ADAM_CHUNK : n;
    f32x4 pv[4], gv[4], mv[4], vv[4];
    bool vec[4];
    int cnt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + threadIdx.x * 4 + 1024 * u;
        cnt[u] = (i >= end) ? 0 : ((i + 4 <= end) ? 4 : (int)(end - i));
        vec[u] = (cnt[u] == 4) && ((i & 3) == 0);
        if (vec[u]) {
            pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
            if (gb) {
                const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(g16 + i);
                gv[u] = f32x4{(float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]};
            } else gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
            mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + i));
            vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + i));
        } else {
            for (int j = 0; j < 4; ++j) { const bool ok = j < cnt[u]; pv[u][j] = ok ? p[i + j] : 0.f; gv[u][j] = ok ? (gb ? (float)g16[i + j] : g[i + j]) : 0.f; mv[u][j] = ok ? m[i + j] : 0.f; vv[u][j] = ok ? v[i + j] : 0.f; }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (cnt[u] == 0) continue;
        const int64_t i = base + threadIdx.x * 4 + 1024 * u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = gv[u][j] * gs;
            mv[u][j] = b1 * mv[u][j] + (1.f - b1) * gj;
            vv[u][j] = b2 * vv[u][j] + (1.f - b2) * gj * gj;
            if (d.mode == 0) {   // transformers.AdamW
                pv[u][j] -= (lr * sqrtf(bc2) / bc1) * (mv[u][j] / (sqrtf(vv[u][j]) + eps));
                if (wd > 0.f) pv[u][j] -= lr * wd * pv[u][j];
            } else {             // torch.optim.AdamW
                pv[u][j] *= (1.f - lr * wd);
                pv[u][j] -= (lr / bc1) * (mv[u][j] / (sqrtf(vv[u][j]) / sqrtf(bc2) + eps));
            }
        }
        if (vec[u]) {
            __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p + i));
            __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m + i));
            __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + i));
            if (p16) store4(p16 + i, pv[u]);
            if (p32) store4(p32 + i, pv[u]);
        } else for (int j = 0; j < cnt[u]; ++j) { p[i + j] = pv[u][j]; m[i + j] = mv[u][j]; v[i + j] = vv[u][j]; if (p16) p16[i + j] = (bf16)pv[u][j]; if (p32) p32[i + j] = pv[u][j]; }
    }
  }
}
__global__ __launch_bounds__(256) void l2norm_multi_kernel(mmf_tensor_list d, float* __restrict__ partials) {
    __shared__ float red[4];
    const int t = blockIdx.y;
    const int64_t n = d.numel[t];
    const int64_t base = (int64_t)blockIdx.x * MT_CHUNK;
    float s = 0.f;
    if (base < n) {
        const float* __restrict__ g = reinterpret_cast<const float*>(d.ptr[t]);
        const int64_t end = (base + MT_CHUNK < n) ? base + MT_CHUNK : n;
        for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {
            if (i + 4 <= end && (i & 3) == 0) { const f32x4 x = load4(g + i); s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]; }
            else for (int64_t j = i; j < end && j < i + 4; ++j) s += g[j] * g[j];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// dst[off_t + i] = scale * src_t[i] for up to MMF_MT_MAX fp32 tensors in one launch ((chunk, tensor) grid like the kernels above): the
// gradient bucket of the data-parallel reducer packed and pre-scaled in ONE pass — dst bf16 (the wire type) or fp32
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(mmf_tensor_list d, mmf_offset_list o, T* __restrict__ dst, float scale) {
    const int t = blockIdx.y;
    const int64_t n = d.numel[t];
    const int64_t base = (int64_t)blockIdx.x * MT_CHUNK;
    if (base >= n) return;
    const float* __restrict__ g = reinterpret_cast<const float*>(d.ptr[t]);
    T* __restrict__ out = dst + o.off[t];
    const int64_t end = (base + MT_CHUNK < n) ? base + MT_CHUNK : n;
    const bool vec = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & (4 * sizeof(T) - 1)) == 0);
    for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {
        if (vec && i + 4 <= end) { f32x4 x = load4(g + i); x *= scale; store4(out + i, x); }
        else for (int64_t j = i; j < end && j < i + 4; ++j) out[j] = (T)(g[j] * scale);
    }
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partials, int n, float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------------
// layout probes
// ------------------------------------------------------------------------------------------------
__global__ void probe_mfma16_kernel(const bf16x8* a, const bf16x8* b, f32x4* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[l], b[l], acc, 0, 0, 0);
    d[l] = acc;
}
__global__ void probe_mfma32_kernel(const bf16x8* a, const bf16x8* b, f32x16* d) {
    const int l = threadIdx.x;
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[l], b[l], acc, 0, 0, 0);
    d[l] = acc;
}
__global__ void probe_tr16_kernel(const uint4* img, const int* addr, s16x4* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) reinterpret_cast<uint4*>(lds)[i] = img[i];
    __syncthreads();
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    out[l] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + addr[l]));
}

template <int NCH, typename... A>
void launch_ln_fwd(int rows, hipStream_t s, A... a) {
    hipLaunchKernelGGL(ln_fwd_kernel<NCH>, dim3((rows + 3) / 4), dim3(256), 0, s, a...);
}
template <int NCH, typename... A>
void launch_ln_bwd(int grid, hipStream_t s, A... a) {
    hipLaunchKernelGGL(ln_bwd_kernel<NCH>, dim3(grid), dim3(64 * LnbWaves<NCH>::value), 0, s, a...);
}

inline int grid_for(int64_t n, int per_block, int cap) {
    int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" {

static bool ln_h_path(int H) { return (H % 256) == 0 && H <= 1024 && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 1); }
static int layernorm_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int H, float eps,
                              DropoutCfg dc, void* stream) {
    MMF_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null operand");
    MMF_CHECK_ARG(rows > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "layernorm_fwd: need H % 4 == 0 and H <= 2048");
    MMF_CHECK_ARG(dc.thr16 == 0 || ln_h_path(H), "layernorm_dropout_fwd: the fused dropout is built for mmf_layernorm_dropout_fusable(H) widths");
    hipStream_t s = (hipStream_t)stream;
    const int nch = (H + 255) / 256;
    const bf16* xp = (const bf16*)x; bf16* yp = (bf16*)y;
    if (ln_h_path(H)) {     // half a wave per row, 16-byte accesses
        const dim3 grid((rows + 7) / 8);
        switch (H / 256) {
            case 1: hipLaunchKernelGGL(ln_fwd_h_kernel<1>, grid, dim3(256), 0, s, xp, gamma, beta, yp, mean, rstd, rows, eps, dc); break;
            case 2: hipLaunchKernelGGL(ln_fwd_h_kernel<2>, grid, dim3(256), 0, s, xp, gamma, beta, yp, mean, rstd, rows, eps, dc); break;
            case 3: hipLaunchKernelGGL(ln_fwd_h_kernel<3>, grid, dim3(256), 0, s, xp, gamma, beta, yp, mean, rstd, rows, eps, dc); break;
            default: hipLaunchKernelGGL(ln_fwd_h_kernel<4>, grid, dim3(256), 0, s, xp, gamma, beta, yp, mean, rstd, rows, eps, dc); break;
        }
        MMF_CHECK_LAUNCH();
        return 0;
    }
    switch (nch) {
        case 1: launch_ln_fwd<1>(rows, s, xp, gamma, beta, yp, mean, rstd, rows, H, eps); break;
        case 2: launch_ln_fwd<2>(rows, s, xp, gamma, beta, yp, mean, rstd, rows, H, eps); break;
        case 3: launch_ln_fwd<3>(rows, s, xp, gamma, beta, yp, mean, rstd, rows, H, eps); break;
        case 4: launch_ln_fwd<4>(rows, s, xp, gamma, beta, yp, mean, rstd, rows, H, eps); break;
        default: launch_ln_fwd<8>(rows, s, xp, gamma, beta, yp, mean, rstd, rows, H, eps); break;
    }
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int H, float eps, void* stream) {
    return layernorm_fwd_impl(x, gamma, beta, y, mean, rstd, rows, H, eps, DropoutCfg{0u, 0u, 1.f, nullptr}, stream);
}
int mmf_layernorm_dropout_fusable(int H) { return ln_h_path(H) ? 1 : 0; }
int mmf_layernorm_dropout_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int H, float eps,
                              uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, void* stream) {
    return layernorm_fwd_impl(x, gamma, beta, y, mean, rstd, rows, H, eps, DropoutCfg{drop_key, drop_thr16, drop_scale, drop_seed}, stream);
}

int mmf_layernorm_bwd_ws_floats(int H) { return LNB_MAX_GRID * 3 * H; }

// workgroups (= partial rows) of the half-wave backward for `rows` rows
static int lnb_h_grid(int rows) {
    int grid = (rows + 15) / 16;
    if (grid > LNB_MAX_GRID) grid = LNB_MAX_GRID;
    const int tg = 0;
    if (tg > 0 && tg < grid) grid = tg;
    return grid;
}
static bool lnb_h_path(int H, bool dbias) { return (H % 256) == 0 && H <= 1024 && !(H == 1024 && dbias) && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 1); }

int mmf_layernorm_bwd_deferrable(int rows, int H) { return rows > 0 && lnb_h_path(H, false) ? 1 : 0; }
// (library-internal, gemm.hip: the LayerNorm rider of the grouped weight-gradient launch) number of 256-thread blocks of the deferred backward of `rows` rows
// when it is the H = 768 two-rows-in-flight half-wave form — the form the rider implements — else 0
__attribute__((visibility("hidden"))) int mmf_lnb_rider_blocks(int rows, int H) {
    // (few token rows - MMBT at B = 8, 1824 rows - make the tiles' K-loops short and the joint launch measured 0.3 % slower than the two: 2048 rows and up ride)
    if (H != 768 || rows < 2048 || !lnb_h_path(H, false) || (mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & (8 | 32))) return 0;
    const int grid = lnb_h_grid(rows);
    return rows > 8 * grid ? grid : 0;
}

int mmf_layernorm_bwd_reduce_multi(const mmf_ln_reduce_list* d, void* stream) {
    MMF_CHECK_ARG(d && d->n > 0 && d->n <= MMF_MT_MAX, "layernorm_bwd_reduce_multi: bad list");
    LnReduceArgs a;
    int hmax = 0;
    for (int i = 0; i < d->n; ++i) {
        MMF_CHECK_ARG(d->partials[i] && d->dgamma[i] && d->dbeta[i] && mmf_layernorm_bwd_deferrable(d->rows[i], d->H[i]),
                      "layernorm_bwd_reduce_multi: entry was not produced by a deferred mmf_layernorm_bwd");
        a.partials[i] = d->partials[i]; a.o0[i] = d->dgamma[i]; a.o1[i] = d->dbeta[i];
        a.nblk[i] = lnb_h_grid(d->rows[i]); a.H[i] = d->H[i];
        hmax = d->H[i] > hmax ? d->H[i] : hmax;
    }
    hipLaunchKernelGGL(ln_bwd_reduce_multi_kernel, dim3((hmax + 63) / 64, 2, d->n), dim3(1024), 0, (hipStream_t)stream, a);
    MMF_CHECK_LAUNCH();
    return 0;
}

static int layernorm_bwd_impl(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                      void* dlin, uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, float* dgamma,
                      float* dbeta, float* dbias, int accumulate, float* partials, int rows, int H, DropoutCfg din, void* stream) {
    MMF_CHECK_ARG(dy && x && mean && rstd && gamma && dx && partials, "layernorm_bwd: null operand");
    MMF_CHECK_ARG(din.thr16 == 0 || lnb_h_path(H, dbias != nullptr), "layernorm_bwd_din: the fused input dropout is built for mmf_layernorm_dropout_fusable(H) widths");
    MMF_CHECK_ARG(rows > 0 && H > 0 && (H % 4) == 0 && H <= 2048, "layernorm_bwd: need H % 4 == 0 and H <= 2048");
    MMF_CHECK_ARG(drop_thr16 == 0 || dlin, "layernorm_bwd: dropout needs dlin");
    hipStream_t s = (hipStream_t)stream;
    DropoutCfg dc{drop_key, drop_thr16, drop_scale, drop_seed};
    const int tg = 0;
    if (lnb_h_path(H, dbias != nullptr)) {     // half a wave per row, 16-byte accesses
        const int grid = lnb_h_grid(rows);
        const bf16* dyp = (const bf16*)dy; const bf16* xp = (const bf16*)x; bf16* dxp = (bf16*)dx; bf16* dlp = (bf16*)dlin;
        // two rows of a half-wave in flight whenever it owns more than one (MMF_TUN_ALT_FORMS bit 3: one at a time, the schedule of the input-dropout form; its bit-equality test) — except at H = 1024, where
        // the two-row form needs all 256 registers = ONE wave per SIMD and loses to the one-row form at two (isolated, backward + dropout + reduce: 14.1 vs 13.0 us
        // at 3232 rows, 36.5 vs 28.7 at 14592: profiles/r05_experiments.txt section 14)
        const bool two = rows > 8 * grid && !(mmf_amd_get_tunable(MMF_TUN_ALT_FORMS) & 8) && H < 1024;
#define MMF_LNB_H(NC)                                                                                                              \
        /* (input dropout: one row in flight per half-wave — with two the H = 768 form needs all 256 registers and a single wave per SIMD) */ \
        if (din.thr16) hipLaunchKernelGGL((ln_bwd_h_kernel<NC, false, 1, true>), dim3(grid), dim3(256), 0, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, din); \
        else if (dbias) hipLaunchKernelGGL((ln_bwd_h_kernel<NC, true, 1>), dim3(grid), dim3(256), 0, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, din); \
        else if (two) hipLaunchKernelGGL((ln_bwd_h_kernel<NC, false, 2>), dim3(grid), dim3(256), 0, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, din); \
        else hipLaunchKernelGGL((ln_bwd_h_kernel<NC, false, 1>), dim3(grid), dim3(256), 0, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, din);
        switch (H / 256) {
            case 1: MMF_LNB_H(1) break;
            case 2: MMF_LNB_H(2) break;
            case 3: MMF_LNB_H(3) break;
            default: MMF_LNB_H(4) break;
        }
#undef MMF_LNB_H
        MMF_CHECK_LAUNCH();
        if (!dgamma && !dbeta && !dbias) return 0;     // deferred: the caller finishes with mmf_layernorm_bwd_reduce_multi
        hipLaunchKernelGGL(ln_bwd_reduce_h_kernel, dim3((H + 63) / 64, dbias ? 3 : 2), dim3(1024), 0, s, partials, grid, H, dgamma, dbeta, dbias, accumulate);
        MMF_CHECK_LAUNCH();
        return 0;
    }
    MMF_CHECK_ARG(dgamma || dbeta || dbias, "layernorm_bwd: deferring the column sums needs mmf_layernorm_bwd_deferrable(rows, H)");
    const int nch_ = (H + 255) / 256;
    const int grid = grid_for(rows, nch_ >= 5 ? 4 : nch_ >= 4 ? 8 : LNB_WAVES, tg > 0 ? (tg > LNB_MAX_GRID ? LNB_MAX_GRID : tg) : (nch_ >= 4 ? 2 * LNB_GRID : LNB_GRID));
    const int nch = (H + 255) / 256;
    const bf16* dyp = (const bf16*)dy; const bf16* xp = (const bf16*)x; bf16* dxp = (bf16*)dx; bf16* dlp = (bf16*)dlin;
    switch (nch) {
        case 1: launch_ln_bwd<1>(grid, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, H); break;
        case 2: launch_ln_bwd<2>(grid, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, H); break;
        case 3: launch_ln_bwd<3>(grid, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, H); break;
        case 4: launch_ln_bwd<4>(grid, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, H); break;
        case 5: case 6: launch_ln_bwd<6>(grid, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, H); break;
        default: launch_ln_bwd<8>(grid, s, dyp, xp, mean, rstd, gamma, dxp, dlp, dc, partials, rows, H); break;
    }
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((H + 31) / 32, 3), dim3(256), 0, s, partials, grid, H, dgamma, dbeta, dbias,
                       accumulate);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                      void* dlin, uint32_t drop_key, uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, float* dgamma,
                      float* dbeta, float* dbias, int accumulate, float* partials, int rows, int H, void* stream) {
    return layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, dlin, drop_key, drop_thr16, drop_scale, drop_seed, dgamma, dbeta, dbias, accumulate, partials, rows, H,
                              DropoutCfg{0u, 0u, 1.f, nullptr}, stream);
}
int mmf_layernorm_bwd_din(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx, uint32_t in_key, uint32_t in_thr16,
                          float in_scale, const uint32_t* in_seed, float* dgamma, float* dbeta, int accumulate, float* partials, int rows, int H, void* stream) {
    return layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, nullptr, 0u, 0u, 1.f, nullptr, dgamma, dbeta, nullptr, accumulate, partials, rows, H,
                              DropoutCfg{in_key, in_thr16, in_scale, in_seed}, stream);
}

int mmf_amd_take_index_error(void) {
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_index_error), sizeof(int)) != hipSuccess) { mmf_amd_set_error("take_index_error: copy failed"); return -1; }
    if (v) { const int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_index_error), &z, sizeof(int)); }
    return v;
}

int mmf_embed_text_fwd(const int64_t* ids, const int64_t* seg, const float* word, const float* pos, const float* type, void* y,
                       int B, int T, int S, int H, int row0, int pos0, int V, int P, int NT, void* stream) {
    MMF_CHECK_ARG(ids && word && pos && type && y, "embed_text_fwd: null operand");
    MMF_CHECK_ARG(B > 0 && T > 0 && row0 >= 0 && S >= row0 + T && pos0 >= 0 && (H % 4) == 0, "embed_text_fwd: bad shape");
    MMF_CHECK_ARG(P <= 0 || pos0 + T <= P, "embed_text_fwd: sequence longer than the position table (max_position_embeddings)");
    hipLaunchKernelGGL(embed_text_kernel<bf16>, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, seg, word, pos, type,
                       (bf16*)y, B, T, S, H, row0, pos0, V, NT);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_embed_text_f32_fwd(const int64_t* ids, const int64_t* seg, const float* word, const float* pos, const float* type, float* y,
                           int B, int T, int S, int H, int row0, int pos0, int V, int P, int NT, void* stream) {
    MMF_CHECK_ARG(ids && word && pos && type && y, "embed_text_f32_fwd: null operand");
    MMF_CHECK_ARG(B > 0 && T > 0 && row0 >= 0 && S >= row0 + T && pos0 >= 0 && (H % 4) == 0, "embed_text_f32_fwd: bad shape");
    MMF_CHECK_ARG(P <= 0 || pos0 + T <= P, "embed_text_f32_fwd: sequence longer than the position table (max_position_embeddings)");
    hipLaunchKernelGGL(embed_text_kernel<float>, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, seg, word, pos, type,
                       y, B, T, S, H, row0, pos0, V, NT);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_align_pos_fwd(const int64_t* align, const float* pos, const float* typ, const int64_t* typ_idx, float* out, int rows, int A, int H, int P, int NT,
                      void* stream) {
    MMF_CHECK_ARG(align && pos && out && rows > 0 && A > 0 && P > 0 && (H % 4) == 0, "align_pos_fwd: bad operand");
    MMF_CHECK_ARG(!typ || (typ_idx && NT > 0), "align_pos_fwd: a type table needs its indices");
    hipLaunchKernelGGL(align_pos_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, align, pos, typ, typ_idx, out, rows, A, H, P, NT);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_align_pos_bwd(const void* dvis, int ld, int nb, int rpb, int bstride, const int64_t* align, float* dpos, int A, int H, int P, void* stream) {
    MMF_CHECK_ARG(dvis && align && dpos && nb > 0 && rpb > 0 && A > 0 && P > 0 && H > 0 && ld >= H, "align_pos_bwd: bad operand");
    const int rows = nb * rpb;
    hipLaunchKernelGGL(align_pos_bwd_kernel<bf16>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)dvis, ld, rpb, bstride, align, dpos, rows, A,
                       H, P);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_align_pos_f32_bwd(const float* dvis, int ld, int nb, int rpb, int bstride, const int64_t* align, float* dpos, int A, int H, int P, void* stream) {
    MMF_CHECK_ARG(dvis && align && dpos && nb > 0 && rpb > 0 && A > 0 && P > 0 && H > 0 && ld >= H, "align_pos_f32_bwd: bad operand");
    const int rows = nb * rpb;
    hipLaunchKernelGGL(align_pos_bwd_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dvis, ld, rpb, bstride, align, dpos, rows, A, H, P);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_rows_add_embed(const void* x, const int64_t* seg, const float* pos, const float* type, void* y, int B, int L, int S, int H,
                       int row0, int pos0, void* stream) {
    MMF_CHECK_ARG(x && y, "rows_add_embed: null operand");
    MMF_CHECK_ARG(B > 0 && L > 0 && row0 >= 0 && S >= row0 + L && pos0 >= 0 && (H % 4) == 0, "rows_add_embed: bad shape");
    hipLaunchKernelGGL(rows_add_embed_kernel<bf16>, dim3((B * L + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, seg, pos, type,
                       (bf16*)y, B, L, S, H, row0, pos0);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_rows_add_embed_f32(const float* x, const int64_t* seg, const float* pos, const float* type, float* y, int B, int L, int S, int H,
                           int row0, int pos0, void* stream) {
    MMF_CHECK_ARG(x && y, "rows_add_embed_f32: null operand");
    MMF_CHECK_ARG(B > 0 && L > 0 && row0 >= 0 && S >= row0 + L && pos0 >= 0 && (H % 4) == 0, "rows_add_embed_f32: bad shape");
    hipLaunchKernelGGL(rows_add_embed_kernel<float>, dim3((B * L + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, seg, pos, type, y, B, L, S,
                       H, row0, pos0);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_rows_add_table_f32(const float* x, const int64_t* idx, const float* table, void* y, int rows, int D, void* stream) {
    MMF_CHECK_ARG(x && y && rows > 0 && D > 0 && (D % 4) == 0, "rows_add_table_f32: bad operand");
    hipLaunchKernelGGL(rows_add_table_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, idx, table, (bf16*)y, rows, D);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_copy_rows_bf16(const void* src, int src_bstride, void* dst, int dst_bstride, int nb, int rpb, int H, void* stream) {
    MMF_CHECK_ARG(src && dst, "copy_rows: null operand");
    MMF_CHECK_ARG(nb > 0 && rpb > 0 && src_bstride >= rpb && dst_bstride >= rpb && (H % 8) == 0, "copy_rows: bad shape");
    hipLaunchKernelGGL(copy_rows_kernel, dim3((nb * rpb + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, src_bstride,
                       (bf16*)dst, dst_bstride, nb, rpb, H);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_rows_scatter_add_ws_floats(int H) { return FEW_GROUPS * 2 * H; }
int mmf_rows_scatter_add(const void* x, int ld, int nb, int rpb, int bstride, const int64_t* idx, int idx_ld, int per_pos,
                         int idx_base, float* out, int H, int few_buckets, int nbuckets, float* ws, int skip_bucket,
                         void* stream) {
    MMF_CHECK_ARG(x && out, "rows_scatter_add: null operand");
    MMF_CHECK_ARG(!(few_buckets && skip_bucket >= 0), "rows_scatter_add: skip_bucket is for the atomic (large-table) form");
    MMF_CHECK_ARG(nb > 0 && rpb > 0 && (H % 4) == 0 && (ld % 4) == 0, "rows_scatter_add: bad shape");
    MMF_CHECK_ARG(idx || nbuckets <= 0 || (per_pos ? idx_base + rpb <= nbuckets : idx_base < nbuckets), "rows_scatter_add: bucket outside the table");
    const int total = nb * rpb;
    if (few_buckets) {
        MMF_CHECK_ARG(ws && nbuckets >= 1, "rows_scatter_add: few_buckets needs a workspace and the bucket count");
        const int groups = grid_for(total, 4, FEW_GROUPS);
        hipLaunchKernelGGL(scatter_add_few_kernel, dim3(groups, (H + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ld,
                           nb, rpb, bstride, idx, idx_ld, per_pos, idx_base, out, H, ws, nbuckets);
        MMF_CHECK_LAUNCH();
        hipLaunchKernelGGL(scatter_add_few_reduce_kernel, dim3((H + 255) / 256, nbuckets < 2 ? nbuckets : 2), dim3(256), 0,
                           (hipStream_t)stream, ws, groups, H, nbuckets, out);
    } else if (!idx && per_pos && skip_bucket < 0) {
        hipLaunchKernelGGL(scatter_add_pos_kernel, dim3(rpb, (H / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ld, nb, rpb,
                           bstride, idx_base, out, H);
    } else if (idx && total <= SCATTER_UNIQUE_MAX && (H % 16) == 0 && mmf_amd_get_tunable(MMF_TUN_SCATTER_ATOMIC) != 1) {      // deterministic, no atomics (index array given: idx_base / per_pos unused)
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_add_unique_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SCATTER_UNIQUE_MAX * 4);
            if (ae != hipSuccess) { mmf_amd_set_error(hipGetErrorString(ae)); return 2; }
            attr_set = true;
        }
        hipLaunchKernelGGL(scatter_add_unique_kernel, dim3(total), dim3(256), total * 4, (hipStream_t)stream, (const bf16*)x, ld, nb, rpb, bstride, idx,
                           idx_ld, out, H, skip_bucket, nbuckets);
    } else {
        hipLaunchKernelGGL(scatter_add_direct_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                           ld, nb, rpb, bstride, idx, idx_ld, per_pos, idx_base, out, H, skip_bucket, nbuckets);
    }
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_embed_tables_bwd_ws_floats(int S, int H) { return S * 3 * H; }
int mmf_embed_tables_bwd(const void* x, int ld, int B, int T, int R, const int64_t* seg, const int64_t* vt, int pos0, float* dpos, int P, float* dtyp, int NT,
                         float* dtyp_vis, int NTV, float* dpos_vis, int H, float* ws, void* stream) {
    MMF_CHECK_ARG(x && ws && B > 0 && T > 0 && R >= 0 && (H % 4) == 0 && (ld % 4) == 0 && ld >= H, "embed_tables_bwd: bad operand");
    MMF_CHECK_ARG(!dpos || (pos0 >= 0 && pos0 + T <= P), "embed_tables_bwd: positions outside the table");
    MMF_CHECK_ARG((!dtyp || (seg && NT >= 1)) && (!dtyp_vis || (vt && NTV >= 1 && R > 0)) && (!dpos_vis || R > 0), "embed_tables_bwd: a table gradient without its index / rows");
    hipLaunchKernelGGL(embed_tables_bwd_kernel, dim3(T + R, (H / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ld, B, T, R, dtyp ? seg : nullptr,
                       dtyp_vis ? vt : nullptr, pos0, dpos, dtyp, NT, dtyp_vis, NTV, ws, H);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(embed_tables_reduce_kernel, dim3((H + 63) / 64, R > 0 ? 5 : 2), dim3(256), 0, (hipStream_t)stream, ws, T, R, H, dtyp, NT, dtyp_vis, NTV, dpos_vis);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_gather_rows(const void* x, const int64_t* index, void* out, int B, int S, int H, uint32_t drop_key,
                    uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, void* stream) {
    MMF_CHECK_ARG(x && index && out && (H % 4) == 0, "gather_rows: bad operand");
    hipLaunchKernelGGL(gather_rows_kernel<bf16>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, index, (bf16*)out,
                       B, S, H, DropoutCfg{drop_key, drop_thr16, drop_scale, drop_seed}, 0);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_gather_rows_f32(const float* x, const int64_t* index, float* out, int B, int S, int H, void* stream) {
    MMF_CHECK_ARG(x && index && out && (H % 4) == 0, "gather_rows_f32: bad operand");
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, index, out,
                       B, S, H, DropoutCfg{0u, 0u, 1.f, nullptr}, 0);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_scatter_rows(const void* dout, const int64_t* index, void* dx, int B, int S, int H, uint32_t drop_key,
                     uint32_t drop_thr16, float drop_scale, const uint32_t* drop_seed, void* stream) {
    MMF_CHECK_ARG(dout && index && dx && (H % 4) == 0, "scatter_rows: bad operand");
    hipLaunchKernelGGL(gather_rows_kernel<bf16>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout, index,
                       (bf16*)dx, B, S, H, DropoutCfg{drop_key, drop_thr16, drop_scale, drop_seed}, 1);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_scatter_rows_full(const void* dout, const int64_t* index, void* dx, int B, int S, int H, uint32_t drop_key, uint32_t drop_thr16,
                          float drop_scale, const uint32_t* drop_seed, void* stream) {
    MMF_CHECK_ARG(dout && index && dx && B > 0 && S > 0 && (H % 8) == 0, "scatter_rows_full: bad operand (H % 8 == 0)");
    const long rows = (long)B * S;
    hipLaunchKernelGGL(scatter_rows_full_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout, index, (bf16*)dx, B,
                       S, H, DropoutCfg{drop_key, drop_thr16, drop_scale, drop_seed});
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_colsum_ws_floats(int N) { return CS_GROUPS * N; }
int mmf_colsum_bf16(const void* x, int ld, int nb, int rpb, int bstride, int N, float* out, float beta, float* partials,
                    void* stream) {
    MMF_CHECK_ARG(x && out && partials, "colsum: null operand");
    MMF_CHECK_ARG(nb > 0 && rpb > 0 && N > 0 && (ld % 4) == 0, "colsum: bad shape");
    const int groups = grid_for((int64_t)nb * rpb, 4, CS_GROUPS);
    hipLaunchKernelGGL(colsum_kernel, dim3(groups, (N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ld, nb, rpb,
                       bstride, N, partials);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, partials, groups, N, out, beta);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    MMF_CHECK_ARG(src && dst && n >= 0, "cast: null operand");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 2048, 4096)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    MMF_CHECK_ARG(src && dst && n >= 0, "cast: null operand");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 1024, 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, dst, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_make_additive_mask(const int64_t* mask, float* out, int64_t n, void* stream) {
    MMF_CHECK_ARG(mask && out && n > 0, "additive_mask: bad operand");
    hipLaunchKernelGGL(additive_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, out, n);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_visual_masks(const int64_t* input_mask, const int64_t* image_dim, int B, int T, int R, int64_t* image_mask, int64_t* attention_mask,
                     int64_t* visual_embeddings_type, float* mask_add, int64_t* pool_index, void* stream) {
    MMF_CHECK_ARG(input_mask && image_mask && attention_mask && visual_embeddings_type && mask_add && pool_index && B > 0 && T > 0 && R > 0,
                  "visual_masks: bad operand");
    hipLaunchKernelGGL(visual_masks_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, input_mask, image_dim, T, R, image_mask, attention_mask,
                       visual_embeddings_type, mask_add, pool_index);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_dropout_bf16(const void* x, void* y, int64_t n, uint32_t drop_key, uint32_t drop_thr16, float drop_scale,
                     const uint32_t* drop_seed, void* stream) {
    MMF_CHECK_ARG(x && y && n > 0 && n < ((int64_t)1 << 32), "dropout: bad operand");
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 1024, 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n,
                       DropoutCfg{drop_key, drop_thr16, drop_scale, drop_seed});
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_gelu_bwd_bf16(const void* dh, const void* u, void* du, int64_t n, void* stream) {
    MMF_CHECK_ARG(dh && u && du && n > 0, "gelu_bwd: bad operand");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n, 1024, 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dh, (const bf16*)u,
                       (bf16*)du, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_eltwise_bf16(int op, const void* a, const void* b, void* out, int64_t n, void* stream) {
    MMF_CHECK_ARG(a && out && n > 0 && op >= 0 && op <= 3 && (op == 1 || b), "eltwise: bad operand");
    hipLaunchKernelGGL(eltwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, (const bf16*)a,
                       (const bf16*)b, (bf16*)out, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_tanh_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, void* stream) {
    MMF_CHECK_ARG(dy && y && dx && n > 0, "tanh_bwd: bad operand");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)y, (bf16*)dx, n);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_cast2d_f32_to_bf16(const float* src, int lds, void* dst, int ldd, int rows, int cols, void* stream) {
    MMF_CHECK_ARG(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols, "cast2d: bad operand");
    const int64_t n = (int64_t)rows * ldd;
    hipLaunchKernelGGL(cast2d_f32_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, lds, (bf16*)dst,
                       ldd, rows, cols);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_cast2d_bf16_to_f32(const void* src, int lds, float* dst, int ldd, int rows, int cols, void* stream) {
    MMF_CHECK_ARG(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols, "cast2d: bad operand");
    const int64_t n = (int64_t)rows * cols;
    hipLaunchKernelGGL(cast2d_bf16_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, lds,
                       dst, ldd, rows, cols);
    MMF_CHECK_LAUNCH();
    return 0;
}

__global__ void seed_advance_kernel(uint32_t* seed) { seed[0] += 1u; }
}  // extern "C" (kernel must not have C linkage)
extern "C" {
int mmf_seed_advance(uint32_t* seed, void* stream) {
    MMF_CHECK_ARG(seed, "seed_advance: null");
    hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_bce_logits_ws_floats(void) { return BCE_BLOCKS; }
int mmf_bce_logits_fwd(const float* scores, const float* targets, float* loss, float* ws, int B, int N, void* stream) {
    MMF_CHECK_ARG(scores && targets && loss && ws && B > 0 && N > 0, "bce_fwd: bad operand");
    hipLaunchKernelGGL(bce_fwd_partial_kernel, dim3(BCE_BLOCKS), dim3(256), 0, (hipStream_t)stream, scores, targets, ws, (int64_t)B * N);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_fwd_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, BCE_BLOCKS, loss, B);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_bce_logits_bwd(const float* scores, const float* targets, const float* gloss, void* dscores, int ldd, int B, int N,
                       void* stream) {
    MMF_CHECK_ARG(scores && targets && dscores && B > 0 && N > 0 && ldd >= N, "bce_bwd: bad operand");
    const int64_t n = (int64_t)B * ldd;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, targets, gloss,
                       (bf16*)dscores, ldd, B, N);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_vocab_cross_entropy_fwd(const float* logits, int ld, const int64_t* labels, float* lse, float* rowloss, float* loss, float* count,
                                int R, int C, int ignore_index, void* stream) {
    MMF_CHECK_ARG(logits && labels && lse && rowloss && loss && count && R > 0 && C > 0 && ld >= C, "vocab_cross_entropy_fwd: bad operand");
    hipLaunchKernelGGL(vocab_ce_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, lse, rowloss, C, ignore_index);
    hipLaunchKernelGGL(vocab_ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rowloss, labels, loss, count, R, C, ignore_index);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_vocab_cross_entropy_bwd(const float* logits, int ld, const int64_t* labels, const float* lse, const float* count, const float* gloss,
                                void* dlogits, int ldd, int R, int C, int ignore_index, void* stream) {
    MMF_CHECK_ARG(logits && labels && lse && count && dlogits && R > 0 && C > 0 && ld >= C, "vocab_cross_entropy_bwd: bad operand");
    MMF_CHECK_ARG(ldd >= C && (ldd % 8) == 0, "vocab_cross_entropy_bwd: ldd must be a multiple of 8 covering C (the GEMM operand's leading dimension)");
    for (int r0 = 0; r0 < R; r0 += 65535) {       // (grid.y is limited to 65535 rows per launch)
        const int rows = R - r0 < 65535 ? R - r0 : 65535;
        hipLaunchKernelGGL(vocab_ce_bwd_kernel<bf16>, dim3((ldd / 4 + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream,
                           logits + (size_t)r0 * ld, ld, labels + r0, lse + r0, count, gloss, (bf16*)dlogits + (size_t)r0 * ldd, ldd, C,
                           ignore_index);
    }
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_vocab_cross_entropy_f32_bwd(const float* logits, int ld, const int64_t* labels, const float* lse, const float* count, const float* gloss,
                                    float* dlogits, int ldd, int R, int C, int ignore_index, void* stream) {
    MMF_CHECK_ARG(logits && labels && lse && count && dlogits && R > 0 && C > 0 && ld >= C, "vocab_cross_entropy_f32_bwd: bad operand");
    MMF_CHECK_ARG(ldd >= C && (ldd % 4) == 0, "vocab_cross_entropy_f32_bwd: ldd must be a multiple of 4 covering C");
    for (int r0 = 0; r0 < R; r0 += 65535) {
        const int rows = R - r0 < 65535 ? R - r0 : 65535;
        hipLaunchKernelGGL(vocab_ce_bwd_kernel<float>, dim3((ldd / 4 + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream,
                           logits + (size_t)r0 * ld, ld, labels + r0, lse + r0, count, gloss, dlogits + (size_t)r0 * ldd, ldd, C, ignore_index);
    }
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_soft_target_kl_fwd(const float* logits, int ld, const float* target, int ldt, const int64_t* row_label, float* lse, float* tsum,
                           float* rowloss, float* loss, float* count, int R, int C, void* stream) {
    MMF_CHECK_ARG(logits && target && row_label && lse && tsum && rowloss && loss && count && R > 0 && C > 0 && ld >= C && ldt >= C,
                  "soft_target_kl_fwd: bad operand");
    hipLaunchKernelGGL(soft_kl_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, target, ldt, row_label, lse, tsum, rowloss, C);
    hipLaunchKernelGGL(soft_kl_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rowloss, row_label, loss, count, R);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_soft_target_kl_bwd(const float* logits, int ld, const float* target, int ldt, const int64_t* row_label, const float* lse,
                           const float* tsum, const float* count, const float* gloss, void* dlogits, int ldd, int R, int C, void* stream) {
    MMF_CHECK_ARG(logits && target && row_label && lse && tsum && count && dlogits && R > 0 && C > 0 && ld >= C && ldt >= C,
                  "soft_target_kl_bwd: bad operand");
    MMF_CHECK_ARG(ldd >= C && (ldd % 8) == 0, "soft_target_kl_bwd: ldd must be a multiple of 8 covering C (the GEMM operand's leading dimension)");
    for (int r0 = 0; r0 < R; r0 += 65535) {       // (grid.y is limited to 65535 rows per launch)
        const int rows = R - r0 < 65535 ? R - r0 : 65535;
        hipLaunchKernelGGL(soft_kl_bwd_kernel<bf16>, dim3((ldd / 4 + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream,
                           logits + (size_t)r0 * ld, ld, target + (size_t)r0 * ldt, ldt, row_label + r0, lse + r0, tsum + r0, count, gloss,
                           (bf16*)dlogits + (size_t)r0 * ldd, ldd, C);
    }
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_soft_target_kl_f32_bwd(const float* logits, int ld, const float* target, int ldt, const int64_t* row_label, const float* lse,
                               const float* tsum, const float* count, const float* gloss, float* dlogits, int ldd, int R, int C, void* stream) {
    MMF_CHECK_ARG(logits && target && row_label && lse && tsum && count && dlogits && R > 0 && C > 0 && ld >= C && ldt >= C,
                  "soft_target_kl_f32_bwd: bad operand");
    MMF_CHECK_ARG(ldd >= C && (ldd % 4) == 0, "soft_target_kl_f32_bwd: ldd must be a multiple of 4 covering C");
    for (int r0 = 0; r0 < R; r0 += 65535) {
        const int rows = R - r0 < 65535 ? R - r0 : 65535;
        hipLaunchKernelGGL(soft_kl_bwd_kernel<float>, dim3((ldd / 4 + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream,
                           logits + (size_t)r0 * ld, ld, target + (size_t)r0 * ldt, ldt, row_label + r0, lse + r0, tsum + r0, count, gloss,
                           dlogits + (size_t)r0 * ldd, ldd, C);
    }
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_cross_entropy_fwd(const float* logits, const int64_t* labels, float* loss, float* count, int B, int C, int ignore_index, void* stream) {
    MMF_CHECK_ARG(logits && labels && loss && count && B > 0 && C > 0, "cross_entropy_fwd: bad operand");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, logits, labels, loss, count, B, C, ignore_index);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* count, const float* gloss, float* dlogits, int B, int C,
                          int ignore_index, void* stream) {
    MMF_CHECK_ARG(logits && labels && count && dlogits && B > 0 && C > 0, "cross_entropy_bwd: bad operand");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, labels, count, gloss, dlogits, B, C,
                       ignore_index);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_adamw_step(float* p, const float* g, float* m, float* v, void* p16, int64_t n, const int64_t* seg_end,
                   const float* seg_wd, int nseg, float lr, float beta1, float beta2, float eps, int step, int correct_bias,
                   int mode, float grad_scale, void* stream) {
    MMF_CHECK_ARG(p && g && m && v && seg_end && seg_wd && nseg > 0 && n > 0 && step >= 1, "adamw: bad operand");
    float bc1 = 1.f, bc2 = 1.f;
    if (correct_bias) {
        bc1 = 1.f - powf(beta1, (float)step);
        bc2 = 1.f - powf(beta2, (float)step);
    }
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 1024, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)p16, n,
                       seg_end, seg_wd, nseg, lr, beta1, beta2, eps, bc1, bc2, mode, grad_scale);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_adamw_multi(const mmf_adamw_multi_desc* d, void* stream) {
    MMF_CHECK_ARG(d && d->n > 0 && d->n <= MMF_MT_MAX && (d->step >= 1 || d->dev_state), "adamw_multi: bad descriptor");
    for (int i = 0; i < d->n; ++i)
        MMF_CHECK_ARG(d->p[i] && d->g[i] && d->m[i] && d->v[i] && d->numel[i] > 0, "adamw_multi: null tensor");
    float bc1 = 1.f, bc2 = 1.f;
    if (d->correct_bias && !d->dev_state) { bc1 = 1.f - powf(d->beta1, (float)d->step); bc2 = 1.f - powf(d->beta2, (float)d->step); }
    AdamLaunch a;
    a.d = *d;
    int blocks = 0;
    for (int i = 0; i < d->n; ++i) { a.cstart[i] = blocks; blocks += (int)((d->numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK); }
    for (int i = d->n; i <= MMF_MT_MAX; ++i) a.cstart[i] = blocks;
    const int cap = 0;
    const int grid = (cap > 0 && cap < blocks) ? cap : blocks;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, bc1, bc2, blocks);
    MMF_CHECK_LAUNCH();
    return 0;
}
// The two one-thread bookkeeping kernels at the head of a captured training step (dropout seed, optimizer step count / schedule factor) as ONE
// launch: a launch costs ~4.7 us inside a replayed hipGraph whatever it computes (profiles/r05_graph_replay_kernels.txt).
__global__ void step_advance_kernel(uint32_t* seed, float* state, int schedule, float warmup, float total) {
    if (seed) seed[0] += 1u;
    if (state) {
        const float t = state[0] + 1.f;
        state[0] = t;
        float f = 1.f;
        if (schedule == 1) {
            const float s = t - 1.f;
            f = (s < warmup) ? s / fmaxf(1.f, warmup) : fmaxf(0.f, (total - s) / fmaxf(1.f, total - warmup));
        }
        state[1] = f;
    }
}
int mmf_step_advance(uint32_t* seed, float* state, int schedule, float warmup_steps, float total_steps, void* stream) {
    MMF_CHECK_ARG((seed || state) && (schedule == 0 || schedule == 1), "step_advance: bad argument");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed, state, schedule, warmup_steps, total_steps);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_optim_state_advance(float* state, int schedule, float warmup_steps, float total_steps, void* stream) {
    MMF_CHECK_ARG(state && (schedule == 0 || schedule == 1), "optim_state_advance: bad argument");
    hipLaunchKernelGGL(optim_state_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, schedule, warmup_steps, total_steps);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_l2norm_sq_ws_floats(const mmf_tensor_list* d) {
    int64_t mx = 0;
    for (int i = 0; i < d->n; ++i) mx = d->numel[i] > mx ? d->numel[i] : mx;
    return (int)((mx + MT_CHUNK - 1) / MT_CHUNK) * d->n;
}
int mmf_l2norm_sq_multi(const mmf_tensor_list* d, float* out, int accumulate, float* ws, void* stream) {
    MMF_CHECK_ARG(d && d->n > 0 && d->n <= MMF_MT_MAX && out && ws, "l2norm_sq_multi: bad descriptor");
    int64_t mx = 0;
    for (int i = 0; i < d->n; ++i) { MMF_CHECK_ARG(d->ptr[i] && d->numel[i] > 0, "l2norm_sq_multi: null tensor"); mx = d->numel[i] > mx ? d->numel[i] : mx; }
    const unsigned gx = (unsigned)((mx + MT_CHUNK - 1) / MT_CHUNK);
    hipLaunchKernelGGL(l2norm_multi_kernel, dim3(gx, d->n), dim3(256), 0, (hipStream_t)stream, *d, ws);
    MMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, (int)(gx * d->n), out, accumulate);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_pack_f32_multi(const mmf_tensor_list* d, const mmf_offset_list* off, void* dst, int dst_bf16, float scale, void* stream) {
    MMF_CHECK_ARG(d && off && dst && d->n > 0 && d->n <= MMF_MT_MAX, "pack_f32_multi: bad descriptor");
    int64_t mx = 0;
    for (int i = 0; i < d->n; ++i) {
        MMF_CHECK_ARG(d->ptr[i] && d->numel[i] > 0 && off->off[i] >= 0, "pack_f32_multi: null tensor / negative offset");
        mx = d->numel[i] > mx ? d->numel[i] : mx;
    }
    const unsigned gx = (unsigned)((mx + MT_CHUNK - 1) / MT_CHUNK);
    if (dst_bf16) hipLaunchKernelGGL(pack_multi_kernel<bf16>, dim3(gx, d->n), dim3(256), 0, (hipStream_t)stream, *d, *off, (bf16*)dst, scale);
    else hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(gx, d->n), dim3(256), 0, (hipStream_t)stream, *d, *off, (float*)dst, scale);
    MMF_CHECK_LAUNCH();
    return 0;
}

int mmf_probe_mfma16(const void* a, const void* b, float* d, void* stream) {
    hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16x8*)a, (const bf16x8*)b, (f32x4*)d);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_probe_mfma32(const void* a, const void* b, float* d, void* stream) {
    hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16x8*)a, (const bf16x8*)b, (f32x16*)d);
    MMF_CHECK_LAUNCH();
    return 0;
}
int mmf_probe_tr16(const void* img, const int* addr, void* out, void* stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint4*)img, addr, (s16x4*)out);
    MMF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
