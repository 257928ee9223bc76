// mmf_amd :: the half-wave-per-row LayerNorm backward as a device function: the body of ln_bwd_h_kernel (rowops.hip) and of the LayerNorm RIDER of the grouped
// weight-gradient launch (gemm.hip: gemm_wide_grouped_ln_kernel runs the next layer's first LayerNorm backward on the CUs its 216 tiles leave idle).
#pragma once
#include "common.h"

namespace lnk {

typedef float f32x8r __attribute__((ext_vector_type(8)));
DEVI f32x8r load8(const bf16* p) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
    f32x8r r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
    return r;
}
// streaming forms (the LayerNorm rider of the weight-gradient launch: its rows pass the L2 the gradient tiles share their operand panels in)
DEVI f32x8r load8_nt(const bf16* p) {
    typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
    const u32x4r w = __builtin_nontemporal_load(reinterpret_cast<const u32x4r*>(p));
    const bf16x8 v = __builtin_bit_cast(bf16x8, w);
    f32x8r r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
    return r;
}
DEVI void store8_nt(bf16* p, f32x8r v) {
    typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4r, t), reinterpret_cast<u32x4r*>(p));
}
DEVI f32x8r load8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    return f32x8r{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}
DEVI void store8(bf16* p, f32x8r v) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}
DEVI float half_sum(float v) {      // sum over the 32 lanes of this half-wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define LN_BWD_RED_FLOATS(NC) (8 * ((NC) * 256 / 8 + 1) * 8)

// Backward: a workgroup of 4 waves = 8 half-waves; half-wave h owns rows blockIdx * 8 + h + 8 * gridDim * i (two rows each at the
// VQA2 shape: 456 workgroups, two co-resident per CU so that one streams while the other reduces); all loads of a row are issued before anything is reduced.  Column-sum partials (dgamma, dbeta, optionally dbias) are combined across the
// workgroup's 32 half-waves in LDS and written once per workgroup: partials[blk][q][H], q < NQ.
template <int NC, bool DBIAS, int NR, bool DIN = false, int STREAM = 0>      // STREAM: bit 0 non-temporal loads of dy / x (the rider, gemm.hip), bit 1 non-temporal stores of dx / dlin (measured: no gain).  DIN: dropout backward applied to dy as it is loaded (its own instantiation: the register budget of the hot form decides its occupancy)
DEVI void ln_bwd_h_block(const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                         const float* __restrict__ gamma, bf16* __restrict__ dx, bf16* __restrict__ dlin, DropoutCfg drop, float* __restrict__ partials,
                         int rows, DropoutCfg din, const int tid, const int blk, const int nblk, const bool active, float* __restrict__ red_) {
    // `tid` 0..255 within the 256 threads that form block `blk` of `nblk`; `red_`: LN_BWD_RED_FLOATS(NC) floats of LDS of those threads.  `active == false`
    // (a rider half without a block left, gemm.hip): nothing is read or written, only the barriers are kept (they are workgroup-wide).
    constexpr int H = NC * 256, NQ = DBIAS ? 3 : 2;
    float (*red)[H / 8 + 1][8] = reinterpret_cast<float (*)[H / 8 + 1][8]>(red_);        // one quantity at a time: [half-wave][lane chunk][8]   (+1: bank spread)
    const int hl = tid & 31, half = tid >> 5;
    f32x8r ag[NC], ab[NC], al[DBIAS ? NC : 1];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        ag[c] = ab[c] = f32x8r{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (DBIAS) al[c] = ag[c];
    }
    // (gamma is re-read per chunk from L1 / L2 in both passes instead of being held: the register budget is what decides how
    // many workgroups - rows in flight - a CU holds)
    // NR rows of a half-wave are in flight together: every load of both rows is issued before anything is reduced (the VQA2 shape gives each
    // half-wave exactly two rows; with one row at a time the second row's HBM latency was exposed: 13.4 us per launch for 45 MB).
    const int rstride = 8 * nblk;
    for (int row0 = blk * 8 + half; active && row0 < rows; row0 += NR * rstride) {
        f32x8r xv[NR][NC], dv[NR][NC];
        float mu[NR], rs[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int row = row0 + r * rstride;
            const int rowc = row < rows ? row : row0;          // (a missing second row re-reads the first; its results are dropped)
            const size_t off = (size_t)rowc * H + hl * 8;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if constexpr (STREAM & 1) { xv[r][c] = load8_nt(x + off + 256 * c); dv[r][c] = load8_nt(dy + off + 256 * c); }
                else { xv[r][c] = load8(x + off + 256 * c); dv[r][c] = load8(dy + off + 256 * c); }
            }
            mu[r] = mean[rowc]; rs[r] = rstd[rowc];
        }
        if (DIN && din.thr16) {      // the LayerNorm's OUTPUT went through nn.Dropout in the forward (embeddings.py:345): dy = dropout_backward(incoming), element
                              // index row * H + col, rounded to bf16 like the separate mmf_dropout_bf16 launch stored it
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = row0 + r * rstride;
                const int rowc = row < rows ? row : row0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const uint32_t idx = (uint32_t)rowc * (uint32_t)H + (uint32_t)(hl * 8 + 256 * c);
                    const f32x4 s0 = drop_scale4(drop_key(din), idx, din.thr16, din.scale), s1 = drop_scale4(drop_key(din), idx + 4, din.thr16, din.scale);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { dv[r][c][i] = (float)(bf16)(dv[r][c][i] * s0[i]); dv[r][c][i + 4] = (float)(bf16)(dv[r][c][i + 4] * s1[i]); }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int row = row0 + r * rstride;
            if (row >= rows) continue;
            const size_t off = (size_t)row * H + hl * 8;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x8r gm = load8(gamma + hl * 8 + 256 * c);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xv[r][c][i] = (xv[r][c][i] - mu[r]) * rs[r];                  // xhat
                    const float g = dv[r][c][i] * gm[i];
                    s1 += g;
                    s2 += g * xv[r][c][i];
                }
            }
            const float c1 = half_sum(s1) * (1.f / (float)H), c2 = half_sum(s2) * (1.f / (float)H);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x8r gm = load8(gamma + hl * 8 + 256 * c);
                f32x8r d;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    d[i] = rs[r] * (dv[r][c][i] * gm[i] - c1 - xv[r][c][i] * c2);
                    ag[c][i] += dv[r][c][i] * xv[r][c][i];
                    ab[c][i] += dv[r][c][i];
                }
                if constexpr (STREAM & 2) store8_nt(dx + off + 256 * c, d); else store8(dx + off + 256 * c, d);
                if (dlin) {
                    const uint32_t idx = (uint32_t)row * (uint32_t)H + (uint32_t)(hl * 8 + 256 * c);
                    const f32x4 s0 = drop_scale4(drop_key(drop), idx, drop.thr16, drop.scale);
                    const f32x4 s1_ = drop_scale4(drop_key(drop), idx + 4, drop.thr16, drop.scale);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { d[i] *= s0[i]; d[i + 4] *= s1_[i]; }
                    if constexpr (STREAM & 2) store8_nt(dlin + off + 256 * c, d); else store8(dlin + off + 256 * c, d);
                }
                if (DBIAS) {     // the bias gradient uses the same rounding the weight-gradient GEMM will see
#pragma unroll
                    for (int i = 0; i < 8; ++i) al[c][i] += (float)(bf16)d[i];
                }
            }
        }
    }
#pragma unroll 1
    for (int qn = 0; qn < NQ; ++qn) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x8r v = (qn == 0) ? ag[c] : (qn == 1) ? ab[c] : al[DBIAS ? c : 0];
            float* dst = &red[half][hl + 32 * c][0];
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        __syncthreads();
        for (int col = tid; active && col < H; col += 256) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += red[w][col >> 3][col & 7];
            partials[((size_t)blk * 3 + qn) * H + col] = t;
        }
    }
}


}  // namespace lnk
