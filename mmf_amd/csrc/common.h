// mmf_amd :: device-side helpers shared by every gfx950 kernel in this library.
// MI355X / CDNA4 only: wave = 64 lanes, MFMA bf16, 160 KiB LDS. No CUDA/portability paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define MMF_WAVE 64
#define DEVI __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// error plumbing for the C ABI (host side)
// ---------------------------------------------------------------------------------------------
#ifdef __cplusplus
extern "C" {
#endif
void mmf_amd_set_error(const char* msg);
#ifdef __cplusplus
}
#endif

#define MMF_CHECK_ARG(cond, msg)                                                        \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            mmf_amd_set_error(msg " [" #cond "]");                                      \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define MMF_CHECK_LAUNCH()                                                              \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            mmf_amd_set_error(hipGetErrorString(e__));                                  \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// numeric helpers
// ---------------------------------------------------------------------------------------------
DEVI float bf2f(bf16 x) { return (float)x; }
DEVI bf16 f2bf(float x) { return (bf16)x; }  // lowers to v_cvt_pk_bf16_f32 (RNE) on gfx950

DEVI bf16x4 pack4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (bf16)a; r[1] = (bf16)b; r[2] = (bf16)c; r[3] = (bf16)d;
    return r;
}

// exact-erf GELU, as HF `gelu` (transformers ACT2FN["gelu"]): x * 0.5 * (1 + erf(x / sqrt(2))), and its derivative
// Phi(x) + x * phi(x).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 round-off of the product
// and far below the bf16 output rounding); exp(-x^2/2) is shared by erf's tail and by phi.  ~20 VALU ops per
// element instead of the ~70 of libm erff + expf, which made the FFN epilogues VALU-bound.
DEVI void gelu_erf_both(float x, float& h, float& g) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float e = __expf(-ax * ax);  // exp(-x^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    const float cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    h = x * cdf;
    g = cdf + x * 0.39894228040143267794f * e;
}
DEVI float gelu_erf(float x) { float h, g; gelu_erf_both(x, h, g); return h; }
DEVI float gelu_erf_grad(float x) { float h, g; gelu_erf_both(x, h, g); return g; }

// wave-wide (64 lane) reductions
DEVI float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVI float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// counter-based dropout RNG.  One 32-bit hash per PAIR of elements; element e of a dropout site
// keeps iff its 16-bit half >= thr16 (thr16 = round(p * 65536)).  The same (key, linear index)
// reproduces the same decision in backward, whatever the thread mapping.
// ---------------------------------------------------------------------------------------------
// Mixer built from FULL-RATE integer ops only (v_alignbit_b32, v_mad_u32_u24, shifts, xors): 12 instructions, ~24 cycles per wave64
// hash, against ~38 for a "lowbias32"-style finalizer whose three 32-bit multiplies (v_mul_lo_u32) issue at quarter rate — the dropout
// hash is the largest single VALU item of the attention kernels (profiles/r02_attention_timeline.txt).  Avalanche on random inputs:
// every input bit flips every output bit with probability 0.5 +- 0.013 (20000 samples; the 32-bit-multiply mixer measures +- 0.011);
// keep rates and neighbour / row / key correlations of the 16-bit halves on sequential indices are at sampling noise (tools/hash_quality.py).
DEVI uint32_t mix24(uint32_t x) {
    x ^= x >> 16;
    x = __umul24(x, 0xB5297Bu) + __builtin_rotateleft32(x, 9);
    x ^= x >> 13;
    x = __umul24(x, 0x68E31Fu) + __builtin_rotateleft32(x, 11);
    x ^= x >> 15;
    return x;
}
DEVI uint32_t drop_hash(uint32_t key, uint32_t pair_idx) { return mix24(pair_idx + key); }
// keep-scale factors (0 or scale) for the 4 consecutive elements starting at linear index idx4
// (idx4 % 4 == 0).
DEVI f32x4 drop_scale4(uint32_t key, uint32_t idx4, uint32_t thr16, float scale) {
    const uint32_t h0 = drop_hash(key, idx4 >> 1);
    const uint32_t h1 = drop_hash(key, (idx4 >> 1) + 1);
    f32x4 r;
    r[0] = ((h0 & 0xffffu) >= thr16) ? scale : 0.f;
    r[1] = ((h0 >> 16) >= thr16) ? scale : 0.f;
    r[2] = ((h1 & 0xffffu) >= thr16) ? scale : 0.f;
    r[3] = ((h1 >> 16) >= thr16) ? scale : 0.f;
    return r;
}
DEVI float drop_scale1(uint32_t key, uint32_t idx, uint32_t thr16, float scale) {
    const uint32_t h = drop_hash(key, idx >> 1);
    const uint32_t half = (idx & 1u) ? (h >> 16) : (h & 0xffffu);
    return (half >= thr16) ? scale : 0.f;
}

struct DropoutCfg {
    uint32_t key;          // per-site key (host mixes the step seed with the site id)
    uint32_t thr16;        // 0 => dropout disabled
    float scale;           // 1 / (1 - thr16/65536)
    const uint32_t* seed;  // optional device word mixed into the key at run time: lets a captured hipGraph draw
                           // fresh masks on every replay (mmf_seed_advance bumps it inside the graph)
};
// effective key of a site for this launch (wave-uniform)
// (the step seed enters through its own odd multiplier: consecutive replays land far apart in the index stream, not one pair apart)
DEVI uint32_t drop_key(const DropoutCfg& d) { return d.seed ? d.key + d.seed[0] * 0x9E3779B1u : d.key; }
