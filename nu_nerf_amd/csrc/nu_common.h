// nu_common.h -- shared device/host helpers for the NU-NeRF gfx950 hot-path library.
//
// Everything in csrc/ is written for gfx950 (MI355X, CDNA4) only: 64-wide wavefronts,
// fp32-input MFMA (v_mfma_f32_32x32x2_f32) for the MLP contractions, LDS-staged tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nu_nerf.h"  // public C ABI: error codes, NuGemmNT / NuGemmTN / NuPackDesc

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NU_WAVE 64

// round-to-nearest-even fp32 -> bf16 (v_cvt_pk_bf16_f32; NaN stays NaN)
static __device__ inline bf16x4 nu_to_bf16x4(f32x4 v) {
    bf16x4 o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    return o;
}
// exact three-way split of fp32 into bf16 pieces: v = p1 + p2 + p3 (each subtraction below is exact in fp32)
static __device__ inline void nu_split3(f32x4 v, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
#pragma clang fp contract(off)
    p1 = nu_to_bf16x4(v);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = v[e] - (float)p1[e];
    p2 = nu_to_bf16x4(r);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = r[e] - (float)p2[e];
    p3 = nu_to_bf16x4(r);
}

// ceil-div / round-up helpers (host + device)
static __host__ __device__ inline int nu_cdiv(int a, int b) { return (a + b - 1) / b; }
static __host__ __device__ inline int nu_rup(int a, int b) { return nu_cdiv(a, b) * b; }
static __host__ __device__ inline long long nu_cdivl(long long a, long long b) { return (a + b - 1) / b; }

// Softplus(beta=100) with torch's threshold semantics (reference: network/field.py:126-127,
// nn.Softplus(beta=100) -> x*beta > 20 ? x : log1p(exp(x*beta))/beta).
static __device__ inline float nu_softplus100(float x) {
    float bx = 100.0f * x;
    return bx > 20.0f ? x : log1pf(expf(bx)) * 0.01f;
}
// d softplus / d preact expressed through the POST-activation value h = softplus(a):
// sigma(beta*a) = 1 - exp(-beta*h).  Lets the training path store only h.
static __device__ inline float nu_softplus100_grad_from_h(float h) { return -expm1f(-100.0f * h); }

static __device__ inline float nu_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Hardware-transcendental variants for the GEMM epilogues (raw v_exp_f32 / v_log_f32, ~1 ulp each, no denormal range
// fix-ups: every argument here is provably in range).  BRANCH-FREE on purpose: the libm-style __expf / __logf expand to a
// compare + select + ldexp ladder and hipcc turned the two ternaries of the first version into exec-mask branches -- ~50
// instructions per element, and the epilogue touches every hidden activation (measured: 7.4 us of a 28 us tile).
static __device__ inline float nu_softplus100_fast(float x) {
    const float u = fminf(x * 144.26950408889634f, 28.853900817779268f);            // 100 x log2(e), capped at 20 log2(e)
    const float t = __builtin_amdgcn_exp2f(u);                                       // exp(min(100 x, 20))
    const float big = __builtin_amdgcn_logf(1.0f + t) * 0.0069314718055994531f;      // log2(1 + t) ln2 / 100
    const float small = t * (0.01f - t * (0.005f - t * (0.01f / 3.0f)));             // log1p series below 1e-3 (|err| < t^4/4)
    const float l = t < 1e-3f ? small : big;
    return 100.0f * x > 20.0f ? x : l;                                                // torch's threshold: beta x > 20 -> identity
}
static __device__ inline float nu_exp_m100(float h) { return __builtin_amdgcn_exp2f(-144.26950408889634f * h); }   // exp(-100 h) = 1 - softplus'(a)

// sRGB transfer (reference: utils/raw_utils.py:5-17). eps = FLT_EPSILON.
static __device__ inline float nu_linear_to_srgb(float x) {
    const float eps = 1.1920928955078125e-07f;
    float s0 = (323.0f / 25.0f) * x;
    float s1 = (211.0f * powf(fmaxf(x, eps), 5.0f / 12.0f) - 11.0f) / 200.0f;
    return x <= 0.0031308f ? s0 : s1;
}
// derivative of the above w.r.t. x
static __device__ inline float nu_linear_to_srgb_grad(float x) {
    const float eps = 1.1920928955078125e-07f;
    if (x <= 0.0031308f) return 323.0f / 25.0f;
    if (x < eps) return 0.0f;  // clamp region (unreachable: eps < 0.0031308)
    return (211.0f / 200.0f) * (5.0f / 12.0f) * powf(x, 5.0f / 12.0f - 1.0f);
}

static inline int nu_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NU_OK : NU_ERR_LAUNCH;
}

// wave-level inclusive scans / reductions over 64 lanes
static __device__ inline float nu_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ inline float nu_wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
static __device__ inline float nu_wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}
