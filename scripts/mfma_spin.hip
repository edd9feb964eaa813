// Development aid for scripts/determinism_valu_victim.py: kernels that do nothing but issue one kind of MFMA from registers (no LDS, no
// memory traffic in the loop), to tell whether the instruction alone is what a packed-fp32 VALU kernel on another stream reacts to.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/libmfma_spin.so scripts/mfma_spin.hip
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float seed) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    const float s = seed + threadIdx.x * 1e-3f;
    bf16x8 a, b;
    f16x8 ah, bh;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(s + e); b[e] = (__bf16)(s - e); ah[e] = (_Float16)(s + e); bh[e] = (_Float16)(s - e); }
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // v_mfma_f32_32x32x16_bf16
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        } else if (KIND == 1) {   // v_mfma_f32_16x16x32_bf16
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d3, 0, 0, 0);
        } else if (KIND == 2) {   // v_mfma_f32_32x32x2_f32
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c3, 0, 0, 0);
        } else if (KIND == 4) {   // v_cvt_pk_bf16_f32 only (fp32 -> bf16 conversion, the instruction the split / rounding GEMM kernels add to their loops)
            f32x4 t = {s + i, s - i, s * i, s + 2.f * i};
#pragma unroll
            for (int e = 0; e < 4; ++e) d0[e] += (float)(__bf16)t[e] + (float)(__bf16)(t[e] * 1.0001f);
        } else if (KIND == 6) {   // bf16 MFMA fed by fp32 -> bf16 conversions of changing values, through LDS (the shape of a rounding GEMM loop)
            __shared__ bf16x8 st[256];
            f32x4 t0 = {s + i, s - i, s * i, s + 2.f * i}, t1 = {s - 3.f * i, s + 5.f * i, s * 0.5f * i, s - 7.f * i};
            bf16x8 na;
#pragma unroll
            for (int e = 0; e < 4; ++e) { na[e] = (__bf16)t0[e]; na[4 + e] = (__bf16)t1[e]; }
            st[threadIdx.x] = na;
            __syncthreads();
            const bf16x8 fa = st[(threadIdx.x + 64) & 255], fb = st[(threadIdx.x + 128) & 255];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fa, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fb, c3, 0, 0, 0);
            __syncthreads();
        } else if (KIND == 5) {   // LDS traffic only: ds_write_b128 / ds_read_b128
            __shared__ f32x4 buf[256 * 4];
            buf[threadIdx.x + 256 * (i & 3)] = d0;
            __syncthreads();
            d1 += buf[(threadIdx.x * 7 + i) & 1023];
            __syncthreads();
        } else {                  // v_mfma_f32_32x32x16_f16
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
        }
    }
    float r = 0.f;
    for (int e = 0; e < 16; ++e) r += c0[e] + c1[e] + c2[e] + c3[e];
    for (int e = 0; e < 4; ++e) r += d0[e] + d1[e] + d2[e] + d3[e];
    if (r == 12345.678f) out[0] = r;          // keeps the loop alive
}

extern "C" int mfma_spin(int kind, int blocks, int iters, float* out, hipStream_t stream) {
    switch (kind) {
        case 0: hipLaunchKernelGGL((spin<0>), dim3(blocks), dim3(256), 0, stream, out, iters, 0.5f); break;
        case 1: hipLaunchKernelGGL((spin<1>), dim3(blocks), dim3(256), 0, stream, out, iters, 0.5f); break;
        case 2: hipLaunchKernelGGL((spin<2>), dim3(blocks), dim3(256), 0, stream, out, iters, 0.5f); break;
        case 4: hipLaunchKernelGGL((spin<4>), dim3(blocks), dim3(256), 0, stream, out, iters, 0.5f); break;
        case 5: hipLaunchKernelGGL((spin<5>), dim3(blocks), dim3(256), 0, stream, out, iters, 0.5f); break;
        case 6: hipLaunchKernelGGL((spin<6>), dim3(blocks), dim3(256), 0, stream, out, iters, 0.5f); break;
        default: hipLaunchKernelGGL((spin<3>), dim3(blocks), dim3(256), 0, stream, out, iters, 0.5f); break;
    }
    return (int)hipGetLastError();
}
