// Development aid (not part of the library): what costs the fp32-MFMA main loop its last 20 %?
// Builds the NT kernel's k-step out of its pieces and times each combination.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_lds_probe.hip -o scripts/mfma_lds_probe && ./scripts/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSW 36

// MODE bits: 1 = LDS fragment reads, 2 = barrier per k-step, 4 = LDS writes (+ second barrier), 8 = global loads
template <int MODE, int WPC>
__global__ __launch_bounds__(256, WPC) void probe(const float* __restrict__ A, float* out, int iters, int lda) {
    __shared__ __attribute__((aligned(16))) float smem[2][128 * LDSW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, lh = lane >> 5;
    const int c4 = tid & 7, r0 = tid >> 3;
    for (int i = tid; i < 2 * 128 * LDSW; i += 256) (&smem[0][0])[i] = (float)(i & 15) * 1e-3f;
    __syncthreads();
    const int a_off = (wr * 64 + li) * LDSW + 4 * lh;
    const int b_off = (wc * 64 + li) * LDSW + 4 * lh;
    f32x16 acc[2][2] = {};
    f32x4 ra4[4], rb4[4];
    for (int i = 0; i < 4; ++i) { ra4[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f}; rb4[i] = ra4[i]; }
    const float* ap = A + (long long)(blockIdx.x * 128 + r0) * lda + 4 * c4;
    f32x4 a0 = {1e-3f, 2e-3f, 3e-3f, 4e-3f}, a1 = a0, b0 = a0, b1 = a0;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra4[i] = *reinterpret_cast<const f32x4*>(ap + (long long)(32 * i) * lda + (it & 7) * 32);
                rb4[i] = *reinterpret_cast<const f32x4*>(ap + (long long)(32 * i) * lda + ((it + 3) & 7) * 32);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (MODE & 1) {
                a0 = *reinterpret_cast<const f32x4*>(&smem[0][a_off + kk * 8]);
                a1 = *reinterpret_cast<const f32x4*>(&smem[0][a_off + 32 * LDSW + kk * 8]);
                b0 = *reinterpret_cast<const f32x4*>(&smem[1][b_off + kk * 8]);
                b1 = *reinterpret_cast<const f32x4*>(&smem[1][b_off + 32 * LDSW + kk * 8]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
            }
        }
        if (MODE & 2) __syncthreads();
        if (MODE & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(&smem[0][(r0 + 32 * i) * LDSW + 4 * c4]) = ra4[i];
                *reinterpret_cast<f32x4*>(&smem[1][(r0 + 32 * i) * LDSW + 4 * c4]) = rb4[i];
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int WPC>
static void run(const char* what, const float* A, float* out, int lda) {
    const int iters = 2000, blocks = 256 * WPC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE, WPC>), dim3(blocks), dim3(256), 0, 0, A, out, iters, lda);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<MODE, WPC>), dim3(blocks), dim3(256), 0, 0, A, out, iters, lda);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double fl = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("WPC=%d mode=%2d %-46s %8.3f ms  %6.1f TFLOP/s\n", WPC, MODE, what, ms, fl / ms / 1e9);
}

// 8 waves per workgroup (4 x 2), wave tile 32 x 64: the same chunk and LDS image shared by twice the waves (2 workgroups per CU
// = 4 waves per SIMD at <= 128 VGPRs).  MODE bits as above.
template <int MODE, int WPC>
__global__ __launch_bounds__(512, WPC * 2) void probe8(const float* __restrict__ A, float* out, int iters, int lda) {
    __shared__ __attribute__((aligned(16))) float smem[2][128 * LDSW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, lh = lane >> 5;
    const int c4 = tid & 7, r0 = tid >> 3;          // 64 rows per pass, 2 passes
    for (int i = tid; i < 2 * 128 * LDSW; i += 512) (&smem[0][0])[i] = (float)(i & 15) * 1e-3f;
    __syncthreads();
    const int a_off = (wr * 32 + li) * LDSW + 4 * lh;
    const int b_off = (wc * 64 + li) * LDSW + 4 * lh;
    f32x16 acc[2] = {};
    f32x4 ra4[2], rb4[2];
    for (int i = 0; i < 2; ++i) { ra4[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f}; rb4[i] = ra4[i]; }
    const float* ap = A + (long long)(blockIdx.x * 128 + r0) * lda + 4 * c4;
    f32x4 a0 = {1e-3f, 2e-3f, 3e-3f, 4e-3f}, b0 = a0, b1 = a0;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra4[i] = *reinterpret_cast<const f32x4*>(ap + (long long)(64 * i) * lda + (it & 7) * 32);
                rb4[i] = *reinterpret_cast<const f32x4*>(ap + (long long)(64 * i) * lda + ((it + 3) & 7) * 32);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (MODE & 1) {
                a0 = *reinterpret_cast<const f32x4*>(&smem[0][a_off + kk * 8]);
                b0 = *reinterpret_cast<const f32x4*>(&smem[1][b_off + kk * 8]);
                b1 = *reinterpret_cast<const f32x4*>(&smem[1][b_off + 32 * LDSW + kk * 8]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[1], 0, 0, 0);
            }
        }
        if (MODE & 2) __syncthreads();
        if (MODE & 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<f32x4*>(&smem[0][(r0 + 64 * i) * LDSW + 4 * c4]) = ra4[i];
                *reinterpret_cast<f32x4*>(&smem[1][(r0 + 64 * i) * LDSW + 4 * c4]) = rb4[i];
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int WPC>
static void run8(const char* what, const float* A, float* out, int lda) {
    const int iters = 2000, blocks = 256 * WPC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe8<MODE, WPC>), dim3(blocks), dim3(512), 0, 0, A, out, iters, lda);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe8<MODE, WPC>), dim3(blocks), dim3(512), 0, 0, A, out, iters, lda);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double fl = (double)blocks * 8 * iters * 32 * 4096.0;
    printf("8-wave WG WPC=%d mode=%2d %-40s %8.3f ms  %6.1f TFLOP/s\n", WPC, MODE, what, ms, fl / ms / 1e9);
}

int main() {
    const int lda = 256;
    const size_t n = (size_t)1024 * 128 * lda;
    float *A, *out;
    hipMalloc(&A, n * sizeof(float));
    {   // random operands (zero-filled ones read high: the chip holds a higher clock on them)
        float* h = (float*)malloc(n * sizeof(float));
        srand(3);
        for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
        hipMemcpy(A, h, n * sizeof(float), hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&out, 64);
    run<0, 2>("mfma only", A, out, lda);
    run<1, 2>("+ lds frag reads", A, out, lda);
    run<3, 2>("+ lds reads + barrier", A, out, lda);
    run<7, 2>("+ lds reads + writes + 2 barriers", A, out, lda);
    run<15, 2>("+ global loads (full k-step)", A, out, lda);
    run<9, 2>("lds reads + global loads, no barrier", A, out, lda);
    run<0, 3>("mfma only", A, out, lda);
    run<1, 3>("+ lds frag reads", A, out, lda);
    run<3, 3>("+ lds reads + barrier", A, out, lda);
    run<7, 3>("+ lds reads + writes + 2 barriers", A, out, lda);
    run<15, 3>("+ global loads (full k-step)", A, out, lda);
    run8<0, 2>("mfma only", A, out, lda);
    run8<1, 2>("+ lds frag reads", A, out, lda);
    run8<3, 2>("+ lds reads + barrier", A, out, lda);
    run8<7, 2>("+ lds reads + writes + 2 barriers", A, out, lda);
    run8<15, 2>("+ global loads (full k-step)", A, out, lda);
    run8<15, 1>("+ global loads (full k-step)", A, out, lda);
    run<0, 1>("mfma only", A, out, lda);
    run<1, 1>("+ lds frag reads", A, out, lda);
    run<15, 1>("+ global loads (full k-step)", A, out, lda);
    return 0;
}
