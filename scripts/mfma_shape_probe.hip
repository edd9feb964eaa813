// Development aid (not part of the library): does the fp32 MFMA SHAPE change what the chip delivers on random data?
// MI355X_MICROARCH.md "DVFS give-back" (7): for bf16 the 16x16x32 loop delivered ~1.15x the FLOP/s of the 32x32x16 loop at
// equal cycles per FLOP, because the chip holds a higher clock.  Same question for v_mfma_f32_16x16x4_f32 vs
// v_mfma_f32_32x32x2_f32 on a 64 x 64 wave tile with every operand re-read from LDS by ds_read_b128 (the NT kernel's loop).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_shape_probe.hip -o scripts/mfma_shape_probe && ./scripts/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSW 36

__device__ unsigned long long g_clk[2];

// SHAPE 0: 32x32x2 (2 x 2 tiles of 32 x 32 per wave); SHAPE 1: 16x16x4 (4 x 4 tiles of 16 x 16 per wave).
// LDS: 1 = fragments re-read from LDS every k-group, 0 = operands stay in registers.
template <int SHAPE, int LDS, int WPC>
__global__ __launch_bounds__(256, WPC) void probe(const float* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[2][128 * LDSW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    for (int i = tid; i < 2 * 128 * LDSW; i += 256) (&smem[0][0])[i] = src[(blockIdx.x * 977 + i) & 0xfffff];
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float s = 0.f;
    if (SHAPE == 0) {
        const int li = lane & 31, lh = lane >> 5;
        const int a_off = (wr * 64 + li) * LDSW + 4 * lh, b_off = (wc * 64 + li) * LDSW + 4 * lh;
        f32x16 acc[2][2] = {};
        f32x4 a0 = *reinterpret_cast<const f32x4*>(&smem[0][a_off]), a1 = *reinterpret_cast<const f32x4*>(&smem[0][a_off + 32 * LDSW]);
        f32x4 b0 = *reinterpret_cast<const f32x4*>(&smem[1][b_off]), b1 = *reinterpret_cast<const f32x4*>(&smem[1][b_off + 32 * LDSW]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (LDS) {
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
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        // lane (r = l & 15, q = l >> 4) reads 4 consecutive k of row r: element e of the read is k = 4 q + e, the k-step of MFMA e
        const int lr = lane & 15, lq = lane >> 4;
        const int a_off = (wr * 64 + lr) * LDSW + 4 * lq, b_off = (wc * 64 + lr) * LDSW + 4 * lq;
        f32x4 acc[4][4] = {};
        f32x4 a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = *reinterpret_cast<const f32x4*>(&smem[0][a_off + 16 * t * LDSW]);
            b[t] = *reinterpret_cast<const f32x4*>(&smem[1][b_off + 16 * t * LDSW]);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {        // one read set = 16 k: two per 32-deep chunk
                if (LDS) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        a[t] = *reinterpret_cast<const f32x4*>(&smem[0][a_off + 16 * t * LDSW + kk * 16]);
                        b[t] = *reinterpret_cast<const f32x4*>(&smem[1][b_off + 16 * t * LDSW + kk * 16]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    if (blockIdx.x == 0 && tid == 0) { g_clk[0] = clock64() - c0; g_clk[1] = wall_clock64() - w0; }
    if (s == 123.456f) out[0] = s;
}

template <int SHAPE, int LDS, int WPC>
static void run(const char* what, const float* src, float* out) {
    const int iters = 4000, blocks = 256 * WPC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 30; ++w) hipLaunchKernelGGL((probe<SHAPE, LDS, WPC>), dim3(blocks), dim3(256), 0, 0, src, out, iters);   // ~1 s warm
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<SHAPE, LDS, WPC>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    unsigned long long clk[2];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    const double fl = (double)blocks * 4 * iters * 64 * 4096.0;     // per wave and chunk: 64 x 64 x 32 x 2
    printf("%-10s lds=%d WPC=%d %-28s %8.3f ms  %6.1f TFLOP/s  clock %.3f GHz\n", SHAPE ? "16x16x4" : "32x32x2", LDS, WPC, what, ms,
           fl / ms / 1e9, (double)clk[0] / (double)clk[1] * 0.1);
}

int main(int argc, char** argv) {
    const size_t n = 1 << 20;
    float* h = (float*)malloc(n * sizeof(float));
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    srand(7);
    for (size_t i = 0; i < n; ++i) h[i] = zeros ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *out;
    hipMalloc(&src, n * sizeof(float));
    hipMemcpy(src, h, n * sizeof(float), hipMemcpyHostToDevice);
    hipMalloc(&out, 64);
    printf("operands: %s\n", zeros ? "zeros" : "uniform random [-1, 1)");
    run<0, 0, 1>("registers", src, out);
    run<1, 0, 1>("registers", src, out);
    run<0, 1, 1>("lds fragments", src, out);
    run<1, 1, 1>("lds fragments", src, out);
    run<0, 1, 2>("lds fragments", src, out);
    run<1, 1, 2>("lds fragments", src, out);
    run<0, 0, 2>("registers", src, out);
    run<1, 0, 2>("registers", src, out);
    return 0;
}
