// Development aid (not part of the library): stand-alone timing lab for the NT GEMM kernel -- no Python, starts in a second.
// Compiles csrc/gemm.hip into this binary with -DNU_LAB, which adds per-workgroup phase stamps and a start-up stagger.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNU_LAB -Inu_nerf_amd/csrc -Iinclude scripts/gemm_lab.hip -o scripts/gemm_lab
//   ./scripts/gemm_lab [sweep|trace]
#include "../nu_nerf_amd/csrc/gemm.hip"
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale, float shift) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((float)(x & 0xffffff) / 16777216.0f - 0.5f) * scale + shift;
    }
}
static float* dalloc(long long n, unsigned seed, float scale, float shift = 0.f) {
    float* p;
    CK(hipMalloc(&p, n * sizeof(float)));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p, n, seed, scale, shift);
    return p;
}

// bare matrix-pipe loops on random operands held in registers: what each f32 MFMA shape delivers, and at which clock
__device__ unsigned long long lab_clk[4];
template <int SHAPE>
__global__ __launch_bounds__(256) void bare_mfma_kernel(const float* __restrict__ src, float* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = src[(t * 8 + i) & 0xfffff]; y[i] = src[(t * 8 + i + 77777) & 0xfffff]; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], y[j], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], y[j + 1], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j + 1], y[j], a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j + 1], y[j + 1], a3, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    } else {
        f32x4 a[16] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j + (q >> 2)], y[j + (q & 3)], a[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) s += a[q][0] + a[q][1] + a[q][2] + a[q][3];
    }
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { lab_clk[0] = clock64() - c0; lab_clk[1] = wall_clock64() - w0; }
}

// FETCH_SIZE calibration (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern"): streaming reads
// of a known number of bytes with 4 B per lane (the TN kernel's transposing loads) and with 16 B per lane
__global__ __launch_bounds__(256) void calib_read4_kernel(const float* __restrict__ p, long long n, float* out) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += p[i];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void calib_read16_kernel(const f32x4* __restrict__ p, long long n4, float* out) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) { const f32x4 v = p[i]; s += v[0] + v[3]; }
    if (s == 123.456f) out[0] = s;
}

struct Bufs { float *A, *B, *C, *C2, *H, *D, *bias; int M, N, K; };

static double time_nt(const Bufs& b, int epi, int iters) {
    NuGemmNT g = {};
    g.A = b.A; g.lda = b.K; g.B = b.B; g.ldb = b.K; g.M = b.M; g.N = b.N; g.K = b.K; g.C = b.C; g.ldc = b.N; g.C2 = b.C2; g.ldc2 = b.N;
    g.bias = b.bias; g.H = b.H; g.ldh = b.N; g.D = b.D; g.ldd = b.N; g.Cadd = b.D; g.ldadd = b.N; g.alpha = 1.f; g.groups = 1; g.epi = epi;
    for (int i = 0; i < 2; ++i) if (nu_gemm_nt_launch(g, 0)) { printf("launch failed\n"); exit(1); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) nu_gemm_nt_launch(g, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

#define set_int(sym, val) do { int v_ = (val); CK(hipMemcpyToSymbol(HIP_SYMBOL(sym), &v_, sizeof(int))); } while (0)

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "sweep";
    const int M = 540672;   // the step's outer-point count (4224 row tiles)
    Bufs b256 = {dalloc((long long)M * 256, 1, 1.f), dalloc(256 * 256, 2, 0.12f), dalloc((long long)M * 256, 3, 0.f), dalloc((long long)M * 256, 4, 0.f),
                 dalloc((long long)M * 256, 5, 0.02f, 0.01f), dalloc((long long)M * 256, 6, 1.f), dalloc(256, 7, 0.1f), M, 256, 256};
    Bufs b1024 = b256;
    b1024.M = M / 4; b1024.K = 1024; b1024.B = dalloc(256 * 1024, 8, 0.06f);
    CK(hipDeviceSynchronize());
    set_int(nu_lab_skip_epi, 0);
    auto tf = [](const Bufs& b, double ms) { return 2.0 * b.M * b.N * b.K / ms / 1e9; };
    if (!strcmp(mode, "sweep")) {
        // warm the clocks
        for (int i = 0; i < 3; ++i) time_nt(b256, NU_EPI_PLAIN, 10);
        Bufs bin = b256;
        bin.M = 114048;                       // the step's inner-point count: 891 row tiles
        Bufs b64 = b256;
        b64.K = 64;                           // input layers (A is read with lda = 64 from the same buffer)
        for (int pass = 0; pass < 2; ++pass)
        for (int gen = 1; gen <= 2; ++gen) {
            nu_lab_v1 = gen == 1;
            for (int grid : {0, 256}) {
                nu_lab_grid = grid;
                for (int skip = 0; skip < 2; ++skip) {
                    set_int(nu_lab_skip_epi, skip);
                    double r[10];
                    int i = 0;
                    for (int epi : {NU_EPI_PLAIN, NU_EPI_BIAS_RELU, NU_EPI_BIAS_SOFTPLUS, NU_EPI_MUL_DSP, NU_EPI_Q_SP}) r[i++] = tf(b256, time_nt(b256, epi, 20));
                    r[i++] = tf(b1024, time_nt(b1024, NU_EPI_PLAIN, 20));
                    for (int epi : {NU_EPI_BIAS_RELU, NU_EPI_BIAS_SOFTPLUS}) r[i++] = tf(bin, time_nt(bin, epi, 40));
                    r[i++] = tf(b64, time_nt(b64, NU_EPI_BIAS_SOFTPLUS, 40));
                    printf("gen %d grid %3d skip_epi %d : M=540k K=256 plain %6.1f relu %6.1f softplus %6.1f dsp %6.1f q_sp %6.1f | K=1024 %6.1f | M=114k relu %6.1f softplus %6.1f | K=64 softplus %6.1f\n",
                           gen, grid, skip, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]);
                }
            }
        }
        nu_lab_grid = 0; nu_lab_v1 = 0;
        set_int(nu_lab_skip_epi, 0);
    } else if (!strcmp(mode, "levels")) {
        // which part of the epilogue costs the launch its 11-14 %?  0: all of it; 2: scratch round trip + math, no global stores;
        // 1: scratch writes only; 3: nothing; 4: all of it, but every workgroup stores to (and reads its auxiliary operands from) one
        // small region that stays in L2
        for (int i = 0; i < 3; ++i) time_nt(b256, NU_EPI_PLAIN, 10);
        for (int pass = 0; pass < 2; ++pass)
        for (int gen : {2})
        for (int lvl : {0, 4, 2, 1, 3}) {
            set_int(nu_lab_skip_epi, lvl);
            double r[10];
            int i = 0;
            for (int epi : {NU_EPI_PLAIN, NU_EPI_BIAS_RELU, NU_EPI_BIAS_SOFTPLUS, NU_EPI_MUL_DSP, NU_EPI_Q_SP}) r[i++] = tf(b256, time_nt(b256, epi, 20));
            r[i++] = tf(b1024, time_nt(b1024, NU_EPI_PLAIN, 20));
            printf("gen %d skip level %d : M=540k K=256 plain %6.1f relu %6.1f softplus %6.1f dsp %6.1f q_sp %6.1f | K=1024 %6.1f\n",
                   gen, lvl, r[0], r[1], r[2], r[3], r[4], r[5]);
        }
        set_int(nu_lab_skip_epi, 0);
    } else if (!strcmp(mode, "msweep")) {
        // 128- against 64-row tiles of the second-generation kernel by row count, K = 256 (the 16-deep-stage, three-workgroups-per-CU
        // kernel of profiles/r03/gen4_16deep_3wg_nt_kernel.patch was the third column of profiles/r03/lab_row_count_sweep_*.txt)
        for (int i = 0; i < 3; ++i) time_nt(b256, NU_EPI_PLAIN, 10);
        for (int pass = 0; pass < 2; ++pass)
        for (int m : {8192, 16384, 32768, 49152, 67456, 90112, 114048, 135168, 163840, 200704, 270336, 540672}) {
            Bufs b = b256;
            b.M = m;
            double r[2][3];
            for (int v = 0; v < 2; ++v) {
                nu_lab_small = v;
                int i = 0;
                for (int epi : {NU_EPI_BIAS_RELU, NU_EPI_BIAS_SOFTPLUS, NU_EPI_MUL_DSP}) r[v][i++] = tf(b, time_nt(b, epi, m < 100000 ? 80 : 30));
            }
            printf("M %7d : relu / softplus / dsp   128-row %6.1f %6.1f %6.1f | 64-row %6.1f %6.1f %6.1f\n", m,
                   r[0][0], r[0][1], r[0][2], r[1][0], r[1][1], r[1][2]);
        }
        nu_lab_small = -1;
    } else if (!strcmp(mode, "custagger")) {
        // all 256 first-dispatched workgroups reach their epilogues together: does spreading the store bursts over a tile time help?
        for (int i = 0; i < 3; ++i) time_nt(b256, NU_EPI_PLAIN, 10);
        for (int pass = 0; pass < 2; ++pass)
        for (int stag : {0, -60, -120, -225, -450}) {
            set_int(nu_lab_stagger_ticks, stag);
            double r[8];
            int i = 0;
            for (int epi : {NU_EPI_PLAIN, NU_EPI_BIAS_RELU, NU_EPI_BIAS_SOFTPLUS, NU_EPI_MUL_DSP}) r[i++] = tf(b256, time_nt(b256, epi, 20));
            printf("16 phases of %4d ticks (10 ns): M=540k K=256 plain %6.1f relu %6.1f softplus %6.1f dsp %6.1f\n", -stag, r[0], r[1], r[2], r[3]);
        }
        set_int(nu_lab_stagger_ticks, 0);
    } else if (!strcmp(mode, "pairs")) {
        // do the two workgroups of a CU run their epilogues at the same time?  Phase stamps (100 MHz) of co-resident pairs
        // (blocks b and b + 256), relative to the launch's first stamp, with and without a start-up stagger
        static unsigned long long tr[1024][NU_LAB_TILES][3];
        for (int stag : {0, 1800}) {
            set_int(nu_lab_stagger_ticks, stag);
            nu_lab_grid = 512;
            for (int epi : {NU_EPI_PLAIN, NU_EPI_BIAS_SOFTPLUS}) {
                const double ms = time_nt(b256, epi, 3);
                CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(nu_lab_trace), sizeof(tr)));
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < 512; ++w) t0 = tr[w][0][0] < t0 ? tr[w][0][0] : t0;
                printf("stagger %d ticks, epi %d: %.1f TFLOP/s\n", stag, epi, tf(b256, ms));
                double both = 0, either = 0;
                for (int b = 0; b < 256; ++b) {
                    // overlap of the two workgroups' epilogue intervals over tiles 1..5, as a share of their epilogue time
                    for (int t = 1; t < NU_LAB_TILES; ++t)
                        for (int u = 1; u < NU_LAB_TILES; ++u) {
                            const double a0 = (double)(tr[b][t][1] - t0), a1 = (double)(tr[b][t][2] - t0);
                            const double c0 = (double)(tr[b + 256][u][1] - t0), c1 = (double)(tr[b + 256][u][2] - t0);
                            const double lo = a0 > c0 ? a0 : c0, hi = a1 < c1 ? a1 : c1;
                            if (hi > lo) both += hi - lo;
                        }
                    for (int t = 1; t < NU_LAB_TILES; ++t) either += (double)(tr[b][t][2] - tr[b][t][1]);
                }
                printf("  epilogue time of a workgroup that coincides with its CU partner's epilogue: %.1f %%\n", 100.0 * both / either);
                for (int b : {0, 1, 37}) {
                    printf("  pair (%d, %d) us since launch [main start, main end, epilogue end] per tile:\n", b, b + 256);
                    for (int w : {b, b + 256}) {
                        printf("    block %3d:", w);
                        for (int t = 0; t < NU_LAB_TILES; ++t) printf("  %6.1f %6.1f %6.1f |", (tr[w][t][0] - t0) * 0.01, (tr[w][t][1] - t0) * 0.01, (tr[w][t][2] - t0) * 0.01);
                        printf("\n");
                    }
                }
            }
        }
        set_int(nu_lab_stagger_ticks, 0); nu_lab_grid = 0;
    } else if (!strcmp(mode, "small")) {
        // 64-row tiles (gen 2, TMN = 1) against 128-row tiles on point sets that do not fill the chip
        for (int i = 0; i < 3; ++i) time_nt(b256, NU_EPI_PLAIN, 10);
        for (int pass = 0; pass < 2; ++pass)
            for (int m : {2048, 4096, 8192, 16384, 24576, 32768, 49152, 65536, 131072}) {
                Bufs bs = b256;
                bs.M = m;
                double r[2][3];
                for (int sm = 0; sm < 2; ++sm) {
                    nu_lab_small = sm;
                    int i = 0;
                    for (int epi : {NU_EPI_BIAS_RELU, NU_EPI_BIAS_SOFTPLUS, NU_EPI_Q_SP}) r[sm][i++] = time_nt(bs, epi, 200) * 1e3;
                }
                printf("M=%6d K=256 N=256: 128-row tiles relu %6.1f us softplus %6.1f us q_sp %6.1f us | 64-row tiles relu %6.1f softplus %6.1f q_sp %6.1f  (%.0f / %.0f TFLOP/s relu)\n",
                       m, r[0][0], r[0][1], r[0][2], r[1][0], r[1][1], r[1][2], tf(bs, r[0][0] * 1e-3), tf(bs, r[1][0] * 1e-3));
            }
        nu_lab_small = -1;
    } else if (!strcmp(mode, "calib")) {
        // run under `rocprofv3 --pmc FETCH_SIZE`: each launch reads exactly 553,648,128 bytes (the 540672 x 256 fp32 operand)
        const long long n = (long long)M * 256;
        for (int i = 0; i < 3; ++i) {
            hipLaunchKernelGGL(calib_read4_kernel, dim3(4096), dim3(256), 0, 0, b256.A, n, b256.C);
            hipLaunchKernelGGL(calib_read16_kernel, dim3(4096), dim3(256), 0, 0, (const f32x4*)b256.A, n / 4, b256.C);
        }
        CK(hipDeviceSynchronize());
        printf("calib: each launch read %lld bytes\n", n * 4);
    } else if (!strcmp(mode, "mem")) {
        // is the lone workgroup's main loop waiting for memory?  A from an 8 MB window vs streamed from HBM
        for (int small = 0; small < 2; ++small) {
            set_int(nu_lab_small_a, small);
            for (int grid : {256, 512}) {
                nu_lab_grid = grid;
                for (int skip = 1; skip >= 0; --skip) {
                    set_int(nu_lab_skip_epi, skip);
                    printf("gen 2 small_A %d grid %3d skip_epi %d : K=256 plain %6.1f relu %6.1f | K=1024 plain %6.1f TFLOP/s\n", small, grid, skip,
                           tf(b256, time_nt(b256, NU_EPI_PLAIN, 20)), tf(b256, time_nt(b256, NU_EPI_BIAS_RELU, 20)), tf(b1024, time_nt(b1024, NU_EPI_PLAIN, 20)));
                }
            }
        }
        set_int(nu_lab_small_a, 0); set_int(nu_lab_skip_epi, 0); nu_lab_grid = 0;
    } else if (!strcmp(mode, "stages")) {
        // shader cycles one wave spends in each stage of a chunk (third tile of block 0), lone workgroup and two per CU
        set_int(nu_lab_stamps, 1);
        for (int grid : {256, 512}) {
            nu_lab_grid = grid;
            time_nt(b256, NU_EPI_BIAS_RELU, 2);
            long long st[8][8];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(nu_lab_stage), sizeof(st)));
            printf("grid %d: cycles per stage [k-group 0 | k-group 1 + LDS hand-over | k-group 2 + fetch | advance + barrier | k-group 3] (a k-group = 16 MFMAs = 1024 matrix cycles)\n", grid);
            for (int kt = 0; kt < 8; ++kt)
                printf("  chunk %d: %5lld %5lld %5lld %5lld %5lld   total %5lld\n", kt, st[kt][1] - st[kt][0], st[kt][2] - st[kt][1], st[kt][4] - st[kt][2],
                       st[kt][5] - st[kt][4], st[kt][6] - st[kt][5], st[kt][6] - st[kt][0]);
        }
        set_int(nu_lab_stamps, 0); nu_lab_grid = 0;
    } else if (!strcmp(mode, "clock")) {
        // matrix-pipe rate and shader clock of the two f32 MFMA shapes on random register operands, 1..3 waves per SIMD
        float* src = dalloc(1 << 20, 99, 2.f);
        for (int wps : {1, 2, 3}) {
            for (int shape : {32, 16}) {
                const int iters = 20000;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                    CK(hipEventRecord(e0, 0));
                    if (shape == 32) hipLaunchKernelGGL(bare_mfma_kernel<32>, dim3(256 * wps), dim3(256), 0, 0, src, b256.C, iters);
                    else hipLaunchKernelGGL(bare_mfma_kernel<16>, dim3(256 * wps), dim3(256), 0, 0, src, b256.C, iters);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned long long h[4];
                    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(lab_clk), sizeof(h)));
                    // flops: 32x32x2: 4096 per MFMA, 8 per iteration... per wave: iters * (shape == 32 ? 16 * 4096 : 32 * 2048)
                    const double fl = (double)iters * 65536.0 * 4 * 256 * wps;
                    if (rep) printf("bare MFMA %s, %d wave(s)/SIMD: %6.1f TFLOP/s, shader clock %.0f MHz (block 0: %.2f ms)\n",
                                    shape == 32 ? "32x32x2" : "16x16x4", wps, fl / ms / 1e9, (double)h[0] / (double)h[1] * 100.0, ms);
                }
            }
        }
        // the real kernels: clock held by block 0 over its whole life
        for (int v1 = 1; v1 >= 0; --v1)
        for (int grid : {256, 0})
        for (int skip = 0; skip < 2; ++skip) {
            set_int(nu_lab_skip_epi, skip);
            nu_lab_v1 = v1; nu_lab_grid = grid;
            for (int epi : {NU_EPI_PLAIN, NU_EPI_BIAS_SOFTPLUS}) {
                const double ms = time_nt(b256, epi, 30);
                unsigned long long h[2];
                CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(nu_dbg_clk), sizeof(h)));
                const double mhz = (double)h[0] / (double)h[1] * 100.0;
                printf("NT gen %d grid %3d K=256 epi=%d skip_epi=%d: %6.1f TFLOP/s, shader clock %.0f MHz -> %.1f %% of the matrix pipe at that clock\n", v1 ? 1 : 2, grid, epi, skip,
                       tf(b256, ms), mhz, 100.0 * tf(b256, ms) / (157.3 * mhz / 2400.0));
            }
        }
        nu_lab_v1 = 0; nu_lab_grid = 0; set_int(nu_lab_skip_epi, 0);
    } else {
        // traced launches: mean main-loop / epilogue duration per tile (tiles 1..5 of every workgroup), by grid size and epilogue
        static unsigned long long tr[1024][NU_LAB_TILES][3];
        const int lvl = argc > 2 ? atoi(argv[2]) : 0;
        set_int(nu_lab_skip_epi, lvl);
        printf("epilogue ablation level %d\n", lvl);
        for (int grid : {256, 512}) {
            for (int epi : {NU_EPI_PLAIN, NU_EPI_BIAS_SOFTPLUS, NU_EPI_MUL_DSP, NU_EPI_Q_SP}) {
                nu_lab_grid = grid;
                const double ms = time_nt(b256, epi, 3);
                CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(nu_lab_trace), sizeof(tr)));
                double main_us = 0, epi_us = 0, gap_us = 0;
                int n = 0;
                for (int w = 0; w < grid; ++w)
                    for (int t = 1; t < NU_LAB_TILES; ++t) {
                        main_us += (tr[w][t][1] - tr[w][t][0]) * 0.01;
                        epi_us += (tr[w][t][2] - tr[w][t][1]) * 0.01;
                        gap_us += (tr[w][t][0] - tr[w][t - 1][2]) * 0.01;
                        ++n;
                    }
                printf("grid %3d epi %d: %6.1f TFLOP/s  per tile: main loop %6.2f us, epilogue %6.2f us, hand-over to next tile %5.2f us\n",
                       grid, epi, tf(b256, ms), main_us / n, epi_us / n, gap_us / n);
            }
        }
        nu_lab_grid = 0;
    }
    return 0;
}
