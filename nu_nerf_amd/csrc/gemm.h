// gemm.h -- fp32 MFMA GEMM building blocks for the fused MLP ops (gfx950 only).
//
// Two kernels carry >95 % of the training step's FLOPs:
//
//   gemm_nt : C[M, N]  = epi( A[M, K] . B[N, K]^T )        activations x packed weights
//             (forward layers, backward-data sweeps, the SDF reverse/tangent sweeps)
//   gemm_tn : C[N1,N2] = sum_p A[p, N1] . B[p, N2]          weight gradients, split over p
//
// Both run on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD -> 157.3 TFLOP/s chip peak).
// 128x128 workgroup tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// VGPRs), BK = 32, register-prefetched double-buffered LDS.  K is required to be a multiple of
// 32: every operand buffer in this library is allocated with a padded leading dimension whose
// pad columns are zero (weights) / finite (activations).
#pragma once
#include "nu_common.h"

// Epilogue opcodes for gemm_nt.  v = alpha * acc at (row, col).
enum NuEpi {
    NU_EPI_BIAS_NONE = 0,      // C = v + bias[col]
    NU_EPI_BIAS_RELU = 1,      // C = relu(v + bias[col])
    NU_EPI_BIAS_SOFTPLUS = 2,  // C = softplus100(v + bias[col])
    NU_EPI_MUL_DRELU = 3,      // C = v * (H[row,col] > 0)
    NU_EPI_MUL_DSP = 4,        // C = v * sp'(H)              sp' = 1 - exp(-100 H)
    NU_EPI_Q_SP = 5,           // C = v * sp'(H);  C2 = v * D[row,col] * 100 * (1 - sp'(H))
    NU_EPI_B_SP = 6,           // C = v * sp'(H) + Cadd[row,col]
    NU_EPI_PLAIN = 7,          // C = v
    NU_EPI_B_RELU = 8,         // C = v * (H > 0) + Cadd[row,col]
    NU_EPI_COUNT = 9
};

struct NuGemmNT {
    const float* A; int lda;      // [M, lda], lda >= K
    const float* B; int ldb;      // [>=ceil128(N), ldb], ldb >= K   (packed weights, zero padded)
    int M, N, K;                  // K % 32 == 0
    float* C; int ldc;            // primary output
    float* C2; int ldc2;          // secondary output (NU_EPI_Q_SP)
    const float* bias;            // [N]  (bias epilogues; may be null => 0)
    const float* H; int ldh;      // post-activation aux
    const float* D; int ldd;      // delta aux (NU_EPI_Q_SP)
    const float* Cadd; int ldadd; // additive aux (NU_EPI_B_SP)
    int zero_to;                  // cols in [N, zero_to) of C (and C2) are written as 0
    int act_cols;                 // derivative epilogues: cols >= act_cols are written as plain v (0 => all cols)
    float alpha;
    // grouped launch: blockIdx.z in [0, groups); element strides per group
    int groups;
    long long sA, sB, sC, sC2, sBias, sH, sD, sCadd;
    int epi;
};

struct NuGemmTN {
    // C[n1, n2] = sum over pairs, p of A_i[p, n1] * B_i[p, n2]
    const float* A0; int lda0; const float* B0; int ldb0;
    const float* A1; int lda1; const float* B1; int ldb1;   // optional second pair (A1 == null => none)
    int P;          // rows to reduce over
    int N1, N2;     // output extent
    float* slab;    // [S][N1p][N2p] partials, N1p = ceil128(N1), N2p = ceil128(N2)
    float* bias_slab; // optional [S][N1p] : column sums of A0 (+A1 is NOT included)
    int S;          // number of p-splits
    int groups; long long sA0, sB0, sA1, sB1;  // grouped: blockIdx.z = group*S + split
    long long sSlab, sBiasSlab;
};

int nu_gemm_nt_launch(const NuGemmNT& g, hipStream_t stream);
int nu_gemm_tn_launch(const NuGemmTN& g, hipStream_t stream);
// out[n1*ldo + n2] (+)= alpha * sum_s slab[s][n1][n2]
int nu_slab_reduce_launch(const float* slab, int S, int N1, int N2, float* out, int ldo, float alpha,
                          int accumulate, hipStream_t stream);
// out[n1] (+)= sum_s bias_slab[s][n1]
int nu_bias_slab_reduce_launch(const float* bslab, int S, int N1, float* out, int accumulate, hipStream_t stream);
int nu_slab_reduce_strided_launch(const float* slab, int S, int N1, int N2, int N1p, int N2p, float* out, int ldo,
                                  float alpha, int accumulate, hipStream_t stream);
