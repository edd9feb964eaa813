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
// VGPRs), BK = 32, next chunk prefetched into registers, one 36.9 KB LDS buffer (k contiguous per
// operand row, stride 36) so that 2-3 workgroups share a CU.  K is required to be a multiple of
// 32: every operand buffer in this library is allocated with a padded leading dimension whose
// pad columns are zero (weights) / finite (activations).
#pragma once
#include "nu_common.h"

// NuEpi, NuGemmNT, NuGemmTN: see include/nu_nerf.h (the public C ABI)
#include "nu_nerf.h"

int nu_gemm_nt_launch(const NuGemmNT& g, hipStream_t stream);
int nu_gemm_nt_batch_launch(const NuGemmNT* probs, int n, hipStream_t stream);
int nu_gemm_tn_launch(const NuGemmTN& g, hipStream_t stream);
// append one reduction problem (see NuReduceDesc) to a host-side list
int nu_reduce_push(NuReduceDesc* descs, int* ndesc, int cap, const float* slab, int S, int N1, int N2, int rs,
                   long long ss, float* out, int ldo, float alpha, int accumulate);

// context services (gemm_tn.hip): slab space from the deferred-reduction arena, per-launch events
int nu_ctx_take(NuOpCtx* c, long long nbytes, int ndesc_needed, hipStream_t stream, float** out, long long* out_bytes);
void nu_ctx_ev_begin(NuOpCtx* c, hipStream_t stream);
void nu_ctx_ev_end(NuOpCtx* c, hipStream_t stream, double kind, double flops, double bytes);
