// loss.hip -- fused loss assembly of the stage-1 training step.
//
// Replaces, for the loss set of the shipped stage-1 configs, the chain of small eager reductions between the renderer's
// per-ray outputs and the scalar the trainer back-propagates (paths relative to /root/reference):
//   white background + clamp            network/renderer_zerothick.py:783-787
//   compute_rgb_loss (charbonier)       network/renderer_zerothick.py:501-513
//   colour_spec activation              network/renderer_zerothick.py:780-781 (linear_to_srgb(exp(min(., exp_max))))
//   NeRFRenderLoss / EikonalLoss / OuterRegLoss / NormalOrientationLoss      network/loss.py:26-48, :194-213
//   total = sum of means of the 'loss*' entries                              train/trainer_zero.py:153-161
// One forward kernel (per-ray outputs + per-block partial sums, fixed order), one finishing block (terms + total), one
// backward kernel.  Deterministic: no float atomics.  The upstream gradient is read from device memory (no host sync).
#include "nu_common.h"

#define NU_LOSS_TERMS 4          // sum charbonnier, sum eikonal integrand, sum (bkgr - spec)^2 over candidates, sum normal integrand
#define NU_LOSS_BLOCK 256

static __device__ inline float nu_block_sum(float v, float* red) {
    v = nu_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < NU_LOSS_BLOCK / 64; ++i) s += red[i];
    return s;
}

// rays r < R: colour, losses per ray; points p < P: eikonal integrand.  cand (optional, u8[R]): rays that take part in the
// outer regulariser (network/renderer.py:710-725); NULL = all rays.
__global__ __launch_bounds__(NU_LOSS_BLOCK) void loss_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ acc,
                                                                 const float* __restrict__ rgb_bg, const float* __restrict__ spec_raw,
                                                                 const float* __restrict__ gerr, const float* __restrict__ nrm_sum,
                                                                 const float* __restrict__ gt, const unsigned char* __restrict__ cand,
                                                                 int R, int P, int white_bg, float exp_max, float* __restrict__ ray_rgb,
                                                                 float* __restrict__ color_spec, float* __restrict__ loss_rgb,
                                                                 float* __restrict__ partial) {
    __shared__ float red[NU_LOSS_BLOCK / 64];
    const long long i = (long long)blockIdx.x * NU_LOSS_BLOCK + threadIdx.x;
    float s_rgb = 0.f, s_eik = 0.f, s_reg = 0.f, s_nrm = 0.f, s_cnt = 0.f;
    if (i < R) {
        const float a = acc[i];
        float d2 = 0.f;
        const bool c = cand == nullptr || cand[i] != 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float col = rgb[i * 3 + k];
            if (white_bg) col = col + (1.0f - a);
            const float pr = fminf(fmaxf(col, 0.0f), 1.0f);
            ray_rgb[i * 3 + k] = pr;
            const float df = gt[i * 3 + k] - pr;
            d2 += df * df;
            const float sp = nu_linear_to_srgb(expf(fminf(spec_raw[i * 3 + k], exp_max)));
            color_spec[i * 3 + k] = sp;
            if (c) {
                const float e = rgb_bg[i * 3 + k] - sp;
                s_reg += e * e;
            }
        }
        const float l = sqrtf(d2 + 0.001f);
        loss_rgb[i] = l;
        s_rgb = l;
        s_cnt = c ? 1.f : 0.f;
        if (nrm_sum) s_nrm = nrm_sum[i];
    }
    if (i < P) s_eik = gerr[i];
    float* out = partial + (long long)blockIdx.x * (NU_LOSS_TERMS + 1);
    const float t0 = nu_block_sum(s_rgb, red), t1 = nu_block_sum(s_eik, red), t2 = nu_block_sum(s_reg, red),
                t3 = nu_block_sum(s_nrm, red), t4 = nu_block_sum(s_cnt, red);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3; out[4] = t4; }
}

// terms[0..3] = loss_rgb, loss_eikonal, loss_outer_reg, loss_normal (each already weighted, = mean of the entry);
// terms[4] = their sum; terms[5] = candidate count
// point_weight (optional device scalar): this rank's share of a per-point mean over the union of all ranks' points
// (parallel.GradAllReducer.point_weight); it scales the eikonal term and its gradient.  NULL = 1.
__global__ __launch_bounds__(NU_LOSS_BLOCK) void loss_finish_kernel(const float* __restrict__ partial, int nblk, int R, int P,
                                                                    float w_eik, float w_reg, float w_nrm,
                                                                    const float* __restrict__ point_weight, float* __restrict__ terms) {
    __shared__ float red[NU_LOSS_BLOCK / 64];
    float s[NU_LOSS_TERMS + 1] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += NU_LOSS_BLOCK)
#pragma unroll
        for (int k = 0; k < NU_LOSS_TERMS + 1; ++k) s[k] += partial[(long long)b * (NU_LOSS_TERMS + 1) + k];
    float t[NU_LOSS_TERMS + 1];
#pragma unroll
    for (int k = 0; k < NU_LOSS_TERMS + 1; ++k) t[k] = nu_block_sum(s[k], red);
    if (threadIdx.x == 0) {
        const float l_rgb = R > 0 ? t[0] / (float)R : 0.f;
        // (mean of gerr * pw) * w_eik, in the registry's order of operations; no inner point: the module reports zeros(1)
        const float pw = point_weight ? point_weight[0] : 1.0f;
        const float l_eik = P > 0 ? w_eik * (pw * t[1]) / (float)P : 0.f;
        const float l_reg = t[4] > 0.f ? w_reg * t[2] / (3.0f * t[4]) : 0.f;
        const float l_nrm = R > 0 ? w_nrm * t[3] / (float)R : 0.f;
        terms[0] = l_rgb; terms[1] = l_eik; terms[2] = l_reg; terms[3] = l_nrm;
        terms[4] = ((l_rgb + l_eik) + l_reg) + l_nrm;
        terms[5] = t[4];
    }
}

__global__ __launch_bounds__(NU_LOSS_BLOCK) void loss_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ acc,
                                                                 const float* __restrict__ rgb_bg, const float* __restrict__ spec_raw,
                                                                 const float* __restrict__ gt, const unsigned char* __restrict__ cand,
                                                                 const float* __restrict__ ray_rgb, const float* __restrict__ color_spec,
                                                                 const float* __restrict__ loss_rgb, const float* __restrict__ terms,
                                                                 const float* __restrict__ upstream,
                                                                 const float* __restrict__ point_weight, int R, int P, int white_bg,
                                                                 float exp_max, float w_eik, float w_reg, float w_nrm,
                                                                 float* __restrict__ d_rgb, float* __restrict__ d_acc,
                                                                 float* __restrict__ d_rgb_bg, float* __restrict__ d_spec_raw,
                                                                 float* __restrict__ d_gerr, float* __restrict__ d_nrm) {
    const long long i = (long long)blockIdx.x * NU_LOSS_BLOCK + threadIdx.x;
    const float up = upstream[0];
    if (i < R) {
        const float a = acc[i];
        const float inv_l = 1.0f / loss_rgb[i];
        const float g_l = up / (float)R;                                    // d total / d loss_rgb[i]
        const float ncand = terms[5];
        const bool c = cand == nullptr || cand[i] != 0;
        const float g_reg = (c && ncand > 0.f) ? up * w_reg * 2.0f / (3.0f * ncand) : 0.f;
        float da = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float col = rgb[i * 3 + k];
            if (white_bg) col = col + (1.0f - a);
            const float pr = ray_rgb[i * 3 + k];
            // d sqrt(d2 + 1e-3) / d pr = -(gt - pr) / l ; torch.clamp passes the gradient on [0, 1] inclusive
            const float g_pr = -(gt[i * 3 + k] - pr) * inv_l * g_l;
            const float g_col = (col >= 0.0f && col <= 1.0f) ? g_pr : 0.f;
            d_rgb[i * 3 + k] = g_col;
            if (white_bg) da -= g_col;
            const float e = rgb_bg[i * 3 + k] - color_spec[i * 3 + k];
            d_rgb_bg[i * 3 + k] = g_reg * e;
            const float sr = spec_raw[i * 3 + k];
            const float ex = expf(fminf(sr, exp_max));
            d_spec_raw[i * 3 + k] = sr <= exp_max ? -g_reg * e * nu_linear_to_srgb_grad(ex) * ex : 0.f;
        }
        d_acc[i] = da;
        if (d_nrm) d_nrm[i] = up * w_nrm / (float)R;
    }
    if (i < P) d_gerr[i] = (point_weight ? point_weight[0] : 1.0f) * (up * w_eik / (float)P);
}

extern "C" long long nu_loss_workspace_bytes(int R, int P) {
    const long long n = R > P ? R : P;
    return (nu_cdivl(n > 0 ? n : 1, NU_LOSS_BLOCK) * (NU_LOSS_TERMS + 1)) * (long long)sizeof(float);
}

extern "C" int nu_loss_fwd(const float* rgb, const float* acc, const float* rgb_bg, const float* spec_raw, const float* gerr,
                           const float* nrm_sum, const float* gt, const unsigned char* cand, int R, int P, int white_bg,
                           float exp_max, float w_eik, float w_reg, float w_nrm, float* ray_rgb, float* color_spec,
                           float* loss_rgb, float* terms, const float* point_weight, void* workspace, long long workspace_bytes,
                           hipStream_t stream) {
    if (R <= 0 || P < 0) return NU_ERR_ARG;
    if (workspace_bytes < nu_loss_workspace_bytes(R, P)) return NU_ERR_WORKSPACE;
    const long long n = R > P ? R : P;
    const int nblk = (int)nu_cdivl(n, NU_LOSS_BLOCK);
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(nblk), dim3(NU_LOSS_BLOCK), 0, stream, rgb, acc, rgb_bg, spec_raw, gerr, nrm_sum, gt, cand,
                       R, P, white_bg, exp_max, ray_rgb, color_spec, loss_rgb, (float*)workspace);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(NU_LOSS_BLOCK), 0, stream, (const float*)workspace, nblk, R, P, w_eik, w_reg,
                       w_nrm, point_weight, terms);
    return nu_launch_status();
}

extern "C" int nu_loss_bwd(const float* rgb, const float* acc, const float* rgb_bg, const float* spec_raw, const float* gt,
                           const unsigned char* cand, const float* ray_rgb, const float* color_spec, const float* loss_rgb,
                           const float* terms, const float* upstream, const float* point_weight, int R, int P, int white_bg,
                           float exp_max, float w_eik, float w_reg, float w_nrm, float* d_rgb, float* d_acc, float* d_rgb_bg, float* d_spec_raw, float* d_gerr,
                           float* d_nrm, hipStream_t stream) {
    if (R <= 0 || P < 0) return NU_ERR_ARG;
    const long long n = R > P ? R : P;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)nu_cdivl(n, NU_LOSS_BLOCK)), dim3(NU_LOSS_BLOCK), 0, stream, rgb, acc, rgb_bg,
                       spec_raw, gt, cand, ray_rgb, color_spec, loss_rgb, terms, upstream, point_weight, R, P, white_bg, exp_max, w_eik, w_reg, w_nrm,
                       d_rgb, d_acc, d_rgb_bg, d_spec_raw, d_gerr, d_nrm);
    return nu_launch_status();
}
