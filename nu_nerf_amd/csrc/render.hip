// render.hip -- per-ray / per-sample kernels of render_core: mid-point generation + inner/outer
// compaction, NeuS SDF->alpha, NeRF++ density activation, physically-based shading combine, and the
// front-to-back composite, each with its hand-derived backward.  All HBM-bound; ray-major layouts so a
// wavefront reads one ray's samples as a contiguous burst.
//
// Reference semantics restated (paths relative to /root/reference):
//   render_core            network/renderer_zerothick.py:725-820
//   compute_sdf_alpha      network/renderer_zerothick.py:657-685
//   compute_density_alpha  network/renderer_zerothick.py:687-693, :515-516
//   AppShadingNetwork mix  network/field.py:698-740  (+ :658-665 light mixing)
#include "nu_common.h"
#include "gemm.h"

#define NU_PT 8
#define NU_MAXCHUNK 4  // samples per lane: supports up to 256 samples per ray

// ------------------------------------------------------------------------------------------------
// Partition: per sample j of ray r: dist_j = z_{j+1}-z_j (last duplicated), mid = z_j + dist_j/2,
// x = o + d*mid, inner = |x| <= 1.  Pass 1 counts inner samples per ray; an exclusive scan gives ray
// offsets; pass 2 writes compact point records in (ray, sample) order (the order boolean-mask indexing
// gives the reference) for both the inner and the outer set.
// ------------------------------------------------------------------------------------------------
static __device__ inline float nu_norm3(const float* x) {
#pragma clang fp contract(off)
    return sqrtf((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
}
static __device__ inline void nu_sample_point(const float* __restrict__ zrow, int S, int j, const float* o, const float* d,
                                              float* x, float& dist) {
    // contraction off + plain operators (HIP's __fmul_rn/__fadd_rn would still fuse): the inner/outer decision
    // |x| <= 1 should round like the reference's eager ops
#pragma clang fp contract(off)
    const float z0 = zrow[j];
    if (j + 1 < S) dist = zrow[j + 1] - z0;
    else dist = S >= 2 ? zrow[S - 1] - zrow[S - 2] : 0.f;
    const float mid = z0 + dist * 0.5f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float m = d[c] * mid;
        x[c] = o[c] + m;
    }
}

__global__ __launch_bounds__(256) void partition_count_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                              const float* __restrict__ z, int R, int S,
                                                              int* __restrict__ cnt_in) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    float oo[3], dd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { oo[c] = o[r * 3LL + c]; dd[c] = d[r * 3LL + c]; }
    int cnt = 0;
    for (int j = lane; j < S; j += 64) {
        float x[3], dist;
        nu_sample_point(z + (long long)r * S, S, j, oo, dd, x, dist);
        cnt += nu_norm3(x) <= 1.0f ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) cnt_in[r] = cnt;
}

// single-block exclusive scan over R <= 1M ints; also writes totals[0] = sum
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ in, int n, int* __restrict__ out,
                                                    int* __restrict__ totals) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = tid * per;
    int e = b + per;
    e = e < n ? e : n;
    int s = 0;
    for (int i = b; i < e; ++i) s += in[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int t = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = b; i < e; ++i) {
        out[i] = run;
        run += in[i];
    }
    if (tid == 1023) totals[0] = part[1023];
}

__global__ __launch_bounds__(256) void partition_write_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                              const float* __restrict__ z, int R, int S,
                                                              const int* __restrict__ off_in,
                                                              float* __restrict__ pt_in, int* __restrict__ idx_in,
                                                              float* __restrict__ pt_out, int* __restrict__ idx_out,
                                                              unsigned char* __restrict__ inner_rm) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    float oo[3], dd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { oo[c] = o[r * 3LL + c]; dd[c] = d[r * 3LL + c]; }
    // direction as render_core normalises it once more (renderer_zerothick.py:740)
    const float dn = fmaxf(sqrtf(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]), 1e-12f);
    const float du[3] = {dd[0] / dn, dd[1] / dn, dd[2] / dn};
    int base_in = off_in[r];
    int base_out = r * S - base_in;
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        float x[3] = {0.f, 0.f, 0.f}, dist = 0.f;
        bool inner = false;
        if (j < S) {
            nu_sample_point(z + (long long)r * S, S, j, oo, dd, x, dist);
            inner = nu_norm3(x) <= 1.0f;
        }
        const unsigned long long m_in = __ballot(j < S && inner);
        const unsigned long long m_out = __ballot(j < S && !inner);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (j < S) {
            const int flat = r * S + j;
            float* rec;
            if (inner) {
                const int k = base_in + __popcll(m_in & below);
                idx_in[k] = flat;
                rec = pt_in + (long long)k * NU_PT;
            } else {
                const int k = base_out + __popcll(m_out & below);
                idx_out[k] = flat;
                rec = pt_out + (long long)k * NU_PT;
            }
            f32x4 a = {x[0], x[1], x[2], dist};
            f32x4 b = {du[0], du[1], du[2], 0.f};
            *reinterpret_cast<f32x4*>(rec) = a;
            *reinterpret_cast<f32x4*>(rec + 4) = b;
            inner_rm[flat] = inner ? 1 : 0;
        }
        base_in += __popcll(m_in);
        base_out += __popcll(m_out);
    }
}

// counts -> offsets + totals[0] = P_in.  Host reads totals after the stream reaches this point.
extern "C" int nu_partition_count(const float* o, const float* d, const float* z, int R, int S, int* cnt_in,
                                  int* off_in, int* totals, hipStream_t stream) {
    if (R <= 0 || S <= 0) return NU_ERR_ARG;
    hipLaunchKernelGGL(partition_count_kernel, dim3(nu_cdiv(R, 4)), dim3(256), 0, stream, o, d, z, R, S, cnt_in);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, cnt_in, R, off_in, totals);
    return nu_launch_status();
}
extern "C" int nu_partition_write(const float* o, const float* d, const float* z, int R, int S, const int* off_in,
                                  float* pt_in, int* idx_in, float* pt_out, int* idx_out, unsigned char* inner_rm,
                                  hipStream_t stream) {
    if (R <= 0 || S <= 0) return NU_ERR_ARG;
    hipLaunchKernelGGL(partition_write_kernel, dim3(nu_cdiv(R, 4)), dim3(256), 0, stream, o, d, z, R, S, off_in, pt_in,
                       idx_in, pt_out, idx_out, inner_rm);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// NeuS alpha (inner points).  inv_s = clip(exp(10*variance), 1e-6, 1e6) read from the parameter on device.
// ------------------------------------------------------------------------------------------------
struct NuAlphaVals {
    float alpha, gerr;
    float dalpha_dsdf, dalpha_dcos, dalpha_dinvs;  // partials (zero where the final clip is active)
    float norm;
};

static __device__ inline NuAlphaVals nu_neus_alpha(float sdf, const float* n, const float* d, float dist, float inv_s,
                                                   float anneal) {
    NuAlphaVals v;
    const float cosv = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
    const float ra = fmaxf(-cosv * 0.5f + 0.5f, 0.f);
    const float rb = fmaxf(-cosv, 0.f);
    const float it = -(ra * (1.0f - anneal) + rb * anneal);
    const float dit_dcos = -((-cosv * 0.5f + 0.5f > 0.f ? -0.5f : 0.f) * (1.0f - anneal) + (-cosv > 0.f ? -1.0f : 0.f) * anneal);
    const float en = sdf + it * dist * 0.5f;  // estimated next
    const float ep = sdf - it * dist * 0.5f;  // estimated prev
    const float pc = nu_sigmoid(ep * inv_s);
    const float nc = nu_sigmoid(en * inv_s);
    const float num = pc - nc + 1e-5f;
    const float den = pc + 1e-5f;
    const float a = num / den;
    v.alpha = fminf(fmaxf(a, 0.f), 1.f);
    const bool pass = (a >= 0.f) && (a <= 1.f);
    // d a / d pc = (den - num)/den^2 ; d a / d nc = -1/den
    const float da_dpc = (den - num) / (den * den);
    const float da_dnc = -1.0f / den;
    const float dpc = pc * (1.0f - pc);
    const float dnc = nc * (1.0f - nc);
    const float da_dep = da_dpc * dpc * inv_s;
    const float da_den = da_dnc * dnc * inv_s;
    const float g = pass ? 1.f : 0.f;
    v.dalpha_dsdf = g * (da_dep + da_den);
    const float da_dit = (da_den - da_dep) * dist * 0.5f;
    v.dalpha_dcos = g * da_dit * dit_dcos;
    v.dalpha_dinvs = g * (da_dpc * dpc * ep + da_dnc * dnc * en);
    v.norm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    v.gerr = (v.norm - 1.0f) * (v.norm - 1.0f);
    return v;
}

static __device__ inline float nu_inv_s(const float* variance) {
    return fminf(fmaxf(expf(variance[0] * 10.0f), 1e-6f), 1e6f);
}

__global__ __launch_bounds__(256) void neus_alpha_fwd_kernel(const float* __restrict__ YX, int ldy,
                                                             const float* __restrict__ nrm, const float* __restrict__ pt,
                                                             const int* __restrict__ idx, int P,
                                                             const float* __restrict__ variance, float anneal,
                                                             float* __restrict__ alpha_rm, float* __restrict__ gerr,
                                                             float* __restrict__ color_rm) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float inv_s = nu_inv_s(variance);
    const float n[3] = {nrm[p * 3LL], nrm[p * 3LL + 1], nrm[p * 3LL + 2]};
    const f32x4 a = *reinterpret_cast<const f32x4*>(pt + (long long)p * NU_PT);
    const f32x4 b = *reinterpret_cast<const f32x4*>(pt + (long long)p * NU_PT + 4);
    const float d[3] = {b[0], b[1], b[2]};
    NuAlphaVals v = nu_neus_alpha(YX[(long long)p * ldy], n, d, a[3], inv_s, anneal);
    const int k = idx[p];
    alpha_rm[k] = v.alpha;
    gerr[p] = v.gerr;
    // 4th colour channel: max(n . d, 0), composited into loss_normal by the non-zero-thickness renderer
    // (network/renderer.py:693-705); ignored (zero cotangent) by the zero-thickness one
    if (color_rm) color_rm[k * 4LL + 3] = fmaxf(d[0] * n[0] + d[1] * n[1] + d[2] * n[2], 0.f);
}
extern "C" int nu_neus_alpha_fwd(const float* YX, int ldy, const float* nrm, const float* pt, const int* idx, int P,
                                 const float* variance, float anneal, float* alpha_rm, float* gerr, float* color_rm,
                                 hipStream_t stream) {
    if (P <= 0) return NU_OK;
    hipLaunchKernelGGL(neus_alpha_fwd_kernel, dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, YX, ldy, nrm, pt, idx, P,
                       variance, anneal, alpha_rm, gerr, color_rm);
    return nu_launch_status();
}

// backward: dalpha_rm (ray-major), dgerr[p] (per-point cotangent of gradient_error), dn_shade (may be null)
//   -> dYX[p, 0] = d sdf ;  nbar[p, 0:3] = total cotangent of the raw normal ;  dvar += d variance
// The variance gradient is a sum over all points: every block leaves its partial (already scaled by d inv_s / d variance) in a
// slab of the CALLER's workspace, and the deterministic batched split reduction the weight gradients use (nu_slab_reduce_batched)
// adds the slab up in a fixed order -- no float atomics (one atomicAdd per block used to make d variance differ by an ulp from run
// to run), and no state in the library: two engines, or two streams, may train their variances at the same time.
__global__ __launch_bounds__(256) void neus_alpha_bwd_kernel(const float* __restrict__ YX, int ldy,
                                                             const float* __restrict__ nrm, const float* __restrict__ pt,
                                                             const int* __restrict__ idx, int P,
                                                             const float* __restrict__ variance, float anneal,
                                                             const float* __restrict__ dalpha_rm,
                                                             const float* __restrict__ dgerr,
                                                             const float* __restrict__ dn_shade,
                                                             const float* __restrict__ dcolor_rm,
                                                             float* __restrict__ dYX, int lddy, float* __restrict__ nbar,
                                                             float* __restrict__ dvar_partial) {
    __shared__ float red[4];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float dinv = 0.f;
    const float var = variance[0];
    const float raw = expf(var * 10.0f);
    const float inv_s = fminf(fmaxf(raw, 1e-6f), 1e6f);
    if (p < P) {
        const float n[3] = {nrm[p * 3LL], nrm[p * 3LL + 1], nrm[p * 3LL + 2]};
        const f32x4 a = *reinterpret_cast<const f32x4*>(pt + (long long)p * NU_PT);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pt + (long long)p * NU_PT + 4);
        const float d[3] = {b[0], b[1], b[2]};
        NuAlphaVals v = nu_neus_alpha(YX[(long long)p * ldy], n, d, a[3], inv_s, anneal);
        const int kk = idx[p];
        const float ga = dalpha_rm[kk];
        dYX[(long long)p * lddy] = ga * v.dalpha_dsdf;
        const float ndot = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
        const float gno = (dcolor_rm && ndot > 0.f) ? dcolor_rm[kk * 4LL + 3] : 0.f;   // d max(n.d, 0) / d n = d
        const float ge = dgerr ? dgerr[p] : 0.f;
        // d gerr / d n = 2 (|n| - 1) n / |n|
        const float ke = v.norm > 0.f ? ge * 2.0f * (v.norm - 1.0f) / v.norm : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            nbar[p * 3LL + c] = (ga * v.dalpha_dcos + gno) * d[c] + ke * n[c] + (dn_shade ? dn_shade[p * 3LL + c] : 0.f);
        dinv = ga * v.dalpha_dinvs;
    }
    if (dvar_partial) {
        dinv = nu_wave_sum(dinv);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dinv;
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool pass = raw >= 1e-6f && raw <= 1e6f;           // inside the clip range: d inv_s / d variance = 10 inv_s
            dvar_partial[blockIdx.x] = pass ? ((red[0] + red[1]) + (red[2] + red[3])) * raw * 10.0f : 0.f;
        }
    }
}
extern "C" long long nu_neus_alpha_bwd_workspace_bytes(int P) { return (long long)nu_cdiv(P > 0 ? P : 1, 256) * sizeof(float); }
extern "C" int nu_neus_alpha_bwd(const float* YX, int ldy, const float* nrm, const float* pt, const int* idx, int P,
                                 const float* variance, float anneal, const float* dalpha_rm, const float* dgerr,
                                 const float* dn_shade, const float* dcolor_rm, float* dYX, int lddy, float* nbar,
                                 float* dvar, void* workspace, long long workspace_bytes, NuReduceDesc* descs, int* ndesc, int cap,
                                 hipStream_t stream) {
    if (P <= 0) return NU_OK;
    const int nblk = nu_cdiv(P, 256);
    float* partial = nullptr;
    if (dvar) {         // d variance = sum of the per-block partials: one deferred reduction problem (finished by nu_slab_reduce_batched / nu_ctx_flush)
        if (!workspace || !descs || !ndesc || workspace_bytes < nu_neus_alpha_bwd_workspace_bytes(P)) return NU_ERR_WORKSPACE;
        partial = static_cast<float*>(workspace);
        const int rc = nu_reduce_push(descs, ndesc, cap, partial, nblk, 1, 1, 1, 1, dvar, 1, 1.0f, 0);
        if (rc != NU_OK) return rc;
    }
    hipLaunchKernelGGL(neus_alpha_bwd_kernel, dim3(nblk), dim3(256), 0, stream, YX, ldy, nrm, pt, idx, P,
                       variance, anneal, dalpha_rm, dgerr, dn_shade, dcolor_rm, dYX, lddy, nbar, partial);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// NeRF++ activation (outer points): alpha = 1 - exp(-softplus(sigma) dist); c = srgb(exp(min(raw, 5)))
// ------------------------------------------------------------------------------------------------
static __device__ inline float nu_softplus1(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(256) void nerf_act_fwd_kernel(const float* __restrict__ sigma, int lds,
                                                           const float* __restrict__ rgb, int ldr,
                                                           const float* __restrict__ pt, const int* __restrict__ idx, int P,
                                                           float* __restrict__ alpha_rm, float* __restrict__ color_rm) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float dist = pt[(long long)p * NU_PT + 3];
    const int k = idx[p];
    alpha_rm[k] = 1.0f - expf(-nu_softplus1(sigma[(long long)p * lds]) * dist);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        color_rm[k * 4LL + c] = nu_linear_to_srgb(expf(fminf(rgb[(long long)p * ldr + c], 5.0f)));
    color_rm[k * 4LL + 3] = 0.f;
}
extern "C" int nu_nerf_act_fwd(const float* sigma, int lds, const float* rgb, int ldr, const float* pt, const int* idx,
                               int P, float* alpha_rm, float* color_rm, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    hipLaunchKernelGGL(nerf_act_fwd_kernel, dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, sigma, lds, rgb, ldr, pt, idx, P,
                       alpha_rm, color_rm);
    return nu_launch_status();
}

__global__ __launch_bounds__(256) void nerf_act_bwd_kernel(const float* __restrict__ sigma, int lds,
                                                           const float* __restrict__ rgb, int ldr,
                                                           const float* __restrict__ pt, const int* __restrict__ idx, int P,
                                                           const float* __restrict__ dalpha_rm,
                                                           const float* __restrict__ dcolor_rm,
                                                           float* __restrict__ dsigma, int ldds, float* __restrict__ drgb,
                                                           int lddr) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float dist = pt[(long long)p * NU_PT + 3];
    const int k = idx[p];
    const float s = sigma[(long long)p * lds];
    const float sp = nu_softplus1(s);
    const float dsp = s > 20.f ? 1.0f : nu_sigmoid(s);
    dsigma[(long long)p * ldds] = dalpha_rm[k] * expf(-sp * dist) * dist * dsp;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float r = rgb[(long long)p * ldr + c];
        const float e = expf(fminf(r, 5.0f));
        drgb[(long long)p * lddr + c] = r <= 5.0f ? dcolor_rm[k * 4LL + c] * nu_linear_to_srgb_grad(e) * e : 0.f;
    }
}
extern "C" int nu_nerf_act_bwd(const float* sigma, int lds, const float* rgb, int ldr, const float* pt, const int* idx,
                               int P, const float* dalpha_rm, const float* dcolor_rm, float* dsigma, int ldds,
                               float* drgb, int lddr, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    hipLaunchKernelGGL(nerf_act_bwd_kernel, dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, sigma, lds, rgb, ldr, pt, idx, P,
                       dalpha_rm, dcolor_rm, dsigma, ldds, drgb, lddr);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Shading combine.  Raw (pre-activation) head outputs in, sRGB colour out (scattered ray-major).
//   Mraw [P, ldm]: metallic, roughness, albedo(3), transmission          (sigmoid)
//   OLo [3P, 4] : outer_light raw for IDE(n^,1) | IDE(r,rho) | IDE(r,0)  (exp(min(.,exp_max)))
//   ILo [2P, 4] : inner_light raw for rho | 0 ;  IWo [P]: inner_weight raw ;  RLo [P,4]: refrac_light raw
// ------------------------------------------------------------------------------------------------
struct NuLut {
    float A, B, dA_du, dB_du, dA_dv, dB_dv;
};
static __device__ inline NuLut nu_lut(const float* __restrict__ lut, float u, float v) {
    // texel centres at (i+0.5)/256, bilinear, clamp to edge (dr.texture linear/clamp; field.py:719-722)
    NuLut o;
    const float fx_raw = u * 256.0f - 0.5f, fy_raw = v * 256.0f - 0.5f;
    const float fx = fminf(fmaxf(fx_raw, 0.f), 255.f), fy = fminf(fmaxf(fy_raw, 0.f), 255.f);
    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    const int x1 = x0 + 1 < 255 ? x0 + 1 : 255, y1 = y0 + 1 < 255 ? y0 + 1 : 255;
    const float tx = fx - (float)x0, ty = fy - (float)y0;
    const float* l00 = lut + ((long long)y0 * 256 + x0) * 2;
    const float* l01 = lut + ((long long)y0 * 256 + x1) * 2;
    const float* l10 = lut + ((long long)y1 * 256 + x0) * 2;
    const float* l11 = lut + ((long long)y1 * 256 + x1) * 2;
    const float gx = (fx_raw >= 0.f && fx_raw <= 255.f) ? 256.0f : 0.f;
    const float gy = (fy_raw >= 0.f && fy_raw <= 255.f) ? 256.0f : 0.f;
    float val[2], du[2], dv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float top = l00[c] * (1 - tx) + l01[c] * tx;
        const float bot = l10[c] * (1 - tx) + l11[c] * tx;
        val[c] = top * (1 - ty) + bot * ty;
        du[c] = gx * ((l01[c] - l00[c]) * (1 - ty) + (l11[c] - l10[c]) * ty);
        dv[c] = gy * (bot - top);
    }
    o.A = val[0]; o.B = val[1]; o.dA_du = du[0]; o.dB_du = du[1]; o.dA_dv = dv[0]; o.dB_dv = dv[1];
    return o;
}

// S2 = AppShadingNetwork_S2.forward (field.py:909-1010, the stage-2 surface shading): no refraction light -- the transmitted part
// continues along the refracted ray -- colour = (diffuse + specular) (1 - T) + F light0 T, second output (1 - F) T (the factor the
// running transmittance is multiplied with); `internal` (hit from inside the object): the colour is multiplied by zero.
template <bool BWD, bool S2>
__global__ __launch_bounds__(256) void shade_combine_kernel(
    const float* __restrict__ Mraw, int ldm, const float* __restrict__ OLo, const float* __restrict__ ILo,
    const float* __restrict__ IWo, const float* __restrict__ RLo, const float* __restrict__ SD,
    const float* __restrict__ lut, const int* __restrict__ idx, int P, float exp_max,
    // forward outputs
    float* __restrict__ color_rm, float* __restrict__ aux,  // aux [P,4]: occ_prob, transmission, metallic, roughness
    // backward
    const float* __restrict__ dcolor_rm, float* __restrict__ dMraw, float* __restrict__ dOLo, float* __restrict__ dILo,
    float* __restrict__ dIWo, float* __restrict__ dRLo, float* __restrict__ dNoV,
    // S2 only
    float* __restrict__ rc_out = nullptr, const float* __restrict__ d_rc = nullptr, int internal = 0) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float* mr = Mraw + (long long)p * ldm;
    const float met = nu_sigmoid(mr[0]), rho = nu_sigmoid(mr[1]), T = nu_sigmoid(mr[5]);
    const float alb[3] = {nu_sigmoid(mr[2]), nu_sigmoid(mr[3]), nu_sigmoid(mr[4])};
    const float nov = SD[(long long)p * 8 + 3];
    float Ld[3], dir[3], dir0[3], ind[3], ind0[3], refr[3];
    float rLd[3], rdir[3], rdir0[3], rind[3], rind0[3], rrefr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rLd[c] = OLo[(long long)p * 4 + c];
        rdir[c] = OLo[(long long)(P + p) * 4 + c];
        rdir0[c] = OLo[(long long)(2LL * P + p) * 4 + c];
        rind[c] = ILo[(long long)p * 4 + c];
        rind0[c] = ILo[(long long)(P + p) * 4 + c];
        rrefr[c] = S2 ? 0.f : RLo[(long long)p * 4 + c];
        Ld[c] = expf(fminf(rLd[c], exp_max));
        dir[c] = expf(fminf(rdir[c], exp_max));
        dir0[c] = expf(fminf(rdir0[c], exp_max));
        ind[c] = expf(fminf(rind[c], exp_max));
        ind0[c] = expf(fminf(rind0[c], exp_max));
        refr[c] = expf(fminf(rrefr[c], exp_max));
    }
    const float occ = IWo[p] * 0.5f + 0.5f;
    const float oc = fminf(fmaxf(occ, 0.f), 1.f);
    const float t = fminf(fmaxf(1.0f - nov, 0.f), 1.f);
    const float t2 = t * t, t4 = t2 * t2;
    const float sch = 0.04f + 0.96f * t4 * t;
    const float F = fminf(fmaxf(sch, 0.f), 1.f);
    const float u = fminf(fmaxf(nov, 0.f), 1.f), v = fminf(fmaxf(rho, 0.f), 1.f);
    const NuLut L = nu_lut(lut, u, v);
    float lin[3], light[3], light0[3], sa[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float diffuse = (1.0f - met) * alb[c] * Ld[c];
        sa[c] = 0.04f * (1.0f - met) + met * alb[c];
        light[c] = ind[c] * oc + dir[c] * (1.0f - oc);
        light0[c] = ind0[c] * oc + dir0[c] * (1.0f - oc);
        const float spec = (sa[c] * L.A + L.B) * light[c];
        lin[c] = (diffuse + spec) * (1.0f - T) + (S2 ? F * light0[c] : F * light0[c] + (1.0f - F) * refr[c]) * T;
        if (S2 && internal) lin[c] = lin[c] * 0.f;
    }
    const int k = idx[p];
    if (!BWD) {
#pragma unroll
        for (int c = 0; c < 3; ++c) color_rm[k * 4LL + c] = nu_linear_to_srgb(lin[c]);
        if (S2) rc_out[p] = (1.0f - F) * T;
        if (aux) {
            f32x4 a = {occ, T, met, rho};
            *reinterpret_cast<f32x4*>(aux + (long long)p * 4) = a;
        }
        return;
    }
    float dmet = 0.f, drho = 0.f, dT = 0.f, dalb[3], docc_c = 0.f, dF = 0.f, dA = 0.f, dB = 0.f;
    if (S2 && d_rc) {                    // second output (1 - F) T
        dF -= d_rc[p] * T;
        dT += d_rc[p] * (1.0f - F);
    }
    float* gOL0 = dOLo + (long long)p * 4;
    float* gOL1 = dOLo + (long long)(P + p) * 4;
    float* gOL2 = dOLo + (long long)(2LL * P + p) * 4;
    float* gIL0 = dILo + (long long)p * 4;
    float* gIL1 = dILo + (long long)(P + p) * 4;
    float* gRL = S2 ? nullptr : dRLo + (long long)p * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = (S2 && internal) ? 0.f : dcolor_rm[k * 4LL + c] * nu_linear_to_srgb_grad(lin[c]);
        const float diffuse = (1.0f - met) * alb[c] * Ld[c];
        const float specw = sa[c] * L.A + L.B;
        const float spec = specw * light[c];
        const float mixT = S2 ? F * light0[c] : F * light0[c] + (1.0f - F) * refr[c];
        dT += g * (mixT - (diffuse + spec));
        const float gd = g * (1.0f - T);   // d diffuse, d spec
        const float gm = g * T;            // d mixT
        // diffuse
        dmet += gd * (-alb[c] * Ld[c]);
        dalb[c] = gd * (1.0f - met) * Ld[c];
        const float dLd = gd * (1.0f - met) * alb[c];
        // spec
        const float dspecw = gd * light[c];
        const float dlight = gd * specw;
        const float dsa = dspecw * L.A;
        dA += dspecw * sa[c];
        dB += dspecw;
        dmet += dsa * (alb[c] - 0.04f);
        dalb[c] += dsa * met;
        // mixT
        dF += gm * (S2 ? light0[c] : light0[c] - refr[c]);
        const float dlight0 = gm * F;
        const float drefr = gm * (1.0f - F);
        // light mixes
        docc_c += dlight * (ind[c] - dir[c]) + dlight0 * (ind0[c] - dir0[c]);
        const float dind = dlight * oc, ddir = dlight * (1.0f - oc);
        const float dind0 = dlight0 * oc, ddir0 = dlight0 * (1.0f - oc);
        // exp(min(raw, exp_max))
        gOL0[c] = rLd[c] <= exp_max ? dLd * Ld[c] : 0.f;
        gOL1[c] = rdir[c] <= exp_max ? ddir * dir[c] : 0.f;
        gOL2[c] = rdir0[c] <= exp_max ? ddir0 * dir0[c] : 0.f;
        gIL0[c] = rind[c] <= exp_max ? dind * ind[c] : 0.f;
        gIL1[c] = rind0[c] <= exp_max ? dind0 * ind0[c] : 0.f;
        if (!S2) gRL[c] = rrefr[c] <= exp_max ? drefr * refr[c] : 0.f;
    }
    gOL0[3] = 0.f; gOL1[3] = 0.f; gOL2[3] = 0.f; gIL0[3] = 0.f; gIL1[3] = 0.f;
    if (!S2) gRL[3] = 0.f;
    // occlusion: occ = 0.5 raw + 0.5, clamp passes gradient on [0,1]
    dIWo[p] = (occ >= 0.f && occ <= 1.f) ? docc_c * 0.5f : 0.f;
    // Fresnel: F = clamp(0.04 + 0.96 t^5), t = clamp(1 - NoV)
    float dnov = 0.f;
    if (sch >= 0.f && sch <= 1.f) {
        const float dt = dF * 0.96f * 5.0f * t4;
        if (1.0f - nov >= 0.f && 1.0f - nov <= 1.f) dnov -= dt;
    }
    // LUT
    const float du = dA * L.dA_du + dB * L.dB_du;
    const float dv = dA * L.dA_dv + dB * L.dB_dv;
    if (nov >= 0.f && nov <= 1.f) dnov += du;
    if (rho >= 0.f && rho <= 1.f) drho += dv;
    dNoV[p] = dnov;
    float* gm = dMraw + (long long)p * ldm;
    gm[0] = dmet * met * (1.0f - met);
    gm[1] = drho * rho * (1.0f - rho);
#pragma unroll
    for (int c = 0; c < 3; ++c) gm[2 + c] = dalb[c] * alb[c] * (1.0f - alb[c]);
    gm[5] = dT * T * (1.0f - T);
    for (int c = 6; c < ldm; ++c) gm[c] = 0.f;
}

extern "C" int nu_shade_combine_fwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo,
                                    const float* RLo, const float* SD, const float* lut, const int* idx, int P,
                                    float exp_max, float* color_rm, float* aux, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    hipLaunchKernelGGL((shade_combine_kernel<false, false>), dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, Mraw, ldm, OLo, ILo, IWo,
                       RLo, SD, lut, idx, P, exp_max, color_rm, aux, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (const float*)nullptr, 0);
    return nu_launch_status();
}
extern "C" int nu_shade_combine_bwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo,
                                    const float* RLo, const float* SD, const float* lut, const int* idx, int P,
                                    float exp_max, const float* dcolor_rm, float* dMraw, float* dOLo, float* dILo,
                                    float* dIWo, float* dRLo, float* dNoV, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    hipLaunchKernelGGL((shade_combine_kernel<true, false>), dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, Mraw, ldm, OLo, ILo, IWo,
                       RLo, SD, lut, idx, P, exp_max, (float*)nullptr, (float*)nullptr, dcolor_rm, dMraw, dOLo, dILo, dIWo,
                       dRLo, dNoV, (float*)nullptr, (const float*)nullptr, 0);
    return nu_launch_status();
}

// stage-2 surface shading (AppShadingNetwork_S2): colour [P,4] sRGB (scattered through idx), rc [P] = (1 - F) T
extern "C" int nu_s2_shade_combine_fwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo, const float* SD,
                                       const float* lut, const int* idx, int P, float exp_max, int internal, float* color_rm, float* rc,
                                       hipStream_t stream) {
    if (P <= 0) return NU_OK;
    hipLaunchKernelGGL((shade_combine_kernel<false, true>), dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, Mraw, ldm, OLo, ILo, IWo,
                       (const float*)nullptr, SD, lut, idx, P, exp_max, color_rm, (float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, rc, (const float*)nullptr,
                       internal);
    return nu_launch_status();
}
extern "C" int nu_s2_shade_combine_bwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo, const float* SD,
                                       const float* lut, const int* idx, int P, float exp_max, int internal, const float* dcolor_rm,
                                       const float* d_rc, float* dMraw, float* dOLo, float* dILo, float* dIWo, float* dNoV,
                                       hipStream_t stream) {
    if (P <= 0) return NU_OK;
    hipLaunchKernelGGL((shade_combine_kernel<true, true>), dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, Mraw, ldm, OLo, ILo, IWo,
                       (const float*)nullptr, SD, lut, idx, P, exp_max, (float*)nullptr, (float*)nullptr, dcolor_rm, dMraw, dOLo, dILo,
                       dIWo, (float*)nullptr, dNoV, (float*)nullptr, d_rc, internal);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Composite: one wavefront per ray, S <= 256 samples held as contiguous chunks of CH = ceil(S/64) per lane.
//   T_j = prod_{i<j} (1 - a_i + 1e-7);  w_j = a_j T_j;  rgb = sum w_j c_j;  acc = sum w_j
//   background-only composite: a_bg = a * (1 - inner)  (renderer_zerothick.py:757, :777-779)
// ------------------------------------------------------------------------------------------------
static __device__ inline float nu_wave_excl_prod(float v, int lane) {
    float inc = nu_wave_incl_prod(v, lane);
    float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : ex;
}
static __device__ inline float nu_wave_excl_suffix_sum(float v, int lane) {
    // sum over lanes > lane
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(inc, o, 64);
        if (lane + o < 64) inc += t;
    }
    return inc - v;
}

template <int CH>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ alpha, const float* __restrict__ color,
                                                            const unsigned char* __restrict__ inner, int R, int S,
                                                            float* __restrict__ weights, float* __restrict__ rgb,
                                                            float* __restrict__ acc, float* __restrict__ rgb_bg,
                                                            float* __restrict__ aux_sum) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    const long long base = (long long)r * S;
    float a[CH], abg[CH];
    float pl = 1.f, plb = 1.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int j = lane * CH + i;
        a[i] = j < S ? alpha[base + j] : 0.f;
        abg[i] = (j < S && !inner[base + j]) ? a[i] : 0.f;
        if (j < S) { pl *= (1.0f - a[i] + 1e-7f); plb *= (1.0f - abg[i] + 1e-7f); }
    }
    float T = nu_wave_excl_prod(pl, lane), Tb = nu_wave_excl_prod(plb, lane);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f}, sacc = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int j = lane * CH + i;
        if (j < S) {
            const float w = a[i] * T, wb = abg[i] * Tb;
            if (weights) weights[base + j] = w;
            const f32x4 c = *reinterpret_cast<const f32x4*>(color + (base + j) * 4);   // rgb + normal-orientation term
            s[0] += w * c[0]; s[1] += w * c[1]; s[2] += w * c[2]; s[3] += w * c[3];
            sb[0] += wb * c[0]; sb[1] += wb * c[1]; sb[2] += wb * c[2];
            sacc += w;
            T *= (1.0f - a[i] + 1e-7f);
            Tb *= (1.0f - abg[i] + 1e-7f);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = nu_wave_sum(s[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) sb[c] = nu_wave_sum(sb[c]);
    sacc = nu_wave_sum(sacc);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { rgb[r * 3LL + c] = s[c]; rgb_bg[r * 3LL + c] = sb[c]; }
        acc[r] = sacc;
        if (aux_sum) aux_sum[r] = s[3];
    }
}

template <int CH>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ color,
                                                            const unsigned char* __restrict__ inner, int R, int S,
                                                            const float* __restrict__ drgb, const float* __restrict__ dacc,
                                                            const float* __restrict__ drgb_bg, const float* __restrict__ daux,
                                                            float* __restrict__ dalpha, float* __restrict__ dcolor) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    const long long base = (long long)r * S;
    const float g[4] = {drgb[r * 3LL], drgb[r * 3LL + 1], drgb[r * 3LL + 2], daux ? daux[r] : 0.f};
    const float gb[3] = {drgb_bg ? drgb_bg[r * 3LL] : 0.f, drgb_bg ? drgb_bg[r * 3LL + 1] : 0.f,
                         drgb_bg ? drgb_bg[r * 3LL + 2] : 0.f};
    const float ga = dacc ? dacc[r] : 0.f;
    float a[CH], abg[CH], gw[CH], gwb[CH];
    bool out_s[CH];
    float pl = 1.f, plb = 1.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int j = lane * CH + i;
        a[i] = j < S ? alpha[base + j] : 0.f;
        out_s[i] = (j < S) && !inner[base + j];
        abg[i] = out_s[i] ? a[i] : 0.f;
        if (j < S) { pl *= (1.0f - a[i] + 1e-7f); plb *= (1.0f - abg[i] + 1e-7f); }
    }
    float T = nu_wave_excl_prod(pl, lane), Tb = nu_wave_excl_prod(plb, lane);
    float Tj[CH], Tbj[CH];
    float ls = 0.f, lsb = 0.f;  // lane sums of gw*w
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int j = lane * CH + i;
        Tj[i] = T; Tbj[i] = Tb;
        gw[i] = 0.f; gwb[i] = 0.f;
        if (j < S) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(color + (base + j) * 4);
            gw[i] = g[0] * c[0] + g[1] * c[1] + g[2] * c[2] + g[3] * c[3] + ga;
            gwb[i] = gb[0] * c[0] + gb[1] * c[1] + gb[2] * c[2];
            const float w = a[i] * T, wb = abg[i] * Tb;
            f32x4 dc = {w * g[0] + wb * gb[0], w * g[1] + wb * gb[1], w * g[2] + wb * gb[2], w * g[3]};
            *reinterpret_cast<f32x4*>(dcolor + (base + j) * 4) = dc;
            ls += gw[i] * w;
            lsb += gwb[i] * wb;
            T *= (1.0f - a[i] + 1e-7f);
            Tb *= (1.0f - abg[i] + 1e-7f);
        }
    }
    // suffix sums: sum over samples AFTER j of gw*w
    float suf = nu_wave_excl_suffix_sum(ls, lane), sufb = nu_wave_excl_suffix_sum(lsb, lane);
#pragma unroll
    for (int i = CH - 1; i >= 0; --i) {
        const int j = lane * CH + i;
        if (j < S) {
            float da = Tj[i] * gw[i] - suf / (1.0f - a[i] + 1e-7f);
            if (out_s[i]) da += Tbj[i] * gwb[i] - sufb / (1.0f - abg[i] + 1e-7f);
            dalpha[base + j] = da;
            suf += gw[i] * a[i] * Tj[i];
            sufb += gwb[i] * abg[i] * Tbj[i];
        }
    }
}

// colour is [R*S, 4]: rgb + one auxiliary channel composited into aux_sum (loss_normal of network/renderer.py:705)
extern "C" int nu_composite_fwd(const float* alpha, const float* color, const unsigned char* inner, int R, int S,
                                float* weights, float* rgb, float* acc, float* rgb_bg, float* aux_sum, hipStream_t stream) {
    if (R <= 0 || S <= 0 || S > 64 * NU_MAXCHUNK) return NU_ERR_ARG;
    dim3 grid(nu_cdiv(R, 4)), block(256);
    const int ch = nu_cdiv(S, 64);
#define NU_CASE(c) case c: hipLaunchKernelGGL(composite_fwd_kernel<c>, grid, block, 0, stream, alpha, color, inner, R, S, weights, rgb, acc, rgb_bg, aux_sum); break;
    switch (ch) { NU_CASE(1) NU_CASE(2) NU_CASE(3) NU_CASE(4) default: return NU_ERR_ARG; }
#undef NU_CASE
    return nu_launch_status();
}
extern "C" int nu_composite_bwd(const float* alpha, const float* color, const unsigned char* inner, int R, int S,
                                const float* drgb, const float* dacc, const float* drgb_bg, const float* daux,
                                float* dalpha, float* dcolor, hipStream_t stream) {
    if (R <= 0 || S <= 0 || S > 64 * NU_MAXCHUNK) return NU_ERR_ARG;
    dim3 grid(nu_cdiv(R, 4)), block(256);
    const int ch = nu_cdiv(S, 64);
#define NU_CASE(c) case c: hipLaunchKernelGGL(composite_bwd_kernel<c>, grid, block, 0, stream, alpha, color, inner, R, S, drgb, dacc, drgb_bg, daux, dalpha, dcolor); break;
    switch (ch) { NU_CASE(1) NU_CASE(2) NU_CASE(3) NU_CASE(4) default: return NU_ERR_ARG; }
#undef NU_CASE
    return nu_launch_status();
}
