// sampler.hip -- hierarchical per-ray sampler (no-grad), one wavefront per ray, per-ray buffers in LDS.
//
// Reference semantics restated (paths relative to /root/reference):
//   sample_ray   network/renderer_zerothick.py:572-612   (coarse + background z, S up-sampling rounds)
//   upsample     network/renderer_zerothick.py:525-554   (alpha from finite-difference cosine, weights)
//   sample_pdf   network/field.py:468-498                (deterministic inverse-CDF, searchsorted right)
//   cat_z_vals   network/renderer_zerothick.py:556-570   (merge by sort + SDF gather)
#include "nu_common.h"

// sample placement should round like the reference's eager mul/add chains: no FMA contraction in this file
#pragma clang fp contract(off)

#define NU_SMAX 256  // max samples per ray held in LDS

// coarse z:  z[r,j] = near + (far-near)*lin[j] + (u1[r]-0.5)*2/Nc   (u1 == null => no jitter)
// bg z:      zo = lower[k] + (upper[k]-lower[k])*u2[r,k]  (or lin_bg[k] without jitter); out[Nbg-1-k] = far/zo + 1/Nbg
// also emits the coarse sample positions X[r*Nc + j] = o + d*z
__global__ __launch_bounds__(256) void sample_coarse_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                            const float* __restrict__ near, const float* __restrict__ far,
                                                            const float* __restrict__ lin, const float* __restrict__ lower,
                                                            const float* __restrict__ upper, const float* __restrict__ u1,
                                                            const float* __restrict__ u2, int R, int Nc, int Nbg,
                                                            float* __restrict__ zc, float* __restrict__ zbg,
                                                            float* __restrict__ X) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int tot = Nc + Nbg;
    if (i >= (long long)R * tot) return;
    const int r = (int)(i / tot), j = (int)(i - (long long)r * tot);
    const float nr = near[r], fr = far[r];
    if (j < Nc) {
        const float span = fr - nr;
        const float sc = span * lin[j];
        float z = nr + sc;
        if (u1) {
            const float tr = (u1[r] - 0.5f) * 2.0f;
            const float jit = tr / (float)Nc;
            z = z + jit;
        }
        zc[(long long)r * Nc + j] = z;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float m = d[r * 3LL + c] * z;
            X[((long long)r * Nc + j) * 3 + c] = o[r * 3LL + c] + m;
        }
    } else {
        const int k = j - Nc;
        float zo = lower[k];
        if (u2) {
            const float w = (upper[k] - lower[k]) * u2[(long long)r * Nbg + k];
            zo = lower[k] + w;
        }
        const float q = fr / zo;
        zbg[(long long)r * Nbg + (Nbg - 1 - k)] = q + 1.0f / (float)Nbg;
    }
}
extern "C" int nu_sample_coarse(const float* o, const float* d, const float* near, const float* far, const float* lin,
                                const float* lower, const float* upper, const float* u1, const float* u2, int R, int Nc,
                                int Nbg, float* zc, float* zbg, float* X, hipStream_t stream) {
    if (R <= 0) return NU_ERR_ARG;
    const long long n = (long long)R * (Nc + Nbg);
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)nu_cdivl(n, 256)), dim3(256), 0, stream, o, d, near, far, lin,
                       lower, upper, u1, u2, R, Nc, Nbg, zc, zbg, X);
    return nu_launch_status();
}

// One up-sampling round.  inv_s = min(exp(10*variance), inv_s_cap) (clip_sample_variance) or the cap itself.
// Outputs z_new[R, n_new] (ascending) and the new sample positions Xn[R*n_new, 3].
// MODE 0: NeuS `upsample` (cos = min(prev_cos, cos) clipped to [-1e3, 0], zeroed outside the unit sphere).
// MODE 1: `get_weights` of the occlusion probe (field.py:501-521): cos = min(cos, 0), alpha masked by cos < 0,
//         inv_s = exp(10*variance) uncapped; optionally writes wsum[r] = sum of the weights (the hit probability).
template <int MODE>
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                       const float* __restrict__ z, const float* __restrict__ sdf, int R,
                                                       int sn, const float* __restrict__ variance, float inv_s_cap,
                                                       int use_variance, const float* __restrict__ uvals, int n_new,
                                                       float* __restrict__ z_new, float* __restrict__ Xn,
                                                       float* __restrict__ wsum) {
    __shared__ float s_z[4][NU_SMAX], s_s[4][NU_SMAX], s_r[4][NU_SMAX], s_c[4][NU_SMAX + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    const bool live = r < R;
    float oo[3] = {0, 0, 0}, dd[3] = {0, 0, 0};
    if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { oo[c] = o[r * 3LL + c]; dd[c] = d[r * 3LL + c]; }
        for (int j = lane; j < sn; j += 64) {
            const float zz = z[(long long)r * sn + j];
            s_z[w][j] = zz;
            s_s[w][j] = sdf[(long long)r * sn + j];
            const float x0 = oo[0] + dd[0] * zz, x1 = oo[1] + dd[1] * zz, x2 = oo[2] + dd[2] * zz;
            s_r[w][j] = sqrtf((x0 * x0 + x1 * x1) + x2 * x2);
        }
    }
    __syncthreads();
    // non-live waves run the arithmetic on zeros (no global access) so that every wave reaches the barrier below
    float inv_s = inv_s_cap;
    if (use_variance) inv_s = fminf(expf(variance[0] * 10.0f), inv_s_cap);
    if (MODE == 1) inv_s = expf(variance[0] * 10.0f);
    const int ni = sn - 1;                 // intervals
    const int CH = (ni + 63) / 64;         // contiguous intervals per lane (<= 4)
    float al[4], wl[4];
    float pl = 1.f;
    for (int i = 0; i < CH; ++i) {
        const int m = lane * CH + i;
        al[i] = 0.f;
        if (live && m < ni) {
            const float s0 = s_s[w][m], s1 = s_s[w][m + 1], z0 = s_z[w][m], z1 = s_z[w][m + 1];
            const float mid = (s0 + s1) * 0.5f;
            float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
            bool surf = true;
            if (MODE == 0) {
                float prev = 0.f;
                if (m > 0) prev = (s0 - s_s[w][m - 1]) / (z0 - s_z[w][m - 1] + 1e-5f);
                cosv = fminf(prev, cosv);
                cosv = fminf(fmaxf(cosv, -1e3f), 0.f);
                const bool inside = (s_r[w][m] < 1.0f) || (s_r[w][m + 1] < 1.0f);
                cosv = inside ? cosv : 0.f;
            } else {
                surf = cosv < 0.f;
                cosv = fminf(cosv, 0.f);
            }
            const float dz = z1 - z0;
            const float pe = mid - cosv * dz * 0.5f, ne = mid + cosv * dz * 0.5f;
            const float pc = nu_sigmoid(pe * inv_s), nc = nu_sigmoid(ne * inv_s);
            al[i] = surf ? (pc - nc + 1e-5f) / (pc + 1e-5f) : 0.f;
            pl *= (1.0f - al[i] + 1e-7f);
        }
    }
    // exclusive transmittance, weights, their sum
    float inc = nu_wave_incl_prod(pl, lane);
    float T = __shfl_up(inc, 1, 64);
    if (lane == 0) T = 1.f;
    float lsum = 0.f, lraw = 0.f;
    for (int i = 0; i < CH; ++i) {
        const int m = lane * CH + i;
        wl[i] = 0.f;
        if (m < ni) {
            lraw += al[i] * T;
            wl[i] = al[i] * T + 1e-5f;  // sample_pdf adds 1e-5 to every weight
            lsum += wl[i];
            T *= (1.0f - al[i] + 1e-7f);
        }
    }
    const float total = nu_wave_sum(lsum);
    if (wsum) {
        const float raw = nu_wave_sum(lraw);
        if (live && lane == 0) wsum[r] = raw;
    }
    // cdf: cdf[0] = 0, cdf[m+1] = sum_{i<=m} w_i/total
    float lp = 0.f;
    for (int i = 0; i < CH; ++i) {
        const int m = lane * CH + i;
        if (m < ni) { wl[i] = wl[i] / total; lp += wl[i]; }
    }
    float run = nu_wave_incl_sum(lp, lane) - lp;
    if (lane == 0) s_c[w][0] = 0.f;
    for (int i = 0; i < CH; ++i) {
        const int m = lane * CH + i;
        if (m < ni) { run += wl[i]; s_c[w][m + 1] = run; }
    }
    __syncthreads();
    if (!live) return;
    // inverse CDF for n_new deterministic u's; cdf has sn entries, bins = z (sn entries)
    for (int k = lane; k < n_new; k += 64) {
        const float u = uvals[k];
        int lo = 0, hi = sn;  // first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_c[w][mid] > u) hi = mid; else lo = mid + 1;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < sn - 1 ? lo : sn - 1;
        const float c0 = s_c[w][below], c1 = s_c[w][above];
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.0f;
        const float t = (u - c0) / den;
        const float zz = s_z[w][below] + t * (s_z[w][above] - s_z[w][below]);
        z_new[(long long)r * n_new + k] = zz;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Xn[((long long)r * n_new + k) * 3 + c] = oo[c] + dd[c] * zz;
    }
}
extern "C" int nu_upsample(const float* o, const float* d, const float* z, const float* sdf, int R, int sn,
                           const float* variance, float inv_s_cap, int use_variance, const float* uvals, int n_new,
                           float* z_new, float* Xn, hipStream_t stream) {
    if (R <= 0 || sn < 2 || sn > NU_SMAX) return NU_ERR_ARG;
    hipLaunchKernelGGL(upsample_kernel<0>, dim3(nu_cdiv(R, 4)), dim3(256), 0, stream, o, d, z, sdf, R, sn, variance,
                       inv_s_cap, use_variance, uvals, n_new, z_new, Xn, (float*)nullptr);
    return nu_launch_status();
}
// Occlusion-probe weights along secondary rays (get_weights + optional deterministic resampling)
extern "C" int nu_probe_weights(const float* o, const float* d, const float* z, const float* sdf, int R, int sn,
                                const float* variance, const float* uvals, int n_new, float* z_new, float* Xn,
                                float* wsum, hipStream_t stream) {
    if (R <= 0 || sn < 2 || sn > NU_SMAX) return NU_ERR_ARG;
    hipLaunchKernelGGL(upsample_kernel<1>, dim3(nu_cdiv(R, 4)), dim3(256), 0, stream, o, d, z, sdf, R, sn, variance, 0.f, 0,
                       uvals, n_new, z_new, Xn, wsum);
    return nu_launch_status();
}

// merge two ascending lists per ray (old before new on ties), carrying the SDF values along when given
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ z, const float* __restrict__ sdf, int sn,
                                                    const float* __restrict__ zn, const float* __restrict__ sdfn, int nn,
                                                    int R, float* __restrict__ zo, float* __restrict__ sdfo) {
    __shared__ float s_z[4][NU_SMAX], s_n[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    const bool live = r < R;
    if (live) {
        for (int j = lane; j < sn; j += 64) s_z[w][j] = z[(long long)r * sn + j];
        for (int j = lane; j < nn; j += 64) s_n[w][j] = zn[(long long)r * nn + j];
    }
    __syncthreads();
    if (!live) return;
    const int so = sn + nn;
    for (int j = lane; j < sn; j += 64) {
        const float v = s_z[w][j];
        int cnt = 0;
        for (int k = 0; k < nn; ++k) cnt += s_n[w][k] < v ? 1 : 0;
        zo[(long long)r * so + j + cnt] = v;
        if (sdfo) sdfo[(long long)r * so + j + cnt] = sdf[(long long)r * sn + j];
    }
    for (int k = lane; k < nn; k += 64) {
        const float v = s_n[w][k];
        int lo = 0, hi = sn;  // count of old <= v
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_z[w][mid] <= v) lo = mid + 1; else hi = mid;
        }
        // z_new is non-decreasing, so its own index k already orders equal new values
        zo[(long long)r * so + k + lo] = v;
        if (sdfo) sdfo[(long long)r * so + k + lo] = sdfn[(long long)r * nn + k];
    }
}
extern "C" int nu_merge_sorted(const float* z, const float* sdf, int sn, const float* zn, const float* sdfn, int nn, int R,
                               float* zo, float* sdfo, hipStream_t stream) {
    if (R <= 0 || sn > NU_SMAX || nn > 64) return NU_ERR_ARG;
    hipLaunchKernelGGL(merge_kernel, dim3(nu_cdiv(R, 4)), dim3(256), 0, stream, z, sdf, sn, zn, sdfn, nn, R, zo, sdfo);
    return nu_launch_status();
}

// concat [R, a] and [R, b] -> [R, a+b]
__global__ void concat_cols_kernel(const float* __restrict__ A, int a, const float* __restrict__ B, int b, int R,
                                   float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = a + b;
    if (i >= (long long)R * t) return;
    const int r = (int)(i / t), j = (int)(i - (long long)r * t);
    out[i] = j < a ? A[(long long)r * a + j] : B[(long long)r * b + (j - a)];
}
extern "C" int nu_concat_cols(const float* A, int a, const float* B, int b, int R, float* out, hipStream_t stream) {
    const long long n = (long long)R * (a + b);
    if (n <= 0) return NU_ERR_ARG;
    hipLaunchKernelGGL(concat_cols_kernel, dim3((unsigned)nu_cdivl(n, 256)), dim3(256), 0, stream, A, a, B, b, R, out);
    return nu_launch_status();
}
