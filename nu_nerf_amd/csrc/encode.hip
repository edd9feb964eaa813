// encode.hip -- input encodings of the three MLP stacks, one wavefront per point, lane = output column
// so every row write is a contiguous burst.
//
// Reference semantics restated:
//   positional encoding [x, sin(2^k x), cos(2^k x)]_k            network/field.py:14-61
//   integrated directional encoding (72-d)                       utils/ref_utils.py:84-114
//   SDF input-gradient through the embedding (J_emb^T)           network/field.py:158-170 (autograd)
//   shading directions n^, v^, r = 2(v.n)n - v, NoV              network/field.py:686-689
//   NeRF++ inputs (x/|x|, 1/|x|), view = -d                      network/renderer_zerothick.py:687-690
#include "nu_common.h"
#include "ide_table.h"

// Per-point record written by the partition kernel (render.hip): 8 floats
//   [0..2] position x, [3] dist (section length), [4..6] unit ray direction d, [7] unused
#define NU_PT 8

// column -> (is_raw, k, coord, is_cos) for an L-frequency embedding of a D-vector:
// col < D : x[col];  else q = col - D, k = q / (2D), r = q % (2D), c = r % D, cos if r >= D
static __device__ inline float nu_embed_col(const float* x, int D, int col) {
    if (col < D) return x[col];
    const int q = col - D;
    const int k = q / (2 * D);
    const int r = q - k * 2 * D;
    const int c = r >= D ? r - D : r;
    const float a = x[c] * (float)(1 << k);
    return r >= D ? cosf(a) : sinf(a);
}

// ------------------------------------------------------------------------------------------------
// SDF inputs:  E[p, 0:39] (+ zero pad to 64),  U4[p, 217:256] = embedding,  YX[p, 257:260] = x, rest of YX pad = 0
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sdf_embed_kernel(const float* __restrict__ pt, int pt_ld, int P,
                                                        float* __restrict__ E, float* __restrict__ U4,
                                                        float* __restrict__ YX) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < P; p += nwave) {
        float x[3];
        x[0] = pt[(long long)p * pt_ld + 0];
        x[1] = pt[(long long)p * pt_ld + 1];
        x[2] = pt[(long long)p * pt_ld + 2];
        const float v = lane < 39 ? nu_embed_col(x, 3, lane) : 0.f;
        E[(long long)p * 64 + lane] = v;
        if (U4 && lane < 39) U4[(long long)p * 256 + 217 + lane] = v;
        if (YX && lane < 31) YX[(long long)p * 288 + 257 + lane] = lane < 3 ? x[lane] : 0.f;
    }
}

extern "C" int nu_sdf_embed(const float* pt, int pt_ld, int P, float* E, float* U4, float* YX, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(sdf_embed_kernel, dim3(blocks), dim3(256), 0, stream, pt, pt_ld, P, E, U4, YX);
    return nu_launch_status();
}

// n[p, c] = sum_col J[col][c] * (G0[p,col] + Gs[p,col]),  J = d emb / d x evaluated from the stored embedding E.
__global__ __launch_bounds__(256) void embed_jt_kernel(const float* __restrict__ E, const float* __restrict__ G0, int ldg0,
                                                       const float* __restrict__ Gs, int ldgs, int P,
                                                       float* __restrict__ n) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < P; p += nwave) {
        float t = 0.f;
        int c = 0;
        if (lane < 39) {
            float g = G0[(long long)p * ldg0 + lane];
            if (Gs) g += Gs[(long long)p * ldgs + lane];
            if (lane < 3) {
                c = lane;
                t = g;
            } else {
                const int q = lane - 3;
                const int k = q / 6;
                const int r = q - 6 * k;
                c = r >= 3 ? r - 3 : r;
                const float f = (float)(1 << k);
                // d sin(f x)/dx = f cos(f x) (cos sits 3 columns later); d cos(f x)/dx = -f sin(f x)
                const float other = r >= 3 ? -E[(long long)p * 64 + lane - 3] : E[(long long)p * 64 + lane + 3];
                t = g * f * other;
            }
        }
        const float s0 = nu_wave_sum(c == 0 ? t : 0.f);
        const float s1 = nu_wave_sum(c == 1 ? t : 0.f);
        const float s2 = nu_wave_sum(c == 2 ? t : 0.f);
        if (lane < 3) n[(long long)p * 3 + lane] = lane == 0 ? s0 : (lane == 1 ? s1 : s2);
    }
}
extern "C" int nu_embed_jt(const float* E, const float* G0, int ldg0, const float* Gs, int ldgs, int P, float* n,
                           hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(embed_jt_kernel, dim3(blocks), dim3(256), 0, stream, E, G0, ldg0, Gs, ldgs, P, n);
    return nu_launch_status();
}

// q0[p, col] = sum_c J[col][c] * nbar[p, c]  -> Q0[p, 0:39] (pad zero to 64) and Q4[p, 217:256]
__global__ __launch_bounds__(256) void embed_j_kernel(const float* __restrict__ E, const float* __restrict__ nbar, int P,
                                                      float* __restrict__ Q0, float* __restrict__ Q4) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < P; p += nwave) {
        float v = 0.f;
        if (lane < 39) {
            if (lane < 3) {
                v = nbar[(long long)p * 3 + lane];
            } else {
                const int q = lane - 3;
                const int k = q / 6;
                const int r = q - 6 * k;
                const int c = r >= 3 ? r - 3 : r;
                const float f = (float)(1 << k);
                const float other = r >= 3 ? -E[(long long)p * 64 + lane - 3] : E[(long long)p * 64 + lane + 3];
                v = f * other * nbar[(long long)p * 3 + c];
            }
        }
        Q0[(long long)p * 64 + lane] = v;
        if (lane < 39) Q4[(long long)p * 256 + 217 + lane] = v;
    }
}
extern "C" int nu_embed_j(const float* E, const float* nbar, int P, float* Q0, float* Q4, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(embed_j_kernel, dim3(blocks), dim3(256), 0, stream, E, nbar, P, Q0, Q4);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// IDE: lane i < 36 evaluates term i.  Returns (re, im) and optionally the partials.
// ------------------------------------------------------------------------------------------------
// Per-lane constants of term i = lane (loaded once per kernel: the Horner loop then runs out of registers with a uniform
// trip count -- coefficients above degree l - m are zero in the table, and leading zeros leave Horner's arithmetic unchanged)
struct NuIdeLane {
    float c[NU_IDE_DEG];
    float sigma, att1;     // l(l+1)/2 and exp(-sigma) (kappa_inv = 1: the diffuse query)
    int m;
    bool live;
};
static __device__ inline NuIdeLane nu_ide_lane(int lane) {
    NuIdeLane L;
    const int i = lane < NU_IDE_TERMS ? lane : NU_IDE_TERMS - 1;
#pragma unroll
    for (int k = 0; k < NU_IDE_DEG; ++k) L.c[k] = c_ide_mat[i][k];
    const int l = c_ide_l[i];
    L.m = c_ide_m[i];
    L.sigma = 0.5f * (float)l * (float)(l + 1);
    L.att1 = expf(-L.sigma);
    L.live = lane < NU_IDE_TERMS;
    return L;
}
// Direction-dependent factors of one term, shared by every kappa the direction is queried with
struct NuIdeDir {
    float are, aim;        // (x+iy)^m
    float pre, pim;        // (x+iy)^(m-1)   (0 for m = 0)
    float poly, dpoly;     // P(z), P'(z)
};
static __device__ inline NuIdeDir nu_ide_dir(const NuIdeLane& L, float x, float y, float z) {
    NuIdeDir t;
    float ar = 1.f, ai = 0.f, pr = 0.f, pi = 0.f;
#pragma unroll
    for (int j = 0; j < NU_IDE_DEG - 1; ++j) {
        if (j < L.m) {
            pr = ar; pi = ai;
            const float nr = ar * x - ai * y;
            const float ni = ar * y + ai * x;
            ar = nr; ai = ni;
        }
    }
    float poly = 0.f, dpoly = 0.f;
#pragma unroll
    for (int k = NU_IDE_DEG - 1; k >= 0; --k) {
        dpoly = dpoly * z + poly;
        poly = poly * z + L.c[k];
    }
    t.are = ar; t.aim = ai; t.pre = pr; t.pim = pi; t.poly = poly; t.dpoly = dpoly;
    return t;
}
static __device__ inline float nu_ide_att(const NuIdeLane& L, float kinv) { return expf(-L.sigma * kinv); }

// write one 72-d IDE row (+ zero pad up to `padto` columns) given the direction factors and the attenuation
static __device__ inline void nu_ide_store(float* row, int lane, const NuIdeLane& L, const NuIdeDir& t, float att, int padto) {
    if (L.live) {
        row[lane] = t.are * t.poly * att;
        row[36 + lane] = t.aim * t.poly * att;
    }
    for (int c = 72 + lane; c < padto; c += 64) row[c] = 0.f;
}
// per-lane partials of d/d(x, y, z) of sum_i (gre_i re_i + gim_i im_i) for gradient weights already scaled by the
// attenuation (gre, gim = sum over rows of att_row * g_row)
static __device__ inline void nu_ide_dir_grad(const NuIdeLane& L, const NuIdeDir& t, float gre, float gim, float& ax, float& ay,
                                              float& az) {
    const float cr = (float)L.m * t.poly * (gre * t.pre + gim * t.pim);
    const float ci = (float)L.m * t.poly * (gre * t.pim - gim * t.pre);
    ax = cr;
    ay = -ci;
    az = (gre * t.are + gim * t.aim) * t.dpoly;
}

__global__ __launch_bounds__(256) void ide_kernel(const float* __restrict__ dirs, const float* __restrict__ kinv, int P,
                                                  float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const NuIdeLane L = nu_ide_lane(lane);
    for (int p = wave; p < P; p += nwave) {
        const NuIdeDir t = nu_ide_dir(L, dirs[p * 3LL], dirs[p * 3LL + 1], dirs[p * 3LL + 2]);
        nu_ide_store(out + (long long)p * ldo, lane, L, t, nu_ide_att(L, kinv ? kinv[p] : 0.f), ldo);
    }
}
// stand-alone IDE (test entry + the per-ray colour_spec query IDE(d, 0), renderer_zerothick.py:780)
extern "C" int nu_ide(const float* dirs, const float* kappa_inv, int P, float* out, int ldo, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    if (ldo < 72) return NU_ERR_ARG;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(ide_kernel, dim3(blocks), dim3(256), 0, stream, dirs, kappa_inv, P, out, ldo);
    return nu_launch_status();
}

__global__ __launch_bounds__(256) void ide_bwd_kernel(const float* __restrict__ dirs, const float* __restrict__ kinv,
                                                      const float* __restrict__ gout, int ldg, int P,
                                                      float* __restrict__ ddirs, float* __restrict__ dk) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const NuIdeLane L = nu_ide_lane(lane);
    for (int p = wave; p < P; p += nwave) {
        const NuIdeDir t = nu_ide_dir(L, dirs[p * 3LL], dirs[p * 3LL + 1], dirs[p * 3LL + 2]);
        const float att = nu_ide_att(L, kinv ? kinv[p] : 0.f);
        const float* grow = gout + (long long)p * ldg;
        const float gre = L.live ? grow[lane] * att : 0.f, gim = L.live ? grow[36 + lane] * att : 0.f;
        float ax, ay, az;
        nu_ide_dir_grad(L, t, gre, gim, ax, ay, az);
        const float dx = nu_wave_sum(ax), dy = nu_wave_sum(ay), dz = nu_wave_sum(az);
        const float dkk = nu_wave_sum(-L.sigma * t.poly * (gre * t.are + gim * t.aim));
        if (lane == 0) {
            ddirs[p * 3LL] = dx; ddirs[p * 3LL + 1] = dy; ddirs[p * 3LL + 2] = dz;
            if (dk) dk[p] = dkk;
        }
    }
}
extern "C" int nu_ide_bwd(const float* dirs, const float* kappa_inv, const float* gout, int ldg, int P, float* ddirs,
                          float* dkappa, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(ide_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dirs, kappa_inv, gout, ldg, P, ddirs, dkappa);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Shading-stack inputs (P inner points).  Buffers (row-major, zero padded):
//   OLin [3P, 96] : IDE(n^,1) | IDE(r,rho) | IDE(r,0)                    -> outer_light
//   ILin [2P,128] : [E(39), IDE(r,rho)(72)] | [E(39), IDE(r,0)(72)]      -> inner_light
//   IWin [P, 96]  : [E(39), embed(r,6)(39)]                              -> inner_weight (inputs detached)
//   RLin [P, 96]  : [E(39), embed(v^,6)(39)]                             -> refrac_light
//   SD   [P, 8]   : n^(3), NoV, 1/|n|, rho, 0, 0
// ------------------------------------------------------------------------------------------------
static __device__ inline void nu_shade_dirs(const float* n, const float* d, float* nh, float* vh, float* r, float& nov,
                                            float& inv_norm) {
    const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    inv_norm = 1.0f / fmaxf(nn, 1e-12f);
    const float vn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float ivn = 1.0f / fmaxf(vn, 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; ++c) { nh[c] = n[c] * inv_norm; vh[c] = -d[c] * ivn; }
    nov = nh[0] * vh[0] + nh[1] * vh[1] + nh[2] * vh[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = nov * nh[c] * 2.0f - vh[c];
}

// Point on the unit sphere seen from p (inside, moved to radius 0.999 when outside) along `dir`
// (offset_points_to_sphere + get_sphere_intersection + normalize; field.py:447-464, :642-643, :676-677)
struct NuSph {
    float s[3];       // unit point on the sphere
    float pp[3];      // offset origin p'
    float t, root, snorm;
};
static __device__ inline NuSph nu_sph_point(const float* x, const float* dir) {
    NuSph o;
    const float xn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const float sc = xn > 0.999f ? 0.999f / xn : 1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) o.pp[c] = xn > 0.999f ? x[c] / xn * 0.999f : x[c];
    (void)sc;
    const float b = o.pp[0] * dir[0] + o.pp[1] * dir[1] + o.pp[2] * dir[2];
    const float cc = o.pp[0] * o.pp[0] + o.pp[1] * o.pp[1] + o.pp[2] * o.pp[2];
    o.root = sqrtf(b * b - cc + 1.0f + 1e-6f);
    o.t = -b + o.root;
    float sv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) sv[c] = o.pp[c] + dir[c] * o.t;
    o.snorm = fmaxf(sqrtf(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]), 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; ++c) o.s[c] = sv[c] / o.snorm;
    return o;
}
// cotangent of the unit sphere point -> cotangent of dir
static __device__ inline void nu_sph_point_bwd(const NuSph& o, const float* dir, const float* ds_unit, float* ddir) {
    const float dotp = ds_unit[0] * o.s[0] + ds_unit[1] * o.s[1] + ds_unit[2] * o.s[2];
    float ds[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ds[c] = (ds_unit[c] - o.s[c] * dotp) / o.snorm;
    const float b = o.pp[0] * dir[0] + o.pp[1] * dir[1] + o.pp[2] * dir[2];
    const float dt_db = -1.0f + b / o.root;
    const float dsd = ds[0] * dir[0] + ds[1] * dir[1] + ds[2] * dir[2];   // d L / d t
#pragma unroll
    for (int c = 0; c < 3; ++c) ddir[c] += o.t * ds[c] + dsd * dt_db * o.pp[c];
}

// sphere = 0: OLin rows are [IDE(dir, k) | 0] (ld 96).  sphere = 1 (shader_config.sphere_direction): rows are
// [IDE(dir, k) | IDE(sph(x, dir), k2) | 0] (ld 160) with k2 = 1 for the diffuse row and rho for both specular rows.
// rdim = 3 + 6 * refrac_freq columns per half of RLin (prefix of the L=6 embedding), ld_rl = padded row length.
__global__ __launch_bounds__(256) void shade_encode_fwd_kernel(const float* __restrict__ nrm, const float* __restrict__ pt,
                                                               int pt_ld, const float* __restrict__ E,
                                                               const float* __restrict__ Mraw, int ldm, int P, int sphere,
                                                               int ld_ol, int rdim, int ld_rl,
                                                               float* __restrict__ OLin, float* __restrict__ ILin,
                                                               float* __restrict__ IWin, float* __restrict__ RLin,
                                                               float* __restrict__ SD) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const NuIdeLane IL = nu_ide_lane(lane);
    for (int p = wave; p < P; p += nwave) {
        float n[3], d[3], x[3], nh[3], vh[3], r[3], nov, inorm;
#pragma unroll
        for (int c = 0; c < 3; ++c) { n[c] = nrm[p * 3LL + c]; d[c] = pt[(long long)p * pt_ld + 4 + c]; x[c] = pt[(long long)p * pt_ld + c]; }
        nu_shade_dirs(n, d, nh, vh, r, nov, inorm);
        const float rho = nu_sigmoid(Mraw[(long long)p * ldm + 1]);
        const float e = lane < 39 ? E[(long long)p * 64 + lane] : 0.f;

        float* ol0 = OLin + (long long)p * ld_ol;
        float* ol1 = OLin + (long long)(P + p) * ld_ol;
        float* ol2 = OLin + (long long)(2LL * P + p) * ld_ol;
        // two direction evaluations (n^, r) serve all five IDE rows: only the attenuation differs between them
        const NuIdeDir tn = nu_ide_dir(IL, nh[0], nh[1], nh[2]);
        const NuIdeDir tr = nu_ide_dir(IL, r[0], r[1], r[2]);
        const float att_rho = nu_ide_att(IL, rho);
        if (sphere) {
            const NuSph sn = nu_sph_point(x, nh), sr = nu_sph_point(x, r);
            const NuIdeDir tsn = nu_ide_dir(IL, sn.s[0], sn.s[1], sn.s[2]);
            const NuIdeDir tsr = nu_ide_dir(IL, sr.s[0], sr.s[1], sr.s[2]);
            nu_ide_store(ol0, lane, IL, tn, IL.att1, 72);
            nu_ide_store(ol0 + 72, lane, IL, tsn, IL.att1, ld_ol - 72);
            nu_ide_store(ol1, lane, IL, tr, att_rho, 72);
            nu_ide_store(ol1 + 72, lane, IL, tsr, att_rho, ld_ol - 72);
            nu_ide_store(ol2, lane, IL, tr, 1.0f, 72);
            nu_ide_store(ol2 + 72, lane, IL, tsr, att_rho, ld_ol - 72);
        } else {
            nu_ide_store(ol0, lane, IL, tn, IL.att1, ld_ol);
            nu_ide_store(ol1, lane, IL, tr, att_rho, ld_ol);
            nu_ide_store(ol2, lane, IL, tr, 1.0f, ld_ol);
        }

        float* il0 = ILin + (long long)p * 128;
        float* il1 = ILin + (long long)(P + p) * 128;
        if (lane < 39) { il0[lane] = e; il1[lane] = e; }
        nu_ide_store(il0 + 39, lane, IL, tr, att_rho, 128 - 39);
        nu_ide_store(il1 + 39, lane, IL, tr, 1.0f, 128 - 39);

        float* iw = IWin + (long long)p * 96;
        float* rl = RLin + (long long)p * ld_rl;
        if (lane < 39) {
            iw[lane] = e;
            iw[39 + lane] = nu_embed_col(r, 3, lane);
        } else if (lane < 39 + 18) {
            iw[78 + lane - 39] = 0.f;
        }
        if (lane < rdim) {
            rl[lane] = e;                                   // L=rf embedding is a prefix of the L=6 one
            rl[rdim + lane] = nu_embed_col(vh, 3, lane);
        }
        for (int c = 2 * rdim + lane; c < ld_rl; c += 64) rl[c] = 0.f;
        if (lane < 8) {
            float v = 0.f;
            if (lane < 3) v = nh[lane];
            else if (lane == 3) v = nov;
            else if (lane == 4) v = inorm;
            else if (lane == 5) v = rho;
            SD[(long long)p * 8 + lane] = v;
        }
    }
}
extern "C" int nu_shade_encode_fwd(const float* nrm, const float* pt, int pt_ld, const float* E, const float* Mraw,
                                   int ldm, int P, int sphere, int ld_ol, int refrac_dim, int ld_rl, float* OLin,
                                   float* ILin, float* IWin, float* RLin, float* SD, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    if (ld_ol < (sphere ? 144 : 72) || refrac_dim > 39 || ld_rl < 2 * refrac_dim) return NU_ERR_ARG;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(shade_encode_fwd_kernel, dim3(blocks), dim3(256), 0, stream, nrm, pt, pt_ld, E, Mraw, ldm, P, sphere,
                       ld_ol, refrac_dim, ld_rl, OLin, ILin, IWin, RLin, SD);
    return nu_launch_status();
}

// backward: gradients w.r.t. the IDE inputs (dOLin, dILin cols 39..110) and dNoV (from the combine kernel)
//   -> dn_shade[P,3] (w.r.t. the RAW sdf gradient n), and dMraw[p,1] += d rho * rho (1 - rho)
__global__ __launch_bounds__(256) void shade_encode_bwd_kernel(const float* __restrict__ nrm, const float* __restrict__ pt,
                                                               int pt_ld, const float* __restrict__ SD,
                                                               const float* __restrict__ dOLin, int ld_ol, int sphere,
                                                               const float* __restrict__ dILin,
                                                               const float* __restrict__ dNoV, int P,
                                                               float* __restrict__ dn, float* __restrict__ dMraw, int ldm) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const NuIdeLane IL = nu_ide_lane(lane);
    for (int p = wave; p < P; p += nwave) {
        float n[3], d[3], x[3], nh[3], vh[3], r[3], nov, inorm;
#pragma unroll
        for (int c = 0; c < 3; ++c) { n[c] = nrm[p * 3LL + c]; d[c] = pt[(long long)p * pt_ld + 4 + c]; x[c] = pt[(long long)p * pt_ld + c]; }
        nu_shade_dirs(n, d, nh, vh, r, nov, inorm);
        const float rho = SD[(long long)p * 8 + 5];
        float dnh[3], dr[3], drho;
        const float* g0 = dOLin + (long long)p * ld_ol;
        const float* g1 = dOLin + (long long)(P + p) * ld_ol;
        const float* g2 = dOLin + (long long)(2LL * P + p) * ld_ol;
        const float* i0 = dILin + (long long)p * 128 + 39;
        const float* i1 = dILin + (long long)(P + p) * 128 + 39;
        const float att_rho = nu_ide_att(IL, rho);
        const bool lv = IL.live;
        float ax, ay, az;
        {   // IDE(n^, 1): one row
            const NuIdeDir tn = nu_ide_dir(IL, nh[0], nh[1], nh[2]);
            nu_ide_dir_grad(IL, tn, lv ? g0[lane] * IL.att1 : 0.f, lv ? g0[36 + lane] * IL.att1 : 0.f, ax, ay, az);
            dnh[0] = nu_wave_sum(ax); dnh[1] = nu_wave_sum(ay); dnh[2] = nu_wave_sum(az);
        }
        {   // IDE(r, rho) feeds outer_light row 1 and inner_light row 0; IDE(r, 0) outer row 2 and inner row 1:
            // one direction evaluation, gradient weights combined per attenuation before the wave reduction
            const NuIdeDir tr = nu_ide_dir(IL, r[0], r[1], r[2]);
            const float are_k = lv ? g1[lane] + i0[lane] : 0.f, aim_k = lv ? g1[36 + lane] + i0[36 + lane] : 0.f;
            const float are_0 = lv ? g2[lane] + i1[lane] : 0.f, aim_0 = lv ? g2[36 + lane] + i1[36 + lane] : 0.f;
            nu_ide_dir_grad(IL, tr, are_k * att_rho + are_0, aim_k * att_rho + aim_0, ax, ay, az);
            dr[0] = nu_wave_sum(ax); dr[1] = nu_wave_sum(ay); dr[2] = nu_wave_sum(az);
            drho = nu_wave_sum(-IL.sigma * att_rho * tr.poly * (are_k * tr.are + aim_k * tr.aim));
        }
        if (sphere) {
            const NuSph sn = nu_sph_point(x, nh), sr = nu_sph_point(x, r);
            float ds[3];
            const NuIdeDir tsn = nu_ide_dir(IL, sn.s[0], sn.s[1], sn.s[2]);
            nu_ide_dir_grad(IL, tsn, lv ? g0[72 + lane] * IL.att1 : 0.f, lv ? g0[108 + lane] * IL.att1 : 0.f, ax, ay, az);
            ds[0] = nu_wave_sum(ax); ds[1] = nu_wave_sum(ay); ds[2] = nu_wave_sum(az);
            nu_sph_point_bwd(sn, nh, ds, dnh);
            // both specular rows encode the sphere point of r with the point's roughness (field.py:643-646)
            const NuIdeDir tsr = nu_ide_dir(IL, sr.s[0], sr.s[1], sr.s[2]);
            const float sre = lv ? g1[72 + lane] + g2[72 + lane] : 0.f, sim = lv ? g1[108 + lane] + g2[108 + lane] : 0.f;
            nu_ide_dir_grad(IL, tsr, sre * att_rho, sim * att_rho, ax, ay, az);
            ds[0] = nu_wave_sum(ax); ds[1] = nu_wave_sum(ay); ds[2] = nu_wave_sum(az);
            drho += nu_wave_sum(-IL.sigma * att_rho * tsr.poly * (sre * tsr.are + sim * tsr.aim));
            nu_sph_point_bwd(sr, r, ds, dr);
        }
        if (lane == 0) {
            // r = 2 NoV n^ - v^ ; NoV = n^ . v^
            const float dnov = dNoV[p] + 2.0f * (dr[0] * nh[0] + dr[1] * nh[1] + dr[2] * nh[2]);
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = dnh[c] + 2.0f * nov * dr[c] + dnov * vh[c];
            // n^ = n / |n|
            const float dotp = t[0] * nh[0] + t[1] * nh[1] + t[2] * nh[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) dn[p * 3LL + c] = (t[c] - nh[c] * dotp) * inorm;
            dMraw[(long long)p * ldm + 1] += drho * rho * (1.0f - rho);
        }
    }
}
extern "C" int nu_shade_encode_bwd(const float* nrm, const float* pt, int pt_ld, const float* SD, const float* dOLin,
                                   int ld_ol, int sphere, const float* dILin, const float* dNoV, int P, float* dn,
                                   float* dMraw, int ldm, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(shade_encode_bwd_kernel, dim3(blocks), dim3(256), 0, stream, nrm, pt, pt_ld, SD, dOLin, ld_ol, sphere,
                       dILin, dNoV, P, dn, dMraw, ldm);
    return nu_launch_status();
}

// Per-ray mirror query for colour_spec: row = [IDE(d, 0) | IDE(sph(x, d), 0) if sphere | 0]
//   zero-thickness renderer: x unused (renderer_zerothick.py:780-781); standard renderer: x = the ray's candidate point
//   (sample 64, network/renderer.py:710-725)
__global__ __launch_bounds__(256) void spec_encode_kernel(const float* __restrict__ dirs, const float* __restrict__ x, int R,
                                                          int sphere, float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const NuIdeLane IL = nu_ide_lane(lane);
    for (int r = wave; r < R; r += nwave) {
        const float d[3] = {dirs[r * 3LL], dirs[r * 3LL + 1], dirs[r * 3LL + 2]};
        float* row = out + (long long)r * ldo;
        const NuIdeDir td = nu_ide_dir(IL, d[0], d[1], d[2]);
        if (sphere) {
            const float xx[3] = {x[r * 3LL], x[r * 3LL + 1], x[r * 3LL + 2]};
            const NuSph s = nu_sph_point(xx, d);
            nu_ide_store(row, lane, IL, td, 1.0f, 72);
            nu_ide_store(row + 72, lane, IL, nu_ide_dir(IL, s.s[0], s.s[1], s.s[2]), 1.0f, ldo - 72);
        } else {
            nu_ide_store(row, lane, IL, td, 1.0f, ldo);
        }
    }
}
extern "C" int nu_spec_encode(const float* dirs, const float* x, int R, int sphere, float* out, int ldo, hipStream_t stream) {
    if (R <= 0) return NU_OK;
    if (ldo < (sphere ? 144 : 72) || (sphere && !x)) return NU_ERR_ARG;
    int blocks = nu_cdiv(R, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(spec_encode_kernel, dim3(blocks), dim3(256), 0, stream, dirs, x, R, sphere, out, ldo);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// NeRF++ inputs (P outer points):  E4[p, 0:84] (pad to 96) = embed((x/|x|, 1/|x|), 10);
//   U5[p, 256:340] = same embedding (skip concat, pad to 352);  V[p, 256:283] = embed(-d, 4) (pad to 288)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nerf_embed_kernel(const float* __restrict__ pt, int pt_ld, int P,
                                                         float* __restrict__ E4, float* __restrict__ U5,
                                                         float* __restrict__ V) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < P; p += nwave) {
        float x[4], vd[3];
        const float px = pt[(long long)p * pt_ld], py = pt[(long long)p * pt_ld + 1], pz = pt[(long long)p * pt_ld + 2];
        const float nn = sqrtf(px * px + py * py + pz * pz);
        x[0] = px / nn; x[1] = py / nn; x[2] = pz / nn; x[3] = 1.0f / nn;
#pragma unroll
        for (int c = 0; c < 3; ++c) vd[c] = -pt[(long long)p * pt_ld + 4 + c];
        for (int col = lane; col < 96; col += 64) {
            const float v = col < 84 ? nu_embed_col(x, 4, col) : 0.f;
            E4[(long long)p * 96 + col] = v;
            U5[(long long)p * 352 + 256 + col] = v;  // cols 340..351 get the zero pad
        }
        if (lane < 32) V[(long long)p * 288 + 256 + lane] = lane < 27 ? nu_embed_col(vd, 3, lane) : 0.f;
    }
}
extern "C" int nu_nerf_embed(const float* pt, int pt_ld, int P, float* E4, float* U5, float* V, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(nerf_embed_kernel, dim3(blocks), dim3(256), 0, stream, pt, pt_ld, P, E4, U5, V);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Input gradients (stage 2: sample positions depend on the learned IoR through the refracted directions)
// ------------------------------------------------------------------------------------------------
// d L / d x of the SDF network: first-order part J_emb^T (dE + dSkip) plus, when nbar is given, the derivative of
// n = J_emb(x)^T gbar through J_emb itself:  sum_col gbar[col] * nbar_c * emb''_col,  emb'' = -f^2 * emb for sin/cos.
__global__ __launch_bounds__(256) void embed_jt2_kernel(const float* __restrict__ E, const float* __restrict__ dE, int lde,
                                                        const float* __restrict__ dS, int lds, const float* __restrict__ G0,
                                                        int ldg0, const float* __restrict__ Gs, int ldgs,
                                                        const float* __restrict__ nbar, int P, float* __restrict__ dx,
                                                        int accumulate) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < P; p += nwave) {
        float t = 0.f;
        int c = 0;
        if (lane < 39) {
            float g = dE[(long long)p * lde + lane];
            if (dS) g += dS[(long long)p * lds + lane];
            if (lane < 3) {
                c = lane;
                t = g;
            } else {
                const int q = lane - 3;
                const int k = q / 6;
                const int r = q - 6 * k;
                c = r >= 3 ? r - 3 : r;
                const float f = (float)(1 << k);
                const float self = E[(long long)p * 64 + lane];
                const float other = r >= 3 ? -E[(long long)p * 64 + lane - 3] : E[(long long)p * 64 + lane + 3];
                t = g * f * other;
                if (nbar) {
                    float gb = G0[(long long)p * ldg0 + lane];
                    if (Gs) gb += Gs[(long long)p * ldgs + lane];
                    t -= gb * nbar[(long long)p * 3 + c] * f * f * self;
                }
            }
        }
        const float s0 = nu_wave_sum(c == 0 ? t : 0.f);
        const float s1 = nu_wave_sum(c == 1 ? t : 0.f);
        const float s2 = nu_wave_sum(c == 2 ? t : 0.f);
        if (lane < 3) {
            const float v = lane == 0 ? s0 : (lane == 1 ? s1 : s2);
            float* o = dx + (long long)p * 3 + lane;
            *o = accumulate ? *o + v : v;
        }
    }
}
extern "C" int nu_embed_jt2(const float* E, const float* dE, int lde, const float* dS, int lds, const float* G0, int ldg0,
                            const float* Gs, int ldgs, const float* nbar, int P, float* dx, int accumulate,
                            hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(embed_jt2_kernel, dim3(blocks), dim3(256), 0, stream, E, dE, lde, dS, lds, G0, ldg0, Gs, ldgs, nbar, P,
                       dx, accumulate);
    return nu_launch_status();
}

// d L / d x and d L / d dir of the NeRF++ inputs from the gradients of its two embeddings:
//   gE [P, >=84] (+ gS [P, >=84] skip copy) w.r.t. embed((x/|x|, 1/|x|), 10);  gV [P, >=27] w.r.t. embed(-d, 4)
__global__ __launch_bounds__(256) void nerf_embed_bwd_kernel(const float* __restrict__ pt, int pt_ld, const float* __restrict__ E4,
                                                             const float* __restrict__ V, const float* __restrict__ gE, int lde,
                                                             const float* __restrict__ gS, int lds,
                                                             const float* __restrict__ gV, int ldv, int P,
                                                             float* __restrict__ dx, float* __restrict__ ddir) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    for (int p = wave; p < P; p += nwave) {
        // ---- point embedding: 4 inputs, 10 frequencies, 84 columns (two passes over the lanes) ----
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int col = lane; col < 84; col += 64) {
            float g = gE[(long long)p * lde + col];
            if (gS) g += gS[(long long)p * lds + col];
            int c;
            float t;
            if (col < 4) {
                c = col;
                t = g;
            } else {
                const int q = col - 4;
                const int k = q / 8;
                const int r = q - 8 * k;
                c = r >= 4 ? r - 4 : r;
                const float f = (float)(1 << k);
                const float other = r >= 4 ? -E4[(long long)p * 96 + col - 4] : E4[(long long)p * 96 + col + 4];
                t = g * f * other;
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acc4[cc] += (c == cc) ? t : 0.f;
        }
        float g4[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) g4[cc] = nu_wave_sum(acc4[cc]);
        // ---- view embedding: 3 inputs (-d), 4 frequencies, 27 columns stored at V[:, 256:283] ----
        float tv = 0.f;
        int cv = 0;
        if (lane < 27) {
            const float g = gV[(long long)p * ldv + lane];
            if (lane < 3) {
                cv = lane;
                tv = g;
            } else {
                const int q = lane - 3;
                const int k = q / 6;
                const int r = q - 6 * k;
                cv = r >= 3 ? r - 3 : r;
                const float f = (float)(1 << k);
                const float other = r >= 3 ? -V[(long long)p * 288 + 256 + lane - 3] : V[(long long)p * 288 + 256 + lane + 3];
                tv = g * f * other;
            }
        }
        const float v0 = nu_wave_sum(cv == 0 ? tv : 0.f), v1 = nu_wave_sum(cv == 1 ? tv : 0.f), v2 = nu_wave_sum(cv == 2 ? tv : 0.f);
        if (lane == 0) {
            const float x[3] = {pt[(long long)p * pt_ld], pt[(long long)p * pt_ld + 1], pt[(long long)p * pt_ld + 2]};
            const float nn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            const float xh[3] = {x[0] / nn, x[1] / nn, x[2] / nn};
            const float dotp = xh[0] * g4[0] + xh[1] * g4[1] + xh[2] * g4[2];
            // x4 = (x/|x|, 1/|x|):  d(x/|x|) = (I - xh xh^T)/|x| ; d(1/|x|) = -x/|x|^3
#pragma unroll
            for (int c = 0; c < 3; ++c) dx[(long long)p * 3 + c] = (g4[c] - xh[c] * dotp) / nn - g4[3] * xh[c] / (nn * nn);
            ddir[(long long)p * 3] = -v0; ddir[(long long)p * 3 + 1] = -v1; ddir[(long long)p * 3 + 2] = -v2;   // view = -d
        }
    }
}
extern "C" int nu_nerf_embed_bwd(const float* pt, int pt_ld, const float* E4, const float* V, const float* gE, int lde,
                                 const float* gS, int lds, const float* gV, int ldv, int P, float* dx, float* ddir,
                                 hipStream_t stream) {
    if (P <= 0) return NU_OK;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(nerf_embed_bwd_kernel, dim3(blocks), dim3(256), 0, stream, pt, pt_ld, E4, V, gE, lde, gS, lds, gV, ldv, P,
                       dx, ddir);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Generic positional embedding of 3-vectors, n_freq <= 10 (get_embedder(n_freq, 3), network/field.py:14-61):
//   out[p, 0:3] = x, out[p, 3 + 6k + c] = sin(2^k x_c), out[p, 6 + 6k + c] = cos(2^k x_c)
// Used where the width is not the SDF network's 6 frequencies: the 8-frequency position code of AppShadingNetwork_SpecInner
// (field.py:1351), the 2-frequency refraction codes (field.py:1353-1354).  One wave per point, lanes over columns; the
// backward recomputes sin / cos from x:  dx_c = g[c] + sum_k 2^k (cos(a) g_sin - sin(a) g_cos).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_n_fwd_kernel(const float* __restrict__ x, int P, int n_freq, float* __restrict__ out,
                                                          int ldo) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const int ncol = 3 + 6 * n_freq;
    for (int p = wave; p < P; p += nwave) {
        const float xv[3] = {x[p * 3LL], x[p * 3LL + 1], x[p * 3LL + 2]};
        if (lane < ncol) out[(long long)p * ldo + lane] = nu_embed_col(xv, 3, lane);
    }
}
__global__ __launch_bounds__(256) void embed_n_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int ldg, int P,
                                                          int n_freq, float* __restrict__ dx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float* gp = g + (long long)p * ldg;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float xc = x[p * 3LL + c];
        float acc = gp[c];
        for (int k = 0; k < n_freq; ++k) {
            const float f = (float)(1 << k), a = xc * f;
            acc += f * (cosf(a) * gp[3 + 6 * k + c] - sinf(a) * gp[6 + 6 * k + c]);
        }
        dx[p * 3LL + c] = acc;
    }
}
extern "C" int nu_embed_n_fwd(const float* x, int P, int n_freq, float* out, int ldo, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    if (n_freq < 0 || n_freq > 10 || ldo < 3 + 6 * n_freq) return NU_ERR_ARG;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(embed_n_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, P, n_freq, out, ldo);
    return nu_launch_status();
}
extern "C" int nu_embed_n_bwd(const float* x, const float* g, int ldg, int P, int n_freq, float* dx, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    if (n_freq < 0 || n_freq > 10 || ldg < 3 + 6 * n_freq) return NU_ERR_ARG;
    hipLaunchKernelGGL(embed_n_bwd_kernel, dim3(nu_cdiv(P, 256)), dim3(256), 0, stream, x, g, ldg, P, n_freq, dx);
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Stage-2 form of the shading-stack inputs: explicit points / normals / view directions [P,3] (no point records), position code of
// `pos_freq` frequencies computed here (6: AppShadingNetwork / _S2, 8: AppShadingNetwork_SpecInner, field.py:1351), refraction
// codes of `rdim = 3 + 6 refrac_freq` columns (0: no refraction stack, AppShadingNetwork_S2) -- and a backward that returns the
// gradients w.r.t. ALL inputs: stage 2 differentiates positions, normals and view directions (every one of them depends on the
// learned index of refraction), stage 1 only the normal and the roughness.
//   n^ = n / max(|n|, 1e-12), v^ likewise;  NoV = n^ . v^;  r = 2 NoV n^ - v^;  rho = sigmoid(Mraw[:, 1])
//   OLin [3P, ld_ol]: IDE(n^,1) | IDE(r,rho) | IDE(r,0)   (+ IDE of the sphere points of x along n^ / r, kappa 1 / rho / rho)
//   ILin [2P, 128]  : [pe(x), IDE(r,rho)] | [pe(x), IDE(r,0)]
//   IWin [P, 96]    : [pe(x), embed(r,6)]          (inputs of the occlusion head: no gradient, field.py:649)
//   RLin [P, ld_rl] : [embed(x,rf), embed(v^,rf)]
//   SD   [P, 12]    : n^(3), NoV, 1/|n|, rho, 1/|v|, 0, r(3), 0
// (field.py:636-682, :828-907, :1399-1445; one wave per point, lanes over encoding columns)
// ------------------------------------------------------------------------------------------------
static __device__ inline void nu_s2_dirs(const float* n, const float* v, float* nh, float* vh, float* r, float& nov, float& inorm,
                                         float& ivn) {
    inorm = 1.0f / fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
    ivn = 1.0f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; ++c) { nh[c] = n[c] * inorm; vh[c] = v[c] * ivn; }
    nov = nh[0] * vh[0] + nh[1] * vh[1] + nh[2] * vh[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = nov * nh[c] * 2.0f - vh[c];
}
// cotangent of the unit sphere point -> cotangents of dir AND of the offset origin p'
static __device__ inline void nu_sph_point_bwd2(const NuSph& o, const float* dir, const float* ds_unit, float* ddir, float* dpp) {
    const float dotp = ds_unit[0] * o.s[0] + ds_unit[1] * o.s[1] + ds_unit[2] * o.s[2];
    float ds[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ds[c] = (ds_unit[c] - o.s[c] * dotp) / o.snorm;
    const float b = o.pp[0] * dir[0] + o.pp[1] * dir[1] + o.pp[2] * dir[2];
    const float dt_db = -1.0f + b / o.root;
    const float dt = ds[0] * dir[0] + ds[1] * dir[1] + ds[2] * dir[2];    // d L / d t
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ddir[c] += o.t * ds[c] + dt * dt_db * o.pp[c];
        dpp[c] += ds[c] + dt * (dt_db * dir[c] - o.pp[c] / o.root);        // t = -b + sqrt(b^2 - p'.p' + 1 + 1e-6)
    }
}
// per-lane share of J^T g for one embedding column `col` of argument a[3]: adds to the coordinate the column depends on
static __device__ inline void nu_embed_col_grad(const float* a, int col, float g, float& gx, float& gy, float& gz) {
    int c;
    float d;
    if (col < 3) { c = col; d = 1.0f; }
    else {
        const int q = col - 3, k = q / 6, rr = q - k * 6;
        c = rr >= 3 ? rr - 3 : rr;
        const float f = (float)(1 << k), arg = a[c] * f;
        d = rr >= 3 ? -sinf(arg) * f : cosf(arg) * f;
    }
    const float v = g * d;
    gx += c == 0 ? v : 0.f; gy += c == 1 ? v : 0.f; gz += c == 2 ? v : 0.f;
}

__global__ __launch_bounds__(256) void s2_shade_encode_fwd_kernel(const float* __restrict__ xs, const float* __restrict__ nrm,
                                                                  const float* __restrict__ view, const float* __restrict__ Mraw,
                                                                  int ldm, int P, int sphere, int pe, int ld_ol, int rdim, int ld_rl,
                                                                  float* __restrict__ OLin, float* __restrict__ ILin,
                                                                  float* __restrict__ IWin, float* __restrict__ RLin,
                                                                  float* __restrict__ SD) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const NuIdeLane IL = nu_ide_lane(lane);
    for (int p = wave; p < P; p += nwave) {
        float n[3], v[3], x[3], nh[3], vh[3], r[3], nov, inorm, ivn;
#pragma unroll
        for (int c = 0; c < 3; ++c) { n[c] = nrm[p * 3LL + c]; v[c] = view[p * 3LL + c]; x[c] = xs[p * 3LL + c]; }
        nu_s2_dirs(n, v, nh, vh, r, nov, inorm, ivn);
        const float rho = nu_sigmoid(Mraw[(long long)p * ldm + 1]);
        const float e = lane < pe ? nu_embed_col(x, 3, lane) : 0.f;
        float* ol0 = OLin + (long long)p * ld_ol;
        float* ol1 = OLin + (long long)(P + p) * ld_ol;
        float* ol2 = OLin + (long long)(2LL * P + p) * ld_ol;
        const NuIdeDir tn = nu_ide_dir(IL, nh[0], nh[1], nh[2]);
        const NuIdeDir tr = nu_ide_dir(IL, r[0], r[1], r[2]);
        const float att_rho = nu_ide_att(IL, rho);
        if (sphere) {
            const NuSph sn = nu_sph_point(x, nh), sr = nu_sph_point(x, r);
            const NuIdeDir tsn = nu_ide_dir(IL, sn.s[0], sn.s[1], sn.s[2]);
            const NuIdeDir tsr = nu_ide_dir(IL, sr.s[0], sr.s[1], sr.s[2]);
            nu_ide_store(ol0, lane, IL, tn, IL.att1, 72);
            nu_ide_store(ol0 + 72, lane, IL, tsn, IL.att1, ld_ol - 72);
            nu_ide_store(ol1, lane, IL, tr, att_rho, 72);
            nu_ide_store(ol1 + 72, lane, IL, tsr, att_rho, ld_ol - 72);
            nu_ide_store(ol2, lane, IL, tr, 1.0f, 72);
            nu_ide_store(ol2 + 72, lane, IL, tsr, att_rho, ld_ol - 72);
        } else {
            nu_ide_store(ol0, lane, IL, tn, IL.att1, ld_ol);
            nu_ide_store(ol1, lane, IL, tr, att_rho, ld_ol);
            nu_ide_store(ol2, lane, IL, tr, 1.0f, ld_ol);
        }
        float* il0 = ILin + (long long)p * 128;
        float* il1 = ILin + (long long)(P + p) * 128;
        if (lane < pe) { il0[lane] = e; il1[lane] = e; }
        nu_ide_store(il0 + pe, lane, IL, tr, att_rho, 128 - pe);
        nu_ide_store(il1 + pe, lane, IL, tr, 1.0f, 128 - pe);
        float* iw = IWin + (long long)p * 96;
        if (lane < pe) iw[lane] = e;
        if (lane < 39) iw[pe + lane] = nu_embed_col(r, 3, lane);
        for (int c = pe + 39 + lane; c < 96; c += 64) iw[c] = 0.f;
        if (rdim > 0) {
            float* rl = RLin + (long long)p * ld_rl;
            if (lane < rdim) {
                rl[lane] = e;                               // the rf-frequency code is a prefix of the pos_freq one (rf <= pos_freq)
                rl[rdim + lane] = nu_embed_col(vh, 3, lane);
            }
            for (int c = 2 * rdim + lane; c < ld_rl; c += 64) rl[c] = 0.f;
        }
        if (lane < 12) {
            float o = 0.f;
            if (lane < 3) o = nh[lane];
            else if (lane == 3) o = nov;
            else if (lane == 4) o = inorm;
            else if (lane == 5) o = rho;
            else if (lane == 6) o = ivn;
            else if (lane >= 8 && lane < 11) o = r[lane - 8];
            SD[(long long)p * 12 + lane] = o;
        }
    }
}

__global__ __launch_bounds__(256) void s2_shade_encode_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ nrm,
                                                                  const float* __restrict__ view, const float* __restrict__ SD,
                                                                  int P, int sphere, int pe, int ld_ol, int rdim, int ld_rl,
                                                                  const float* __restrict__ dOLin, const float* __restrict__ dILin,
                                                                  const float* __restrict__ dRLin, const float* __restrict__ dNoV,
                                                                  float* __restrict__ dx, float* __restrict__ dn,
                                                                  float* __restrict__ dv, float* __restrict__ drho_raw) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const NuIdeLane IL = nu_ide_lane(lane);
    const bool lv = IL.live;
    for (int p = wave; p < P; p += nwave) {
        float n[3], v[3], x[3], nh[3], vh[3], r[3], nov, inorm, ivn;
#pragma unroll
        for (int c = 0; c < 3; ++c) { n[c] = nrm[p * 3LL + c]; v[c] = view[p * 3LL + c]; x[c] = xs[p * 3LL + c]; }
        nu_s2_dirs(n, v, nh, vh, r, nov, inorm, ivn);
        const float rho = SD[(long long)p * 12 + 5];
        const float att_rho = nu_ide_att(IL, rho);
        const float* g0 = dOLin ? dOLin + (long long)p * ld_ol : nullptr;
        const float* g1 = dOLin ? dOLin + (long long)(P + p) * ld_ol : nullptr;
        const float* g2 = dOLin ? dOLin + (long long)(2LL * P + p) * ld_ol : nullptr;
        const float* i0 = dILin ? dILin + (long long)p * 128 : nullptr;
        const float* i1 = dILin ? dILin + (long long)(P + p) * 128 : nullptr;
        const float* rl = (dRLin && rdim > 0) ? dRLin + (long long)p * ld_rl : nullptr;
        auto G = [&](const float* row, int col) -> float { return (row && lv) ? row[col] : 0.f; };
        float dnh[3], dr[3], drho, dpp[3] = {0.f, 0.f, 0.f};
        float ax, ay, az;
        {   // IDE(n^, 1)
            const NuIdeDir tn = nu_ide_dir(IL, nh[0], nh[1], nh[2]);
            nu_ide_dir_grad(IL, tn, G(g0, lane) * IL.att1, G(g0, 36 + lane) * IL.att1, ax, ay, az);
            dnh[0] = nu_wave_sum(ax); dnh[1] = nu_wave_sum(ay); dnh[2] = nu_wave_sum(az);
        }
        {   // IDE(r, rho): outer row 1 + inner row 0;  IDE(r, 0): outer row 2 + inner row 1
            const NuIdeDir tr = nu_ide_dir(IL, r[0], r[1], r[2]);
            const float are_k = G(g1, lane) + G(i0, pe + lane), aim_k = G(g1, 36 + lane) + G(i0, pe + 36 + lane);
            const float are_0 = G(g2, lane) + G(i1, pe + lane), aim_0 = G(g2, 36 + lane) + G(i1, pe + 36 + lane);
            nu_ide_dir_grad(IL, tr, are_k * att_rho + are_0, aim_k * att_rho + aim_0, ax, ay, az);
            dr[0] = nu_wave_sum(ax); dr[1] = nu_wave_sum(ay); dr[2] = nu_wave_sum(az);
            drho = nu_wave_sum(-IL.sigma * att_rho * tr.poly * (are_k * tr.are + aim_k * tr.aim));
        }
        if (sphere) {
            const NuSph sn = nu_sph_point(x, nh), sr = nu_sph_point(x, r);
            float ds[3];
            const NuIdeDir tsn = nu_ide_dir(IL, sn.s[0], sn.s[1], sn.s[2]);
            nu_ide_dir_grad(IL, tsn, G(g0, 72 + lane) * IL.att1, G(g0, 108 + lane) * IL.att1, ax, ay, az);
            ds[0] = nu_wave_sum(ax); ds[1] = nu_wave_sum(ay); ds[2] = nu_wave_sum(az);
            nu_sph_point_bwd2(sn, nh, ds, dnh, dpp);
            const NuIdeDir tsr = nu_ide_dir(IL, sr.s[0], sr.s[1], sr.s[2]);
            const float sre = G(g1, 72 + lane) + G(g2, 72 + lane), sim = G(g1, 108 + lane) + G(g2, 108 + lane);
            nu_ide_dir_grad(IL, tsr, sre * att_rho, sim * att_rho, ax, ay, az);
            ds[0] = nu_wave_sum(ax); ds[1] = nu_wave_sum(ay); ds[2] = nu_wave_sum(az);
            drho += nu_wave_sum(-IL.sigma * att_rho * tsr.poly * (sre * tsr.are + sim * tsr.aim));
            nu_sph_point_bwd2(sr, r, ds, dr, dpp);
        }
        // position codes (both inner_light rows, the refraction stack's first half) and the view code (its second half)
        float ex = 0.f, ey = 0.f, ez = 0.f, wx = 0.f, wy = 0.f, wz = 0.f;
        if (lane < pe) {
            float g = (i0 ? i0[lane] + i1[lane] : 0.f) + ((rl && lane < rdim) ? rl[lane] : 0.f);
            nu_embed_col_grad(x, lane, g, ex, ey, ez);
        }
        if (rl && lane < rdim) nu_embed_col_grad(vh, lane, rl[rdim + lane], wx, wy, wz);
        ex = nu_wave_sum(ex); ey = nu_wave_sum(ey); ez = nu_wave_sum(ez);
        wx = nu_wave_sum(wx); wy = nu_wave_sum(wy); wz = nu_wave_sum(wz);
        if (lane == 0) {
            // r = 2 NoV n^ - v^ ; NoV = n^ . v^
            const float dnov = (dNoV ? dNoV[p] : 0.f) + 2.0f * (dr[0] * nh[0] + dr[1] * nh[1] + dr[2] * nh[2]);
            const float dve[3] = {wx, wy, wz};
            float tn_[3], tv_[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                tn_[c] = dnh[c] + 2.0f * nov * dr[c] + dnov * vh[c];
                tv_[c] = -dr[c] + dnov * nh[c] + dve[c];
            }
            const float dn_ = tn_[0] * nh[0] + tn_[1] * nh[1] + tn_[2] * nh[2];
            const float dv_ = tv_[0] * vh[0] + tv_[1] * vh[1] + tv_[2] * vh[2];
            // p' = x inside radius 0.999, else 0.999 x / |x|
            const float xn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            float dxs[3] = {dpp[0], dpp[1], dpp[2]};
            if (xn > 0.999f) {
                const float xd = (dpp[0] * x[0] + dpp[1] * x[1] + dpp[2] * x[2]) / (xn * xn);
#pragma unroll
                for (int c = 0; c < 3; ++c) dxs[c] = 0.999f / xn * (dpp[c] - x[c] * xd);
            }
            const float dem[3] = {ex, ey, ez};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dn[p * 3LL + c] = (tn_[c] - nh[c] * dn_) * inorm;
                dv[p * 3LL + c] = (tv_[c] - vh[c] * dv_) * ivn;
                dx[p * 3LL + c] = dem[c] + dxs[c];
            }
            drho_raw[p] = drho * rho * (1.0f - rho);
        }
    }
}

extern "C" int nu_s2_shade_encode_fwd(const float* x, const float* nrm, const float* view, const float* Mraw, int ldm, int P, int sphere,
                                      int pos_freq, int ld_ol, int refrac_freq, int ld_rl, float* OLin, float* ILin, float* IWin,
                                      float* RLin, float* SD, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    const int pe = 3 + 6 * pos_freq, rdim = refrac_freq >= 0 ? 3 + 6 * refrac_freq : 0;
    if (pos_freq < 0 || pe + 72 > 128 || pe + 39 > 96 || ld_ol < (sphere ? 144 : 72) || rdim > pe || (rdim > 0 && (ld_rl < 2 * rdim || !RLin)))
        return NU_ERR_ARG;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(s2_shade_encode_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, nrm, view, Mraw, ldm, P, sphere, pe, ld_ol, rdim,
                       ld_rl, OLin, ILin, IWin, RLin, SD);
    return nu_launch_status();
}
extern "C" int nu_s2_shade_encode_bwd(const float* x, const float* nrm, const float* view, const float* SD, int P, int sphere, int pos_freq,
                                      int ld_ol, int refrac_freq, int ld_rl, const float* dOLin, const float* dILin, const float* dRLin,
                                      const float* dNoV, float* dx, float* dn, float* dv, float* drho_raw, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    const int pe = 3 + 6 * pos_freq, rdim = refrac_freq >= 0 ? 3 + 6 * refrac_freq : 0;
    if (pos_freq < 0 || pe + 72 > 128 || rdim > pe) return NU_ERR_ARG;
    int blocks = nu_cdiv(P, 4);
    blocks = blocks < 8192 ? blocks : 8192;
    hipLaunchKernelGGL(s2_shade_encode_bwd_kernel, dim3(blocks), dim3(256), 0, stream, x, nrm, view, SD, P, sphere, pe, ld_ol, rdim, ld_rl,
                       dOLin, dILin, dRLin, dNoV, dx, dn, dv, drho_raw);
    return nu_launch_status();
}
