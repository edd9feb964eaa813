// lbvh.hip -- linear BVH build + closest-hit traversal over a triangle mesh (gfx950).
//
// Replaces the reference's OptiX path: GAS build (network/tracing_optix.py:20-23, :142-146), the
// 1xN launch (:74-117, :154-158) and the three device programs of cuda/triangle.cu:48-99
//   raygen: one closest-hit trace per ray, tmin = 0, tmax = 1e16, no face culling
//   miss  : (hit, index) = (0.0, 10000000)      closesthit: (1.0, primitive index)
// Build = Morton codes of triangle centroids -> sort -> Karras 2012 radix-tree hierarchy -> bottom-up
// AABB refit -> 4-wide records (each node's grandchildren).  Traversal = four lanes per ray over the 4-wide records, one
// stack per ray in LDS, nearest entry first (lbvh_trace_quad_kernel); the one-ray-per-lane kernel over the binary nodes
// (per-lane stack in LDS, wavefront-interleaved) is kept behind NU_LBVH_QUAD=0.
//
// Hit indices are defined bit-exactly against the brute-force oracle (oracle/lbvh_oracle.py): the
// ray/triangle test below is written with explicit single-rounding fp32 operations in a fixed order, ties in t go
// to the lowest face id, and boxes are padded so that traversal never culls a triangle the test would accept.
#include "nu_common.h"
#include <stdlib.h>

// Bit-exact parity with the oracle needs one rounding per operation.  HIP's __fmul_rn/__fadd_rn are header-defined
// plain operators that still carry the 'contract' flag, so the arithmetic is written with ordinary operators and FMA
// contraction is switched off for this whole file.
#pragma clang fp contract(off)

#define NU_MISS_INDEX 10000000
#define NU_STACK 64

struct NuBvhNode {           // 64 bytes: both child boxes inline
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int left, right;         // >= 0: internal node index; < 0: leaf, sorted position = -1 - value
    int parent, pad;
};

struct NuBvhHeader {
    int n_faces, n_pad, n_verts, pad;
    float bmin[3], bmax[3], eps, pad2;
};

// ---- memory layout inside the caller's buffer -------------------------------------------------
static __host__ __device__ inline long long nu_align256(long long x) { return (x + 255) / 256 * 256; }
// 4-wide view of the same tree (one record per binary internal node: its grandchildren, or a child that is a leaf), traversed
// by the small-batch kernel with four lanes per ray.  Entry = 32 bytes: box, child reference.
#define NU_WIDE_EMPTY ((int)0x80000000)
struct NuBvhWideEntry {
    float bmin[3], bmax[3];
    int ref;                 // >= 0: internal node index (its wide record); < 0: leaf, sorted position = -1 - value; NU_WIDE_EMPTY: unused
    int pad;
};
struct NuBvhLayout {
    long long header, keys, nodes, leaf_parent, counters, tris, ids, bounds_i, wide, total;
};
static __host__ __device__ inline NuBvhLayout nu_bvh_layout(int n) {
    int npad = 1;
    while (npad < n) npad <<= 1;
    NuBvhLayout L;
    long long o = 0;
    L.header = o; o = nu_align256(o + sizeof(NuBvhHeader));
    L.keys = o; o = nu_align256(o + (long long)npad * 8);
    L.nodes = o; o = nu_align256(o + (long long)(n > 1 ? n - 1 : 1) * sizeof(NuBvhNode));
    L.leaf_parent = o; o = nu_align256(o + (long long)n * 4);
    L.counters = o; o = nu_align256(o + (long long)n * 4);
    L.tris = o; o = nu_align256(o + (long long)n * 48);      // 3 vertices x (x,y,z,pad)
    L.ids = o; o = nu_align256(o + (long long)n * 4);
    L.bounds_i = o; o = nu_align256(o + 32);
    L.wide = o; o = nu_align256(o + (long long)(n > 1 ? n - 1 : 1) * 4 * sizeof(NuBvhWideEntry));
    L.total = o;
    return L;
}
extern "C" long long nu_lbvh_bytes(int n_faces) { return nu_bvh_layout(n_faces).total; }

// order-preserving float <-> int maps for atomicMin/Max
static __device__ inline int nu_f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
static __device__ inline float nu_ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void lbvh_init_kernel(char* buf, NuBvhLayout L, int n, int npad, int nv) {
    NuBvhHeader* h = (NuBvhHeader*)(buf + L.header);
    int* b = (int*)(buf + L.bounds_i);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        h->n_faces = n; h->n_pad = npad; h->n_verts = nv;
        for (int c = 0; c < 3; ++c) { b[c] = 0x7fffffff; b[3 + c] = (int)0x80000000; }
    }
    int* cnt = (int*)(buf + L.counters);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) cnt[i] = 0;
}

__global__ void lbvh_bounds_kernel(const float* __restrict__ V, const int* __restrict__ F, int n, char* buf, NuBvhLayout L) {
    int* b = (int*)(buf + L.bounds_i);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c[3];
    for (int k = 0; k < 3; ++k)
        c[k] = (V[F[i * 3] * 3LL + k] + V[F[i * 3 + 1] * 3LL + k] + V[F[i * 3 + 2] * 3LL + k]) * (1.0f / 3.0f);
    for (int k = 0; k < 3; ++k) { atomicMin(&b[k], nu_f2ord(c[k])); atomicMax(&b[3 + k], nu_f2ord(c[k])); }
}

static __device__ inline unsigned nu_expand10(unsigned v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void lbvh_morton_kernel(const float* __restrict__ V, const int* __restrict__ F, int n, int npad, char* buf,
                                   NuBvhLayout L) {
    const int* b = (const int*)(buf + L.bounds_i);
    unsigned long long* keys = (unsigned long long*)(buf + L.keys);
    NuBvhHeader* h = (NuBvhHeader*)(buf + L.header);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = nu_ord2f(b[k]); hi[k] = nu_ord2f(b[3 + k]); }
    if (i == 0) {
        float diag = 0.f;
        for (int k = 0; k < 3; ++k) { h->bmin[k] = lo[k]; h->bmax[k] = hi[k]; diag += (hi[k] - lo[k]) * (hi[k] - lo[k]); }
        h->eps = 1e-5f * fmaxf(sqrtf(diag), 1e-3f);
    }
    if (i >= npad) return;
    if (i >= n) { keys[i] = ~0ull; return; }
    unsigned code = 0;
    for (int k = 0; k < 3; ++k) {
        const float c = (V[F[i * 3] * 3LL + k] + V[F[i * 3 + 1] * 3LL + k] + V[F[i * 3 + 2] * 3LL + k]) * (1.0f / 3.0f);
        const float ext = fmaxf(hi[k] - lo[k], 1e-20f);
        float u = (c - lo[k]) / ext;
        u = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        code |= nu_expand10((unsigned)u) << (2 - k);
    }
    keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

__global__ void lbvh_bitonic_kernel(unsigned long long* keys, int npad, int j, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    const int l = i ^ j;
    if (l > i) {
        const unsigned long long a = keys[i], b = keys[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[l] = a; }
    }
}

// gather triangles in sorted order (+ ids)
__global__ void lbvh_gather_kernel(const float* __restrict__ V, const int* __restrict__ F, int n, char* buf, NuBvhLayout L) {
    const unsigned long long* keys = (const unsigned long long*)(buf + L.keys);
    float* tris = (float*)(buf + L.tris);
    int* ids = (int*)(buf + L.ids);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int f = (int)(keys[p] & 0xffffffffu);
    ids[p] = f;
    for (int v = 0; v < 3; ++v) {
        const int vi = F[f * 3 + v];
        for (int k = 0; k < 3; ++k) tris[p * 12LL + v * 4 + k] = V[vi * 3LL + k];
        tris[p * 12LL + v * 4 + 3] = 0.f;
    }
}

static __device__ inline int nu_delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll(keys[i] ^ keys[j]);
}

// Karras 2012: one thread per internal node
__global__ void lbvh_hierarchy_kernel(int n, char* buf, NuBvhLayout L) {
    const unsigned long long* keys = (const unsigned long long*)(buf + L.keys);
    NuBvhNode* nodes = (NuBvhNode*)(buf + L.nodes);
    int* leaf_parent = (int*)(buf + L.leaf_parent);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (nu_delta(keys, n, i, i + 1) - nu_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = nu_delta(keys, n, i, i - d);
    int lmax = 2;
    while (nu_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (nu_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = nu_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (nu_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = (lo == gamma) ? -1 - gamma : gamma;
    const int right = (hi == gamma + 1) ? -1 - (gamma + 1) : gamma + 1;
    nodes[i].left = left;
    nodes[i].right = right;
    if (i == 0) nodes[i].parent = -1;
    if (left >= 0) nodes[left].parent = i; else leaf_parent[gamma] = i;
    if (right >= 0) nodes[right].parent = i; else leaf_parent[gamma + 1] = i;
}

// bottom-up refit: leaf threads climb; the second arriver at a node owns it
__global__ void lbvh_refit_kernel(int n, char* buf, NuBvhLayout L) {
    NuBvhNode* nodes = (NuBvhNode*)(buf + L.nodes);
    const int* leaf_parent = (const int*)(buf + L.leaf_parent);
    int* counters = (int*)(buf + L.counters);
    const float* tris = (const float*)(buf + L.tris);
    const NuBvhHeader* h = (const NuBvhHeader*)(buf + L.header);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float eps = h->eps;
    float bmin[3], bmax[3];
    for (int k = 0; k < 3; ++k) {
        const float a = tris[p * 12LL + k], b = tris[p * 12LL + 4 + k], c = tris[p * 12LL + 8 + k];
        bmin[k] = fminf(a, fminf(b, c)) - eps;
        bmax[k] = fmaxf(a, fmaxf(b, c)) + eps;
    }
    int child = -1 - p;
    int node = leaf_parent[p];
    while (node >= 0) {
        NuBvhNode* nd = &nodes[node];
        volatile float* dmin = (nd->left == child) ? nd->lmin : nd->rmin;
        volatile float* dmax = (nd->left == child) ? nd->lmax : nd->rmax;
        for (int k = 0; k < 3; ++k) { dmin[k] = bmin[k]; dmax[k] = bmax[k]; }
        __threadfence();
        const int prev = atomicAdd(&counters[node], 1);
        if (prev == 0) return;      // sibling subtree not finished: its thread continues upwards
        __threadfence();
        volatile NuBvhNode* vn = nd;
        for (int k = 0; k < 3; ++k) {
            bmin[k] = fminf(vn->lmin[k], vn->rmin[k]);
            bmax[k] = fmaxf(vn->lmax[k], vn->rmax[k]);
        }
        child = node;
        node = nd->parent;
    }
}

// Every (j, k) step of the sorting network whose partner distance j fits inside one NU_SORT_TILE-key tile is executed from
// LDS: one launch covers k = 2 .. NU_SORT_TILE completely, and for each larger k the tail j = NU_SORT_TILE/2 .. 1.  Same
// comparators as lbvh_bitonic_kernel in the same order, hence the same permutation; 10 launches instead of 120 at 32768 keys.
#define NU_SORT_TILE 4096
__global__ __launch_bounds__(256) void lbvh_bitonic_local_kernel(unsigned long long* keys, int npad, int k_first, int k_last) {
    __shared__ unsigned long long sk[NU_SORT_TILE];
    const int base = blockIdx.x * NU_SORT_TILE;
    const int tile = npad < NU_SORT_TILE ? npad : NU_SORT_TILE;
    for (int t = threadIdx.x; t < tile; t += 256) sk[t] = keys[base + t];
    __syncthreads();
    for (int k = k_first; k <= k_last; k <<= 1) {
        for (int j = (k >> 1) < (tile >> 1) ? (k >> 1) : (tile >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < tile; t += 256) {
                const int l = t ^ j;
                if (l > t) {
                    const unsigned long long a = sk[t], b = sk[l];
                    const bool up = ((base + t) & k) == 0;
                    if ((a > b) == up) { sk[t] = b; sk[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < tile; t += 256) keys[base + t] = sk[t];
}

// 4-wide records: node i -> the children of its two children (a child that is a leaf stands for itself); after the refit
__global__ void lbvh_widen_kernel(int n, char* buf, NuBvhLayout L) {
    const NuBvhNode* nodes = (const NuBvhNode*)(buf + L.nodes);
    NuBvhWideEntry* wide = (NuBvhWideEntry*)(buf + L.wide);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const NuBvhNode nd = nodes[i];
    NuBvhWideEntry e[4];
    int ne = 0;
    auto put = [&](const float* bmin, const float* bmax, int ref) {
        for (int k = 0; k < 3; ++k) { e[ne].bmin[k] = bmin[k]; e[ne].bmax[k] = bmax[k]; }
        e[ne].ref = ref; e[ne].pad = 0;
        ++ne;
    };
    for (int side = 0; side < 2; ++side) {
        const int c = side ? nd.right : nd.left;
        if (c < 0) {
            put(side ? nd.rmin : nd.lmin, side ? nd.rmax : nd.lmax, c);
        } else {
            const NuBvhNode nc = nodes[c];
            put(nc.lmin, nc.lmax, nc.left);
            put(nc.rmin, nc.rmax, nc.right);
        }
    }
    for (; ne < 4; ) {
        const float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        put(lo, hi, NU_WIDE_EMPTY);
    }
    for (int k = 0; k < 4; ++k) wide[i * 4LL + k] = e[k];
}

extern "C" int nu_lbvh_build(const float* V, int n_verts, const int* F, int n_faces, void* bvh, long long bvh_bytes,
                             hipStream_t stream) {
    if (n_faces <= 0 || n_verts <= 0) return NU_ERR_ARG;
    const NuBvhLayout L = nu_bvh_layout(n_faces);
    if (bvh_bytes < L.total) return NU_ERR_WORKSPACE;
    int npad = 1;
    while (npad < n_faces) npad <<= 1;
    char* buf = (char*)bvh;
    const int T = 256;
    hipLaunchKernelGGL(lbvh_init_kernel, dim3(nu_cdiv(n_faces, T) < 1024 ? nu_cdiv(n_faces, T) : 1024), dim3(T), 0, stream, buf, L,
                       n_faces, npad, n_verts);
    hipLaunchKernelGGL(lbvh_bounds_kernel, dim3(nu_cdiv(n_faces, T)), dim3(T), 0, stream, V, F, n_faces, buf, L);
    hipLaunchKernelGGL(lbvh_morton_kernel, dim3(nu_cdiv(npad, T)), dim3(T), 0, stream, V, F, n_faces, npad, buf, L);
    unsigned long long* keys = (unsigned long long*)(buf + L.keys);
    {
        const int tile = npad < NU_SORT_TILE ? npad : NU_SORT_TILE;
        const int nblk = npad / tile;
        hipLaunchKernelGGL(lbvh_bitonic_local_kernel, dim3(nblk), dim3(256), 0, stream, keys, npad, 2, tile);
        for (int k = tile << 1; k <= npad; k <<= 1) {
            for (int j = k >> 1; j >= tile; j >>= 1)
                hipLaunchKernelGGL(lbvh_bitonic_kernel, dim3(nu_cdiv(npad, T)), dim3(T), 0, stream, keys, npad, j, k);
            hipLaunchKernelGGL(lbvh_bitonic_local_kernel, dim3(nblk), dim3(256), 0, stream, keys, npad, k, k);
        }
    }
    hipLaunchKernelGGL(lbvh_gather_kernel, dim3(nu_cdiv(n_faces, T)), dim3(T), 0, stream, V, F, n_faces, buf, L);
    if (n_faces > 1) {
        hipLaunchKernelGGL(lbvh_hierarchy_kernel, dim3(nu_cdiv(n_faces - 1, T)), dim3(T), 0, stream, n_faces, buf, L);
        hipLaunchKernelGGL(lbvh_refit_kernel, dim3(nu_cdiv(n_faces, T)), dim3(T), 0, stream, n_faces, buf, L);
        hipLaunchKernelGGL(lbvh_widen_kernel, dim3(nu_cdiv(n_faces - 1, T)), dim3(T), 0, stream, n_faces, buf, L);
    }
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// ray / triangle (Moeller-Trumbore, both facings), single-rounding fp32 ops in a fixed order
// ------------------------------------------------------------------------------------------------
static __device__ inline float nu_dot3_rn(const float* a, const float* b) {
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];     // contraction is off in this file: one rounding per op
}
static __device__ inline void nu_cross_rn(const float* a, const float* b, float* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
// returns true and t when the ray hits the triangle with tmin < t < tmax
static __device__ inline bool nu_ray_tri(const float* o, const float* d, const float* v0, const float* v1, const float* v2,
                                         float tmin, float tmax, float& t) {
    float e1[3], e2[3], pv[3], tv[3], qv[3];
    for (int k = 0; k < 3; ++k) { e1[k] = v1[k] - v0[k]; e2[k] = v2[k] - v0[k]; }
    nu_cross_rn(d, e2, pv);
    const float det = nu_dot3_rn(e1, pv);
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    for (int k = 0; k < 3; ++k) tv[k] = o[k] - v0[k];
    const float u = nu_dot3_rn(tv, pv) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    nu_cross_rn(tv, e1, qv);
    const float v = nu_dot3_rn(d, qv) * inv;
    if (!(v >= 0.0f && (u + v) <= 1.0f)) return false;
    t = nu_dot3_rn(e2, qv) * inv;
    return t > tmin && t < tmax;
}

static __device__ inline bool nu_ray_box(const float* o, const float* invd, const float* bmin, const float* bmax, float tmin,
                                         float tbest, float& tnear) {
    float t0 = tmin, t1 = tbest;
    for (int k = 0; k < 3; ++k) {
        const float a = (bmin[k] - o[k]) * invd[k], b = (bmax[k] - o[k]) * invd[k];
        t0 = fmaxf(t0, fminf(a, b));      // fminf/fmaxf drop the NaN of 0 * inf
        t1 = fminf(t1, fmaxf(a, b));
    }
    tnear = t0;
    return t0 <= t1 * 1.0000005f;
}

// STACK entries per lane live in LDS, wavefront-interleaved ([entry][lane]: a push or pop of all 64 lanes is one conflict-free
// row).  The first pass runs with a SHORT stack (16 entries = 4 KB per wave, so LDS no longer pins the kernel at 8 waves per
// CU: measured 1.22 -> 1.85 G rays/s on 20480 faces, 0.61 -> 1.33 on 327680; 8 / 12 / 24 / 32 entries are slower); a lane whose traversal would need more marks its ray (hit = -1) and the second pass re-traces exactly those rays with
// the full 64-entry stack (a Morton tree over 30-bit codes + index tie-break is at most 62 deep).  RETRACE: only marked rays.
// RPW rays per wave: a traversal is a chain of dependent node fetches (latency-bound), so a small batch is faster spread thin --
// 4096 rays as 64 per wave are 64 waves on 16 CUs (78 us); as 16 per wave they are 256 single-wave workgroups, one per CU.
template <int STACK, bool RETRACE, int RPW = 64>
__global__ __launch_bounds__(RPW == 64 ? 256 : 64) void lbvh_trace_kernel(const char* __restrict__ buf, NuBvhLayout L,
                                                                           const float* __restrict__ rays, int N, float tmin, float tmax,
                                                                           float* __restrict__ hit, int* __restrict__ idx,
                                                                           float* __restrict__ tout) {
    __shared__ int stack[RPW == 64 ? 4 : 1][STACK][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = RPW == 64 ? blockIdx.x * blockDim.x + threadIdx.x : blockIdx.x * RPW + lane;
    if (r >= N || lane >= RPW) return;
    if (RETRACE && hit[r] >= 0.0f) return;
    bool overflow = false;
    const NuBvhHeader* h = (const NuBvhHeader*)(buf + L.header);
    const NuBvhNode* nodes = (const NuBvhNode*)(buf + L.nodes);
    const float* tris = (const float*)(buf + L.tris);
    const int* ids = (const int*)(buf + L.ids);
    const int n = h->n_faces;
    float o[3], d[3], invd[3];
    for (int k = 0; k < 3; ++k) { o[k] = rays[r * 6LL + k]; d[k] = rays[r * 6LL + 3 + k]; invd[k] = 1.0f / d[k]; }
    float best_t = tmax;
    int best_id = NU_MISS_INDEX;
    bool found = false;

    auto test_leaf = [&](int pos) {
        const float* tp = tris + pos * 12LL;
        float t;
        if (nu_ray_tri(o, d, tp, tp + 4, tp + 8, tmin, tmax, t)) {
            const int id = ids[pos];
            if (!found || t < best_t || (t == best_t && id < best_id)) { best_t = t; best_id = id; found = true; }
        }
    };

    if (n == 1) {
        test_leaf(0);
    } else {
        int sp = 0;
        int node = 0;
#ifdef NU_LBVH_STATS
        int steps = 0;
#endif
        while (true) {
#ifdef NU_LBVH_STATS
            ++steps;
#endif
            const NuBvhNode nd = nodes[node];
            float tl, tr;
            // inclusive in best_t: an equal-t hit with a lower face id must still be found
            const bool hl = nu_ray_box(o, invd, nd.lmin, nd.lmax, tmin, best_t, tl);
            const bool hr = nu_ray_box(o, invd, nd.rmin, nd.rmax, tmin, best_t, tr);
            int next = -1;
            if (hl && nd.left < 0) test_leaf(-1 - nd.left);
            if (hr && nd.right < 0) test_leaf(-1 - nd.right);
            const bool il = hl && nd.left >= 0, ir = hr && nd.right >= 0;
            if (il && ir) {
                const bool left_first = tl <= tr;
                next = left_first ? nd.left : nd.right;
                if (sp < STACK) stack[w][sp++][lane] = left_first ? nd.right : nd.left;
                else overflow = true;
            } else if (il) {
                next = nd.left;
            } else if (ir) {
                next = nd.right;
            }
            if (next < 0) {
                if (sp == 0) break;
                next = stack[w][--sp][lane];
            }
            node = next;
        }
#ifdef NU_LBVH_STATS
        best_t = (float)steps; found = true;
#endif
    }
    if (overflow && STACK < NU_STACK) { hit[r] = -1.0f; return; }      // incomplete: the second pass re-traces this ray
    hit[r] = found ? 1.0f : 0.0f;
    idx[r] = best_id;
    if (tout) tout[r] = found ? best_t : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// FOUR lanes per ray over the 4-wide records (16 rays per single-wave workgroup).  A traversal is a chain of dependent fetches --
// a launch of 4 096 rays lasts as long as its longest chain (56 records at ~0.7 us; the average ray visits 13) -- so the lever
// is fewer and shorter links: the four lanes of a ray test the four entries of a wide record at once
// (two binary levels per fetch), leaf entries are intersected by the lanes that hold them, the hits are ordered inside the
// quad with DPP quad permutes and the ray's stack (one per quad, in LDS) takes the farther ones.  Closest hit, ties in t to the
// lowest face id, the same box and triangle tests as the one-lane-per-ray kernel: identical (hit, index, t).
// ------------------------------------------------------------------------------------------------
#define NU_WSTACK 96            // <= 3 pushes per wide level, <= 31 wide levels (a Morton tree over 62-bit keys)
template <int CTRL> static __device__ __forceinline__ int nu_quad_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> static __device__ __forceinline__ float nu_quad_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define NU_QP_XOR1 0xB1         // quad_perm:[1,0,3,2]
#define NU_QP_XOR2 0x4E         // quad_perm:[2,3,0,1]
#define NU_QP_XOR3 0x1B         // quad_perm:[3,2,1,0]
__global__ __launch_bounds__(64) void lbvh_trace_quad_kernel(const char* __restrict__ buf, NuBvhLayout L, const float* __restrict__ rays,
                                                             int N, float tmin, float tmax, float* __restrict__ hit,
                                                             int* __restrict__ idx, float* __restrict__ tout) {
    __shared__ int stack[NU_WSTACK][16];
    const int lane = threadIdx.x & 63, sub = lane & 3, q = lane >> 2;
    const int r = blockIdx.x * 16 + q;
    if (r >= N) return;                                    // whole quads leave together
    const NuBvhHeader* h = (const NuBvhHeader*)(buf + L.header);
    const NuBvhWideEntry* wide = (const NuBvhWideEntry*)(buf + L.wide);
    const float* tris = (const float*)(buf + L.tris);
    const int* ids = (const int*)(buf + L.ids);
    const int n = h->n_faces;
    float o[3], d[3], invd[3];
    for (int k = 0; k < 3; ++k) { o[k] = rays[r * 6LL + k]; d[k] = rays[r * 6LL + 3 + k]; invd[k] = 1.0f / d[k]; }
    float best_t = tmax;
    int best_id = NU_MISS_INDEX;
    int found = 0;

    // candidate of this lane -> best of the quad -> the ray's running best (every lane of the quad holds the same triple)
    auto merge = [&](int cf, float ct, int ci) {
#define NU_BETTER(f2, t2, i2) ((f2) && (!cf || (t2) < ct || ((t2) == ct && (i2) < ci)))
        { const int f2 = nu_quad_i<NU_QP_XOR1>(cf); const float t2 = nu_quad_f<NU_QP_XOR1>(ct); const int i2 = nu_quad_i<NU_QP_XOR1>(ci);
          if (NU_BETTER(f2, t2, i2)) { cf = 1; ct = t2; ci = i2; } }
        { const int f2 = nu_quad_i<NU_QP_XOR2>(cf); const float t2 = nu_quad_f<NU_QP_XOR2>(ct); const int i2 = nu_quad_i<NU_QP_XOR2>(ci);
          if (NU_BETTER(f2, t2, i2)) { cf = 1; ct = t2; ci = i2; } }
#undef NU_BETTER
        if (cf && (!found || ct < best_t || (ct == best_t && ci < best_id))) { best_t = ct; best_id = ci; found = 1; }
    };
    auto test_leaf = [&](int pos, int& cf, float& ct, int& ci) {
        const float* tp = tris + pos * 12LL;
        float t;
        if (nu_ray_tri(o, d, tp, tp + 4, tp + 8, tmin, tmax, t)) { cf = 1; ct = t; ci = ids[pos]; }
    };

    if (n == 1) {
        int cf = 0, ci = NU_MISS_INDEX;
        float ct = 0.f;
        if (sub == 0) test_leaf(0, cf, ct, ci);
        merge(cf, ct, ci);
    } else {
        int sp = 0;
#ifdef NU_LBVH_STATS
        int steps = 0;
#endif
        // The loop is arranged so that the two fetches of a step -- the triangle of a leaf entry that was hit, the next record --
        // are in flight TOGETHER: the triangle is requested, the next record is chosen (that needs the box distances only) and
        // requested, and only then is the triangle intersected.  The running best it may improve prunes from the next record on.
        NuBvhWideEntry e = wide[sub];
        while (true) {
#ifdef NU_LBVH_STATS
            ++steps;
#endif
            float tn;
            // inclusive in best_t: an equal-t hit with a lower face id must still be found
            const bool hb = e.ref != NU_WIDE_EMPTY && nu_ray_box(o, invd, e.bmin, e.bmax, tmin, best_t, tn);
            const bool leaf = hb && e.ref < 0;
            const int pos = leaf ? -1 - e.ref : 0;
            float tv[9];
            int tid = NU_MISS_INDEX;
            if (leaf) {
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int k = 0; k < 3; ++k) tv[v * 3 + k] = tris[pos * 12LL + v * 4 + k];
                tid = ids[pos];
            }
            // internal entries that were hit: nearest first, the others on the ray's stack (farthest deepest)
            const int want = (hb && e.ref >= 0) ? 1 : 0;
            int rank = 0, cnt = want;
            { const int w2 = nu_quad_i<NU_QP_XOR1>(want); const float t2 = nu_quad_f<NU_QP_XOR1>(tn); const int s2 = sub ^ 1;
              cnt += w2; rank += (w2 && (t2 < tn || (t2 == tn && s2 < sub))) ? 1 : 0; }
            { const int w2 = nu_quad_i<NU_QP_XOR2>(want); const float t2 = nu_quad_f<NU_QP_XOR2>(tn); const int s2 = sub ^ 2;
              cnt += w2; rank += (w2 && (t2 < tn || (t2 == tn && s2 < sub))) ? 1 : 0; }
            { const int w2 = nu_quad_i<NU_QP_XOR3>(want); const float t2 = nu_quad_f<NU_QP_XOR3>(tn); const int s2 = sub ^ 3;
              cnt += w2; rank += (w2 && (t2 < tn || (t2 == tn && s2 < sub))) ? 1 : 0; }
            int next = (want && rank == 0) ? e.ref : -1;
            next = max(next, nu_quad_i<NU_QP_XOR1>(next));
            next = max(next, nu_quad_i<NU_QP_XOR2>(next));
            if (want && rank > 0) stack[sp + (cnt - 1 - rank)][q] = e.ref;
            sp += cnt > 0 ? cnt - 1 : 0;
            bool done = false;
            if (cnt == 0) {
                // (measured and dropped: keeping the box distance beside each stacked record and skipping, at the pop, what the
                // running best has overtaken -- 12.9 -> 11.2 records per ray on average, but the longest chain of a batch, which
                // is what a launch waits for, stays at 56 records and the pop loop costs more than it saves: 38.8 -> 44.1 us;
                // subtrees of up to four triangles as ONE entry whose triangles are fetched together -- longest chain 56 -> 42
                // records, but every step then carries four triangle fetches: 39 -> 44 us at 4 096 rays, 2.7 -> 1.0 G rays/s at 2^20;
                // eight lanes per ray over 8-wide records -- three binary levels per fetch, profiles/r03/lbvh_eight_lanes_per_ray.patch --
                // bit-exact too, 37 / 32 us against 42 / 31 at 4 096 rays and 1.65 against 2.67 G rays/s at 2^20: a link costs
                // ~0.55 us whatever the mesh size (320 .. 81 920 faces), and the longest chain shrinks less than the fan-out grows)
                if (sp == 0) done = true;
                else next = stack[--sp][q];
            }
            NuBvhWideEntry e2 = e;
            if (!done) e2 = wide[next * 4LL + sub];
            // the leaf entries of THIS record
            int cf = 0, ci = NU_MISS_INDEX;
            float ct = 0.f;
            if (leaf) {
                float t;
                if (nu_ray_tri(o, d, tv, tv + 3, tv + 6, tmin, tmax, t)) { cf = 1; ct = t; ci = tid; }
            }
            merge(cf, ct, ci);
            if (done) break;
            e = e2;
        }
#ifdef NU_LBVH_STATS
        best_t = (float)steps; found = 1;                  // development build: t_out reports the number of records visited
#endif
    }
    if (sub == 0) {
        hit[r] = found ? 1.0f : 0.0f;
        idx[r] = best_id;
        if (tout) tout[r] = found ? best_t : 0.0f;
    }
}

extern "C" int nu_lbvh_trace(const void* bvh, int n_faces, const float* rays, int N, float tmin, float tmax, float* hit,
                             int* idx, float* t_out, hipStream_t stream) {
    if (N <= 0) return NU_OK;
    const NuBvhLayout L = nu_bvh_layout(n_faces);
    // the four-lanes-per-ray kernel is the default at every batch size (4 096 rays: 75 -> 39 us object-aimed, 49 -> 32 us camera rays;
    // 2^20 rays: 1.82 -> 2.67 and 4.22 -> 4.96 G rays/s; profiles/r03/lbvh_bench.txt); NU_LBVH_QUAD=0 selects the one-lane-per-ray
    // kernels below (development A/B)
    static const int quad_env = getenv("NU_LBVH_QUAD") ? atoi(getenv("NU_LBVH_QUAD")) : 1;
    if (quad_env != 0) {
        hipLaunchKernelGGL(lbvh_trace_quad_kernel, dim3(nu_cdiv(N, 16)), dim3(64), 0, stream, (const char*)bvh, L, rays, N, tmin, tmax,
                           hit, idx, t_out);
        return nu_launch_status();
    }
    static const int full_only = getenv("NU_LBVH_FULL_STACK") ? atoi(getenv("NU_LBVH_FULL_STACK")) : 0;   // development A/B
    if (full_only) {
        hipLaunchKernelGGL((lbvh_trace_kernel<NU_STACK, false>), dim3(nu_cdiv(N, 256)), dim3(256), 0, stream, (const char*)bvh, L, rays, N,
                           tmin, tmax, hit, idx, t_out);
        return nu_launch_status();
    }
    static const int short_env = getenv("NU_LBVH_SHORT") ? atoi(getenv("NU_LBVH_SHORT")) : 16;   // development sweep: 16 measured best
    static const int rpw_env = getenv("NU_LBVH_RPW") ? atoi(getenv("NU_LBVH_RPW")) : 0;          // development switch: rays per wave
    // small batches (a training step traces 4096 rays or fewer per bounce): 16 or 32 rays per single-wave workgroup, so that the
    // batch covers the chip's 256 CUs; from 16 384 rays on, full waves
    const int rpw = rpw_env ? rpw_env : (N <= 8192 ? 16 : (N <= 16384 ? 32 : 64));
    if (rpw == 16 || rpw == 32) {
        const dim3 grid(nu_cdiv(N, rpw)), block(64);
        if (rpw == 16) {
            hipLaunchKernelGGL((lbvh_trace_kernel<16, false, 16>), grid, block, 0, stream, (const char*)bvh, L, rays, N, tmin, tmax, hit, idx, t_out);
            hipLaunchKernelGGL((lbvh_trace_kernel<NU_STACK, true, 16>), grid, block, 0, stream, (const char*)bvh, L, rays, N, tmin, tmax, hit, idx, t_out);
        } else {
            hipLaunchKernelGGL((lbvh_trace_kernel<16, false, 32>), grid, block, 0, stream, (const char*)bvh, L, rays, N, tmin, tmax, hit, idx, t_out);
            hipLaunchKernelGGL((lbvh_trace_kernel<NU_STACK, true, 32>), grid, block, 0, stream, (const char*)bvh, L, rays, N, tmin, tmax, hit, idx, t_out);
        }
        return nu_launch_status();
    }
#define NU_TRACE1(S) hipLaunchKernelGGL((lbvh_trace_kernel<S, false>), dim3(nu_cdiv(N, 256)), dim3(256), 0, stream, (const char*)bvh, L, \
                                        rays, N, tmin, tmax, hit, idx, t_out)
    if (short_env == 8) NU_TRACE1(8); else if (short_env == 12) NU_TRACE1(12);
    else if (short_env == 24) NU_TRACE1(24); else if (short_env == 32) NU_TRACE1(32); else NU_TRACE1(16);
#undef NU_TRACE1
    hipLaunchKernelGGL((lbvh_trace_kernel<NU_STACK, true>), dim3(nu_cdiv(N, 256)), dim3(256), 0, stream, (const char*)bvh, L, rays, N,
                       tmin, tmax, hit, idx, t_out);
    return nu_launch_status();
}

// brute-force closest hit on the device (same triangle test; O(N*F)): cross-check + tiny meshes
__global__ __launch_bounds__(256) void brute_trace_kernel(const float* __restrict__ V, const int* __restrict__ F, int nf,
                                                          const float* __restrict__ rays, int N, float tmin, float tmax,
                                                          float* __restrict__ hit, int* __restrict__ idx, float* __restrict__ tout) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    float o[3], d[3];
    for (int k = 0; k < 3; ++k) { o[k] = rays[r * 6LL + k]; d[k] = rays[r * 6LL + 3 + k]; }
    float best_t = tmax;
    int best_id = NU_MISS_INDEX;
    bool found = false;
    for (int f = 0; f < nf; ++f) {
        float v[3][3];
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 3; ++k) v[a][k] = V[F[f * 3 + a] * 3LL + k];
        float t;
        if (nu_ray_tri(o, d, v[0], v[1], v[2], tmin, tmax, t)) {
            if (!found || t < best_t) { best_t = t; best_id = f; found = true; }   // ascending f: ties keep the lowest id
        }
    }
    hit[r] = found ? 1.0f : 0.0f;
    idx[r] = best_id;
    if (tout) tout[r] = found ? best_t : 0.0f;
}
extern "C" int nu_brute_trace(const float* V, const int* F, int n_faces, const float* rays, int N, float tmin, float tmax,
                              float* hit, int* idx, float* t_out, hipStream_t stream) {
    if (N <= 0) return NU_OK;
    hipLaunchKernelGGL(brute_trace_kernel, dim3(nu_cdiv(N, 256)), dim3(256), 0, stream, V, F, n_faces, rays, N, tmin, tmax, hit,
                       idx, t_out);
    return nu_launch_status();
}
