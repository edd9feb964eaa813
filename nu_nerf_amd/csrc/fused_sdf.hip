// fused_sdf.hip -- the SDF network's no-gradient forward as ONE kernel: x -> embedding -> eight 256-wide softplus layers -> sdf.
//
// Replaces, for the evaluations that keep nothing (the hierarchical sampler's 64 + 3 x 16 SDF queries per ray, the occlusion
// probe, extract_fields, the stage-2 inner up-sampler):
//   SDFNetwork.forward -> sdf                    network/field.py:133-153 (embedding :47-61, skip concat :142-143)
//   as called from upsample / cat_z_vals         network/renderer_zerothick.py:525-570, :572-612
// which the layer-by-layer path runs as 1 + 8 + 1 launches whose activations round-trip HBM although nothing reads them again.
// On the point sets these callers produce (8 192 rows at the reference's default batch of 512 rays) a 256 -> 256 layer is 128
// tiles for 256 CUs and each launch is latency, not arithmetic.
//
// Design.  A workgroup owns a tile of 32 TMN rows (points) for ALL layers.  The activations of the tile live in LDS, k-contiguous
// per row (stride 260 floats: ds_read_b128 fragment reads hit every 16-byte slot once per 16 rows), and ARE the A operand of every
// layer; the weights stream through two LDS stages in 32-deep chunks exactly as in gemm_nt2_kernel (registers -> other stage in the
// middle of a chunk, global -> registers behind it, one barrier per chunk, one memory instruction per MFMA gap) -- across layer
// boundaries too, so the next layer's first chunk is already staged when a layer's epilogue runs.  Four waves, each 32 TMN rows x 64
// of the 256 output columns; a layer's epilogue (bias + softplus(beta = 100)) writes the accumulators straight back over the
// activation tile: every wave has finished reading it at the last chunk's barrier.  Layer 4's skip input [h4 (217) | embedding
// (39)] (the 1 / sqrt 2 is folded into the packed weights) is formed by re-inserting the tile's embedding, kept in LDS.  The sdf
// head is one wave per row with the arithmetic of skinny_fwd_kernel<1, 256>.
// Arithmetic order is that of the layered path (same MFMA, same k order, same softplus, same head reduction): BIT-identical
// results, so the sampler's parity tests hold unchanged.
#include "gemm.h"

#define FS_ALD 260                 // activation row stride (floats)
#define FS_ELD 40                  // embedding row stride
#define FS_BLD 36                  // weight-stage row stride (as NT_LDS)
#define FS_BSTAGE (256 * FS_BLD)   // floats per weight stage: 256 output columns x 32 k (+ 4 pad)

static __device__ inline float fs_embed_col(const float* x, int col) {      // get_embedder(6, 3) column (encode.hip: nu_embed_col)
    if (col < 3) return x[col];
    const int q = col - 3;
    const int k = q / 6;
    const int r = q - k * 6;
    const int c = r >= 3 ? r - 3 : r;
    const float a = x[c] * (float)(1 << k);
    return r >= 3 ? cosf(a) : sinf(a);
}

struct FsNet {                     // what the kernel needs of NuSdfNet (kernel-argument block)
    const float* Wp[8]; const float* bias[8];
    const float* w8; const float* b8;
    int Kp[8], N[8];
};

template <int TMN>
__global__ __launch_bounds__(256, 1) void sdf_fused_fwd_kernel(FsNet net, const float* __restrict__ X, int x_ld, int P,
                                                               float* __restrict__ sdf_out) {
    constexpr int TM = 32 * TMN;
    __shared__ __attribute__((aligned(16))) float act[TM * FS_ALD];
    __shared__ __attribute__((aligned(16))) float emb[TM * FS_ELD];
    __shared__ __attribute__((aligned(16))) float bst[2 * FS_BSTAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wc = tid >> 6;                                  // this wave's 64 output columns
    const int li = lane & 31, lh = lane >> 5;
    const int c4 = tid & 7, r0 = tid >> 3;                    // weight loader: rows r0 + 32 i (i < 8), 16-byte slot c4
    const int a0_off = li * FS_ALD + 4 * lh, a1_off = a0_off + 32 * FS_ALD;
    const int b0_off = (wc * 64 + li) * FS_BLD + 4 * lh, b1_off = b0_off + 32 * FS_BLD;
    const int w_off = r0 * FS_BLD + 4 * c4;
    const int ntiles = (P + TM - 1) / TM;

    // ---- weight loader: a cursor over the chunks of all eight layers, in order (past the end it stays on the last chunk) ----
    int ld_l = 0, ld_kt = 0;
    const float* bp = nullptr;                                // this thread's slot of row r0 of the chunk at the cursor
    long long bstep = 0;                                      // 32 rows further
    f32x4 rb4[8];
    auto set_cursor = [&]() {
        bp = net.Wp[ld_l] + (long long)r0 * net.Kp[ld_l] + 4 * c4 + ld_kt * 32;
        bstep = 32LL * net.Kp[ld_l];
    };
    auto advance = [&]() {
        if (++ld_kt == net.Kp[ld_l] / 32) {
            if (ld_l == 7) { --ld_kt; return; }
            ld_kt = 0;
            ++ld_l;
        }
        set_cursor();
    };
    struct Frag { f32x4 a0, a1, b0, b1; };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        // ---- embedding of the tile's points: columns 0..38 (zero pad to 64) -> act, and a copy for layer 4's skip input ----
        for (int idx = tid; idx < TM * 64; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            int p = row0 + r;
            p = p < P ? p : P - 1;
            float x[3] = {X[(long long)p * x_ld], X[(long long)p * x_ld + 1], X[(long long)p * x_ld + 2]};
            const float v = c < 39 ? fs_embed_col(x, c) : 0.f;
            act[r * FS_ALD + c] = v;
            if (c < FS_ELD) emb[r * FS_ELD + c] = v;
        }
        // ---- weight pipeline prologue: chunk 0 -> stage 0, chunk 1 -> registers ----
        ld_l = 0; ld_kt = 0;
        set_cursor();
#pragma unroll
        for (int i = 0; i < 8; ++i) rb4[i] = *reinterpret_cast<const f32x4*>(bp + i * bstep);
        advance();
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&bst[w_off + 32 * i * FS_BLD]) = rb4[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) rb4[i] = *reinterpret_cast<const f32x4*>(bp + i * bstep);
        advance();
        __syncthreads();
        int cur = 0;
        Frag F0, F1;
        F0.b0 = *reinterpret_cast<const f32x4*>(&bst[b0_off]);
        F0.b1 = *reinterpret_cast<const f32x4*>(&bst[b1_off]);

        for (int l = 0; l < 8; ++l) {
            const int nk = net.Kp[l] / 32;
            f32x16 acc[TMN][2];
#pragma unroll
            for (int i = 0; i < TMN; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            // first A fragments of the layer (the activation tile was rewritten by the previous layer's epilogue)
            F0.a0 = *reinterpret_cast<const f32x4*>(&act[a0_off]);
            if (TMN == 2) F0.a1 = *reinterpret_cast<const f32x4*>(&act[a1_off]);
#define FS_PIN __builtin_amdgcn_sched_barrier(0);
#define FS_M(F, e, i, j) if constexpr (i < TMN) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a##i[e], F.b##j[e], acc[i][j], 0, 0, 0); FS_PIN }
#define FS_RA0(F, k0, kk) F.a0 = *reinterpret_cast<const f32x4*>(&act[a0_off + (k0) + (kk) * 8]); FS_PIN
#define FS_RA1(F, k0, kk) if constexpr (TMN == 2) { F.a1 = *reinterpret_cast<const f32x4*>(&act[a1_off + (k0) + (kk) * 8]); FS_PIN }
#define FS_RB(F, S, kk, m) F.m = *reinterpret_cast<const f32x4*>(&(S)[m##_off + (kk) * 8]); FS_PIN
#define FS_W(i) *reinterpret_cast<f32x4*>(&so[w_off + 32 * (i) * FS_BLD]) = rb4[i]; FS_PIN
#define FS_L(i) rb4[i] = *reinterpret_cast<const f32x4*>(bp + (i) * bstep); FS_PIN
            const float* sc = &bst[cur * FS_BSTAGE];
            for (int kt = 0; kt < nk; ++kt) {
                float* so = &bst[(cur ^ 1) * FS_BSTAGE];
                const int k0 = kt * 32;
                // k-group 0: fragment reads of k-group 1
                FS_M(F0, 0, 0, 0) FS_M(F0, 0, 0, 1) FS_RA0(F1, k0, 1) FS_M(F0, 0, 1, 0) FS_M(F0, 0, 1, 1)
                FS_M(F0, 1, 0, 0) FS_M(F0, 1, 0, 1) FS_RA1(F1, k0, 1) FS_M(F0, 1, 1, 0) FS_M(F0, 1, 1, 1)
                FS_M(F0, 2, 0, 0) FS_M(F0, 2, 0, 1) FS_RB(F1, sc, 1, b0) FS_M(F0, 2, 1, 0) FS_M(F0, 2, 1, 1)
                FS_M(F0, 3, 0, 0) FS_M(F0, 3, 0, 1) FS_RB(F1, sc, 1, b1) FS_M(F0, 3, 1, 0) FS_M(F0, 3, 1, 1)
                // k-group 1: the next weight chunk registers -> other stage; fragment reads of k-group 2
                FS_M(F1, 0, 0, 0) FS_W(0) FS_M(F1, 0, 0, 1) FS_RA0(F0, k0, 2) FS_M(F1, 0, 1, 0) FS_W(1) FS_M(F1, 0, 1, 1)
                FS_M(F1, 1, 0, 0) FS_W(2) FS_M(F1, 1, 0, 1) FS_RA1(F0, k0, 2) FS_M(F1, 1, 1, 0) FS_W(3) FS_M(F1, 1, 1, 1)
                FS_M(F1, 2, 0, 0) FS_W(4) FS_M(F1, 2, 0, 1) FS_RB(F0, sc, 2, b0) FS_M(F1, 2, 1, 0) FS_W(5) FS_M(F1, 2, 1, 1)
                FS_M(F1, 3, 0, 0) FS_W(6) FS_M(F1, 3, 0, 1) FS_RB(F0, sc, 2, b1) FS_M(F1, 3, 1, 0) FS_W(7) FS_M(F1, 3, 1, 1)
                // k-group 2: the chunk after that global -> registers; fragment reads of k-group 3
                FS_M(F0, 0, 0, 0) FS_L(0) FS_M(F0, 0, 0, 1) FS_RA0(F1, k0, 3) FS_M(F0, 0, 1, 0) FS_L(1) FS_M(F0, 0, 1, 1)
                FS_M(F0, 1, 0, 0) FS_L(2) FS_M(F0, 1, 0, 1) FS_RA1(F1, k0, 3) FS_M(F0, 1, 1, 0) FS_L(3) FS_M(F0, 1, 1, 1)
                FS_M(F0, 2, 0, 0) FS_L(4) FS_M(F0, 2, 0, 1) FS_RB(F1, sc, 3, b0) FS_M(F0, 2, 1, 0) FS_L(5) FS_M(F0, 2, 1, 1)
                FS_M(F0, 3, 0, 0) FS_L(6) FS_M(F0, 3, 0, 1) FS_RB(F1, sc, 3, b1) FS_M(F0, 3, 1, 0) FS_L(7) FS_M(F0, 3, 1, 1)
                advance();
                __syncthreads();        // the other stage is complete; every wave holds its last fragments of this chunk
                // k-group 3: first fragments of the next chunk (weights: the other stage, also across a layer boundary).  The activation
                // reads are unconditional -- a branch around them would make hipcc drain the weight loads at the join -- and after a
                // layer's last chunk they return bytes nobody uses (the next layer re-reads its first fragments behind the epilogue)
                FS_M(F1, 0, 0, 0) FS_M(F1, 0, 0, 1) FS_RA0(F0, k0 + 32, 0) FS_M(F1, 0, 1, 0) FS_M(F1, 0, 1, 1)
                FS_M(F1, 1, 0, 0) FS_M(F1, 1, 0, 1) FS_RA1(F0, k0 + 32, 0) FS_M(F1, 1, 1, 0) FS_M(F1, 1, 1, 1)
                FS_M(F1, 2, 0, 0) FS_M(F1, 2, 0, 1) FS_RB(F0, so, 0, b0) FS_M(F1, 2, 1, 0) FS_M(F1, 2, 1, 1)
                FS_M(F1, 3, 0, 0) FS_M(F1, 3, 0, 1) FS_RB(F0, so, 0, b1) FS_M(F1, 3, 1, 0) FS_M(F1, 3, 1, 1)
                cur ^= 1;
                sc = so;
            }
#undef FS_M
#undef FS_RA0
#undef FS_RA1
#undef FS_RB
#undef FS_W
#undef FS_L
#undef FS_PIN
            // ---- epilogue: h = softplus(acc + bias) back over the activation tile (all reads of it retired at the last barrier) ----
            const int N = net.N[l];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int col = wc * 64 + tn * 32 + li;
                const float bv = col < N ? net.bias[l][col] : 0.f;
#pragma unroll
                for (int tm = 0; tm < TMN; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        // columns [N, 256) (layer 3: the slots of the skip embedding) are not this layer's
                        if (col < N) act[row * FS_ALD + col] = nu_softplus100_fast(acc[tm][tn][r] + bv);
                    }
            }
            if (l == 3) {               // layer 4's input: [h4 (217) | embedding (39)]
                for (int idx = tid; idx < TM * 39; idx += 256) {
                    const int r = idx / 39, c = idx - r * 39;
                    act[r * FS_ALD + 217 + c] = emb[r * FS_ELD + c];
                }
            }
            __syncthreads();
        }
        // ---- sdf head: one wave per row, the arithmetic of skinny_fwd_kernel<1, 256> (16 bytes per lane, xor-shuffle reduction) ----
        {
#pragma clang fp contract(off)
            const f32x4 w = *reinterpret_cast<const f32x4*>(net.w8 + 4 * lane);
            const float b = net.b8[0];
            for (int r = wc; r < TM; r += 4) {
                const f32x4 h = *reinterpret_cast<const f32x4*>(&act[r * FS_ALD + 4 * lane]);
                float a = 0.f;
                a += (h[0] * w[0] + h[1] * w[1]) + (h[2] * w[2] + h[3] * w[3]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
                if (lane == 0 && row0 + r < P) sdf_out[row0 + r] = a + b;
            }
        }
        __syncthreads();            // the tile's LDS is free for the next tile
    }
}

// sdf[P] = SDFNetwork(x)[..., 0] for X [P, x_ld] (first three floats of a row = x), exact fp32, nothing kept.
extern "C" int nu_sdf_fused_fwd(const NuSdfNet* net, const float* X, int x_ld, int P, float* sdf, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    if (!net || !X || !sdf || x_ld < 3) return NU_ERR_ARG;
    FsNet n;
    for (int l = 0; l < 8; ++l) {
        const NuLin& L = net->lin[l];
        const int Kexp = l == 0 ? 64 : 256;
        if (L.Kp != Kexp || L.N > 256 || L.N < 1 || !L.Wp || !L.bias) return NU_ERR_ARG;     // SDFNetwork(dims 39 -> 8 x 256 -> 257, skip at 4)
        n.Wp[l] = L.Wp; n.bias[l] = L.bias; n.Kp[l] = L.Kp; n.N[l] = L.N;
    }
    if (net->lin[3].N != 217 || net->lin[8].Kp != 256) return NU_ERR_ARG;
    n.w8 = net->lin[8].Wp; n.b8 = net->lin[8].bias;
    // 64-row tiles when they still give every CU a workgroup, else 32-row tiles (twice the workgroups, half the work each)
    static const int tm_env = getenv("NU_FUSED_SDF_TM") ? atoi(getenv("NU_FUSED_SDF_TM")) : 0;       // development switch: 32 / 64
    const bool tm64 = tm_env ? tm_env == 64 : nu_cdiv(P, 64) >= 192;      // (measured: 16 384 points 157 vs 175 us, 8 192 points 153 vs 93)
    const int ntiles = nu_cdiv(P, tm64 ? 64 : 32);
    const int grid = ntiles < 256 ? ntiles : 256;
    if (tm64) hipLaunchKernelGGL((sdf_fused_fwd_kernel<2>), dim3(grid), dim3(256), 0, stream, n, X, x_ld, P, sdf);
    else hipLaunchKernelGGL((sdf_fused_fwd_kernel<1>), dim3(grid), dim3(256), 0, stream, n, X, x_ld, P, sdf);
    return nu_launch_status();
}
