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
    __shared__ __attribute__((aligned(16))) float act[TM * FS_ALD + 4];     // (+ 4: a layer's last chunk prefetches the fragment slot behind the last row; never used)
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

// ------------------------------------------------------------------------------------------------
// The same network in the bf16-STORAGE arithmetic (cfg mlp_dtype 'bf16', BASELINE config 4): bf16 weight tables (NuLin.Wp16),
// activations rounded to bf16 (RNE) between layers, products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, bias + softplus
// and the sdf head in fp32 -- value for value what the layered path computes when it stores H[1..8] as bf16
// (nu_sdf_mlp_fwd(..., want_feat = 0) under NuOpCtx.h16: same k order per accumulator, same roundings), so the results are
// BIT-identical.  The layered path moves 1 KB per point and layer through HBM for 131 kFLOP at 16x the fp32 MFMA rate (config 4:
// 56 launches, 3.9 ms of a 39 ms step, at 0.4 of the HBM roofline); here a point costs 12 bytes in and 4 out.
// A workgroup owns 64 (or 32) rows for all layers: activation tile [TM][256] bf16 in LDS (row stride 528 B: conflict-free 16-byte
// fragment reads), the weights stream through two LDS stages of [256][32 k] bf16 (row stride 80 B) -- 80 KB per workgroup, two
// workgroups per CU, eight waves hiding each other's LDS latency; the chunk after next travels global -> registers meanwhile.
// ------------------------------------------------------------------------------------------------
#define FH_ALD 264                 // activation row stride (bf16 elements; 528 B)
#define FH_ELD 40                  // embedding row stride (bf16 elements)
#define FH_BLD 40                  // weight-stage row stride (bf16 elements; 80 B = 32 k + 16 B pad)
#define FH_BSTAGE (256 * FH_BLD)   // elements per weight stage

struct FsNet16 {
    const __bf16* Wp[8]; const float* bias[8];
    const float* w8; const float* b8;
    int Kp[8], N[8];
};

// WR = 2: 512 threads, two row groups of four waves share one weight stream (128 rows per workgroup: half the weight bytes per point
// through L2 -> LDS, which is what bounds this kernel -- every tile re-streams the whole network, 0.95 MB); one workgroup per CU.
template <int TMN, int WR>
__global__ __launch_bounds__(256 * WR, WR == 1 ? 2 : 1) void sdf_fused16_fwd_kernel(FsNet16 net, const float* __restrict__ X, int x_ld, int P,
                                                                 float* __restrict__ sdf_out) {
    constexpr int TM = 32 * TMN * WR;
    constexpr int NT = 256 * WR;                               // threads
    __shared__ __attribute__((aligned(16))) __bf16 act[TM * FH_ALD];
    __shared__ __attribute__((aligned(16))) __bf16 emb[TM * FH_ELD];
    __shared__ __attribute__((aligned(16))) __bf16 bst[2 * FH_BSTAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wc = (tid >> 6) & 3;                            // this wave's 64 output columns
    const int wr = tid >> 8;                                  // ... and its group of 32 TMN rows
    const int li = lane & 31, lh = lane >> 5;
    constexpr int NLD = 4 / WR;                                // 16-byte pieces of a chunk per thread
    const int c4 = tid & 3, r0 = tid >> 2;                    // weight loader: rows r0 + 64 WR i (i < NLD), 16-byte slot c4 of the 64-byte chunk row
    const int a_off = (wr * 32 * TMN + li) * FH_ALD + 8 * lh;                   // lane (r, h) holds k = 8 h .. 8 h + 7 of a 16-deep MFMA step
    const int b_off = (wc * 64 + li) * FH_BLD + 8 * lh;
    const int w_off = r0 * FH_BLD + 8 * c4;
    const int ntiles = (P + TM - 1) / TM;

    // weight loader: a cursor over the 32-deep chunks of all eight layers, in order (past the end it stays on the last chunk)
    int ld_l = 0, ld_kt = 0;
    const __bf16* bp = nullptr;
    long long bstep = 0;                                      // 64 rows further (elements)
    f32x4 rb4[NLD];
    auto set_cursor = [&]() {
        bp = net.Wp[ld_l] + (long long)r0 * net.Kp[ld_l] + 8 * c4 + ld_kt * 32;
        bstep = 64LL * WR * net.Kp[ld_l];
    };
    auto advance = [&]() {
        if (++ld_kt == net.Kp[ld_l] / 32) {
            if (ld_l == 7) { --ld_kt; return; }
            ld_kt = 0;
            ++ld_l;
        }
        set_cursor();
    };
    auto load_regs = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) rb4[i] = *reinterpret_cast<const f32x4*>(bp + i * bstep);
    };
    auto store_regs = [&](int st) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<f32x4*>(&bst[st * FH_BSTAGE + w_off + 64 * WR * i * FH_BLD]) = rb4[i];
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        // embedding of the tile's points, rounded to bf16 as the layered path rounds it on its way into LDS: columns 0..38 (zero pad
        // to 64) -> act, and a copy for layer 4's skip input
        for (int idx = tid; idx < TM * 64; idx += NT) {
            const int r = idx >> 6, c = idx & 63;
            int p = row0 + r;
            p = p < P ? p : P - 1;
            float x[3] = {X[(long long)p * x_ld], X[(long long)p * x_ld + 1], X[(long long)p * x_ld + 2]};
            const __bf16 v = (__bf16)(c < 39 ? fs_embed_col(x, c) : 0.f);
            act[r * FH_ALD + c] = v;
            if (c < FH_ELD) emb[r * FH_ELD + c] = v;
        }
        // weight pipeline prologue: chunk 0 -> stage 0, chunk 1 -> registers
        ld_l = 0; ld_kt = 0;
        set_cursor();
        load_regs();
        advance();
        store_regs(0);
        load_regs();
        advance();
        __syncthreads();
        int cur = 0;

        for (int l = 0; l < 8; ++l) {
            const int nk = net.Kp[l] / 32;
            f32x16 acc[TMN][2];
#pragma unroll
            for (int i = 0; i < TMN; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            for (int kt = 0; kt < nk; ++kt) {
                // stage `cur` holds this chunk, the registers the next one: hand it to the other stage (its readers left at the last
                // barrier) and fetch the chunk after it
                store_regs(cur ^ 1);
                load_regs();
                advance();
                const __bf16* sc = &bst[cur * FH_BSTAGE];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&sc[b_off + 16 * ks]);
                    const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&sc[b_off + 32 * FH_BLD + 16 * ks]);
                    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&act[a_off + kt * 32 + 16 * ks]);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                    if constexpr (TMN == 2) {
                        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&act[a_off + 32 * FH_ALD + kt * 32 + 16 * ks]);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
                    }
                }
                __syncthreads();        // the other stage is complete; every wave is done with this one (and, after a layer's last chunk, with the tile)
                cur ^= 1;
            }
            // epilogue: h = bf16(softplus(acc + bias)) back over the activation tile
            const int N = net.N[l];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int col = wc * 64 + tn * 32 + li;
                const float bv = col < N ? net.bias[l][col] : 0.f;
#pragma unroll
                for (int tm = 0; tm < TMN; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wr * 32 * TMN + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (col < N) act[row * FH_ALD + col] = (__bf16)nu_softplus100_fast(acc[tm][tn][r] + bv);
                    }
            }
            if (l == 3) {               // layer 4's input: [h4 (217) | embedding (39)]
                for (int idx = tid; idx < TM * 39; idx += NT) {
                    const int r = idx / 39, c = idx - r * 39;
                    act[r * FH_ALD + 217 + c] = emb[r * FH_ELD + c];
                }
            }
            __syncthreads();
        }
        // sdf head: one wave per row, the arithmetic of skinny_fwd_kernel<1, 256, true> (4 bf16 per lane widened exactly, fp32 products)
        {
#pragma clang fp contract(off)
            const f32x4 w = *reinterpret_cast<const f32x4*>(net.w8 + 4 * lane);
            const float b = net.b8[0];
            for (int r = tid >> 6; r < TM; r += 4 * WR) {
                const uint2 raw = *reinterpret_cast<const uint2*>(&act[r * FH_ALD + 4 * lane]);
                f32x4 h;
                h[0] = __uint_as_float(raw.x << 16); h[1] = __uint_as_float(raw.x & 0xffff0000u);
                h[2] = __uint_as_float(raw.y << 16); h[3] = __uint_as_float(raw.y & 0xffff0000u);
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

// 128 rows, 512 threads, and the weights of a WHOLE layer in flight.  With one 32-deep chunk in flight per workgroup the kernel above
// waits an L2 round trip (~1.3 us) per chunk for 0.2 us of MFMAs.  Here every thread holds its two 16-byte pieces of ALL eight chunks
// of the next layer in registers (a ring of 8 sets, 64 VGPRs): a set is handed to the LDS stage one chunk before it is needed and
// re-issued for the layer after at once, so a load has a layer's worth of compute (8 chunks) to arrive.  The chunk loop is unrolled
// (static register indices); layer 0 (K = 64: two chunks) is fetched on its own while the ring fills with layer 1.
__global__ __launch_bounds__(512, 1) void sdf_fused16r_fwd_kernel(FsNet16 net, const float* __restrict__ X, int x_ld, int P,
                                                                  float* __restrict__ sdf_out) {
    constexpr int TM = 128;
    __shared__ __attribute__((aligned(16))) __bf16 act[TM * FH_ALD];
    __shared__ __attribute__((aligned(16))) __bf16 emb[TM * FH_ELD];
    __shared__ __attribute__((aligned(16))) __bf16 bst[2 * FH_BSTAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wc = (tid >> 6) & 3;                            // this wave's 64 output columns
    const int wr = tid >> 8;                                  // ... and its 64 rows
    const int li = lane & 31, lh = lane >> 5;
    const int c4 = tid & 3, r0 = tid >> 2;                    // weight loader: rows r0 and r0 + 128, 16-byte slot c4 of the 64-byte chunk row
    const int a_off = (wr * 64 + li) * FH_ALD + 8 * lh;
    const int b_off = (wc * 64 + li) * FH_BLD + 8 * lh;
    const int w_off = r0 * FH_BLD + 8 * c4;
    const int ntiles = (P + TM - 1) / TM;
    const long long woff = (long long)r0 * 256 + 8 * c4;      // this thread's slot in a [256][256] weight table (layers 1..7)

    f32x4 ring[8][2];
    auto ring_load = [&](int i, const __bf16* W) {            // chunk i of the layer whose table is W
        ring[i][0] = *reinterpret_cast<const f32x4*>(W + woff + 32 * i);
        ring[i][1] = *reinterpret_cast<const f32x4*>(W + woff + 32 * i + 128LL * 256);
    };
    auto ring_store = [&](int i, int st) {
        *reinterpret_cast<f32x4*>(&bst[st * FH_BSTAGE + w_off]) = ring[i][0];
        *reinterpret_cast<f32x4*>(&bst[st * FH_BSTAGE + w_off + 128 * FH_BLD]) = ring[i][1];
    };
    auto chunk_mfma = [&](f32x16 (&acc)[2][2], int st, int k0) {
        const __bf16* sc = &bst[st * FH_BSTAGE];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&sc[b_off + 16 * ks]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&sc[b_off + 32 * FH_BLD + 16 * ks]);
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&act[a_off + k0 + 16 * ks]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&act[a_off + 32 * FH_ALD + k0 + 16 * ks]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    auto epilogue = [&](f32x16 (&acc)[2][2], int l) {          // h = bf16(softplus(acc + bias)) back over the activation tile
        const int N = net.N[l];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = wc * 64 + tn * 32 + li;
            const float bv = col < N ? net.bias[l][col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (col < N) act[row * FH_ALD + col] = (__bf16)nu_softplus100_fast(acc[tm][tn][r] + bv);
                }
        }
        if (l == 3) {               // layer 4's input: [h4 (217) | embedding (39)]
            for (int idx = tid; idx < TM * 39; idx += 512) {
                const int r = idx / 39, c = idx - r * 39;
                act[r * FH_ALD + 217 + c] = emb[r * FH_ELD + c];
            }
        }
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        // the ring fills with layer 1 while the tile's embedding is computed and layer 0 runs
#pragma unroll
        for (int i = 0; i < 8; ++i) ring_load(i, net.Wp[1]);
        for (int idx = tid; idx < TM * 64; idx += 512) {
            const int r = idx >> 6, c = idx & 63;
            int p = row0 + r;
            p = p < P ? p : P - 1;
            float x[3] = {X[(long long)p * x_ld], X[(long long)p * x_ld + 1], X[(long long)p * x_ld + 2]};
            const __bf16 v = (__bf16)(c < 39 ? fs_embed_col(x, c) : 0.f);
            act[r * FH_ALD + c] = v;
            if (c < FH_ELD) emb[r * FH_ELD + c] = v;
        }
        {   // layer 0: [256][64] bf16, both chunks fetched here
            f32x4 t0[2], t1[2];
            const __bf16* W0 = net.Wp[0] + (long long)r0 * 64 + 8 * c4;
            t0[0] = *reinterpret_cast<const f32x4*>(W0);           t0[1] = *reinterpret_cast<const f32x4*>(W0 + 128LL * 64);
            t1[0] = *reinterpret_cast<const f32x4*>(W0 + 32);      t1[1] = *reinterpret_cast<const f32x4*>(W0 + 32 + 128LL * 64);
            *reinterpret_cast<f32x4*>(&bst[w_off]) = t0[0];
            *reinterpret_cast<f32x4*>(&bst[w_off + 128 * FH_BLD]) = t0[1];
            *reinterpret_cast<f32x4*>(&bst[FH_BSTAGE + w_off]) = t1[0];
            *reinterpret_cast<f32x4*>(&bst[FH_BSTAGE + w_off + 128 * FH_BLD]) = t1[1];
            __syncthreads();
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            chunk_mfma(acc, 0, 0);
            chunk_mfma(acc, 1, 32);
            __syncthreads();            // every wave is done with the tile's embedding columns and both stages
            ring_store(0, 0);           // layer 1's first chunk (visible after the epilogue's barrier)
            ring_load(0, net.Wp[2]);
            epilogue(acc, 0);
            __syncthreads();
        }
        for (int l = 1; l < 8; ++l) {
            const __bf16* Wn = net.Wp[l < 7 ? l + 1 : 7];      // the layer after this one (layer 7: re-reads its own table, never used)
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                if (kt + 1 < 8) {       // chunk kt + 1 -> the other stage (its readers left at the last barrier); its set goes out again
                    ring_store(kt + 1, (kt + 1) & 1);
                    ring_load(kt + 1, Wn);
                }
                chunk_mfma(acc, kt & 1, kt * 32);
                __syncthreads();
            }
            if (l < 7) {                // the next layer's first chunk (stage 0 was last read at chunk 6)
                ring_store(0, 0);
                ring_load(0, net.Wp[l < 6 ? l + 2 : 7]);
            }
            epilogue(acc, l);
            __syncthreads();
        }
        {   // sdf head: one wave per row, the arithmetic of skinny_fwd_kernel<1, 256, true>
#pragma clang fp contract(off)
            const f32x4 w = *reinterpret_cast<const f32x4*>(net.w8 + 4 * lane);
            const float b = net.b8[0];
            for (int r = tid >> 6; r < TM; r += 8) {
                const uint2 raw = *reinterpret_cast<const uint2*>(&act[r * FH_ALD + 4 * lane]);
                f32x4 h;
                h[0] = __uint_as_float(raw.x << 16); h[1] = __uint_as_float(raw.x & 0xffff0000u);
                h[2] = __uint_as_float(raw.y << 16); h[3] = __uint_as_float(raw.y & 0xffff0000u);
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

extern "C" int nu_sdf_fused16_fwd(const NuSdfNet* net, const float* X, int x_ld, int P, float* sdf, hipStream_t stream) {
    if (P <= 0) return NU_OK;
    if (!net || !X || !sdf || x_ld < 3) return NU_ERR_ARG;
    FsNet16 n;
    for (int l = 0; l < 8; ++l) {
        const NuLin& L = net->lin[l];
        const int Kexp = l == 0 ? 64 : 256;
        if (L.Kp != Kexp || L.N > 256 || L.N < 1 || !L.Wp16 || !L.bias) return NU_ERR_ARG;
        n.Wp[l] = static_cast<const __bf16*>(L.Wp16); n.bias[l] = L.bias; n.Kp[l] = L.Kp; n.N[l] = L.N;
    }
    if (net->lin[3].N != 217 || net->lin[8].Kp != 256) return NU_ERR_ARG;
    n.w8 = net->lin[8].Wp; n.b8 = net->lin[8].bias;      // the 1-wide head stays fp32 (skinny_fwd_h16)
    static const int tm_env = getenv("NU_FUSED_SDF16_TM") ? atoi(getenv("NU_FUSED_SDF16_TM")) : 0;     // development switch: 32 / 64 / 128
    const int tm = tm_env ? tm_env : (nu_cdiv(P, 128) >= 192 ? 128 : (nu_cdiv(P, 64) >= 384 ? 64 : 32));
    const int ntiles = nu_cdiv(P, tm);
    static const bool ring_off = getenv("NU_FUSED_SDF16_RING") && atoi(getenv("NU_FUSED_SDF16_RING")) == 0;    // development switch (A/B)
    if (tm == 128 && !ring_off) {
        hipLaunchKernelGGL(sdf_fused16r_fwd_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(512), 0, stream, n, X, x_ld, P, sdf);
    } else if (tm == 128) {
        hipLaunchKernelGGL((sdf_fused16_fwd_kernel<2, 2>), dim3(ntiles < 256 ? ntiles : 256), dim3(512), 0, stream, n, X, x_ld, P, sdf);
    } else {
        const int grid = ntiles < 512 ? ntiles : 512;
        if (tm == 64) hipLaunchKernelGGL((sdf_fused16_fwd_kernel<2, 1>), dim3(grid), dim3(256), 0, stream, n, X, x_ld, P, sdf);
        else hipLaunchKernelGGL((sdf_fused16_fwd_kernel<1, 1>), dim3(grid), dim3(256), 0, stream, n, X, x_ld, P, sdf);
    }
    return nu_launch_status();
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
