// gemm_tn.hip -- weight-gradient (TN) GEMM kernels, the deterministic split reduction and the raw-GEMM C ABI (gfx950).
// See gemm.h for the contract.
//
// Replaces the autograd weight / bias gradients of every `lin(x)` in network/field.py:133-150, :158-170, :265-289, :371-408.
#include "gemm_epi.h"
#include <stdlib.h>

// development aid: {shader cycles, 100 MHz wall ticks} of block 0 of the last mfma_peak launch
__device__ unsigned long long nu_dbg_clk[2];

// ------------------------------------------------------------------------------------------------
// TN kernel (weight gradients): split over the reduced (point) dimension, partial slabs out.
// ------------------------------------------------------------------------------------------------
// The operands are transposed on their way into LDS ([column][k], k contiguous, row stride 36): each
// thread fetches 16 consecutive reduced rows of ONE column (a wave load = 256 contiguous bytes of one row), so the
// inner loop is exactly the NT kernel's (one ds_read_b128 feeds four MFMA k-steps) and nothing consumes a global
// load before the hand-over to LDS -- the loads stay in flight under the 64 MFMAs of the current chunk.
// BIG = operands of 4 GiB or more (64-bit element offsets instead of one uniform base + a 32-bit byte offset).
// Several weight-gradient problems in ONE launch: the (tile, split) blocks of problem i are the linear block ids
// [blk0[i], blk0[i + 1]), tile fastest.  Each problem keeps its own operands, extents, split count and slab (NuGemmTN).
#define NU_TN_BATCH_MAX 16
struct NuGemmTNBatch {
    NuGemmTN p[NU_TN_BATCH_MAX];
    int blk0[NU_TN_BATCH_MAX + 1];
    int n, pad_;
};
// which problem a block of a batched launch belongs to, and its (tile, split) there; `tiles` = output tiles of that problem
static __device__ __forceinline__ int tn_batch_decode(const NuGemmTNBatch& b, int tile_edge, int& bx, int& split) {
    int pi = 0;
    for (int i = 1; i < b.n; ++i) pi = ((int)blockIdx.x >= b.blk0[i]) ? i : pi;
    const int tiles = ((b.p[pi].N1 + tile_edge - 1) / tile_edge) * ((b.p[pi].N2 + tile_edge - 1) / tile_edge);
    const int local = (int)blockIdx.x - b.blk0[pi];
    split = local / tiles;
    bx = local - split * tiles;
    return pi;
}

template <bool BIG, int PREC>
static __device__ __forceinline__ void tn_body(const NuGemmTN& g, const int bx, const int split, const int grp) {
    constexpr bool BF16 = PREC == 1;
    constexpr bool SPLIT = PREC == 2;                 // exact three-way bf16 split, six partial products (see the NT kernel)
    constexpr int kPlane = 128 * NT_LDSH;
    __shared__ __attribute__((aligned(16))) float smem[SPLIT ? 1 : 2][SPLIT ? 3 * kPlane : 128 * NT_LDS];
    __bf16* const hA = reinterpret_cast<__bf16*>(&smem[0][0]);      // bf16 image(s), as in the NT kernel
    __bf16* const hB = hA + (SPLIT ? 3 : 1) * kPlane;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int t2 = (g.N2 + 127) / 128;
    // (tile, split) = (blockIdx.x, blockIdx.y).  Every tile of one split reads the same p-range of both operands, and with this
    // grid a split's tiles land on different XCDs: rocprofv3 FETCH_SIZE (x2-corrected; calibrated on this access width by
    // scripts/gemm_lab calib) shows 1.9x the algorithmic bytes per launch leaving the L2s.  Giving a split's tiles linear ids
    // that are congruent mod 8 (one XCD) was measured 7 % SLOWER in the step (108.9 -> 101.7 TFLOP/s) and is not done.
    const int n1t = bx / t2, n2t = bx - n1t * t2;
    const int n1_0 = n1t * 128, n2_0 = n2t * 128;
    const int N1p = ((g.N1 + 127) / 128) * 128, N2p = t2 * 128;

    int rows_per = (g.P + g.S - 1) / g.S;
    rows_per = ((rows_per + TBK - 1) / TBK) * TBK;
    const int p_begin = split * rows_per;
    int p_end = p_begin + rows_per;
    p_end = p_end < g.P ? p_end : g.P;
    const int ntile = p_end > p_begin ? (p_end - p_begin + TBK - 1) / TBK : 0;
    const int npair = g.A1 ? 2 : 1;
    const int total = ntile * npair;

    const int c = tid & 127;    // column of the 128-wide operand tile this thread fetches
    const int kg = tid >> 7;    // which 16 of the chunk's 32 reduced rows (wave-uniform)
    const bool do_bias = (g.bias_slab != nullptr) && (n2t == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra4[4], rb4[4];
    int nvalid = 16;            // rows of the pending chunk that exist (ragged tail of the last split)
    bool pend_pair0 = true;
    float bs = 0.f;
    auto load_tile = [&](int t) {
        const int pair = t >= ntile ? 1 : 0;
        const int kt = t - pair * ntile;
        // bf16-stored operands (mode 1 only; flags per operand): 2-byte elements, widened exactly on load
        const bool a16 = BF16 && (g.bf16 & (pair ? NU_TN_A1_16 : NU_TN_A0_16)) != 0;
        const bool b16 = BF16 && (g.bf16 & (pair ? NU_TN_B1_16 : NU_TN_B0_16)) != 0;
        const unsigned eA = a16 ? 2u : 4u, eB = b16 ? 2u : 4u;
        const char* __restrict__ A = (const char*)(pair ? g.A1 : g.A0) + (long long)grp * (pair ? g.sA1 : g.sA0) * eA;
        const char* __restrict__ B = (const char*)(pair ? g.B1 : g.B0) + (long long)grp * (pair ? g.sB1 : g.sB0) * eB;
        const int lda = pair ? g.lda1 : g.lda0;
        const int ldb = pair ? g.ldb1 : g.ldb0;
        int ca = n1_0 + c, cb = n2_0 + c;
        ca = ca < lda ? ca : lda - 1;      // columns past the operand only feed slab rows the reducer never reads
        cb = cb < ldb ? cb : ldb - 1;
        const int pbase = p_begin + kt * TBK + kg * 16;
        nvalid = p_end - pbase;
        pend_pair0 = pair == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int pr = pbase + 4 * i + e;
                pr = pr < p_end ? pr : p_end - 1;   // ragged tail: re-read the last row, zeroed at the hand-over
                const char* pa;
                const char* pb;
                if (BIG) {
                    pa = A + ((long long)pr * lda + ca) * eA;
                    pb = B + ((long long)pr * ldb + cb) * eB;
                } else {
                    pa = A + ((unsigned)pr * (unsigned)lda + (unsigned)ca) * eA;
                    pb = B + ((unsigned)pr * (unsigned)ldb + (unsigned)cb) * eB;
                }
                ra4[i][e] = a16 ? __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(pa) << 16) : *reinterpret_cast<const float*>(pa);
                rb4[i][e] = b16 ? __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(pb) << 16) : *reinterpret_cast<const float*>(pb);
            }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = 4 * i + e < nvalid;
                ra4[i][e] = ok ? ra4[i][e] : 0.f;
                rb4[i][e] = ok ? rb4[i][e] : 0.f;
            }
        if (do_bias && pend_pair0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bs += (ra4[i][0] + ra4[i][1]) + (ra4[i][2] + ra4[i][3]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SPLIT) {
                bf16x4 p1, p2, p3;
                nu_split3(ra4[i], p1, p2, p3);
                __bf16* q = &hA[c * NT_LDSH + kg * 16 + 4 * i];
                *reinterpret_cast<bf16x4*>(q) = p1;
                *reinterpret_cast<bf16x4*>(q + kPlane) = p2;
                *reinterpret_cast<bf16x4*>(q + 2 * kPlane) = p3;
                nu_split3(rb4[i], p1, p2, p3);
                q = &hB[c * NT_LDSH + kg * 16 + 4 * i];
                *reinterpret_cast<bf16x4*>(q) = p1;
                *reinterpret_cast<bf16x4*>(q + kPlane) = p2;
                *reinterpret_cast<bf16x4*>(q + 2 * kPlane) = p3;
            } else if (BF16) {
                *reinterpret_cast<bf16x4*>(&hA[c * NT_LDSH + kg * 16 + 4 * i]) = nu_to_bf16x4(ra4[i]);
                *reinterpret_cast<bf16x4*>(&hB[c * NT_LDSH + kg * 16 + 4 * i]) = nu_to_bf16x4(rb4[i]);
            } else {
                float* s0 = &smem[0][0];
                *reinterpret_cast<f32x4*>(&s0[c * NT_LDS + kg * 16 + 4 * i]) = ra4[i];
                *reinterpret_cast<f32x4*>(&s0[128 * NT_LDS + c * NT_LDS + kg * 16 + 4 * i]) = rb4[i];
            }
        }
    };

    if (total > 0) {
        load_tile(0);
        store_tile();
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 64 + li) * NT_LDS + 4 * lh;
    const int b_off = (wc * 64 + li) * NT_LDS + 4 * lh;
    const int ah_off = (wr * 64 + li) * NT_LDSH + 8 * lh;
    const int bh_off = (wc * 64 + li) * NT_LDSH + 8 * lh;
    // which of the wave's four 32 x 32 blocks hold at least one real output (wave-uniform)
    const int wr_u = __builtin_amdgcn_readfirstlane(wr), wc_u = __builtin_amdgcn_readfirstlane(wc);
    const bool vr0 = n1_0 + wr_u * 64 < g.N1, vr1 = n1_0 + wr_u * 64 + 32 < g.N1;
    const bool vc0 = n2_0 + wc_u * 64 < g.N2, vc1 = n2_0 + wc_u * 64 + 32 < g.N2;
    const bool v00 = vr0 && vc0, v01 = vr0 && vc1, v10 = vr1 && vc0, v11 = vr1 && vc1;
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) load_tile(t + 1);
        if (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[2][3], b[2][3];
#pragma unroll
                for (int t2_ = 0; t2_ < 2; ++t2_)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        a[t2_][p] = *reinterpret_cast<const bf16x8*>(&hA[p * kPlane + ah_off + 32 * t2_ * NT_LDSH + 16 * ks]);
                        b[t2_][p] = *reinterpret_cast<const bf16x8*>(&hB[p * kPlane + bh_off + 32 * t2_ * NT_LDSH + 16 * ks]);
                    }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        f32x16 cacc = acc[tm][tn];
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], cacc, 0, 0, 0);
                        acc[tm][tn] = cacc;
                    }
            }
        } else if (BF16) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&hA[ah_off + 16 * ks]);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&hA[ah_off + 32 * NT_LDSH + 16 * ks]);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&hB[bh_off + 16 * ks]);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&hB[bh_off + 32 * NT_LDSH + 16 * ks]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float* As = &smem[0][0];
                const float* Bs = As + 128 * NT_LDS;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[a_off + kk * 8]);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(&As[a_off + 32 * NT_LDS + kk * 8]);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bs[b_off + kk * 8]);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bs[b_off + 32 * NT_LDS + kk * 8]);
                // (32 x 32 blocks that lie wholly in the padding of N1 / N2 -- 288 = 2 x 128 + 32, 96, 217, 257 = 2 x 128 + 1 -- are
                // skipped, wave-uniformly: their slab entries are never read)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (v00) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                    if (v01) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                    if (v10) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                    if (v11) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (t + 1 < total) {
            store_tile();
            __syncthreads();
        }
    }

    float* __restrict__ slab = g.slab + (long long)grp * g.sSlab + (long long)split * N1p * N2p;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n2_0 + wc * 64 + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n1_0 + wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(long long)row * N2p + col] = acc[tm][tn][r];
            }
        }

    if (do_bias) {
        float* red = &smem[0][0];
        red[kg * 128 + c] = bs;
        __syncthreads();
        if (tid < 128) g.bias_slab[(long long)grp * g.sBiasSlab + (long long)split * N1p + n1_0 + tid] = red[tid] + red[128 + tid];
    }
}

template <bool BIG, int PREC>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(NuGemmTN g) { tn_body<BIG, PREC>(g, blockIdx.x, blockIdx.y, blockIdx.z); }
template <bool BIG>
__global__ __launch_bounds__(256, 2) void gemm_tnb_kernel(NuGemmTNBatch b) {
    int bx, split;
    const int pi = tn_batch_decode(b, 128, bx, split);
    tn_body<BIG, 0>(b.p[pi], bx, split, 0);
}

// ------------------------------------------------------------------------------------------------
// TN kernel, bf16 arithmetic (mode 1), operands stored as bf16 or fp32 per flag.  The first-generation mode-1 kernel above
// fetches one scalar per lane and transposes on the way into LDS; with 2-byte elements that is 4x slower than with floats
// (measured: the load instruction count stays, the bytes per instruction halve).  Here each lane fetches 16 bytes of ONE row
// (8 bf16 columns, or 2 x 16 bytes of fp32 rounded on the way in), the LDS image stays row-major [p][128 columns] and the
// MFMA operands -- 8 consecutive p of one column per lane -- come out of it through ds_read_b64_tr_b16, the hardware
// transposed read: per 16-lane group a 4 (p) x 16 (column) block, lane 4q + c supplying the address of row q, columns
// 4c .. 4c+3.  Rows are 320 bytes (256 + 64 pad) so the four rows of a block sit 16 banks apart: conflict-free reads.
// 20 KB of LDS, chunk = 32 reduced rows (two MFMA k-steps), next chunk in registers under the MFMAs.
// ------------------------------------------------------------------------------------------------
#define TN16_ROWB 320
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ bf16x8 tn16_frag(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * TN16_ROWB));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256, 3) void gemm_tn16_kernel(NuGemmTN g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * TBK * TN16_ROWB];
    char* const sA = smem;
    char* const sB = smem + TBK * TN16_ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int t2 = (g.N2 + 127) / 128;
    const int n1t = blockIdx.x / t2, n2t = blockIdx.x - n1t * t2;
    const int split = blockIdx.y;
    const int n1_0 = n1t * 128, n2_0 = n2t * 128;
    const int grp = blockIdx.z;
    const int N1p = ((g.N1 + 127) / 128) * 128, N2p = t2 * 128;

    int rows_per = (g.P + g.S - 1) / g.S;
    rows_per = ((rows_per + TBK - 1) / TBK) * TBK;
    const int p_begin = split * rows_per;
    int p_end = p_begin + rows_per;
    p_end = p_end < g.P ? p_end : g.P;
    const int ntile = p_end > p_begin ? (p_end - p_begin + TBK - 1) / TBK : 0;
    const int npair = g.A1 ? 2 : 1;
    const int total = ntile * npair;

    const int cg = tid & 15;        // 8-column group of the 128-wide operand tile
    const int rr = tid >> 4;        // row inside a 16-row pass
    const bool do_bias = (g.bias_slab != nullptr) && (n2t == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    uint4 ra[2][2], rb[2][2];       // [pass][half]: a bf16 operand uses half 0 only
    int nvalid = 0;                 // rows of the pending chunk that exist (ragged tail of the last split)
    bool pend_pair0 = true, pa16 = false, pb16 = false;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int t) {
        const int pair = t >= ntile ? 1 : 0;
        const int kt = t - pair * ntile;
        const bool a16 = (g.bf16 & (pair ? NU_TN_A1_16 : NU_TN_A0_16)) != 0;
        const bool b16 = (g.bf16 & (pair ? NU_TN_B1_16 : NU_TN_B0_16)) != 0;
        const int eA = a16 ? 2 : 4, eB = b16 ? 2 : 4;
        const char* __restrict__ A = (const char*)(pair ? g.A1 : g.A0) + (long long)grp * (pair ? g.sA1 : g.sA0) * eA;
        const char* __restrict__ B = (const char*)(pair ? g.B1 : g.B0) + (long long)grp * (pair ? g.sB1 : g.sB0) * eB;
        const int lda = pair ? g.lda1 : g.lda0;
        const int ldb = pair ? g.ldb1 : g.ldb0;
        int ca = n1_0 + 8 * cg, cb = n2_0 + 8 * cg;
        ca = ca <= lda - 8 ? ca : lda - 8;      // column groups past the operand only feed slab rows the reducer never reads
        cb = cb <= ldb - 8 ? cb : ldb - 8;
        const int pbase = p_begin + kt * TBK;
        nvalid = p_end - pbase;
        pend_pair0 = pair == 0;
        pa16 = a16; pb16 = b16;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            int pr = pbase + rr + 16 * ps;
            pr = pr < p_end ? pr : p_end - 1;   // ragged tail: re-read the last row, zeroed at the hand-over
            const char* pa = A + ((long long)pr * lda + ca) * eA;
            const char* pb = B + ((long long)pr * ldb + cb) * eB;
            ra[ps][0] = *reinterpret_cast<const uint4*>(pa);
            if (!a16) ra[ps][1] = *reinterpret_cast<const uint4*>(pa + 16);
            rb[ps][0] = *reinterpret_cast<const uint4*>(pb);
            if (!b16) rb[ps][1] = *reinterpret_cast<const uint4*>(pb + 16);
        }
    };
    auto pack8 = [](uint4 lo, uint4 hi) -> uint4 {       // 8 fp32 -> 8 bf16 (RNE)
        const bf16x4 l = nu_to_bf16x4(__builtin_bit_cast(f32x4, lo)), h = nu_to_bf16x4(__builtin_bit_cast(f32x4, hi));
        const uint2 l2 = __builtin_bit_cast(uint2, l), h2 = __builtin_bit_cast(uint2, h);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const bool ok = rr + 16 * ps < nvalid;
            uint4 va = pa16 ? ra[ps][0] : pack8(ra[ps][0], ra[ps][1]);
            uint4 vb = pb16 ? rb[ps][0] : pack8(rb[ps][0], rb[ps][1]);
            if (do_bias && pend_pair0 && ok) {           // column sums of the operand as stored (fp32 operands: unrounded)
                if (pa16) {
                    const unsigned w[4] = {ra[ps][0].x, ra[ps][0].y, ra[ps][0].z, ra[ps][0].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bs[2 * e] += __uint_as_float(w[e] << 16);
                        bs[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                    }
                } else {
                    const f32x4 l = __builtin_bit_cast(f32x4, ra[ps][0]), h = __builtin_bit_cast(f32x4, ra[ps][1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bs[e] += l[e]; bs[4 + e] += h[e]; }
                }
            }
            if (!ok) { va = make_uint4(0u, 0u, 0u, 0u); vb = va; }
            *reinterpret_cast<uint4*>(sA + (rr + 16 * ps) * TN16_ROWB + cg * 16) = va;
            *reinterpret_cast<uint4*>(sB + (rr + 16 * ps) * TN16_ROWB + cg * 16) = vb;
        }
    };

    if (total > 0) {
        load_tile(0);
        store_tile();
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    const int g4 = lane >> 4, tq = (lane & 15) >> 2, tc = lane & 3;
    // transposed-read address of this lane: row 8 (g4 >> 1) + q of the k-step, columns 16 (g4 & 1) + 4 c of the 32-column tile
    const int a_off = (8 * (g4 >> 1) + tq) * TN16_ROWB + (wr * 64 + 16 * (g4 & 1) + 4 * tc) * 2;
    const int b_off = (8 * (g4 >> 1) + tq) * TN16_ROWB + (wc * 64 + 16 * (g4 & 1) + 4 * tc) * 2;
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) load_tile(t + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 a0 = tn16_frag(sA + a_off + ks * 16 * TN16_ROWB);
            const bf16x8 a1 = tn16_frag(sA + a_off + ks * 16 * TN16_ROWB + 64);
            const bf16x8 b0 = tn16_frag(sB + b_off + ks * 16 * TN16_ROWB);
            const bf16x8 b1 = tn16_frag(sB + b_off + ks * 16 * TN16_ROWB + 64);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (t + 1 < total) {
            store_tile();
            __syncthreads();
        }
    }

    float* __restrict__ slab = g.slab + (long long)grp * g.sSlab + (long long)split * N1p * N2p;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n2_0 + wc * 64 + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n1_0 + wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(long long)row * N2p + col] = acc[tm][tn][r];
            }
        }

    if (do_bias) {                  // 16 row-threads hold partial sums of the same 8 columns
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rr * 128 + 8 * cg + e] = bs[e];
        __syncthreads();
        if (tid < 128) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += red[r * 128 + tid];
            g.bias_slab[(long long)grp * g.sBiasSlab + (long long)split * N1p + n1_0 + tid] = sum;
        }
    }
}

// The same kernel with a 256 x 256 output tile (512 threads, 8 waves as 4 x 2, wave tile 64 x 128, 128 accumulator VGPRs) for the
// shapes whose N1 and N2 are multiples of 256: every operand row crosses HBM ONCE per split instead of once per output-tile row /
// column -- these launches are HBM-bound, the 128-wide kernel's doubled operand traffic is what they wait for.  Rows are 576 bytes.
#define TN16B_ROWB 576
static __device__ __forceinline__ bf16x8 tn16b_frag(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * TN16B_ROWB));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__global__ __launch_bounds__(512, 1) void gemm_tn16x256_kernel(NuGemmTN g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * TBK * TN16B_ROWB];
    char* const sA = smem;
    char* const sB = smem + TBK * TN16B_ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;     // 8 waves as 4 (n1) x 2 (n2); wave tile 64 x 128
    const int t2 = g.N2 / 256;
    const int n1t = blockIdx.x / t2, n2t = blockIdx.x - n1t * t2;
    const int split = blockIdx.y;
    const int n1_0 = n1t * 256, n2_0 = n2t * 256;
    const int grp = blockIdx.z;
    const int N1p = g.N1, N2p = g.N2;                 // multiples of 256

    int rows_per = (g.P + g.S - 1) / g.S;
    rows_per = ((rows_per + TBK - 1) / TBK) * TBK;
    const int p_begin = split * rows_per;
    int p_end = p_begin + rows_per;
    p_end = p_end < g.P ? p_end : g.P;
    const int ntile = p_end > p_begin ? (p_end - p_begin + TBK - 1) / TBK : 0;
    const int npair = g.A1 ? 2 : 1;
    const int total = ntile * npair;

    const int cg = tid & 31;        // 8-column group of the 256-wide operand tile
    const int rr = tid >> 5;        // row inside a 16-row pass
    const bool do_bias = (g.bias_slab != nullptr) && (n2t == 0);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    uint4 ra[2][2], rb[2][2];       // [pass][half]: a bf16 operand uses half 0 only
    int nvalid = 0;                 // rows of the pending chunk that exist (ragged tail of the last split)
    bool pend_pair0 = true, pa16 = false, pb16 = false;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int t) {
        const int pair = t >= ntile ? 1 : 0;
        const int kt = t - pair * ntile;
        const bool a16 = (g.bf16 & (pair ? NU_TN_A1_16 : NU_TN_A0_16)) != 0;
        const bool b16 = (g.bf16 & (pair ? NU_TN_B1_16 : NU_TN_B0_16)) != 0;
        const int eA = a16 ? 2 : 4, eB = b16 ? 2 : 4;
        const char* __restrict__ A = (const char*)(pair ? g.A1 : g.A0) + (long long)grp * (pair ? g.sA1 : g.sA0) * eA;
        const char* __restrict__ B = (const char*)(pair ? g.B1 : g.B0) + (long long)grp * (pair ? g.sB1 : g.sB0) * eB;
        const int lda = pair ? g.lda1 : g.lda0;
        const int ldb = pair ? g.ldb1 : g.ldb0;
        int ca = n1_0 + 8 * cg, cb = n2_0 + 8 * cg;
        ca = ca <= lda - 8 ? ca : lda - 8;      // column groups past the operand only feed slab rows the reducer never reads
        cb = cb <= ldb - 8 ? cb : ldb - 8;
        const int pbase = p_begin + kt * TBK;
        nvalid = p_end - pbase;
        pend_pair0 = pair == 0;
        pa16 = a16; pb16 = b16;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            int pr = pbase + rr + 16 * ps;
            pr = pr < p_end ? pr : p_end - 1;   // ragged tail: re-read the last row, zeroed at the hand-over
            const char* pa = A + ((long long)pr * lda + ca) * eA;
            const char* pb = B + ((long long)pr * ldb + cb) * eB;
            ra[ps][0] = *reinterpret_cast<const uint4*>(pa);
            if (!a16) ra[ps][1] = *reinterpret_cast<const uint4*>(pa + 16);
            rb[ps][0] = *reinterpret_cast<const uint4*>(pb);
            if (!b16) rb[ps][1] = *reinterpret_cast<const uint4*>(pb + 16);
        }
    };
    auto pack8 = [](uint4 lo, uint4 hi) -> uint4 {       // 8 fp32 -> 8 bf16 (RNE)
        const bf16x4 l = nu_to_bf16x4(__builtin_bit_cast(f32x4, lo)), h = nu_to_bf16x4(__builtin_bit_cast(f32x4, hi));
        const uint2 l2 = __builtin_bit_cast(uint2, l), h2 = __builtin_bit_cast(uint2, h);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const bool ok = rr + 16 * ps < nvalid;
            uint4 va = pa16 ? ra[ps][0] : pack8(ra[ps][0], ra[ps][1]);
            uint4 vb = pb16 ? rb[ps][0] : pack8(rb[ps][0], rb[ps][1]);
            if (do_bias && pend_pair0 && ok) {           // column sums of the operand as stored (fp32 operands: unrounded)
                if (pa16) {
                    const unsigned w[4] = {ra[ps][0].x, ra[ps][0].y, ra[ps][0].z, ra[ps][0].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bs[2 * e] += __uint_as_float(w[e] << 16);
                        bs[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                    }
                } else {
                    const f32x4 l = __builtin_bit_cast(f32x4, ra[ps][0]), h = __builtin_bit_cast(f32x4, ra[ps][1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bs[e] += l[e]; bs[4 + e] += h[e]; }
                }
            }
            if (!ok) { va = make_uint4(0u, 0u, 0u, 0u); vb = va; }
            *reinterpret_cast<uint4*>(sA + (rr + 16 * ps) * TN16B_ROWB + cg * 16) = va;
            *reinterpret_cast<uint4*>(sB + (rr + 16 * ps) * TN16B_ROWB + cg * 16) = vb;
        }
    };

    if (total > 0) {
        load_tile(0);
        store_tile();
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    const int g4 = lane >> 4, tq = (lane & 15) >> 2, tc = lane & 3;
    // transposed-read address of this lane: row 8 (g4 >> 1) + q of the k-step, columns 16 (g4 & 1) + 4 c of the 32-column tile
    const int a_off = (8 * (g4 >> 1) + tq) * TN16B_ROWB + (wr * 64 + 16 * (g4 & 1) + 4 * tc) * 2;
    const int b_off = (8 * (g4 >> 1) + tq) * TN16B_ROWB + (wc * 128 + 16 * (g4 & 1) + 4 * tc) * 2;
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) load_tile(t + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 a0 = tn16b_frag(sA + a_off + ks * 16 * TN16B_ROWB);
            const bf16x8 a1 = tn16b_frag(sA + a_off + ks * 16 * TN16B_ROWB + 64);
            bf16x8 b[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) b[tn] = tn16b_frag(sB + b_off + ks * 16 * TN16B_ROWB + 64 * tn);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[tn], acc[0][tn], 0, 0, 0);
                acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[tn], acc[1][tn], 0, 0, 0);
            }
        }
        __syncthreads();
        if (t + 1 < total) {
            store_tile();
            __syncthreads();
        }
    }

    float* __restrict__ slab = g.slab + (long long)grp * g.sSlab + (long long)split * N1p * N2p;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int col = n2_0 + wc * 128 + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n1_0 + wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(long long)row * N2p + col] = acc[tm][tn][r];
            }
        }

    if (do_bias) {                  // 16 row-threads hold partial sums of the same 8 columns
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rr * 256 + 8 * cg + e] = bs[e];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += red[r * 256 + tid];
            g.bias_slab[(long long)grp * g.sBiasSlab + (long long)split * N1p + n1_0 + tid] = sum;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// TN kernel, exact fp32, 256 x 256 output tile per workgroup (512 threads, 8 waves as 4 x 2, each wave 64 x 128 = 2 x 4
// MFMA tiles, 128 accumulator VGPRs; one workgroup per CU).  For the dominant weight-gradient shape (N1, N2 multiples of
// 256) every operand row is fetched ONCE per split instead of once per output-tile row / column (the 128 x 128 kernel above
// moved 1.9x its algorithmic bytes out of the L2s, profiles/r02/traffic_pmc.json) and the 32 transposing scalar loads per
// thread and chunk now feed 128 MFMAs per wave instead of 64: half the load-issue time per matrix cycle.
// Same slab layout, same bias sums, same ragged-tail rules as gemm_tn_kernel.
// ------------------------------------------------------------------------------------------------
// NJ = 4: the 256 x 256 tile (512 threads, 8 waves as 4 x 2, wave tile 64 x 128).  NJ = 2: the same pipeline on a 128 x 128 tile (256
// threads, 4 waves as 2 x 2, wave tile 64 x 64, two workgroups per CU) for the shapes the big tile does not fit (N1 or N2 not a
// multiple of 256, or too few tiles to fill the chip) -- measured no faster there than the first-generation gemm_tn_kernel, which
// stays the default for those shapes (see nu_gemm_tn_launch).
template <bool BIG, int NJ>
static __device__ __forceinline__ void tn2_body(const NuGemmTN& g, const int bx, const int split, const int grp) {
    constexpr int T = 64 * NJ;                       // tile edge: 256 or 128
    __shared__ __attribute__((aligned(16))) float smem[2][2][T * NT_LDS];     // [stage][A | B][column][k (+4 pad)]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;           // wave tile: rows (N1) 64 wr .. +64, columns (N2) 32 NJ wc .. + 32 NJ
    const int t2 = (g.N2 + T - 1) / T;
    const int n1t = bx / t2, n2t = bx - n1t * t2;
    const int n1_0 = n1t * T, n2_0 = n2t * T;
    const int N1p = ((g.N1 + 127) / 128) * 128, N2p = ((g.N2 + 127) / 128) * 128;      // slab extents (nu_wgrad_workspace_bytes)

    int rows_per = (g.P + g.S - 1) / g.S;
    rows_per = ((rows_per + TBK - 1) / TBK) * TBK;
    const int p_begin = split * rows_per;
    int p_end = p_begin + rows_per;
    p_end = p_end < g.P ? p_end : g.P;
    const int ntile = p_end > p_begin ? (p_end - p_begin + TBK - 1) / TBK : 0;
    const int npair = g.A1 ? 2 : 1;
    const int total = ntile * npair;

    const int c = tid & (T - 1);    // column of the T-wide operand tile this thread fetches
    const int kg = tid / T;         // which 16 of the chunk's 32 reduced rows (wave-uniform)
    const bool do_bias = (g.bias_slab != nullptr) && (n2t == 0);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra4[4], rb4[4];
    int nvalid = 16;
    bool pend_pair0 = true;
    float bs = 0.f;
    // The loader is cut into pieces that sit BETWEEN groups of 8 MFMAs (pinned with sched_barrier): the two waves of a SIMD leave
    // every barrier in the same phase, so whatever a wave issues in a block at the top of the chunk -- 32 scalar loads, their
    // address arithmetic, 8 ds_write_b128 -- is time in which NEITHER issues MFMAs (measured: matrix pipe 74 % busy).
    const char* __restrict__ ldA = nullptr;
    const char* __restrict__ ldB = nullptr;
    int ld_lda = 0, ld_ldb = 0, ld_ca = 0, ld_cb = 0, ld_pbase = 0, ld_pbase_u = 0;
    int nvalid_ld = 16;
    bool pair0_ld = true;
    auto load_begin = [&](int t) {
        const int pair = t >= ntile ? 1 : 0;
        const int kt = t - pair * ntile;
        ldA = (const char*)(pair ? g.A1 + (long long)grp * g.sA1 : g.A0 + (long long)grp * g.sA0);
        ldB = (const char*)(pair ? g.B1 + (long long)grp * g.sB1 : g.B0 + (long long)grp * g.sB0);
        ld_lda = pair ? g.lda1 : g.lda0;
        ld_ldb = pair ? g.ldb1 : g.ldb0;
        ld_ca = n1_0 + c; ld_cb = n2_0 + c;
        ld_ca = ld_ca < ld_lda ? ld_ca : ld_lda - 1;
        ld_cb = ld_cb < ld_ldb ? ld_cb : ld_ldb - 1;
        ld_pbase = p_begin + kt * TBK + kg * 16;
        ld_pbase_u = p_begin + kt * TBK;                  // (wave-uniform part)
        nvalid_ld = p_end - ld_pbase;
        pair0_ld = pair == 0;
    };
    auto load_piece = [&](int i, int e0, int e1) {          // rows 4 i + e0 .. 4 i + e1 - 1 of this thread's 16
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            int pr = ld_pbase + 4 * i + e;
            pr = pr < p_end ? pr : p_end - 1;
            if (BIG) {
                ra4[i][e] = reinterpret_cast<const float*>(ldA)[(long long)pr * ld_lda + ld_ca];
                rb4[i][e] = reinterpret_cast<const float*>(ldB)[(long long)pr * ld_ldb + ld_cb];
            } else {
                const unsigned oa = ((unsigned)pr * (unsigned)ld_lda + (unsigned)ld_ca) * 4u;
                const unsigned ob = ((unsigned)pr * (unsigned)ld_ldb + (unsigned)ld_cb) * 4u;
                ra4[i][e] = *reinterpret_cast<const float*>(ldA + oa);
                rb4[i][e] = *reinterpret_cast<const float*>(ldB + ob);
            }
        }
    };
    auto store_piece = [&](int st, int i) {                 // the chunk in registers: nvalid / pend_pair0 describe it
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = 4 * i + e < nvalid;
            ra4[i][e] = ok ? ra4[i][e] : 0.f;
            rb4[i][e] = ok ? rb4[i][e] : 0.f;
        }
        if (do_bias && pend_pair0) bs += (ra4[i][0] + ra4[i][1]) + (ra4[i][2] + ra4[i][3]);
        *reinterpret_cast<f32x4*>(&smem[st][0][c * NT_LDS + kg * 16 + 4 * i]) = ra4[i];
        *reinterpret_cast<f32x4*>(&smem[st][1][c * NT_LDS + kg * 16 + 4 * i]) = rb4[i];
    };
    auto load_tile = [&](int t) {
        load_begin(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_piece(i, 0, 4);
        nvalid = nvalid_ld; pend_pair0 = pair0_ld;
    };
    auto store_tile = [&](int st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_piece(st, i);
    };

    // two LDS stages: chunk t+1 waits in registers (fetched during chunk t-1), goes to the other stage in the MIDDLE of chunk t's
    // MFMAs, and the registers are re-issued for chunk t+2 at once: one barrier per chunk, no store between barriers
    if (total > 0) {
        load_tile(0);
        store_tile(0);
        if (total > 1) load_tile(1);
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 64 + li) * NT_LDS + 4 * lh;
    const int b_off = (wc * 32 * NJ + li) * NT_LDS + 4 * lh;
    // Fragments are double-buffered in registers and EVERY memory instruction of a chunk sits alone between two MFMAs (see
    // gemm_nt2_kernel: an in-order wave issues nothing while one of its own instructions issues, and the two waves of a SIMD belong
    // to this one workgroup and run in phase -- whatever one of them issues in a block, the other issues at the same time).
    // A chunk is 16 slots of 8 MFMAs (k-group kk = slot / 4, k-step e = slot % 4):
    //   slot 0 / 4 / 8 / 9      register piece 0 / 1 / 2 / 3 of chunk t+1 -> the other LDS stage (zeroing of the ragged tail, bias sums)
    //   slots 1-2, 5-6, 10 + 12, 13-14   the same pieces re-issued global -> registers for chunk t+2 (one scalar load per gap)
    //   slots 3, 7, 11          the 6 fragment reads of the next k-group; barrier behind slot 11; slot 15: first fragments of chunk t+1
    struct FragT { f32x4 a[2]; f32x4 b[NJ]; };
    FragT F0, F1;
    // One scalar load: the ROW of the operand is wave-uniform (kg = tid / T is the same for a whole wave), so its address is a
    // scalar 64-bit base (two or three SALU instructions) and the lane only adds its column offset -- global_load_dword v, v_off, s[base]
    // -- no per-lane multiply-add in front of each of the 32 loads of a chunk (they cost the narrow tile a third of its issue slots).
    // The rows are visited in increasing order, so the base advances by one row stride per load (and stops at the last row of the
    // split: the ragged tail re-reads it, zeroed at the hand-over).
    const int kg_u = __builtin_amdgcn_readfirstlane(kg);
    const char* rpA = nullptr;
    const char* rpB = nullptr;
    int prA = 0, prB = 0;                                   // (wave-uniform) row the bases point at
    auto ld_rows_begin = [&]() {
        int pr = ld_pbase_u + kg_u * 16;
        pr = pr < p_end ? pr : p_end - 1;
        prA = prB = pr;
        rpA = ldA + (long long)pr * ld_lda * 4;
        rpB = ldB + (long long)pr * ld_ldb * 4;
    };
    auto ld_one = [&](bool isA, int i, int e) {
        if (isA) {
            ra4[i][e] = *reinterpret_cast<const float*>(rpA + (unsigned)ld_ca * 4u);
            const bool more = prA + 1 < p_end;
            rpA += more ? (long long)ld_lda * 4 : 0;
            prA += more ? 1 : 0;
        } else {
            rb4[i][e] = *reinterpret_cast<const float*>(rpB + (unsigned)ld_cb * 4u);
            const bool more = prB + 1 < p_end;
            rpB += more ? (long long)ld_ldb * 4 : 0;
            prB += more ? 1 : 0;
        }
    };
    // the chunk in registers (nvalid / pend_pair0 describe it): zero the ragged tail, then (a gap later) bias sums + hand-over
    auto st_zero = [&](bool isA, int i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (isA) ra4[i][e] = (4 * i + e < nvalid) ? ra4[i][e] : 0.f;
            else rb4[i][e] = (4 * i + e < nvalid) ? rb4[i][e] : 0.f;
        }
    };
    bool bias_on = false;
    auto st_write = [&](bool isA, int st, int i) {
        if (isA) {
            const float rs = (ra4[i][0] + ra4[i][1]) + (ra4[i][2] + ra4[i][3]);
            bs += bias_on ? rs : 0.f;                    // (a select, not a branch: see the note at the loop)
            *reinterpret_cast<f32x4*>(&smem[st][0][c * NT_LDS + kg * 16 + 4 * i]) = ra4[i];
        } else {
            *reinterpret_cast<f32x4*>(&smem[st][1][c * NT_LDS + kg * 16 + 4 * i]) = rb4[i];
        }
    };
    int cur = 0;
    if (total > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) F0.a[i] = *reinterpret_cast<const f32x4*>(&smem[0][0][a_off + 32 * i * NT_LDS]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) F0.b[j] = *reinterpret_cast<const f32x4*>(&smem[0][1][b_off + 32 * j * NT_LDS]);
    }
#define TN_PIN __builtin_amdgcn_sched_barrier(0);
#define TN_M(F, e, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[i][e], F.b[j][e], acc[i][j], 0, 0, 0); TN_PIN
    // the k-th MFMA of a k-step: (k / 4, k % 4) on the 64 x 128 wave tile, (k / 2, k % 2) on the 64 x 64 one
#define TN_MK(F, e, k) if constexpr (NJ == 4) { TN_M(F, e, (k) / 4, (k) % 4) } else if constexpr ((k) < 4) { TN_M(F, e, (k) / 2, (k) % 2) }
    // one k-step (8 or 4 MFMAs) with one auxiliary statement behind each MFMA (the wide tile has seven slots, the narrow one four)
#define TN_SLOT(F, e, ...) TN_SLOT_(F, e, __VA_ARGS__)      /* (one more level: the list macros below expand first) */
#define TN_SLOT_(F, e, X0, X1, X2, X3, X4, X5, X6)                                                   \
    TN_MK(F, e, 0) X0; TN_PIN TN_MK(F, e, 1) X1; TN_PIN TN_MK(F, e, 2) X2; TN_PIN TN_MK(F, e, 3) X3; TN_PIN     \
    if constexpr (NJ == 4) { TN_MK(F, e, 4) X4; TN_PIN TN_MK(F, e, 5) X5; TN_PIN TN_MK(F, e, 6) X6; TN_PIN TN_MK(F, e, 7) }
#define TN_RDA(F, ST, kk, i) F.a[i] = *reinterpret_cast<const f32x4*>(&smem[ST][0][a_off + 32 * (i) * NT_LDS + (kk) * 8])
#define TN_RDB(F, ST, kk, j) F.b[j] = *reinterpret_cast<const f32x4*>(&smem[ST][1][b_off + 32 * (j) * NT_LDS + (kk) * 8])
#define TN_RDB4(F, ST, kk, j) (void)0; if constexpr (NJ == 4) { TN_RDB(F, ST, kk, j); }
#define TN_READS(F, ST, kk) TN_RDA(F, ST, kk, 0), TN_RDA(F, ST, kk, 1), TN_RDB(F, ST, kk, 0), TN_RDB(F, ST, kk, 1), TN_RDB4(F, ST, kk, 2), TN_RDB4(F, ST, kk, 3), (void)0
#define TN_LD2(i, e) ld_one(true, i, e), ld_one(false, i, e), ld_one(true, i, (e) + 1), ld_one(false, i, (e) + 1), (void)0, (void)0, (void)0
#define TN_ST(i) st_zero(true, i), st_write(true, cur ^ 1, i), st_zero(false, i), st_write(false, cur ^ 1, i), (void)0, (void)0, (void)0
    // No instruction of the loop body is conditional: a branch around a load makes hipcc wait vmcnt(0) at the join (every scalar
    // load then waits for all loads before it -- measured 66 instead of 115 TFLOP/s).  Past the end of the split the loader
    // re-reads the last chunk (valid addresses, data never used) and the hand-over writes a stage nobody reads again; only the
    // bias sum must not see those chunks (a select).
    for (int t = 0; t < total; ++t) {
        bias_on = do_bias && pend_pair0 && t + 1 < total;
        load_begin(t + 2 < total ? t + 2 : total - 1);
        ld_rows_begin();
        TN_SLOT(F0, 0, TN_ST(0))
        TN_SLOT(F0, 1, TN_LD2(0, 0))
        TN_SLOT(F0, 2, TN_LD2(0, 2))
        TN_SLOT(F0, 3, TN_READS(F1, cur, 1))
        TN_SLOT(F1, 0, TN_ST(1))
        TN_SLOT(F1, 1, TN_LD2(1, 0))
        TN_SLOT(F1, 2, TN_LD2(1, 2))
        TN_SLOT(F1, 3, TN_READS(F0, cur, 2))
        TN_SLOT(F0, 0, TN_ST(2))
        TN_SLOT(F0, 1, TN_ST(3))
        TN_SLOT(F0, 2, TN_LD2(2, 0))
        TN_SLOT(F0, 3, TN_READS(F1, cur, 3))
        __syncthreads();            // the other stage is complete; every wave holds its last fragments of this one
        TN_SLOT(F1, 0, TN_LD2(2, 2))
        TN_SLOT(F1, 1, TN_LD2(3, 0))
        TN_SLOT(F1, 2, TN_LD2(3, 2))
        TN_SLOT(F1, 3, TN_READS(F0, cur ^ 1, 0))      // (after the last chunk: stale bytes, never used)
        nvalid = nvalid_ld; pend_pair0 = pair0_ld;
        cur ^= 1;
    }
#undef TN_PIN
#undef TN_M
#undef TN_MK
#undef TN_SLOT
#undef TN_SLOT_
#undef TN_RDA
#undef TN_RDB
#undef TN_RDB4
#undef TN_READS
#undef TN_LD2
#undef TN_ST

    float* __restrict__ slab = g.slab + (long long)grp * g.sSlab + (long long)split * N1p * N2p;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < NJ; ++tn) {
            const int col = n2_0 + wc * 32 * NJ + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n1_0 + wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(long long)row * N2p + col] = acc[tm][tn][r];
            }
        }
    if (do_bias) {
        float* red = &smem[0][0][0];
        red[kg * T + c] = bs;
        __syncthreads();
        if (tid < T) g.bias_slab[(long long)grp * g.sBiasSlab + (long long)split * N1p + n1_0 + tid] = red[tid] + red[T + tid];
    }
}

template <bool BIG, int NJ>
__global__ __launch_bounds__(128 * NJ, 2) void gemm_tn2_kernel(NuGemmTN g) { tn2_body<BIG, NJ>(g, blockIdx.x, blockIdx.y, blockIdx.z); }
template <bool BIG, int NJ>
__global__ __launch_bounds__(128 * NJ, 2) void gemm_tn2b_kernel(NuGemmTNBatch b) {
    int bx, split;
    const int pi = tn_batch_decode(b, 64 * NJ, bx, split);
    tn2_body<BIG, NJ>(b.p[pi], bx, split, 0);
}

// The split both launch paths use: enough workgroups for the chip, as few and as large slabs as possible.  The 256-wide kernel
// takes the shapes whose N1 and N2 are multiples of 256 in exact fp32 (one workgroup per CU: 256 of them).
// Does a weight-gradient launch with this split take the 256 x 256-tile kernels?  Only when that grid fills the chip: with few
// reduced rows (small batches: P / 256 splits at most) the 128-tile kernels put four times as many workgroups on the CUs.
static bool nu_tn_big_tile(int N1, int N2, int groups, int prec, int S) {
    static const bool tn128_env = getenv("NU_TN_128") && atoi(getenv("NU_TN_128")) != 0;      // development switch: 128 x 128 tiles only
    if ((prec & 3) == 2 || tn128_env || (N1 % 256) != 0 || (N2 % 256) != 0) return false;
    return (long long)(N1 / 256) * (N2 / 256) * (groups > 0 ? groups : 1) * S >= 192;
}
// The split both launch paths use: enough workgroups for the chip, as few and as large slabs as possible.
extern "C" int nu_wgrad_pick_split(int P, int N1, int N2, int groups, int prec) {
    if (groups < 1) groups = 1;
    // at least 256 reduced rows per split -- 128 for the few-thousand-row point sets of the small batches, where the launch is
    // latency-bound and twice the workgroups are worth the extra slabs (7 168 rows: 31.8 -> 25.0 us, profiles/r03/bench_tn_small.txt)
    const int rows_min = P <= 8192 ? 128 : 256;
    const int cap = (P + rows_min - 1) / rows_min > 0 ? (P + rows_min - 1) / rows_min : 1;
    if ((N1 % 256) == 0 && (N2 % 256) == 0) {
        int S = 256 / ((N1 / 256) * (N2 / 256) * groups);
        if (S < 1) S = 1;
        if (S > cap) S = cap;
        if (nu_tn_big_tile(N1, N2, groups, prec, S)) return S;
    }
    int S = 512 / (nu_cdiv(N1, 128) * nu_cdiv(N2, 128) * groups);
    if (S < 1) S = 1;
    if (S > cap) S = cap;
    // (a 128-tile split must not look like a 256-tile one to the launcher: it never does, S x tiles256 stays below 192 here
    // exactly when the 256-tile grid was too small above; with cap large the 256-tile branch has already returned)
    return S;
}

int nu_gemm_tn_launch(const NuGemmTN& g, hipStream_t stream) {
    if (g.N1 <= 0 || g.N2 <= 0 || g.S <= 0) return NU_ERR_ARG;
    if ((g.lda0 & 3) || (g.ldb0 & 3) || (g.A1 && ((g.lda1 & 3) || (g.ldb1 & 3)))) return NU_ERR_ARG;
    const int prec = g.bf16 & 3;
    if (prec == 3 || ((g.bf16 & ~3) && prec != 1)) return NU_ERR_ARG;
    const bool big_tile = nu_tn_big_tile(g.N1, g.N2, g.groups, g.bf16, g.S);
    if ((g.bf16 & 3) == 0) {         // exact fp32: the pipelined kernel, 256 x 256 tiles where the shape allows, else 128 x 128
        // 128 x 128 tiles: the first-generation kernel stays the default -- the pipelined one measured 85.5 vs 89.8 TFLOP/s on the
        // 1024 x 288 shape and 109 vs 110 on 256 x 256 with 128 splits (profiles/r03): at 64 MFMAs per chunk and wave the 32 scalar
        // loads are a third of the issue slots however they are placed.  NU_TN_V1=0 selects the pipelined kernel (development A/B).
        static const bool tn_v1 = !(getenv("NU_TN_V1") && atoi(getenv("NU_TN_V1")) == 0);
        const long long mld = (g.lda0 > g.ldb0 ? g.lda0 : g.ldb0) > (g.A1 ? (g.lda1 > g.ldb1 ? g.lda1 : g.ldb1) : 0)
                                  ? (g.lda0 > g.ldb0 ? g.lda0 : g.ldb0) : (g.lda1 > g.ldb1 ? g.lda1 : g.ldb1);
        const bool big = (long long)g.P * mld * 4 >= (1LL << 32);
        if (big_tile) {
            dim3 grid2((g.N1 / 256) * (g.N2 / 256), g.S, g.groups > 0 ? g.groups : 1);
            if (big) hipLaunchKernelGGL((gemm_tn2_kernel<true, 4>), grid2, dim3(512), 0, stream, g);
            else hipLaunchKernelGGL((gemm_tn2_kernel<false, 4>), grid2, dim3(512), 0, stream, g);
            return nu_launch_status();
        }
        if (!tn_v1) {
            dim3 grid2(nu_cdiv(g.N1, 128) * nu_cdiv(g.N2, 128), g.S, g.groups > 0 ? g.groups : 1);
            if (big) hipLaunchKernelGGL((gemm_tn2_kernel<true, 2>), grid2, dim3(256), 0, stream, g);
            else hipLaunchKernelGGL((gemm_tn2_kernel<false, 2>), grid2, dim3(256), 0, stream, g);
            return nu_launch_status();
        }
    }
    dim3 grid(nu_cdiv(g.N1, 128) * nu_cdiv(g.N2, 128), g.S, g.groups > 0 ? g.groups : 1), block(256);
    const long long max_ld = (g.lda0 > g.ldb0 ? g.lda0 : g.ldb0) > (g.A1 ? (g.lda1 > g.ldb1 ? g.lda1 : g.ldb1) : 0)
                                 ? (g.lda0 > g.ldb0 ? g.lda0 : g.ldb0) : (g.lda1 > g.ldb1 ? g.lda1 : g.ldb1);
    const bool big = (long long)g.P * max_ld * 4 >= (1LL << 32);
    if (prec == 1) {
        // vector-load kernel: 16-byte aligned operands, rows of 8 or more elements (bf16 rows: a multiple of 8)
        auto ok = [](const float* p, int ld, bool h) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && ld >= 8 && (!h || (ld & 7) == 0)); };
        const bool vec = ok(g.A0, g.lda0, g.bf16 & NU_TN_A0_16) && ok(g.B0, g.ldb0, g.bf16 & NU_TN_B0_16) &&
                         ok(g.A1, g.lda1, g.bf16 & NU_TN_A1_16) && ok(g.B1, g.ldb1, g.bf16 & NU_TN_B1_16) &&
                         ((g.sA0 | g.sB0 | g.sA1 | g.sB1) & 7) == 0;
        static const bool tn_scalar_env = getenv("NU_TN_SCALAR") && atoi(getenv("NU_TN_SCALAR")) != 0;      // development switch
        if (vec && !tn_scalar_env) {
            if (big_tile) {
                dim3 grid2((g.N1 / 256) * (g.N2 / 256), g.S, g.groups > 0 ? g.groups : 1);
                hipLaunchKernelGGL(gemm_tn16x256_kernel, grid2, dim3(512), 0, stream, g);
            } else {
                hipLaunchKernelGGL(gemm_tn16_kernel, grid, block, 0, stream, g);
            }
            return nu_launch_status();
        }
    }
    if (big) {
        if (prec == 2) hipLaunchKernelGGL((gemm_tn_kernel<true, 2>), grid, block, 0, stream, g);
        else if (prec == 1) hipLaunchKernelGGL((gemm_tn_kernel<true, 1>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((gemm_tn_kernel<true, 0>), grid, block, 0, stream, g);
    } else {
        if (prec == 2) hipLaunchKernelGGL((gemm_tn_kernel<false, 2>), grid, block, 0, stream, g);
        else if (prec == 1) hipLaunchKernelGGL((gemm_tn_kernel<false, 1>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((gemm_tn_kernel<false, 0>), grid, block, 0, stream, g);
    }
    return nu_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Deterministic split reduction, batched: out[n1*ldo + n2] (+)= alpha * sum_s slab[s*ss + n1*rs + n2] for up to
// NU_REDUCE_MAX independent problems per launch (descriptors travel in the kernel argument block, so there is no
// host -> device copy).  A problem with few outputs and many slabs (bias sums, the skinny heads) spreads each
// output over G threads (fixed order: thread g takes slabs g, g+G, ...; then g = 0 adds the partials in order).
// ------------------------------------------------------------------------------------------------
struct NuReduceBatch {
    NuReduceDesc d[NU_REDUCE_MAX];
    int n;
};

__global__ __launch_bounds__(256) void slab_reduce_batched_kernel(NuReduceBatch b) {
    __shared__ float red[256];
    int di = 0;
    for (int i = 1; i < b.n; ++i) di = ((int)blockIdx.x >= b.d[i].blk_begin) ? i : di;
    const float* __restrict__ slab = b.d[di].slab;
    float* __restrict__ out = b.d[di].out;
    const int S = b.d[di].S, N1 = b.d[di].N1, N2 = b.d[di].N2, rs = b.d[di].rs, G = b.d[di].G;
    const long long ss = b.d[di].ss;
    const int opb = 256 / G;
    const int o = threadIdx.x % opb, g = threadIdx.x / opb;
    const int idx = ((int)blockIdx.x - b.d[di].blk_begin) * opb + o;
    const bool live = idx < N1 * N2;
    float v = 0.f;
    int n1 = 0, n2 = 0;
    if (live) {
        n1 = idx / N2;
        n2 = idx - n1 * N2;
        const float* p = slab + (long long)n1 * rs + n2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int s = g;
        for (; s + 3 * G < S; s += 4 * G) {
            s0 += p[(long long)s * ss];
            s1 += p[(long long)(s + G) * ss];
            s2 += p[(long long)(s + 2 * G) * ss];
            s3 += p[(long long)(s + 3 * G) * ss];
        }
        for (; s < S; s += G) s0 += p[(long long)s * ss];
        v = (s0 + s1) + (s2 + s3);
    }
    if (G > 1) {
        red[threadIdx.x] = v;
        __syncthreads();
        if (g == 0)
            for (int gg = 1; gg < G; ++gg) v += red[gg * opb + o];
    }
    if (live && g == 0) {
        float* q = out + (long long)n1 * b.d[di].ldo + n2;
        v *= b.d[di].alpha;
        *q = b.d[di].accumulate ? *q + v : v;
    }
}

static int nu_reduce_launch_chunk(const NuReduceDesc* descs, int n, hipStream_t stream) {
    NuReduceBatch b;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        b.d[i] = descs[i];
        const long long nout = (long long)descs[i].N1 * descs[i].N2;
        int G = 1;
        while (G < 16 && nout * G < 16384 && 2 * G <= descs[i].S) G *= 2;
        b.d[i].G = G;
        b.d[i].blk_begin = blocks;
        blocks += (int)nu_cdivl(nout, 256 / G);
    }
    b.n = n;
    if (blocks <= 0) return NU_OK;
    hipLaunchKernelGGL(slab_reduce_batched_kernel, dim3(blocks), dim3(256), 0, stream, b);
    return nu_launch_status();
}

// An accumulating problem may depend on an earlier one writing the same output: it gets a launch of its own, after
// everything queued before it (stream order).
extern "C" int nu_slab_reduce_batched(const NuReduceDesc* descs, int n, hipStream_t stream) {
    int i = 0;
    while (i < n) {
        if (descs[i].N1 <= 0 || descs[i].N2 <= 0 || descs[i].S <= 0) return NU_ERR_ARG;
        int j = i + 1;
        if (!descs[i].accumulate)
            while (j < n && j - i < NU_REDUCE_MAX && !descs[j].accumulate) ++j;
        const int rc = nu_reduce_launch_chunk(descs + i, j - i, stream);
        if (rc) return rc;
        i = j;
    }
    return NU_OK;
}

int nu_reduce_push(NuReduceDesc* descs, int* ndesc, int cap, const float* slab, int S, int N1, int N2, int rs,
                   long long ss, float* out, int ldo, float alpha, int accumulate) {
    if (*ndesc >= cap) return NU_ERR_WORKSPACE;
    NuReduceDesc& d = descs[(*ndesc)++];
    d.slab = slab; d.out = out; d.ss = ss; d.S = S; d.N1 = N1; d.N2 = N2; d.rs = rs; d.ldo = ldo;
    d.accumulate = accumulate; d.G = 1; d.blk_begin = 0; d.alpha = alpha;
    return NU_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI (test / bench entry points for the raw GEMMs)
// ------------------------------------------------------------------------------------------------
extern "C" int nu_gemm_nt(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                          float* C2, int ldc2, const float* bias, const float* H, int ldh, const float* D, int ldd,
                          const float* Cadd, int ldadd, int zero_to, float alpha, int epi, hipStream_t stream) {
    NuGemmNT g = {};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.C = C; g.ldc = ldc; g.C2 = C2; g.ldc2 = ldc2; g.bias = bias; g.H = H; g.ldh = ldh; g.D = D; g.ldd = ldd;
    g.Cadd = Cadd; g.ldadd = ldadd; g.zero_to = zero_to; g.alpha = alpha; g.groups = 1; g.epi = epi;
    return nu_gemm_nt_launch(g, stream);
}

extern "C" long long nu_gemm_tn_workspace_bytes(int N1, int N2, int S) {
    return (long long)S * nu_rup(N1, 128) * (nu_rup(N2, 128) + 1) * sizeof(float);
}

// C[N1,N2] = A0^T B0 (+ A1^T B1); bias_out[N1] = column sums of A0 (optional)
extern "C" int nu_gemm_tn(const float* A0, int lda0, const float* B0, int ldb0, const float* A1, int lda1,
                          const float* B1, int ldb1, int P, int N1, int N2, float* C, int ldc, float* bias_out,
                          int S, void* workspace, long long workspace_bytes, hipStream_t stream) {
    NuGemmTN g = {};
    g.A0 = A0; g.lda0 = lda0; g.B0 = B0; g.ldb0 = ldb0; g.A1 = A1; g.lda1 = lda1; g.B1 = B1; g.ldb1 = ldb1;
    g.P = P; g.N1 = N1; g.N2 = N2; g.S = S; g.groups = 1;
    return nu_wgrad(&g, C, ldc, 0, bias_out, 0, workspace, workspace_bytes, stream);
}

extern "C" int nu_gemm_nt_size(void) { return (int)sizeof(NuGemmNT); }
extern "C" int nu_gemm_tn_size(void) { return (int)sizeof(NuGemmTN); }

// struct-pointer entry points (what the Python host layer binds; one pointer argument keeps ctypes cheap)
extern "C" int nu_gemm_nt_ex(const NuGemmNT* g, hipStream_t stream) { return nu_gemm_nt_launch(*g, stream); }

// Weight-gradient GEMM + deterministic split reduction.
//   dW[N1, N2] (ld = ldw) = A0^T B0 (+ A1^T B1);   db[N1] = column sums of A0 (optional)
// grouped: `groups` independent problems at element strides (sA*, sB*, sW, sDb).
extern "C" long long nu_wgrad_workspace_bytes(int N1, int N2, int S, int groups) {
    return (long long)(groups > 0 ? groups : 1) * S * nu_rup(N1, 128) * (nu_rup(N2, 128) + 1) * sizeof(float);
}
// Deferred form: launches the split GEMM into `workspace` (which must stay untouched until the reductions ran)
// and appends the reduction problems to descs[*ndesc ...]; nu_slab_reduce_batched(descs, *ndesc) finishes them.
extern "C" int nu_wgrad_enqueue(const NuGemmTN* gin, float* dW, int ldw, long long sW, float* db, long long sDb,
                                void* workspace, long long workspace_bytes, NuReduceDesc* descs, int* ndesc, int cap,
                                hipStream_t stream) {
    NuGemmTN g = *gin;
    const int groups = g.groups > 0 ? g.groups : 1;
    if (g.P <= 0) return NU_ERR_ARG;
    if (workspace_bytes < nu_wgrad_workspace_bytes(g.N1, g.N2, g.S, groups)) return NU_ERR_WORKSPACE;
    if (*ndesc + groups * (db ? 2 : 1) > cap) return NU_ERR_WORKSPACE;
    const int N1p = nu_rup(g.N1, 128), N2p = nu_rup(g.N2, 128);
    const long long slab_per = (long long)g.S * N1p * N2p;
    const long long bias_per = (long long)g.S * N1p;
    g.slab = (float*)workspace;
    g.sSlab = slab_per;
    g.bias_slab = db ? g.slab + slab_per * groups : nullptr;
    g.sBiasSlab = bias_per;
    g.groups = groups;
    int rc = nu_gemm_tn_launch(g, stream);
    for (int z = 0; z < groups && rc == NU_OK; ++z) {
        rc = nu_reduce_push(descs, ndesc, cap, g.slab + z * slab_per, g.S, g.N1, g.N2, N2p, (long long)N1p * N2p,
                            dW + z * sW, ldw, 1.0f, 0);
        if (rc == NU_OK && db)
            rc = nu_reduce_push(descs, ndesc, cap, g.bias_slab + z * bias_per, g.S, g.N1, 1, 1, N1p, db + z * sDb, 1, 1.0f, 0);
    }
    return rc;
}

extern "C" int nu_wgrad(const NuGemmTN* gin, float* dW, int ldw, long long sW, float* db, long long sDb,
                        void* workspace, long long workspace_bytes, hipStream_t stream) {
    const int groups = gin->groups > 0 ? gin->groups : 1;
    if (groups > NU_REDUCE_MAX / 2) return NU_ERR_ARG;
    NuReduceDesc descs[NU_REDUCE_MAX];
    int n = 0;
    int rc = nu_wgrad_enqueue(gin, dW, ldw, sW, db, sDb, workspace, workspace_bytes, descs, &n, NU_REDUCE_MAX, stream);
    if (rc) return rc;
    return nu_slab_reduce_batched(descs, n, stream);
}

// ------------------------------------------------------------------------------------------------
// Context services shared by the network-level entries (mlp_ops.hip) and the host bindings: the deferred-reduction arena, optional
// per-launch events, and the QUEUE of weight gradients of a pass.
//
// Every weight gradient of a backward pass is independent of the others (they read activations and cotangents that exist by the
// end of the pass and write their own dW), so the pass queues them (nu_wgrad_defer) and nu_wgrad_flush launches the whole queue
// as ONE launch per tile class -- 128 x 128 and 256 x 256 output tiles -- with the splits chosen for the queue as a whole:
// every workgroup reduces the same number of rows, the grid fills the chip once, and a 512-ray batch's nine SDF weight gradients
// (14 k rows each: 110 splits of 128 rows apiece when launched alone) become one launch of 16 splits of 896 rows per problem.
// The split of a problem is a function of the queue's contents only (never of the arena's state), so a pass gives the same bits
// however its slabs end up being flushed.
// ------------------------------------------------------------------------------------------------
extern "C" int nu_wgrad_item_size(void) { return (int)sizeof(NuWgradItem); }

static int ctx_reduce_now(NuOpCtx* c, hipStream_t stream) {
    if (c->ndesc > 0) {
        const int rc = nu_slab_reduce_batched(c->descs, c->ndesc, stream);
        if (rc != NU_OK) return rc;
        c->ndesc = 0;
    }
    c->arena_off = 0;
    return NU_OK;
}
extern "C" int nu_wgrad_flush(NuOpCtx* c, hipStream_t stream);
// the batched reductions of everything LAUNCHED so far (queued weight gradients stay queued): what a full arena forces mid-pass
extern "C" int nu_ctx_reduce(NuOpCtx* c, hipStream_t stream) { return ctx_reduce_now(c, stream); }
extern "C" int nu_ctx_flush(NuOpCtx* c, hipStream_t stream) {
    const int rc = nu_wgrad_flush(c, stream);
    if (rc != NU_OK) return rc;
    return ctx_reduce_now(c, stream);
}
// `nbytes` of slab space that stays untouched until the next flush (stream order makes reuse after a flush safe)
int nu_ctx_take(NuOpCtx* c, long long nbytes, int ndesc_needed, hipStream_t stream, float** out, long long* out_bytes) {
    const long long n = (nbytes + 255) / 256 * 64;          // floats, 256-byte granules
    if (n > c->arena_floats) return NU_ERR_WORKSPACE;
    if (c->arena_off + n > c->arena_floats || c->ndesc + ndesc_needed > c->cap) {
        // a flush on THIS stream reduces every slab taken so far and hands their space out again: only safe when every producer
        // is ordered before it, i.e. not while a second stream feeds the same arena (engine.py _fork / _join)
        if (c->forked) return NU_ERR_WORKSPACE;
        const int rc = ctx_reduce_now(c, stream);
        if (rc != NU_OK) return rc;
    }
    *out = c->arena + c->arena_off;
    *out_bytes = n * 4;
    c->arena_off += n;
    return NU_OK;
}
void nu_ctx_ev_begin(NuOpCtx* c, hipStream_t stream) {
    if (c->ev && c->nev + 2 <= c->ev_cap) (void)hipEventRecord(static_cast<hipEvent_t>(c->ev[c->nev]), stream);
}
void nu_ctx_ev_end(NuOpCtx* c, hipStream_t stream, double kind, double flops, double bytes) {
    if (c->ev && c->nev + 2 <= c->ev_cap) {
        (void)hipEventRecord(static_cast<hipEvent_t>(c->ev[c->nev + 1]), stream);
        double* m = c->ev_meta + 3 * (c->nev / 2);
        m[0] = kind; m[1] = flops; m[2] = bytes;
        c->nev += 2;
    }
}

static inline bool tn_item_big(const NuGemmTN& g) {       // an operand of 4 GiB or more: 64-bit offsets (single launches only)
    const long long m0 = g.lda0 > g.ldb0 ? g.lda0 : g.ldb0, m1 = g.A1 ? (g.lda1 > g.ldb1 ? g.lda1 : g.ldb1) : 0;
    return (long long)g.P * (m0 > m1 ? m0 : m1) * 4 >= (1LL << 32);
}

// One launch of the items idx[0..m) (all of tile class `klass`: 1 = 128 x 128 tiles, 2 = 256 x 256), splits already chosen.
static int wgrad_launch_batch(NuOpCtx* c, NuWgradItem* it, const int* idx, int m, int klass, hipStream_t stream) {
    long long bytes = 0;
    int ndesc = 0, nprob = 0;
    for (int k = 0; k < m; ++k) {
        const NuGemmTN& g = it[idx[k]].g;
        bytes += (nu_wgrad_workspace_bytes(g.N1, g.N2, g.S, g.groups) + 255) / 256 * 256;
        ndesc += g.groups * (it[idx[k]].db ? 2 : 1);
        nprob += g.groups;
    }
    if (nprob > NU_TN_BATCH_MAX) return NU_ERR_ARG;
    float* ws;
    long long nb;
    int rc = nu_ctx_take(c, bytes, ndesc, stream, &ws, &nb);
    if (rc != NU_OK) return rc;
    NuGemmTNBatch b;
    int np = 0, blocks = 0;
    double flops = 0, abytes = 0;
    const int edge = klass == 2 ? 256 : 128;
    for (int k = 0; k < m; ++k) {
        NuWgradItem& w = it[idx[k]];
        const NuGemmTN& g = w.g;
        const int N1p = nu_rup(g.N1, 128), N2p = nu_rup(g.N2, 128);
        const long long slab_per = (long long)g.S * N1p * N2p, bias_per = (long long)g.S * N1p;
        float* slab = ws;
        float* bias_slab = w.db ? slab + slab_per * g.groups : nullptr;
        ws += (nu_wgrad_workspace_bytes(g.N1, g.N2, g.S, g.groups) + 255) / 256 * 64;
        for (int z = 0; z < g.groups; ++z) {
            NuGemmTN& q = b.p[np];
            q = g;
            q.groups = 1;
            q.A0 = g.A0 + z * g.sA0; q.B0 = g.B0 + z * g.sB0;
            q.A1 = g.A1 ? g.A1 + z * g.sA1 : nullptr; q.B1 = g.B1 ? g.B1 + z * g.sB1 : nullptr;
            q.slab = slab + z * slab_per; q.bias_slab = bias_slab ? bias_slab + z * bias_per : nullptr;
            q.sSlab = 0; q.sBiasSlab = 0;
            b.blk0[np++] = blocks;
            blocks += nu_cdiv(g.N1, edge) * nu_cdiv(g.N2, edge) * g.S;
            rc = nu_reduce_push(c->descs, &c->ndesc, c->cap, q.slab, g.S, g.N1, g.N2, N2p, (long long)N1p * N2p, w.dW + z * w.sW, w.ldw, 1.0f, 0);
            if (rc == NU_OK && w.db)
                rc = nu_reduce_push(c->descs, &c->ndesc, c->cap, q.bias_slab, g.S, g.N1, 1, 1, N1p, w.db + z * w.sDb, 1, 1.0f, 0);
            if (rc != NU_OK) return rc;
        }
        flops += w.flops; abytes += w.bytes;
    }
    b.blk0[np] = blocks;
    b.n = np; b.pad_ = 0;
    nu_ctx_ev_begin(c, stream);
    if (klass == 2) hipLaunchKernelGGL((gemm_tn2b_kernel<false, 4>), dim3(blocks), dim3(512), 0, stream, b);
    else hipLaunchKernelGGL((gemm_tnb_kernel<false>), dim3(blocks), dim3(256), 0, stream, b);
    rc = nu_launch_status();
    nu_ctx_ev_end(c, stream, 1.0, flops, abytes);
    return rc;
}

// one item on its own (arithmetic modes and operand sizes the batch kernels do not cover): the split rule of a lone launch
static int wgrad_launch_single(NuOpCtx* c, NuWgradItem& w, hipStream_t stream) {
    NuGemmTN& g = w.g;
    g.S = nu_wgrad_pick_split(g.P, g.N1, g.N2, g.groups, g.bf16 & 3);
    float* ws;
    long long nb;
    int rc = nu_ctx_take(c, nu_wgrad_workspace_bytes(g.N1, g.N2, g.S, g.groups), 2 * g.groups, stream, &ws, &nb);
    if (rc != NU_OK) return rc;
    nu_ctx_ev_begin(c, stream);
    rc = nu_wgrad_enqueue(&g, w.dW, w.ldw, w.sW, w.db, w.sDb, ws, nb, c->descs, &c->ndesc, c->cap, stream);
    nu_ctx_ev_end(c, stream, 1.0, w.flops, w.bytes);
    return rc;
}

extern "C" int nu_wgrad_flush(NuOpCtx* c, hipStream_t stream) {
    const int n = c->npend;
    if (n <= 0 || c->pend == nullptr) return NU_OK;
    NuWgradItem* it = c->pend;
    c->npend = 0;
    static const bool batch_on = !(getenv("NU_TN_BATCH") && atoi(getenv("NU_TN_BATCH")) == 0);      // development switch (A/B)
    static const bool tn128_env = getenv("NU_TN_128") && atoi(getenv("NU_TN_128")) != 0;
    // ---- plan: tile class and split of every item (a function of the queue only) ----
    int klass[NU_WGRAD_QUEUE_MAX];
    if (n > NU_WGRAD_QUEUE_MAX) return NU_ERR_ARG;
    int pmax = 0;
    for (int i = 0; i < n; ++i) {
        NuGemmTN& g = it[i].g;
        if (g.groups < 1) g.groups = 1;
        klass[i] = (batch_on && (g.bf16 & 3) == 0 && (g.bf16 & ~3) == 0 && !tn_item_big(g) && g.groups <= NU_TN_BATCH_MAX) ? 1 : 0;
        if (klass[i] && g.P > pmax) pmax = g.P;
    }
    const int rows_min = pmax <= 8192 ? 128 : 256;       // (see nu_wgrad_pick_split)
    // rows per workgroup so that the class's grid is at most `target` workgroups: all of them resident, all of one length
    auto plan = [&](int k, int edge, int target) -> long long {
        long long work = 0;
        for (int i = 0; i < n; ++i)
            if (klass[i] == k) work += (long long)it[i].g.P * nu_cdiv(it[i].g.N1, edge) * nu_cdiv(it[i].g.N2, edge) * it[i].g.groups;
        if (work == 0) return 0;
        long long rows = nu_cdivl(work, target);
        rows = rows < rows_min ? rows_min : rows;
        rows = nu_cdivl(rows, TBK) * TBK;
        while (true) {
            long long wgs = 0;
            for (int i = 0; i < n; ++i)
                if (klass[i] == k) wgs += nu_cdivl(it[i].g.P, rows) * nu_cdiv(it[i].g.N1, edge) * nu_cdiv(it[i].g.N2, edge) * it[i].g.groups;
            if (wgs <= target || rows >= (1LL << 30)) { for (int i = 0; i < n; ++i) if (klass[i] == k) it[i].g.S = (int)nu_cdivl(it[i].g.P, rows); return wgs; }
            rows += TBK;
        }
    };
    // 256 x 256 tiles (one workgroup per CU) for the shapes that allow them, when that grid fills the chip
    for (int i = 0; i < n; ++i)
        if (klass[i] == 1 && !tn128_env && (it[i].g.N1 % 256) == 0 && (it[i].g.N2 % 256) == 0) klass[i] = 2;
    if (plan(2, 256, 256) < 192)
        for (int i = 0; i < n; ++i) if (klass[i] == 2) klass[i] = 1;
    // 128 x 128 tiles: FOUR workgroups per CU are resident (126 VGPRs, 36.9 KB of LDS) and this kernel is bound by the issue of its
    // scalar transposing loads, which more waves hide: a grid of 1024 measured 121.4 TFLOP/s over a step's weight gradients against
    // 118.0 at 512 (same box, alternating; 512 rays: 7.61 vs 7.74 ms/step)
    static const int c1_target = getenv("NU_TN_C1_TARGET") ? atoi(getenv("NU_TN_C1_TARGET")) : 1024;     // development switch
    plan(1, 128, c1_target);
    // ---- launch: singles first, then one launch per class (more when the batch table or the arena cannot take a class at once) ----
    for (int i = 0; i < n; ++i)
        if (klass[i] == 0) { const int rc = wgrad_launch_single(c, it[i], stream); if (rc != NU_OK) return rc; }
    for (int k = 2; k >= 1; --k) {
        int idx[NU_WGRAD_QUEUE_MAX], m = 0, nprob = 0;
        long long bytes = 0;
        for (int i = 0; i <= n; ++i) {
            const bool mine = i < n && klass[i] == k;
            long long need = 0;
            if (mine) need = (nu_wgrad_workspace_bytes(it[i].g.N1, it[i].g.N2, it[i].g.S, it[i].g.groups) + 255) / 256 * 256;
            const bool full = mine && m > 0 && (nprob + it[i].g.groups > NU_TN_BATCH_MAX || bytes + need > c->arena_floats * 4);
            if ((i == n || full) && m > 0) {
                const int rc = wgrad_launch_batch(c, it, idx, m, k, stream);
                if (rc != NU_OK) return rc;
                m = 0; nprob = 0; bytes = 0;
            }
            if (mine) { idx[m++] = i; nprob += it[i].g.groups; bytes += need; }
        }
    }
    return NU_OK;
}

// Queue one weight gradient of the current pass: dW[N1, N2] (ld = ldw, group stride sW) = A0^T B0 (+ A1^T B1), db = column sums
// of A0 (optional).  The operands must stay valid and unmodified until nu_wgrad_flush / nu_ctx_flush; g->S is ignored (the
// flush chooses the splits).  Without a queue in the context (NuOpCtx.pend == NULL) the launch happens at once.
extern "C" int nu_wgrad_defer(NuOpCtx* c, const NuGemmTN* g, float* dW, int ldw, long long sW, float* db, long long sDb, double flops,
                              double bytes, hipStream_t stream) {
    if (g->P <= 0) return NU_OK;
    NuWgradItem w;
    w.g = *g; w.dW = dW; w.ldw = ldw; w.pad_ = 0; w.sW = sW; w.db = db; w.sDb = sDb; w.flops = flops; w.bytes = bytes;
    if (w.g.groups < 1) w.g.groups = 1;
    if (c->pend == nullptr || c->pend_cap <= 0) return wgrad_launch_single(c, w, stream);
    if (c->npend >= c->pend_cap || c->npend >= NU_WGRAD_QUEUE_MAX) {
        const int rc = nu_wgrad_flush(c, stream);
        if (rc != NU_OK) return rc;
    }
    c->pend[c->npend++] = w;
    return NU_OK;
}

// development aid: bare fp32-MFMA issue loop (no memory) -- what the matrix pipe delivers at the clock it holds
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float x = (float)threadIdx.x * 1e-3f, y = 1.0f + (float)blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { nu_dbg_clk[0] = clock64() - c0; nu_dbg_clk[1] = wall_clock64() - w0; }
}
// development aid: effective shader clock (MHz) seen by block 0 of the last instrumented launch (synchronises)
extern "C" double nu_debug_clock_mhz(void) {
    unsigned long long h[2] = {0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(nu_dbg_clk), sizeof(h)) != hipSuccess || h[1] == 0) return -1.0;
    return (double)h[0] / (double)h[1] * 100.0;
}
extern "C" int nu_debug_mfma_peak(float* out, int blocks, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, stream, out, iters);
    return nu_launch_status();
}

