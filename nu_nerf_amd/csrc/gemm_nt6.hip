// gemm_nt6.hip -- the exact 3-way bf16 split ("bf16x6", NuGemmNT.bf16 mode 2) NT GEMM with PRE-SPLIT weights (gfx950).
//
// Same contract and the same arithmetic as gemm_nt_kernel<EPI, 2> (gemm_nt.hip): C = epi(A . B^T) with both operands written as
// hi + mid + lo bf16 planes and the six partial products >= 2^-16 accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- per
// accumulator the same six MFMAs per 16-deep k-step in the same order, so the results are bit-identical to that kernel.  What
// differs is where the time goes:
//   * the WEIGHT operand arrives already split (NuGemmNT.B6: three bf16 planes written once per step by the pack launch,
//     layout below), so no workgroup splits it again -- in the old kernel every 128-row tile re-split its 128 x K weight tile;
//   * one workgroup owns 128 rows x 256 columns (two 128 x 128 output tiles side by side): the activation rows are split ONCE
//     for both (they are read as fp32, the storage of the step does not change);
//   * 16-deep chunks through two LDS stages, the hand-over of the next chunk (global -> registers -> split -> LDS) spread one
//     memory instruction at a time between the 48 MFMAs of a chunk, the loader cursor running across tile boundaries as in
//     gemm_nt2_kernel; per accumulator the dependent MFMAs sit two apart.
// Replaces, in this arithmetic mode, the same reference code as gemm_nt.hip (network/field.py:133-150, :158-170, :265-289,
// :371-408 and their autograd backward).
//
// B6 layout (include/nu_nerf.h NuGemmNT.B6): the table [rows][ld] is cut into blocks of 256 rows; inside a block the 16-wide k-groups
// follow one another, each as [256 rows][hi x16 | mid x16 | lo x16] (the two 8-element halves of each x16 swapped in rows with bit 3
// set, see below) -- the 24 576 bytes a workgroup needs for one chunk of its 256
// columns are CONTIGUOUS in memory (eight full cache lines per wave-instruction) and are copied to LDS in the same order; row
// offsets that are multiples of 256 and group strides carry over from the fp32 table's element offsets with a factor 3.
#include "gemm_epi.h"
#include <stdlib.h>

#define N6_APL 4096                         // bytes of one A plane of a stage: [128 rows][16 k] bf16
#define N6_BOFF (3 * N6_APL)
#define N6_BIMG 24576                       // the B image of a stage: [256 rows][3 planes][16 k] bf16, rows 96 B apart (fragment reads:
                                            // slot (6 r + 2 p + h) mod 16 of the 256-B bank row -- every slot four times, conflict-free)
#define N6_STAGE (N6_BOFF + N6_BIMG)        // 36 864 B; the epilogue scratch (4 waves x 32 x EPI_LDS floats = 34 816 B) aliases a stage

static_assert(4 * 32 * EPI_LDS * 4 <= N6_STAGE, "epilogue scratch must fit one stage");

template <int EPI, bool BATCH>
static __device__ __forceinline__ void nt6_run(const NuGemmNT* __restrict__ probs, const int* __restrict__ slot0, const int nprob,
                                               char* __restrict__ smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int z = BATCH ? 0 : blockIdx.z;
    auto P = [&](int pi) -> const NuGemmNT& { return probs[BATCH ? pi : 0]; };
    const int li = lane & 31, lh = lane >> 5;
    // fragment addresses inside a stage (bytes): lane (r, h) holds k = 8h .. 8h+7 of row r of a 32-row block -- 1 KB contiguous per fragment
    // (the two 16-byte halves of a row's 32-byte plane segment are SWAPPED in rows with bit 3 set, in LDS and in the B6 tables: a
    // ds_read_b128 serves 16 lanes = 16 consecutive rows of one half per cycle, and rows r and r + 8 would otherwise meet in the same
    // 16-byte slot of the 256-byte bank row -- measured: a third of the LDS cycles were bank-conflict cycles without the swap)
    const int hsw = (lh ^ ((li >> 3) & 1)) * 16;
    const int a_off = (wr * 64 + li) * 32 + hsw;                      // + plane * N6_APL + tm * 1024
    const int bf_off = N6_BOFF + (wc * 64 + li) * 96 + hsw;           // + plane * 32 + sub * 12288 + tn * 3072

    // slots: as gemm_nt_kernel, over 256-column tiles (the column tiles of one row tile sit 8 slots apart)
    const int nslots = BATCH ? slot0[nprob] : ((((P(0).M + TBM - 1) / TBM) + 7) / 8) * 8 * ((P(0).N + 255) / 256);
    auto slot_tile = [&](int j, int& pi, int& mt, int& nt) -> bool {
        if (BATCH)
            while (pi + 1 < nprob && j >= slot0[pi + 1]) ++pi;
        const NuGemmNT& q = P(pi);
        const int ntn = (q.N + 255) / 256, mtiles = (q.M + TBM - 1) / TBM;
        const int jl = BATCH ? j - slot0[pi] : j;
        const int grp = jl / (8 * ntn);
        const int rem = jl - grp * 8 * ntn;
        nt = rem >> 3;
        mt = grp * 8 + (rem & 7);
        return mt < mtiles;
    };
    auto next_valid = [&](int j, int& pi, int& mt, int& nt) -> int {
        while (j < nslots && !slot_tile(j, pi, mt, nt)) j += gridDim.x;
        return j;
    };

    int pi = 0, mt = 0, nt = 0;
    int j = next_valid(blockIdx.x, pi, mt, nt);
    if (j >= nslots) return;

    // ---- loader: a cursor over the chunks of this workgroup's tiles, in order ----
    // A: thread -> (rows tid >> 2 and 64 + tid >> 2, quarter tid & 3): 4 floats of each -- a wave-instruction reads 16 rows x 64
    // contiguous bytes.  B: the chunk image is 1536 contiguous 16-byte pieces, piece tid + 256 i to thread tid (i < 6; rows >= 128 of
    // the tile's weight rows are pieces >= 768, i.e. i >= 3).
    const int arow = tid >> 2, aq = tid & 3;
    const int aw_off = arow * 32 + (((aq >> 1) ^ ((arow >> 3) & 1)) * 16) + (aq & 1) * 8;   // LDS byte offset of this thread's first A piece inside a plane (rows arow and arow + 64: same bit 3)
    const int bw_off = N6_BOFF + tid * 16;                            // ... of its first B piece
    int ld_j = j, ld_kt = 0, ld_pi = pi, ld_nk = 1;
    bool ld_has1 = true;                                              // the loader's tile has its second 128-column half
    const float* ap = nullptr;
    const char* bbase = nullptr;
    // The A pieces are requested TWO chunks ahead of their hand-over (two register sets, used in turn: the activations stream from
    // HBM, ~2 us under load, and a chunk is 1.3 us of MFMAs for the two workgroups of a CU); the B pieces one chunk ahead (weights:
    // L2 hits).  So the cursor is the A loader's; the B loader follows it one chunk behind (lb_*).
    f32x4 ra[2][2], rb[6];
    const char* lb_base = nullptr;
    int lb_koff = 0;
    bool lb_has1 = true;
    long long a_r2 = 0;                                               // floats from this thread's first A row to its second
    auto set_ptrs = [&](int pi_, int mt_, int nt_) {
        const NuGemmNT& q = P(pi_);
        ld_nk = q.K / 16;
        ld_has1 = nt_ * 256 + 128 < q.N;
        int r = mt_ * TBM + arow;
        r = r < q.M ? r : q.M - 1;
        int r2 = r + 64;
        r2 = r2 < q.M ? r2 : q.M - 1;
        ap = q.A + (long long)z * q.sA + (long long)r * q.lda + 4 * aq;
        a_r2 = (long long)(r2 - r) * q.lda;
        bbase = reinterpret_cast<const char*>(q.B6) + ((long long)z * q.sB + (long long)nt_ * 256 * q.ldb) * 6 + tid * 16;
    };
    int ld_koff = 0;                                                  // first k of the chunk at the cursor
    auto advance = [&]() {
        if (ld_j >= nslots) return;
        if (++ld_kt == ld_nk) {
            ld_kt = 0;
            int m2 = 0, n2 = 0;
            ld_j = next_valid(ld_j + gridDim.x, ld_pi, m2, n2);
            if (ld_j < nslots) set_ptrs(ld_pi, m2, n2);
        }
        if (ld_j < nslots) ld_koff = ld_kt * 16;
    };
#define N6_PIN __builtin_amdgcn_sched_barrier(0);
#define N6_LA(S, i) ra[S][i] = *reinterpret_cast<const f32x4*>(ap + ld_koff + (i) * a_r2); N6_PIN
#define N6_LB(i) if ((i) < 3 || lb_has1) rb[i] = *reinterpret_cast<const f32x4*>(lb_base + ((long long)lb_koff * (N6_BIMG / 16) + (i) * 4096)); N6_PIN
    // the B loader takes over the cursor's chunk, then the cursor moves on
    auto step = [&]() {
        lb_base = bbase; lb_koff = ld_koff; lb_has1 = ld_has1;
        advance();
    };
    // the two A pieces split into their three planes (VALU), then one 8-byte write per plane and row
    uint2 wa[2][3];
    auto split_a = [&](const f32x4 (&r)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x4 p1, p2, p3;
            nu_split3(r[i], p1, p2, p3);
            wa[i][0] = __builtin_bit_cast(uint2, p1);
            wa[i][1] = __builtin_bit_cast(uint2, p2);
            wa[i][2] = __builtin_bit_cast(uint2, p3);
        }
    };
#define N6_WA(S, i, p) *reinterpret_cast<uint2*>((S) + (p) * N6_APL + (i) * 2048 + aw_off) = wa[i][p]; N6_PIN
#define N6_WB(S, i) if ((i) < 3 || wr_has1) *reinterpret_cast<f32x4*>((S) + bw_off + (i) * 4096) = rb[i]; N6_PIN
    bool wr_has1 = true;                                              // ... of the chunk that sits in the registers

    NtEpiArgs<EPI> ea;
    if (!BATCH) ea = nt_epi_args<EPI>(P(0), z);
    constexpr bool kMaskR = NtEpiArgs<EPI>::kMaskR;

    // ---- prologue: chunk 0 -> stage 0; chunk 1 -> registers (A set 0, B); chunk 2's A piece -> A set 1; first fragments ----
    set_ptrs(pi, mt, nt);
    N6_LA(0, 0) N6_LA(0, 1)                                           // A of chunk 0
    step();                                                           // B loader on chunk 0, cursor on chunk 1
    N6_LB(0) N6_LB(1) N6_LB(2) N6_LB(3) N6_LB(4) N6_LB(5)
    wr_has1 = lb_has1;
    split_a(ra[0]);
    N6_WA(smem, 0, 0) N6_WA(smem, 0, 1) N6_WA(smem, 0, 2) N6_WA(smem, 1, 0) N6_WA(smem, 1, 1) N6_WA(smem, 1, 2)
    N6_WB(smem, 0) N6_WB(smem, 1) N6_WB(smem, 2) N6_WB(smem, 3) N6_WB(smem, 4) N6_WB(smem, 5)
    N6_LA(0, 0) N6_LA(0, 1)                                           // A of chunk 1
    step();                                                           // B loader on chunk 1, cursor on chunk 2
    N6_LB(0) N6_LB(1) N6_LB(2) N6_LB(3) N6_LB(4) N6_LB(5)
    wr_has1 = lb_has1;
    N6_LA(1, 0) N6_LA(1, 1)                                           // A of chunk 2
    step();                                                           // B loader on chunk 2, cursor on chunk 3
    __syncthreads();
    bf16x8 a[2][3], bA[3], bB[3];
#define N6_RA(S, tm, p) a[tm][p] = *reinterpret_cast<const bf16x8*>((S) + a_off + (p) * N6_APL + (tm) * 1024); N6_PIN
#define N6_RB(BS, S, c, p) BS[p] = *reinterpret_cast<const bf16x8*>((S) + bf_off + (p) * 32 + ((c) >> 1) * 12288 + ((c) & 1) * 3072); N6_PIN
    N6_RA(smem, 0, 0) N6_RA(smem, 1, 0) N6_RA(smem, 0, 1) N6_RA(smem, 1, 1) N6_RA(smem, 0, 2) N6_RA(smem, 1, 2)
    N6_RB(bA, smem, 0, 0) N6_RB(bA, smem, 0, 1) N6_RB(bA, smem, 0, 2)
    int cur = 0;

    while (true) {
        const NuGemmNT& g = P(pi);
        if (BATCH) ea = nt_epi_args<EPI>(g, 0);
        const int nk = g.K / 16;
        const int ntn128 = (g.N + TBN - 1) / TBN;
        const int m0 = mt * TBM, n0 = nt * 256;
        const bool has1 = n0 + 128 < g.N;
        unsigned long long* mwave0 = nt_mask_words<EPI>(ea, g, mt, 2 * nt, z, ntn128, wid);
        unsigned long long* mwave1 = has1 ? nt_mask_words<EPI>(ea, g, mt, 2 * nt + 1, z, ntn128, wid) : nullptr;
        unsigned long long mword0 = 0, mword1 = 0;
        if (kMaskR && mwave0) mword0 = mwave0[lane];
        if (kMaskR && mwave1) mword1 = mwave1[lane];

        f32x16 acc[2][2][2];                                          // [128-column half][tm][tn]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[s][i][jj][r] = 0.0f;

        // One chunk = four 32-column strips of 12 MFMAs (six products x two 32-row blocks; per accumulator the order of
        // gemm_nt_kernel<EPI, 2>: the 2^-16 terms, the 2^-8 terms, the leading term).  Every memory instruction sits alone between two
        // MFMAs (see gemm_nt2_kernel):
        //   strip 0: fragments of strip 1; the NEXT chunk goes registers -> other stage (B pieces; the A piece, split at the top)
        //   strip 1: fragments of strip 2; then the requests, B first and A last: B pieces of the chunk after the next (weights: L2
        //            hits, wanted one chunk later), A piece of the chunk after THAT into the set just emptied (activations: HBM, wanted
        //            two chunks later).  The order matters: memory results return in order, so a wait for the B pieces leaves only
        //            requests issued AFTER them in flight
        //   strip 2: fragments of strip 3
        //   barrier (the other stage is complete, this one fully consumed)
        //   strip 3: the next chunk's first fragments, each plane as soon as its last product has issued
#define N6_M(c, tm, pa, BS, pb) acc[(c) >> 1][tm][(c) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][pa], BS[pb], acc[(c) >> 1][tm][(c) & 1], 0, 0, 0); N6_PIN
#define N6_STRIP(c, BS, on, X0, X1, X2, X3, X4, X5, X6, X7, X8, X9, X10, X11)                        \
        if (on) {                                                                                     \
            N6_M(c, 0, 2, BS, 0) X0 N6_M(c, 1, 2, BS, 0) X1                                           \
            N6_M(c, 0, 0, BS, 2) X2 N6_M(c, 1, 0, BS, 2) X3                                           \
            N6_M(c, 0, 1, BS, 1) X4 N6_M(c, 1, 1, BS, 1) X5                                           \
            N6_M(c, 0, 1, BS, 0) X6 N6_M(c, 1, 1, BS, 0) X7                                           \
            N6_M(c, 0, 0, BS, 1) X8 N6_M(c, 1, 0, BS, 1) X9                                           \
            N6_M(c, 0, 0, BS, 0) X10 N6_M(c, 1, 0, BS, 0) X11                                         \
        } else {                                                                                      \
            X0 X1 X2 X3 X4 X5 X6 X7 X8 X9 X10 X11                                                     \
        }
#define N6_NOP
        const char* sc = smem + cur * N6_STAGE;
        // one chunk; S: the A register set that holds the next chunk's piece (and is refilled, two chunks ahead, as soon as it is split)
#define N6_ITER(S)                                                                                                                  \
        {                                                                                                                           \
            char* so = smem + (cur ^ 1) * N6_STAGE;                                                                                 \
            split_a(ra[S]);                                                                                                         \
            N6_PIN                                                                                                                  \
            N6_STRIP(0, bA, true, N6_WB(so, 0), N6_WB(so, 1), N6_RB(bB, sc, 1, 0), N6_WB(so, 2), N6_WB(so, 3), N6_RB(bB, sc, 1, 1),         \
                     N6_WB(so, 4), N6_WB(so, 5), N6_RB(bB, sc, 1, 2), N6_WA(so, 0, 0), N6_WA(so, 0, 1), N6_WA(so, 0, 2))                    \
            N6_STRIP(1, bB, true, N6_WA(so, 1, 0), N6_WA(so, 1, 1), N6_WA(so, 1, 2), N6_RB(bA, sc, 2, 0), N6_LB(0), N6_LB(1),               \
                     N6_RB(bA, sc, 2, 1), N6_LB(2), N6_LB(3), N6_RB(bA, sc, 2, 2), N6_LB(4), N6_LB(5))                                      \
            N6_STRIP(2, bA, has1, N6_LA(S, 0), N6_LA(S, 1), N6_RB(bB, sc, 3, 0), N6_NOP, N6_RB(bB, sc, 3, 1), N6_NOP,                       \
                     N6_NOP, N6_RB(bB, sc, 3, 2), N6_NOP, N6_NOP, N6_NOP, N6_NOP)                                                           \
            __syncthreads();        /* the other stage is complete; every wave holds its last fragments of this one */             \
            /* (after the very last chunk of the launch the reads below return stale bytes that are never used) */                  \
            N6_STRIP(3, bB, has1, N6_NOP, N6_NOP, N6_RA(so, 0, 2), N6_RA(so, 1, 2), N6_RB(bA, so, 0, 0), N6_RB(bA, so, 0, 1),       \
                     N6_RB(bA, so, 0, 2), N6_NOP, N6_RA(so, 0, 1), N6_RA(so, 1, 1), N6_NOP, N6_NOP)                                 \
            N6_RA(so, 0, 0) N6_RA(so, 1, 0)                                                                                         \
            wr_has1 = lb_has1;                                                                                                      \
            step();                                                                                                                 \
            cur ^= 1;                                                                                                               \
            sc = so;                                                                                                                \
        }
        for (int kt = 0; kt < nk; kt += 2) {        // K % 32 == 0: an even number of chunks per tile, the sets keep their turn across tiles
            N6_ITER(0)
            N6_ITER(1)
        }
#undef N6_ITER
        // ---- epilogue: the stage consumed last is free until the next chunk's hand-over ----
        float* scr = reinterpret_cast<float*>(smem + (cur ^ 1) * N6_STAGE) + wid * (32 * EPI_LDS);
        nt_epilogue<EPI, 4, false, 2>(g, ea, acc[0], scr, m0, n0, mwave0, (unsigned)mword0, (unsigned)(mword0 >> 32), lane, wid);
        if (has1)
            nt_epilogue<EPI, 4, false, 2>(g, ea, acc[1], scr, m0, n0 + 128, mwave1, (unsigned)mword1, (unsigned)(mword1 >> 32), lane, wid);
        int mtn = 0, ntnx = 0;
        const int jn = next_valid(j + gridDim.x, pi, mtn, ntnx);
        if (jn >= nslots) break;
        __syncthreads();            // every wave is done with the scratch before the next hand-over writes that stage
        j = jn; mt = mtn; nt = ntnx;
    }
#undef N6_M
#undef N6_STRIP
#undef N6_NOP
#undef N6_RA
#undef N6_RB
#undef N6_WA
#undef N6_WB
#undef N6_LA
#undef N6_LB
#undef N6_PIN
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt6_kernel(NuGemmNT g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * N6_STAGE];
    nt6_run<EPI, false>(&g, nullptr, 1, smem);
}

// launch (called by nu_gemm_nt_launch after the common argument checks): mode 2 with a pre-split weight table
int nu_gemm_nt6_launch(const NuGemmNT& g, int groups, hipStream_t stream) {
    if ((g.ldb & 15) || ((uintptr_t)g.B6 & 15) || (g.sB & 15) || (g.K & 15)) return NU_ERR_ARG;
    static const int grid_env = getenv("NU_NT_GRID") ? atoi(getenv("NU_NT_GRID")) : 0;
    const long long nslots = (long long)nu_rup(nu_cdiv(g.M, TBM), 8) * nu_cdiv(g.N, 256);
    long long per = nu_rup(nu_cdiv(grid_env ? grid_env : 512, groups), 8);           // two workgroups per CU
    if (per > nslots) per = nslots;
    dim3 grid((unsigned)per, 1, groups), block(256);
    switch (g.epi) {
#define NU_CASE6(E) case E: hipLaunchKernelGGL((gemm_nt6_kernel<E>), grid, block, 0, stream, g); break;
        NU_CASE6(NU_EPI_BIAS_NONE)
        NU_CASE6(NU_EPI_BIAS_RELU)
        NU_CASE6(NU_EPI_BIAS_SOFTPLUS)
        NU_CASE6(NU_EPI_MUL_DRELU)
        NU_CASE6(NU_EPI_MUL_DSP)
        NU_CASE6(NU_EPI_Q_SP)
        NU_CASE6(NU_EPI_B_SP)
        NU_CASE6(NU_EPI_PLAIN)
        NU_CASE6(NU_EPI_B_RELU)
#undef NU_CASE6
        default: return NU_ERR_ARG;
    }
    return nu_launch_status();
}
