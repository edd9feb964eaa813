// gemm.hip -- fp32 MFMA GEMM kernels (gfx950).  See gemm.h for the contract.
//
// Replaces the reference's eager `lin(x)` / autograd matmul chains:
//   network/field.py:133-150 (SDFNetwork.forward), :158-170 (.gradient), :265-289 (NeRFNetwork),
//   :371-408 (make_predictor stacks) and their autograd backward / double backward.
#include "gemm_epi.h"
#include <stdlib.h>

template <int EPI, int PREC>
__global__ __launch_bounds__(256, PREC == 2 ? 2 : NT_WPC) void gemm_nt_kernel(NuGemmNT g) {
    constexpr bool BF16 = PREC == 1;
    constexpr bool SPLIT = PREC == 2;
    constexpr int kPlane = TBM * NT_LDSH;                       // bf16 elements of one [128][32 (+8 pad)] image
    // fp32: 2 x 128 x 36 floats (36864 B).  bf16: 2 images.  split: 6 images (61440 B).  The epilogue scratch aliases it.
    __shared__ __attribute__((aligned(16))) float smem[SPLIT ? 1 : 2][SPLIT ? 3 * kPlane : TBM * NT_LDS];
    // bf16 image: [128 rows][32 k] per operand, rows padded to NT_LDSH elements (80 B: b128 reads stay conflict-free)
    __bf16* const hA = reinterpret_cast<__bf16*>(&smem[0][0]);
    __bf16* const hB = hA + (SPLIT ? 3 : 1) * kPlane;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int z = blockIdx.z;
    const int ntn = (g.N + TBN - 1) / TBN;
    const int mtiles = (g.M + TBM - 1) / TBM;
    // slot space grouped so that the ntn column tiles of one row tile sit 8 slots apart (same XCD under
    // round-robin placement: the second reader of an A tile hits that XCD's L2 -- speed only)
    const int nslots = ((mtiles + 7) / 8) * 8 * ntn;

    const float* __restrict__ A = g.A + (long long)z * g.sA;
    const float* __restrict__ B = g.B + (long long)z * g.sB;
    const int c4 = tid & 7;
    const int r0 = tid >> 3;
    const int nk = g.K / TBK;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 64 + li) * NT_LDS + 4 * lh;
    const int b_off = (wc * 64 + li) * NT_LDS + 4 * lh;
    const int ah_off = (wr * 64 + li) * NT_LDSH + 8 * lh;   // lane (r, h) holds k = 8h .. 8h+7 of a 16-deep MFMA step
    const int bh_off = (wc * 64 + li) * NT_LDSH + 8 * lh;

    auto slot_tile = [&](int j, int& mt, int& nt) -> bool {
        const int grp = j / (8 * ntn);
        const int rem = j - grp * 8 * ntn;
        nt = rem >> 3;
        mt = grp * 8 + (rem & 7);
        return mt < mtiles;
    };
    auto next_valid = [&](int j, int& mt, int& nt) -> int {
        while (j < nslots && !slot_tile(j, mt, nt)) j += gridDim.x;
        return j;
    };

    int mt = 0, nt = 0;
    int j = next_valid(blockIdx.x, mt, nt);
    if (j >= nslots) return;

    const float* ap[4];
    const float* bp[4];
    f32x4 ra4[4], rb4[4];
    auto set_ptrs = [&](int mt_, int nt_) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ra = mt_ * TBM + r0 + 32 * i;
            ra = ra < g.M ? ra : g.M - 1;
            ap[i] = A + (long long)ra * g.lda + 4 * c4;
            bp[i] = B + (long long)(nt_ * TBN + r0 + 32 * i) * g.ldb + 4 * c4;
        }
    };
    auto load_regs = [&](int koff) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra4[i] = *reinterpret_cast<const f32x4*>(ap[i] + koff);
            rb4[i] = *reinterpret_cast<const f32x4*>(bp[i] + koff);
        }
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SPLIT) {
                bf16x4 p1, p2, p3;
                nu_split3(ra4[i], p1, p2, p3);
                __bf16* q = &hA[(r0 + 32 * i) * NT_LDSH + 4 * c4];
                *reinterpret_cast<bf16x4*>(q) = p1;
                *reinterpret_cast<bf16x4*>(q + kPlane) = p2;
                *reinterpret_cast<bf16x4*>(q + 2 * kPlane) = p3;
                nu_split3(rb4[i], p1, p2, p3);
                q = &hB[(r0 + 32 * i) * NT_LDSH + 4 * c4];
                *reinterpret_cast<bf16x4*>(q) = p1;
                *reinterpret_cast<bf16x4*>(q + kPlane) = p2;
                *reinterpret_cast<bf16x4*>(q + 2 * kPlane) = p3;
            } else if (BF16) {
                *reinterpret_cast<bf16x4*>(&hA[(r0 + 32 * i) * NT_LDSH + 4 * c4]) = nu_to_bf16x4(ra4[i]);
                *reinterpret_cast<bf16x4*>(&hB[(r0 + 32 * i) * NT_LDSH + 4 * c4]) = nu_to_bf16x4(rb4[i]);
            } else {
                float* s0 = &smem[0][0];
                *reinterpret_cast<f32x4*>(&s0[(r0 + 32 * i) * NT_LDS + 4 * c4]) = ra4[i];
                *reinterpret_cast<f32x4*>(&s0[TBM * NT_LDS + (r0 + 32 * i) * NT_LDS + 4 * c4]) = rb4[i];
            }
        }
    };

    set_ptrs(mt, nt);
    load_regs(0);
    store_regs();
    __syncthreads();

    const NtEpiArgs<EPI> ea = nt_epi_args<EPI>(g, z);
    constexpr bool kMaskR = NtEpiArgs<EPI>::kMaskR;

    while (true) {
        int mtn = 0, ntnx = 0;
        const int jn = next_valid(j + gridDim.x, mtn, ntnx);
        const bool has_next = jn < nslots;
        const int m0 = mt * TBM, n0 = nt * TBN;
        // ReLU sign bits of this wave's 64x64 slab: 64 ballot words [tm][i][e], one per lane.  The reader fetches its word
        // here, a whole main loop ahead of the epilogue (the point of the exercise: no load latency left in the epilogue)
        unsigned long long* mwave = nt_mask_words<EPI>(ea, g, mt, nt, z, ntn, wid);
        unsigned long long mword = 0;
        if (kMaskR && mwave) mword = mwave[lane];
        const unsigned mlo = (unsigned)mword, mhi = (unsigned)(mword >> 32);

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;

        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                load_regs((kt + 1) * TBK);
            } else if (has_next) {
                set_ptrs(mtn, ntnx);
                load_regs(0);
            }
            if (SPLIT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 a[2][3], b[2][3];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            a[t][p] = *reinterpret_cast<const bf16x8*>(&hA[p * kPlane + ah_off + 32 * t * NT_LDSH + 16 * ks]);
                            b[t][p] = *reinterpret_cast<const bf16x8*>(&hB[p * kPlane + bh_off + 32 * t * NT_LDSH + 16 * ks]);
                        }
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) {
                            f32x16 c = acc[tm][tn];
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], c, 0, 0, 0);   // 2^-16 terms
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], c, 0, 0, 0);   // 2^-8 terms
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], c, 0, 0, 0);   // leading term
                            acc[tm][tn] = c;
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
                const float* As = &smem[0][0];
                const float* Bs = As + TBM * NT_LDS;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[a_off + kk * 8]);
                    f32x4 a1 = *reinterpret_cast<const f32x4*>(&As[a_off + 32 * NT_LDS + kk * 8]);
                    f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bs[b_off + kk * 8]);
                    f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bs[b_off + 32 * NT_LDS + kk * 8]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
                    }
                }
            }
            __syncthreads();   // every wave is done reading this chunk (and, after the last one, the scratch is free)
            if (kt + 1 < nk) {
                store_regs();
                __syncthreads();
            }
        }

        // ---- epilogue ----
        nt_epilogue<EPI, 2>(g, ea, acc, &smem[0][0] + wid * (32 * EPI_LDS), m0, n0, mwave, mlo, mhi, lane, wid);
        if (!has_next) break;
        __syncthreads();   // every wave is done with the scratch
        store_regs();
        __syncthreads();
        j = jn; mt = mtn; nt = ntnx;
    }
}

// ------------------------------------------------------------------------------------------------
// NT kernel, exact fp32 MFMA, second generation: the main loop is software-pipelined INSIDE each wave.
//
// What the phase stamps of scripts/gemm_lab.hip showed about the first-generation kernel above (K = 256, 128 x 128 tiles):
// a workgroup alone on its CU spends 22 us in a main loop whose 512 MFMAs per wave take 13.9 us -- every k-chunk pays
// fragment-read latency (the ds_reads of the next k-group issue right before the last MFMA of the current one), a vmcnt
// wait + eight ds_write_b128 + lgkmcnt(0) between two barriers, and a second fragment-read latency after them; three
// co-resident workgroups only hide part of that (82 % of the matrix pipe without any epilogue).  Here
//   * LDS holds TWO stages (73.7 KB, two workgroups per CU, up to 256 VGPRs): the next chunk is written to the other stage
//     in the MIDDLE of the current chunk's MFMAs, so a chunk has ONE barrier and no store sits between barriers;
//   * fragments are double-buffered in registers: the reads of k-group kk+1 issue before the MFMAs of kk;
//   * the barrier sits after the third k-group; the first fragments of the NEXT chunk are read right behind it, under the
//     16 MFMAs of the fourth k-group -- the matrix pipe never waits for LDS at a chunk boundary;
//   * global loads run two chunks ahead of the MFMAs (one chunk in registers, one in LDS), across tile boundaries.
// __builtin_amdgcn_sched_barrier(0) pins the order of the stages; inside a stage the compiler schedules freely.
// The epilogue (shared with the first generation) uses the stage that was just consumed as its scratch.
// ------------------------------------------------------------------------------------------------
#define NT2_STAGE (2 * TBM * NT_LDS)   // floats per stage: [A 128 x 36 | B 128 x 36]
// TMN = 2: 128 x 128 tiles (wave tile 64 x 64).  TMN = 1: 64 x 128 tiles (4 waves as 2 x 2, wave tile 32 x 64) for launches whose
// 128-row tiles would leave CUs empty (point sets of a few thousand rows: the reference's default batch of 512 rays, the stage-2
// segments): twice the tiles, half the work each; same stages, same chunk flow, the A pieces 2-3 and the a1 fragments drop out.
// BATCH: ONE persistent grid walks the tiles of several independent problems (NuGemmNTBatch: own M, N, K, pointers, epilogue
// arguments; the same epilogue kind and tile height).  A problem's slots form a contiguous range of the slot space (each range a
// multiple of 8 slots, so a tile's column tiles still share an XCD); the loader cursor and the compute cursor each remember which
// problem they are in, and everything that used to be a launch constant (operand bases, K, the epilogue arguments) is re-read
// from the kernel-argument block at a tile switch.  What it buys: the last, partly filled round of a persistent launch is paid
// once per list instead of once per problem, and launches whose tiles do not fill the chip (the light predictors of a 512-ray
// batch, the stage-2 networks) share it.
struct NuGemmNTBatch {
    NuGemmNT p[NU_NT_BATCH_MAX];
    int slot0[NU_NT_BATCH_MAX + 1];       // first slot of problem i; slot0[n] = all slots
    int n, pad_;
};

template <int EPI, int TMN, bool BATCH>
static __device__ __forceinline__ void nt2_run(const NuGemmNT* __restrict__ probs, const int* __restrict__ slot0, const int nprob,
                                               float* __restrict__ smem) {
    constexpr int BM = 64 * TMN;                            // tile rows
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int z = BATCH ? 0 : blockIdx.z;
    auto P = [&](int pi) -> const NuGemmNT& { return probs[BATCH ? pi : 0]; };
    const int c4 = tid & 7;
    const int r0 = tid >> 3;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 32 * TMN + li) * NT_LDS + 4 * lh;
    const int b_off = TBM * NT_LDS + (wc * 64 + li) * NT_LDS + 4 * lh;
    const int a0_off = a_off, a1_off = a_off + 32 * NT_LDS, b0_off = b_off, b1_off = b_off + 32 * NT_LDS;   // the four fragments
    const int w_off = r0 * NT_LDS + 4 * c4;                 // this thread's slot of a staged operand row group

    // slot order inside a problem: see gemm_nt_kernel
    const int nslots = BATCH ? slot0[nprob] : ((((P(0).M + BM - 1) / BM) + 7) / 8) * 8 * ((P(0).N + TBN - 1) / TBN);
    auto slot_tile = [&](int j, int& pi, int& mt, int& nt) -> bool {
        if (BATCH)
            while (pi + 1 < nprob && j >= slot0[pi + 1]) ++pi;      // (the cursors only move forward)
        const NuGemmNT& q = P(pi);
        const int ntn = (q.N + TBN - 1) / TBN, mtiles = (q.M + BM - 1) / BM;
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
    int ld_j = j, ld_kt = 0, ld_pi = pi, ld_nk = 1;         // next chunk to fetch (ld_nk: chunks per tile of the loader's problem)
    const float* ap[4];
    const float* bp[4];
    f32x4 ra4[4], rb4[4];
    auto set_ptrs = [&](int pi_, int mt_, int nt_) {
        const NuGemmNT& q = P(pi_);
        const float* __restrict__ A = q.A + (long long)z * q.sA;
        const float* __restrict__ B = q.B + (long long)z * q.sB;
        ld_nk = q.K / TBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ra = mt_ * BM + r0 + 32 * (i < 2 * TMN ? i : 0);
            ra = ra < q.M ? ra : q.M - 1;
            ap[i] = A + (long long)ra * q.lda + 4 * c4;
            bp[i] = B + (long long)(nt_ * TBN + r0 + 32 * i) * q.ldb + 4 * c4;
        }
    };
    // The loader is split in three so that its pieces can sit BETWEEN MFMAs (an in-order wave issues nothing while it waits
    // for the matrix pipe; a ds_write_b128 takes ~43 cycles to issue, a global_load_dwordx4 ~34, an MFMA occupies the pipe 64):
    //   load_piece(i)   global -> registers, rows r0 + 32 i of A and B, chunk at the cursor (always a valid address: past the
    //                   last chunk the cursor stays on the last tile and the data is never used)
    //   advance()       moves the cursor to the next chunk (scalar bookkeeping; pointer set-up once per tile)
    //   write_piece(st, i)   registers -> LDS stage st
    int ld_koff = 0;
    auto load_piece = [&](int i) {
        if (i < 2 * TMN) ra4[i] = *reinterpret_cast<const f32x4*>(ap[i] + ld_koff);
        rb4[i] = *reinterpret_cast<const f32x4*>(bp[i] + ld_koff);
    };
    auto advance = [&]() {
        if (ld_j >= nslots) return;
        if (++ld_kt == ld_nk) {
            ld_kt = 0;
            int m2 = 0, n2 = 0;
            ld_j = next_valid(ld_j + gridDim.x, ld_pi, m2, n2);
            if (ld_j < nslots) set_ptrs(ld_pi, m2, n2);
        }
        if (ld_j < nslots) ld_koff = ld_kt * TBK;
    };
    auto write_piece = [&](int st, int i) {
        float* s0 = &smem[st * NT2_STAGE];
        if (i < 2 * TMN) *reinterpret_cast<f32x4*>(&s0[w_off + 32 * i * NT_LDS]) = ra4[i];
        *reinterpret_cast<f32x4*>(&s0[TBM * NT_LDS + w_off + 32 * i * NT_LDS]) = rb4[i];
    };
    struct Frag { f32x4 a0, a1, b0, b1; };
    auto read_frag = [&](Frag& f, int st, int kk) {
        const float* s0 = &smem[st * NT2_STAGE];
        f.a0 = *reinterpret_cast<const f32x4*>(&s0[a_off + kk * 8]);
        if (TMN == 2) f.a1 = *reinterpret_cast<const f32x4*>(&s0[a_off + 32 * NT_LDS + kk * 8]);
        f.b0 = *reinterpret_cast<const f32x4*>(&s0[b_off + kk * 8]);
        f.b1 = *reinterpret_cast<const f32x4*>(&s0[b_off + 32 * NT_LDS + kk * 8]);
    };

    NtEpiArgs<EPI> ea;
    if (!BATCH) ea = nt_epi_args<EPI>(P(0), z);
    constexpr bool kMaskR = NtEpiArgs<EPI>::kMaskR;

    // ---- prologue: chunk 0 -> stage 0, chunk 1 -> registers, first fragments ----
    set_ptrs(pi, mt, nt);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(i);
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) write_piece(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(i);
    advance();
    __syncthreads();
    Frag F0, F1;
    read_frag(F0, 0, 0);
    int cur = 0;

    while (true) {
        const NuGemmNT& g = P(pi);
        if (BATCH) ea = nt_epi_args<EPI>(g, 0);
        const int nk = g.K / TBK;
        const int ntn = (g.N + TBN - 1) / TBN;
        // the 128 x 128 tile this tile is (part of), the 64 x 64 slab of it this wave works in, and the wave's 32-row block there
        const int mt128 = TMN == 2 ? mt : (mt >> 1);
        const int slab = TMN == 2 ? wid : (((mt & 1) << 1) | wc), tm0 = TMN == 2 ? 0 : wr;
        const int m0 = mt128 * TBM, n0 = nt * TBN;
        unsigned long long* mwave = nt_mask_words<EPI>(ea, g, mt128, nt, z, ntn, slab);
        unsigned long long mword = 0;
        if (kMaskR && mwave) mword = mwave[lane];
        const unsigned mlo = (unsigned)mword, mhi = (unsigned)(mword >> 32);

        f32x16 acc[TMN][2];
#pragma unroll
        for (int i = 0; i < TMN; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.0f;
#define NT2_PIN __builtin_amdgcn_sched_barrier(0);
#define NT2_M(F, e, i, j) if constexpr (i < TMN) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a##i[e], F.b##j[e], acc[i][j], 0, 0, 0); NT2_PIN }
        // One chunk = 4 k-groups of 16 MFMAs.  An in-order wave issues nothing while one of its own instructions issues, and the
        // matrix pipe runs dry when that takes longer than the MFMA in front of it executes (64 cycles): a ds_write_b128 costs
        // ~43 issue cycles, a global_load_dwordx4 ~34, a ds_read_b128 ~20 -- ONE of them hides behind an MFMA, two in a row do not
        // (measured with per-workgroup phase stamps: 4 reads + 8 writes in two gaps cost a lone workgroup 200 of a k-group's 1024 cycles).
        // So every memory instruction of the chunk sits ALONE between two MFMAs, pinned by sched_barrier(0):
        //   k-group 0: the 4 fragment reads of k-group 1
        //   k-group 1: the NEXT chunk goes registers -> other LDS stage (8 ds_write_b128); the 4 fragment reads of k-group 2
        //   k-group 2: the chunk AFTER that goes global -> registers (8 global_load_dwordx4); the 4 fragment reads of k-group 3
        //   barrier (the other stage is complete, this one fully consumed)
        //   k-group 3: the first fragments of the next chunk
        const float* sc = &smem[cur * NT2_STAGE];
        for (int kt = 0; kt < nk; ++kt) {
            float* so = &smem[(cur ^ 1) * NT2_STAGE];
#define NT2_RD(F, S, kk, m) F.m = *reinterpret_cast<const f32x4*>(&(S)[m##_off + (kk) * 8]); NT2_PIN
#define NT2_RD1(F, S, kk) if constexpr (TMN == 2) { NT2_RD(F, S, kk, a1) }
#define NT2_WA(i) if constexpr (i < 2 * TMN) { *reinterpret_cast<f32x4*>(&so[w_off + 32 * (i) * NT_LDS]) = ra4[i]; NT2_PIN }
#define NT2_WB(i) *reinterpret_cast<f32x4*>(&so[TBM * NT_LDS + w_off + 32 * (i) * NT_LDS]) = rb4[i]; NT2_PIN
#define NT2_LA(i) if constexpr (i < 2 * TMN) { ra4[i] = *reinterpret_cast<const f32x4*>(ap[i] + ld_koff); NT2_PIN }
#define NT2_LB(i) rb4[i] = *reinterpret_cast<const f32x4*>(bp[i] + ld_koff); NT2_PIN
            NT2_M(F0, 0, 0, 0) NT2_M(F0, 0, 0, 1) NT2_RD(F1, sc, 1, a0) NT2_M(F0, 0, 1, 0) NT2_M(F0, 0, 1, 1)
            NT2_M(F0, 1, 0, 0) NT2_M(F0, 1, 0, 1) NT2_RD1(F1, sc, 1) NT2_M(F0, 1, 1, 0) NT2_M(F0, 1, 1, 1)
            NT2_M(F0, 2, 0, 0) NT2_M(F0, 2, 0, 1) NT2_RD(F1, sc, 1, b0) NT2_M(F0, 2, 1, 0) NT2_M(F0, 2, 1, 1)
            NT2_M(F0, 3, 0, 0) NT2_M(F0, 3, 0, 1) NT2_RD(F1, sc, 1, b1) NT2_M(F0, 3, 1, 0) NT2_M(F0, 3, 1, 1)
            NT2_M(F1, 0, 0, 0) NT2_WA(0) NT2_M(F1, 0, 0, 1) NT2_RD(F0, sc, 2, a0) NT2_M(F1, 0, 1, 0) NT2_WB(0) NT2_M(F1, 0, 1, 1)
            NT2_M(F1, 1, 0, 0) NT2_WA(1) NT2_M(F1, 1, 0, 1) NT2_RD1(F0, sc, 2) NT2_M(F1, 1, 1, 0) NT2_WB(1) NT2_M(F1, 1, 1, 1)
            NT2_M(F1, 2, 0, 0) NT2_WA(2) NT2_M(F1, 2, 0, 1) NT2_RD(F0, sc, 2, b0) NT2_M(F1, 2, 1, 0) NT2_WB(2) NT2_M(F1, 2, 1, 1)
            NT2_M(F1, 3, 0, 0) NT2_WA(3) NT2_M(F1, 3, 0, 1) NT2_RD(F0, sc, 2, b1) NT2_M(F1, 3, 1, 0) NT2_WB(3) NT2_M(F1, 3, 1, 1)
            NT2_M(F0, 0, 0, 0) NT2_LA(0) NT2_M(F0, 0, 0, 1) NT2_RD(F1, sc, 3, a0) NT2_M(F0, 0, 1, 0) NT2_LB(0) NT2_M(F0, 0, 1, 1)
            NT2_M(F0, 1, 0, 0) NT2_LA(1) NT2_M(F0, 1, 0, 1) NT2_RD1(F1, sc, 3) NT2_M(F0, 1, 1, 0) NT2_LB(1) NT2_M(F0, 1, 1, 1)
            NT2_M(F0, 2, 0, 0) NT2_LA(2) NT2_M(F0, 2, 0, 1) NT2_RD(F1, sc, 3, b0) NT2_M(F0, 2, 1, 0) NT2_LB(2) NT2_M(F0, 2, 1, 1)
            NT2_M(F0, 3, 0, 0) NT2_LA(3) NT2_M(F0, 3, 0, 1) NT2_RD(F1, sc, 3, b1) NT2_M(F0, 3, 1, 0) NT2_LB(3) NT2_M(F0, 3, 1, 1)
            advance();
            __syncthreads();        // the other stage is complete; every wave holds its last fragments of this one
            // (after the very last chunk the reads below return stale bytes that are never used)
            NT2_M(F1, 0, 0, 0) NT2_M(F1, 0, 0, 1) NT2_RD(F0, so, 0, a0) NT2_M(F1, 0, 1, 0) NT2_M(F1, 0, 1, 1)
            NT2_M(F1, 1, 0, 0) NT2_M(F1, 1, 0, 1) NT2_RD1(F0, so, 0) NT2_M(F1, 1, 1, 0) NT2_M(F1, 1, 1, 1)
            NT2_M(F1, 2, 0, 0) NT2_M(F1, 2, 0, 1) NT2_RD(F0, so, 0, b0) NT2_M(F1, 2, 1, 0) NT2_M(F1, 2, 1, 1)
            NT2_M(F1, 3, 0, 0) NT2_M(F1, 3, 0, 1) NT2_RD(F0, so, 0, b1) NT2_M(F1, 3, 1, 0) NT2_M(F1, 3, 1, 1)
            cur ^= 1;
            sc = so;
        }
#undef NT2_M
#undef NT2_RD
#undef NT2_RD1
#undef NT2_WA
#undef NT2_WB
#undef NT2_LA
#undef NT2_LB
#undef NT2_PIN
        // ---- epilogue: the stage consumed last (cur ^ 1 after the flip) is free until the next chunk's hand-over ----
        nt_epilogue<EPI, 4, false, TMN>(g, ea, acc, &smem[(cur ^ 1) * NT2_STAGE] + wid * (32 * EPI_LDS), m0, n0, mwave, mlo, mhi, lane, slab, tm0);
        int mtn = 0, ntnx = 0;
        const int jn = next_valid(j + gridDim.x, pi, mtn, ntnx);
        if (jn >= nslots) break;
        __syncthreads();            // every wave is done with the scratch before the next hand-over writes that stage
        j = jn; mt = mtn; nt = ntnx;
    }
}

template <int EPI, int TMN>
__global__ __launch_bounds__(256, 2) void gemm_nt2_kernel(NuGemmNT g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * NT2_STAGE];
    nt2_run<EPI, TMN, false>(&g, nullptr, 1, smem);
}
template <int EPI, int TMN>
__global__ __launch_bounds__(256, 2) void gemm_nt2b_kernel(NuGemmNTBatch b) {
    __shared__ __attribute__((aligned(16))) float smem[2 * NT2_STAGE];
    nt2_run<EPI, TMN, true>(b.p, b.slot0, b.n, smem);
}

int nu_gemm_nt16_launch(const NuGemmNT& g, int groups, long long nslots, hipStream_t stream);      // gemm_nt16.hip
int nu_gemm_nt6_launch(const NuGemmNT& g, int groups, hipStream_t stream);                          // gemm_nt6.hip

// argument checks every NT launch path shares
static int nt_check(const NuGemmNT& g) {
    if (g.N <= 0 || g.K <= 0 || (g.K % TBK) != 0 || g.lda < g.K || g.ldb < g.K) return NU_ERR_ARG;
    if ((g.lda & 3) || (g.ldb & 3)) return NU_ERR_ARG;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15)) return NU_ERR_ARG;
    const int ntn = nu_cdiv(g.N, TBN);
    const long long nslots = (long long)nu_rup(nu_cdiv(g.M, 64), 8) * ntn;
    if (nslots > 0x7fffffffLL) return NU_ERR_ARG;
    const int groups = g.groups > 0 ? g.groups : 1;
    if (g.mask && (g.epi == NU_EPI_BIAS_RELU || g.epi == NU_EPI_MUL_DRELU || g.epi == NU_EPI_B_RELU)) {
        // the sign-bit path lives in the 16-byte epilogue only: every lane's 4 columns inside N, matrices aligned
        if ((g.N & 3) || (g.ldc & 3) || ((uintptr_t)g.C & 15) || g.mask_nct <= 0 || g.mask_ct0 < 0) return NU_ERR_ARG;
        if (g.epi == NU_EPI_B_RELU && ((g.ldadd & 3) || ((uintptr_t)g.Cadd & 15))) return NU_ERR_ARG;
        if (g.epi == NU_EPI_BIAS_RELU && ((long long)g.mask_ct0 + (long long)groups * ntn > g.mask_nct || (g.N & 63))) return NU_ERR_ARG;
        if (g.act_cols > 0 && (g.act_cols & 63)) return NU_ERR_ARG;
    }
    const int prec = g.bf16 & 3;
    if (prec == 3) return NU_ERR_ARG;
    if ((g.bf16 & ~3) && !(prec == 1 && (g.bf16 & NU_GEMM_B16)) && !(prec == 2 && (g.bf16 & ~3) == NU_GEMM_PRESPLIT_ALWAYS && g.B6))
        return NU_ERR_ARG;                                                                  // storage flags need the bf16-storage kernel
    return NU_OK;
}

// 64-row tiles when the 128-row tiles cannot give every CU its two workgroups (point sets of a few thousand rows) ... and when
// they shorten the last round: the persistent grid walks ceil(tiles / 512) rounds, so 1054 tiles of 128 rows (a 67 k-row point
// set, the outer points of a 512-ray batch) take three rounds for 2.06 rounds of work; as 2108 tiles of 64 rows they take 5 for
// 4.12.  A 64-row tile costs ~5 % more per FLOP (half the reuse of the weight tile).
static bool nt_small_tiles(long long t128, long long t64) {
    static const int small_env = getenv("NU_NT_SMALL") ? atoi(getenv("NU_NT_SMALL")) : -1;     // development switch: 0 never, 1 always
    auto round_eff = [&](long long tiles) { return (double)tiles / (double)(nu_cdivl(tiles, 512) * 512); };
    return small_env >= 0 ? small_env != 0 : (t128 < 512 || 0.95 * round_eff(t64) > round_eff(t128));
}

int nu_gemm_nt_launch(const NuGemmNT& g, hipStream_t stream) {
    if (g.M <= 0) return NU_OK;
    const int rc = nt_check(g);
    if (rc != NU_OK) return rc;
    const int ntn = nu_cdiv(g.N, TBN);
    const long long nslots = (long long)nu_rup(nu_cdiv(g.M, TBM), 8) * ntn;
    const int groups = g.groups > 0 ? g.groups : 1;
    // persistent: NT_WPC workgroups per CU (256 CUs) shared over the groups, a multiple of 8 so the XCD grouping holds
    static const int grid_env = getenv("NU_NT_GRID") ? atoi(getenv("NU_NT_GRID")) : 0;
    const int prec = g.bf16 & 3;
    const int grid_target = grid_env ? grid_env : 256 * (prec == 2 ? 2 : NT_WPC);   // workgroups the build keeps resident
    long long per = nu_rup(nu_cdiv(grid_target, groups), 8);
    if (per > nslots) per = nslots;
    dim3 grid((unsigned)per, 1, groups), block(256);
    static const bool v1 = getenv("NU_NT_V1") && atoi(getenv("NU_NT_V1")) != 0;   // development switch: first-generation fp32 kernel
    if (prec == 1 && (g.bf16 & NU_GEMM_B16)) return nu_gemm_nt16_launch(g, groups, nslots, stream);      // bf16 storage
    // pre-split weight planes: gemm_nt6_kernel's 128 x 256 tiles, when they still fill the chip (both kernels give the same bits;
    // measured, scripts/bench_nt6.py: 65 536 rows x 256 columns = 512 tiles 145 vs 128 TFLOP/s, 16 384 rows = 128 tiles 65 vs 95)
    static const int nt6_env = getenv("NU_NT6") ? atoi(getenv("NU_NT6")) : -1;          // development switch: 0 never, 1 always
    // ... and for the epilogues without a second auxiliary matrix: Q_SP / B_SP / B_RELU run twice per 128 x 256 tile with more live
    // registers than the build has (58-63 spilled) -- in the step 215 vs 168 us (Q_SP), 169 vs 155 (B_SP), 755 vs 543 (B_RELU)
    const bool nt6_epi = g.epi != NU_EPI_Q_SP && g.epi != NU_EPI_B_SP && g.epi != NU_EPI_B_RELU;
    if (prec == 2 && g.B6 && nt6_env != 0 &&
        (nt6_env == 1 || (g.bf16 & NU_GEMM_PRESPLIT_ALWAYS) || (nt6_epi && (long long)nu_cdiv(g.M, TBM) * nu_cdiv(g.N, 256) * groups >= 320)))
        return nu_gemm_nt6_launch(g, groups, stream);
    if (prec == 0 && !v1) {
        const long long t128 = (long long)nu_cdiv(g.M, TBM) * ntn * groups, t64 = (long long)nu_cdiv(g.M, 64) * ntn * groups;
        const bool small = nt_small_tiles(t128, t64);
        const long long nslots2 = small ? (long long)nu_rup(nu_cdiv(g.M, 64), 8) * ntn : nslots;
        long long per2 = nu_rup(nu_cdiv(grid_env ? grid_env : 512, groups), 8);       // two workgroups per CU
        if (per2 > nslots2) per2 = nslots2;
        dim3 grid2((unsigned)per2, 1, groups);
        switch (g.epi) {
#define NU_CASE2(E) case E: if (small) hipLaunchKernelGGL((gemm_nt2_kernel<E, 1>), grid2, block, 0, stream, g); \
                            else hipLaunchKernelGGL((gemm_nt2_kernel<E, 2>), grid2, block, 0, stream, g); break;
            NU_CASE2(NU_EPI_BIAS_NONE)
            NU_CASE2(NU_EPI_BIAS_RELU)
            NU_CASE2(NU_EPI_BIAS_SOFTPLUS)
            NU_CASE2(NU_EPI_MUL_DRELU)
            NU_CASE2(NU_EPI_MUL_DSP)
            NU_CASE2(NU_EPI_Q_SP)
            NU_CASE2(NU_EPI_B_SP)
            NU_CASE2(NU_EPI_PLAIN)
            NU_CASE2(NU_EPI_B_RELU)
#undef NU_CASE2
            default: return NU_ERR_ARG;
        }
        return nu_launch_status();
    }
    switch (g.epi) {
#define NU_CASE(E) case E: if (prec == 2) hipLaunchKernelGGL((gemm_nt_kernel<E, 2>), grid, block, 0, stream, g); \
                           else if (prec == 1) hipLaunchKernelGGL((gemm_nt_kernel<E, 1>), grid, block, 0, stream, g); \
                           else hipLaunchKernelGGL((gemm_nt_kernel<E, 0>), grid, block, 0, stream, g); break;
        NU_CASE(NU_EPI_BIAS_NONE)
        NU_CASE(NU_EPI_BIAS_RELU)
        NU_CASE(NU_EPI_BIAS_SOFTPLUS)
        NU_CASE(NU_EPI_MUL_DRELU)
        NU_CASE(NU_EPI_MUL_DSP)
        NU_CASE(NU_EPI_Q_SP)
        NU_CASE(NU_EPI_B_SP)
        NU_CASE(NU_EPI_PLAIN)
        NU_CASE(NU_EPI_B_RELU)
#undef NU_CASE
        default: return NU_ERR_ARG;
    }
    return nu_launch_status();
}

// Several independent problems in ONE persistent launch (exact fp32, one epilogue kind).  A problem with `groups` > 1 is expanded
// into its groups (operand pointers advanced by the strides, sign-bit column tiles by mask_ct0).  Whatever the batch kernel does
// not cover -- other arithmetic modes, epilogue kinds without a batch build, NU_NT_BATCH=0 -- runs as one launch per problem: same
// results either way, bit for bit (a tile's arithmetic does not depend on which launch it belongs to).
int nu_gemm_nt_batch_launch(const NuGemmNT* probs, int n, hipStream_t stream) {
    static const bool batch_on = !(getenv("NU_NT_BATCH") && atoi(getenv("NU_NT_BATCH")) == 0);     // development switch (A/B)
    static const bool v1 = getenv("NU_NT_V1") && atoi(getenv("NU_NT_V1")) != 0;
    static const int grid_env = getenv("NU_NT_GRID") ? atoi(getenv("NU_NT_GRID")) : 0;
    int i = 0;
    while (i < n) {
        // the longest run of problems starting at i that one launch can take
        NuGemmNTBatch b;
        int nb = 0, last = i;
        long long t128 = 0, t64 = 0;
        const int epi = probs[i].epi;
        const bool epi_ok = epi == NU_EPI_BIAS_NONE || epi == NU_EPI_BIAS_RELU || epi == NU_EPI_MUL_DRELU || epi == NU_EPI_PLAIN;
        for (int k = i; k < n && batch_on && !v1 && epi_ok; ++k) {
            const NuGemmNT& g = probs[k];
            if (g.M <= 0) { last = k + 1; continue; }
            const int groups = g.groups > 0 ? g.groups : 1;
            if (g.epi != epi || (g.bf16 & 3) != 0 || (g.bf16 & ~3) != 0 || nb + groups > NU_NT_BATCH_MAX) break;
            const int rc = nt_check(g);
            if (rc != NU_OK) return rc;
            const int ntn = nu_cdiv(g.N, TBN);
            for (int z = 0; z < groups; ++z) {
                NuGemmNT& q = b.p[nb++];
                q = g;
                q.groups = 1;
                q.A = g.A + z * g.sA; q.B = g.B + z * g.sB; q.C = g.C + z * g.sC;
                q.C2 = g.C2 ? g.C2 + z * g.sC2 : nullptr; q.bias = g.bias ? g.bias + z * g.sBias : nullptr;
                q.H = g.H ? g.H + z * g.sH : nullptr; q.D = g.D ? g.D + z * g.sD : nullptr; q.Cadd = g.Cadd ? g.Cadd + z * g.sCadd : nullptr;
                q.mask_ct0 = g.mask_ct0 + z * ntn;
                t128 += (long long)nu_cdiv(g.M, TBM) * ntn;
                t64 += (long long)nu_cdiv(g.M, 64) * ntn;
            }
            last = k + 1;
        }
        if (nb < 2) {                  // nothing to share a launch with: the single-problem path (all modes)
            const int rc = nu_gemm_nt_launch(probs[i], stream);
            if (rc != NU_OK) return rc;
            ++i;
            continue;
        }
        const bool small = nt_small_tiles(t128, t64);
        long long slots = 0;
        for (int k = 0; k < nb; ++k) {
            b.slot0[k] = (int)slots;
            slots += (long long)nu_rup(nu_cdiv(b.p[k].M, small ? 64 : TBM), 8) * nu_cdiv(b.p[k].N, TBN);
            if (slots > 0x7fffffffLL) return NU_ERR_ARG;
        }
        b.slot0[nb] = (int)slots;
        b.n = nb; b.pad_ = 0;
        long long per = nu_rup(grid_env ? grid_env : 512, 8);       // two workgroups per CU
        if (per > slots) per = slots;
        dim3 grid((unsigned)per, 1, 1), block(256);
        switch (epi) {
#define NU_CASEB(E) case E: if (small) hipLaunchKernelGGL((gemm_nt2b_kernel<E, 1>), grid, block, 0, stream, b); \
                            else hipLaunchKernelGGL((gemm_nt2b_kernel<E, 2>), grid, block, 0, stream, b); break;
            NU_CASEB(NU_EPI_BIAS_NONE)
            NU_CASEB(NU_EPI_BIAS_RELU)
            NU_CASEB(NU_EPI_MUL_DRELU)
            NU_CASEB(NU_EPI_PLAIN)
#undef NU_CASEB
            default: return NU_ERR_ARG;
        }
        const int rc = nu_launch_status();
        if (rc != NU_OK) return rc;
        i = last;
    }
    return NU_OK;
}
extern "C" int nu_gemm_nt_batch(const NuGemmNT* problems_host, int n, hipStream_t stream) {
    if (n < 0 || (n > 0 && problems_host == nullptr)) return NU_ERR_ARG;
    return nu_gemm_nt_batch_launch(problems_host, n, stream);
}
