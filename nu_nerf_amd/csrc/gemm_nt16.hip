// gemm_nt16.hip -- NT GEMM kernels for bf16 STORAGE (cfg mlp_dtype 'bf16', BASELINE config 4).  See gemm.h for the contract.
//
// Replaces the same reference code as gemm_nt.hip (network/field.py:133-150, :158-170, :265-289, :371-408) when the hidden
// activations and weight tables live in HBM as bf16.
#include "gemm_epi.h"
#include <stdlib.h>

#define NT2_STAGE (2 * TBM * NT_LDS)   // floats per stage: [A 128 x 36 | B 128 x 36]

// ------------------------------------------------------------------------------------------------
// NT kernel, bf16 STORAGE (cfg mlp_dtype 'bf16', BASELINE config 4): hidden activations and the weight tables live in HBM as
// bf16, products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, epilogue arithmetic in fp32.  At 16x the fp32 MFMA rate
// these GEMMs are HBM-bound (a 256 -> 256 layer moves 1 KB per point in bf16 against 131 kFLOP), so the design goal is bytes:
// every [P, 256] activation crosses HBM as 512 B, once.
//
// Same tile walk, staging and pipeline as the second-generation fp32 kernel: a 64-deep bf16 chunk of a 128-row operand is
// byte-for-byte the geometry of a 32-deep fp32 chunk (128 B per row + 16 B pad), so one 16-byte LDS fragment read IS the
// 8 x bf16 operand of one MFMA (lane (r, h) holds k = 8h .. 8h+7) and a chunk is 4 k-groups of 4 MFMAs.  Operand A may
// still be fp32 (NU_GEMM_A16 clear: network inputs, buffers that elementwise kernels also touch): it is then fetched as two
// 16-byte pieces per slot and rounded (RNE) on its way into LDS.  K % 32 == 0; a trailing half chunk is zero-filled.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt16_kernel(NuGemmNT g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * NT2_STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int z = blockIdx.z;
    const int ntn = (g.N + TBN - 1) / TBN;
    const int mtiles = (g.M + TBM - 1) / TBM;
    const int nslots = ((mtiles + 7) / 8) * 8 * ntn;       // slot order: see gemm_nt_kernel
    const bool a16 = (g.bf16 & NU_GEMM_A16) != 0;
    const int ea_b = a16 ? 2 : 4;
    const char* __restrict__ A = reinterpret_cast<const char*>(g.A) + (long long)z * g.sA * ea_b;
    const char* __restrict__ B = reinterpret_cast<const char*>(g.B) + (long long)z * g.sB * 2;
    const int c8 = tid & 7;                                 // 16-byte LDS slot = 8 consecutive k
    const int r0 = tid >> 3;
    const int nk = (g.K + 63) / 64;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 64 + li) * NT_LDS + 4 * lh;     // in floats (16-byte slots), as in the fp32 kernel
    const int b_off = TBM * NT_LDS + (wc * 64 + li) * NT_LDS + 4 * lh;
    const int w_off = r0 * NT_LDS + 4 * c8;

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

    int ld_j = j, ld_kt = 0;                                // loader cursor: next chunk to fetch
    const char* ap[4];
    const char* bp[4];
    f32x4 ra4[4], ra4b[4], rb4[4];
    bool rz = false;                                        // the chunk in registers is a half chunk and this slot is past K
    auto set_ptrs = [&](int mt_, int nt_) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ra = mt_ * TBM + r0 + 32 * i;
            ra = ra < g.M ? ra : g.M - 1;
            ap[i] = A + ((long long)ra * g.lda + 8 * c8) * ea_b;
            bp[i] = B + ((long long)(nt_ * TBN + r0 + 32 * i) * g.ldb + 8 * c8) * 2;
        }
    };
    int ld_koff = 0;                                        // element offset of this thread's slot in the chunk at the cursor
    auto load_piece = [&](int i) {
        if (a16) {
            ra4[i] = *reinterpret_cast<const f32x4*>(ap[i] + (long long)ld_koff * 2);
        } else {
            ra4[i] = *reinterpret_cast<const f32x4*>(ap[i] + (long long)ld_koff * 4);
            ra4b[i] = *reinterpret_cast<const f32x4*>(ap[i] + (long long)ld_koff * 4 + 16);
        }
        rb4[i] = *reinterpret_cast<const f32x4*>(bp[i] + (long long)ld_koff * 2);
    };
    auto set_koff = [&]() {
        const int k0 = ld_kt * 64;
        rz = k0 + 8 * c8 >= g.K;                            // slots past K re-read the chunk's first half (valid memory), zeroed below
        ld_koff = rz ? k0 - 8 * c8 + 8 * (c8 & 3) : k0;     // = k0 + 8 (c8 - 4) relative to this thread's own slot
    };
    auto advance = [&]() {
        if (ld_j >= nslots) return;
        if (++ld_kt == nk) {
            ld_kt = 0;
            int m2 = 0, n2 = 0;
            ld_j = next_valid(ld_j + gridDim.x, m2, n2);
            if (ld_j < nslots) set_ptrs(m2, n2);
        }
    };
    auto write_piece = [&](int st, int i, bool zero) {
        float* s0 = &smem[st * NT2_STAGE];
        f32x4 va = ra4[i], vb = rb4[i];
        if (!a16) {
            const bf16x4 lo = nu_to_bf16x4(ra4[i]), hi = nu_to_bf16x4(ra4b[i]);
            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
            va = __builtin_bit_cast(f32x4, make_uint4(l2.x, l2.y, h2.x, h2.y));
        }
        if (zero) { va = f32x4{0.f, 0.f, 0.f, 0.f}; vb = va; }
        *reinterpret_cast<f32x4*>(&s0[w_off + 32 * i * NT_LDS]) = va;
        *reinterpret_cast<f32x4*>(&s0[TBM * NT_LDS + w_off + 32 * i * NT_LDS]) = vb;
    };
    struct Frag { f32x4 a0, a1, b0, b1; };
    auto read_frag = [&](Frag& f, int st, int kk) {
        const float* s0 = &smem[st * NT2_STAGE];
        f.a0 = *reinterpret_cast<const f32x4*>(&s0[a_off + kk * 8]);
        f.a1 = *reinterpret_cast<const f32x4*>(&s0[a_off + 32 * NT_LDS + kk * 8]);
        f.b0 = *reinterpret_cast<const f32x4*>(&s0[b_off + kk * 8]);
        f.b1 = *reinterpret_cast<const f32x4*>(&s0[b_off + 32 * NT_LDS + kk * 8]);
    };

    const NtEpiArgs<EPI> ea = nt_epi_args<EPI, true>(g, z);
    constexpr bool kMaskR = NtEpiArgs<EPI>::kMaskR;

    // ---- prologue: chunk 0 -> stage 0, chunk 1 -> registers, first fragments ----
    set_ptrs(mt, nt);
    set_koff();
    bool wz = rz;
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(i);
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) write_piece(0, i, wz);
    set_koff();
    wz = rz;
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(i);
    advance();
    __syncthreads();
    Frag F0, F1;
    read_frag(F0, 0, 0);
    int cur = 0;

    while (true) {
        const int m0 = mt * TBM, n0 = nt * TBN;
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
#define NT16_GROUP(F)                                                                                                           \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.a0), __builtin_bit_cast(bf16x8, F.b0), acc[0][0], 0, 0, 0); \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.a0), __builtin_bit_cast(bf16x8, F.b1), acc[0][1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.a1), __builtin_bit_cast(bf16x8, F.b0), acc[1][0], 0, 0, 0); \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.a1), __builtin_bit_cast(bf16x8, F.b1), acc[1][1], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#define NT16_PIN __builtin_amdgcn_sched_barrier(0);
        // one chunk = 4 k-groups of 4 MFMAs; the next chunk goes registers -> the other stage under k-groups 0-1, the chunk after
        // that global -> registers under k-group 2; one barrier per chunk; the next chunk's first fragments land under k-group 3
        for (int kt = 0; kt < nk; ++kt) {
            NT16_GROUP(F0)
            read_frag(F1, cur, 1); NT16_PIN
            write_piece(cur ^ 1, 0, wz); write_piece(cur ^ 1, 1, wz); NT16_PIN
            NT16_GROUP(F1)
            read_frag(F0, cur, 2); NT16_PIN
            write_piece(cur ^ 1, 2, wz); write_piece(cur ^ 1, 3, wz); NT16_PIN
            NT16_GROUP(F0)
            read_frag(F1, cur, 3); NT16_PIN
            set_koff();
            wz = rz;
            load_piece(0); load_piece(1); load_piece(2); load_piece(3); NT16_PIN
            advance();
            __syncthreads();        // the other stage is complete; every wave holds its last fragments of this one
            read_frag(F0, cur ^ 1, 0);      // (after the very last chunk: stale bytes, never used)
            NT16_PIN
            NT16_GROUP(F1)
            cur ^= 1;
        }
#undef NT16_GROUP
#undef NT16_PIN
        // ---- epilogue: the stage consumed last is free until the next chunk's hand-over ----
        nt_epilogue<EPI, 4, true>(g, ea, acc, &smem[(cur ^ 1) * NT2_STAGE] + wid * (32 * EPI_LDS), m0, n0, mwave, mlo, mhi, lane, wid);
        int mtn = 0, ntnx = 0;
        const int jn = next_valid(j + gridDim.x, mtn, ntnx);
        if (jn >= nslots) break;
        __syncthreads();            // every wave is done with the scratch before the next hand-over writes that stage
        j = jn; mt = mtn; nt = ntnx;
    }
}

// bf16-storage NT kernel, occupancy variant: ONE LDS buffer (36.9 KB), three workgroups per CU, the next chunk prefetched into
// registers under the MFMAs (the first-generation flow).  With 2 048 MFMA cycles per K = 256 tile the kernel lives on memory
// latency and the epilogue, which more resident waves hide better than a deeper per-wave pipeline.
template <int EPI, bool A16>
__global__ __launch_bounds__(256, 3) void gemm_nt16b_kernel(NuGemmNT g) {
    __shared__ __attribute__((aligned(16))) float smem[NT2_STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int z = blockIdx.z;
    const int ntn = (g.N + TBN - 1) / TBN;
    const int mtiles = (g.M + TBM - 1) / TBM;
    const int nslots = ((mtiles + 7) / 8) * 8 * ntn;
    constexpr bool a16 = A16;
    constexpr int ea_b = a16 ? 2 : 4;
    const char* __restrict__ A = reinterpret_cast<const char*>(g.A) + (long long)z * g.sA * ea_b;
    const char* __restrict__ B = reinterpret_cast<const char*>(g.B) + (long long)z * g.sB * 2;
    const int c8 = tid & 7;
    const int r0 = tid >> 3;
    const int nk = (g.K + 63) / 64;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 64 + li) * NT_LDS + 4 * lh;
    const int b_off = TBM * NT_LDS + (wc * 64 + li) * NT_LDS + 4 * lh;
    const int w_off = r0 * NT_LDS + 4 * c8;

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

    // addresses: one wave-uniform base per tile (SGPRs) + a 32-bit byte offset per staged row (the row clamp at the M edge
    // is the only per-tile part), so no 64-bit per-lane pointer lives in VGPRs
    const char* abase = A;
    const char* bbase = B;
    unsigned oa[4], ob[4];
    f32x4 ra4[4], ra4b[A16 ? 1 : 4], rb4[4];
    bool rz = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) ob[i] = ((unsigned)(r0 + 32 * i) * (unsigned)g.ldb + 8u * c8) * 2u;
    auto set_ptrs = [&](int mt_, int nt_) {
        const int mtu = __builtin_amdgcn_readfirstlane(mt_), ntu = __builtin_amdgcn_readfirstlane(nt_);
        abase = A + (long long)mtu * TBM * g.lda * ea_b;
        bbase = B + (long long)ntu * TBN * g.ldb * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int rl = r0 + 32 * i;
            rl = mtu * TBM + rl < g.M ? rl : g.M - 1 - mtu * TBM;
            oa[i] = ((unsigned)rl * (unsigned)g.lda + 8u * c8) * (unsigned)ea_b;
        }
    };
    auto load_regs = [&](int kt) {
        const int k0 = kt * 64;
        rz = k0 + 8 * c8 >= g.K;                            // half chunk: slots past K re-read the first half, zeroed at the hand-over
        const unsigned back = rz ? 32u : 0u;
        const char* ak = abase + (long long)k0 * ea_b;
        const char* bk = bbase + (long long)k0 * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (a16) {
                ra4[i] = *reinterpret_cast<const f32x4*>(ak + (oa[i] - back * 2u));
            } else {
                ra4[i] = *reinterpret_cast<const f32x4*>(ak + (oa[i] - back * 4u));
                ra4b[i] = *reinterpret_cast<const f32x4*>(ak + (oa[i] - back * 4u) + 16);
            }
            rb4[i] = *reinterpret_cast<const f32x4*>(bk + (ob[i] - back * 2u));
        }
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 va = ra4[i], vb = rb4[i];
            if constexpr (!A16) {
                const bf16x4 lo = nu_to_bf16x4(ra4[i]), hi = nu_to_bf16x4(ra4b[i]);
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                va = __builtin_bit_cast(f32x4, make_uint4(l2.x, l2.y, h2.x, h2.y));
            }
            if (rz) { va = f32x4{0.f, 0.f, 0.f, 0.f}; vb = va; }
            *reinterpret_cast<f32x4*>(&smem[w_off + 32 * i * NT_LDS]) = va;
            *reinterpret_cast<f32x4*>(&smem[TBM * NT_LDS + w_off + 32 * i * NT_LDS]) = vb;
        }
    };

    set_ptrs(mt, nt);
    load_regs(0);
    store_regs();
    __syncthreads();

    const NtEpiArgs<EPI> ea = nt_epi_args<EPI, true>(g, z);
    constexpr bool kMaskR = NtEpiArgs<EPI>::kMaskR;

    while (true) {
        int mtn = 0, ntnx = 0;
        const int jn = next_valid(j + gridDim.x, mtn, ntnx);
        const bool has_next = jn < nslots;
        const int m0 = mt * TBM, n0 = nt * TBN;
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
                load_regs(kt + 1);
            } else if (has_next) {
                set_ptrs(mtn, ntnx);
                load_regs(0);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&smem[a_off + kk * 8]));
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&smem[a_off + 32 * NT_LDS + kk * 8]));
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&smem[b_off + kk * 8]));
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&smem[b_off + 32 * NT_LDS + kk * 8]));
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();   // every wave is done reading this chunk (and, after the last one, the scratch is free)
            if (kt + 1 < nk) {
                store_regs();
                __syncthreads();
            }
        }
        nt_epilogue<EPI, 2, true>(g, ea, acc, smem + wid * (32 * EPI_LDS), m0, n0, mwave, mlo, mhi, lane, wid);
        if (!has_next) break;
        __syncthreads();   // every wave is done with the scratch
        store_regs();
        __syncthreads();
        j = jn; mt = mtn; nt = ntnx;
    }
}


// bf16-storage launch (called by nu_gemm_nt_launch after the common argument checks)
int nu_gemm_nt16_launch(const NuGemmNT& g, int groups, long long nslots, hipStream_t stream) {
    if ((g.ldb & 7) || ((g.bf16 & NU_GEMM_A16) && (g.lda & 7))) return NU_ERR_ARG;
    static const int grid_env = getenv("NU_NT_GRID") ? atoi(getenv("NU_NT_GRID")) : 0;
    static const int v16 = getenv("NU_NT16_V") ? atoi(getenv("NU_NT16_V")) : 2;     // development switch: 1 = pipelined, 2 = occupancy
    long long per2 = nu_rup(nu_cdiv(grid_env ? grid_env : (v16 == 2 ? 768 : 512), groups), 8);
    if (per2 > nslots) per2 = nslots;
    dim3 grid2((unsigned)per2, 1, groups), block(256);
    switch (g.epi) {
#define NU_CASE16(E) case E: if (v16 == 2 && (g.bf16 & NU_GEMM_A16)) hipLaunchKernelGGL((gemm_nt16b_kernel<E, true>), grid2, block, 0, stream, g); \
                             else if (v16 == 2) hipLaunchKernelGGL((gemm_nt16b_kernel<E, false>), grid2, block, 0, stream, g); \
                             else hipLaunchKernelGGL((gemm_nt16_kernel<E>), grid2, block, 0, stream, g); break;
        NU_CASE16(NU_EPI_BIAS_NONE)
        NU_CASE16(NU_EPI_BIAS_RELU)
        NU_CASE16(NU_EPI_BIAS_SOFTPLUS)
        NU_CASE16(NU_EPI_MUL_DRELU)
        NU_CASE16(NU_EPI_MUL_DSP)
        NU_CASE16(NU_EPI_Q_SP)
        NU_CASE16(NU_EPI_B_SP)
        NU_CASE16(NU_EPI_PLAIN)
        NU_CASE16(NU_EPI_B_RELU)
#undef NU_CASE16
        default: return NU_ERR_ARG;
    }
    return nu_launch_status();
}
