// gemm.hip -- fp32 MFMA GEMM kernels (gfx950).  See gemm.h for the contract.
//
// Replaces the reference's eager `lin(x)` / autograd matmul chains:
//   network/field.py:133-150 (SDFNetwork.forward), :158-170 (.gradient), :265-289 (NeRFNetwork),
//   :371-408 (make_predictor stacks) and their autograd backward / double backward.
#include "gemm.h"
#include <stdlib.h>

#define TBM 128
#define TBN 128
#define TBK 32
#define NT_LDS 36   // padded row stride (floats): 36*r mod 64 hits every 16-B slot once per 16 rows
#define NT_LDSH 40  // bf16 image: padded row stride in elements (80 B)
#define NT_WPC 3    // workgroups per CU the NT kernel is built for (LDS 36.9 KB, <= 168 VGPRs)

// ------------------------------------------------------------------------------------------------
// NT kernel: persistent workgroups walk the output tiles; the first k-tile of the NEXT output tile is
// prefetched into registers under the last MFMAs of the current one, so only the very first tile of a
// workgroup exposes global-load latency.  Epilogue goes through a wave-private LDS transpose so that
// global stores (and the aux loads of the derivative epilogues) are 16 B per lane, 256 B per row.
// ------------------------------------------------------------------------------------------------
#define EPI_LDS 68  // row stride (floats) of the per-wave 32x64 epilogue scratch

template <int EPI>
static __device__ inline float nu_epi_apply(float v, float bv, float h, float d, float ca, float& out2) {
    out2 = 0.f;
    if (EPI == NU_EPI_BIAS_NONE) return v + bv;
    if (EPI == NU_EPI_BIAS_RELU) return fmaxf(v + bv, 0.0f);
    if (EPI == NU_EPI_BIAS_SOFTPLUS) return nu_softplus100_fast(v + bv);
    if (EPI == NU_EPI_MUL_DRELU) return h > 0.0f ? v : 0.0f;
    if (EPI == NU_EPI_MUL_DSP) return v * (1.0f - nu_exp_m100(h));
    if (EPI == NU_EPI_Q_SP) {
        const float e = nu_exp_m100(h);   // sp' = 1 - e ; sp''/sp' = 100 e
        out2 = v * d * 100.0f * e;
        return v * (1.0f - e);
    }
    if (EPI == NU_EPI_B_SP) return v * (1.0f - nu_exp_m100(h)) + ca;
    if (EPI == NU_EPI_B_RELU) return (h > 0.0f ? v : 0.0f) + ca;
    return v;
}

// development aid: {shader cycles, 100 MHz wall ticks} of block 0 of the last mfma_peak launch
__device__ unsigned long long nu_dbg_clk[2];

#ifdef NU_LAB   // scripts/gemm_lab.hip only (never defined for libnunerf.so): per-workgroup phase stamps and ablation switches
#define NU_LAB_TILES 6
__device__ unsigned long long nu_lab_trace[1024][NU_LAB_TILES][3];   // [wg][tile]{main-loop start, main-loop end, epilogue end} 100 MHz
__device__ unsigned nu_lab_hwid[1024][2];
__device__ int nu_lab_stagger_ticks; // > 0: the second half of the grid starts this many 10 ns ticks late (gen 2)
__device__ int nu_lab_skip_epi;      // epilogue ablation: 1 scratch writes only, 2 no global stores, 3 nothing, 4 stores to an L2-resident region
static int nu_lab_grid = 0;          // persistent grid size override (0: default)
static int nu_lab_v1 = 0;            // 1: first-generation fp32 NT kernel
static int nu_lab_small = -1;        // 64-row tiles: -1 library rule, 0 never, 1 always (gen 2)
__device__ int nu_lab_stamps;        // 1: wave 0 of block 0 stamps the stages of its third tile (perturbs that tile)
__device__ long long nu_lab_stage[8][8];
__device__ int nu_lab_small_a;       // 1: every tile reads A from an 8 MB window (memory-latency ablation, gen 2)
#endif

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the NT kernels: accumulators -> wave-private LDS scratch (32 rows at a time) -> row-contiguous float4
// rows, fused bias / activation / derivative epilogues, ReLU sign-bit words.
// ------------------------------------------------------------------------------------------------
template <int EPI>
struct NtEpiArgs {       // per launch (and group)
    float* C; float* C2; const float* bias; const float* H; const float* D; const float* Cadd;
    unsigned long long* mask;
    int zero_to, act_cols;
    bool vec_ok;
    bool c16, x16;           // bf16-storage kernel only: C / C2 are bf16; H / D / Cadd are bf16
    static constexpr bool kMaskW = (EPI == NU_EPI_BIAS_RELU);                                  // writes ReLU sign bits
    static constexpr bool kMaskR = (EPI == NU_EPI_MUL_DRELU || EPI == NU_EPI_B_RELU);          // reads them instead of H
    static constexpr bool kNeedH = (EPI == NU_EPI_MUL_DRELU || EPI == NU_EPI_MUL_DSP || EPI == NU_EPI_Q_SP ||
                                    EPI == NU_EPI_B_SP || EPI == NU_EPI_B_RELU);
    static constexpr bool kNeedD = (EPI == NU_EPI_Q_SP);
    static constexpr bool kNeedAdd = (EPI == NU_EPI_B_SP || EPI == NU_EPI_B_RELU);
    static constexpr bool kBias = (EPI <= NU_EPI_BIAS_SOFTPLUS);
};

// H16: the bf16-storage kernel (matrices flagged 16-bit are __bf16 behind their float* fields; strides stay in elements)
template <int EPI, bool H16 = false>
static __device__ inline NtEpiArgs<EPI> nt_epi_args(const NuGemmNT& g, int z) {
    typedef NtEpiArgs<EPI> E;
    E a;
    a.c16 = H16 && (g.bf16 & NU_GEMM_C16) != 0;
    a.x16 = H16 && (g.bf16 & NU_GEMM_X16) != 0;
    const int ec = a.c16 ? 2 : 4, ex = a.x16 ? 2 : 4;      // element bytes
    auto off = [](const float* p, long long elems, int eb) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + elems * eb); };
    a.C = const_cast<float*>(off(g.C, (long long)z * g.sC, ec));
    a.C2 = g.C2 ? const_cast<float*>(off(g.C2, (long long)z * g.sC2, ec)) : nullptr;
    a.bias = g.bias ? g.bias + (long long)z * g.sBias : nullptr;
    a.H = g.H ? off(g.H, (long long)z * g.sH, ex) : nullptr;
    a.D = g.D ? off(g.D, (long long)z * g.sD, ex) : nullptr;
    a.Cadd = g.Cadd ? off(g.Cadd, (long long)z * g.sCadd, ex) : nullptr;
    a.zero_to = g.zero_to > g.N ? g.zero_to : g.N;
    a.act_cols = g.act_cols > 0 ? g.act_cols : 0x7fffffff;
    a.mask = (E::kMaskW || E::kMaskR) ? g.mask : nullptr;
    // vector path: 4 elements per lane (16 B fp32 / 8 B bf16): every touched matrix aligned to that, ld % 4 == 0 (wave-uniform test)
    const uintptr_t mc = a.c16 ? 7 : 15, mx = a.x16 ? 7 : 15;
    bool v = (((uintptr_t)a.C & mc) == 0) && ((g.ldc & 3) == 0);
    if (E::kNeedH) v = v && (((uintptr_t)a.H & mx) == 0) && ((g.ldh & 3) == 0);
    if (E::kNeedD) v = v && (((uintptr_t)a.D & mx) == 0) && ((g.ldd & 3) == 0) && (((uintptr_t)a.C2 & mc) == 0) && ((g.ldc2 & 3) == 0);
    if (E::kNeedAdd) v = v && (((uintptr_t)a.Cadd & mx) == 0) && ((g.ldadd & 3) == 0);
    a.vec_ok = v;
    return a;
}

// element accessors of the epilogue: `is16` selects __bf16 storage (compiled out of the fp32 kernels)
static __device__ __forceinline__ f32x4 nu_bf16x4_to_f32(uint2 r) {
    f32x4 o;
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
    return o;
}
template <bool H16> static __device__ __forceinline__ f32x4 nt_ld4(const char* p, bool is16) {
    if (H16 && is16) return nu_bf16x4_to_f32(*reinterpret_cast<const uint2*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}
template <bool H16> static __device__ __forceinline__ void nt_st4(char* p, f32x4 v, bool is16) {
    if (H16 && is16) *reinterpret_cast<bf16x4*>(p) = nu_to_bf16x4(v);
    else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool H16> static __device__ __forceinline__ float nt_ld1(const float* base, long long idx, bool is16) {
    if (H16 && is16) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[idx] << 16);
    return base[idx];
}
template <bool H16> static __device__ __forceinline__ void nt_st1(float* base, long long idx, float v, bool is16) {
    if (H16 && is16) reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
    else base[idx] = v;
}

// sign-bit words of the wave's 64 x 64 slab of tile (mt, nt) of group z (nullptr: no mask for this tile)
template <int EPI>
static __device__ inline unsigned long long* nt_mask_words(const NtEpiArgs<EPI>& ea, const NuGemmNT& g, int mt, int nt, int z, int ntn, int wid) {
    const int ct = z * ntn + nt;
    return (ea.mask && ct < g.mask_nct) ? ea.mask + ((((long long)mt * g.mask_nct + ct) * 4 + wid) * 64) : nullptr;
}

// NBH: row groups whose auxiliary loads are in flight together in the fast path (register budget of the caller)
// TMN / tm0: the wave holds TMN 32-row blocks of the 64 x 64 slab `wid` of the 128 x 128 tile at (m0, n0), starting at block tm0
// (2 / 0: the whole slab -- the 128-row-tile kernels; 1 / 0 or 1: one block -- the 64-row-tile kernel, where the two waves that
// share a slab's sign-bit words belong to different workgroups).
template <int EPI, int NBH, bool H16 = false, int TMN = 2>
static __device__ __forceinline__ void nt_epilogue(const NuGemmNT& g, const NtEpiArgs<EPI>& ea, f32x16 (&acc)[TMN][2], float* scr,
                                                   int m0, int n0, unsigned long long* mwave, unsigned mlo, unsigned mhi,
                                                   int lane, int wid, int lab_skip, int tm0 = 0) {
    typedef NtEpiArgs<EPI> E;
    constexpr bool kMaskW = E::kMaskW, kMaskR = E::kMaskR, kNeedH = E::kNeedH, kNeedD = E::kNeedD, kNeedAdd = E::kNeedAdd, kBias = E::kBias;
    float* const C = ea.C; float* const C2 = ea.C2;
    const float* const bias = ea.bias; const float* const H = ea.H; const float* const D = ea.D; const float* const Cadd = ea.Cadd;
    const int zero_to = ea.zero_to, act_cols = ea.act_cols;
    const bool vec_ok = ea.vec_ok;
    const bool c16 = H16 && ea.c16, x16 = H16 && ea.x16;
    const int ec = c16 ? 2 : 4, ex = x16 ? 2 : 4;      // element bytes (4 and 4 in the fp32 kernels: folded)
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wid >> 1, wc = wid & 1;
    unsigned wlo = 0, whi = 0;      // writer: lane l accumulates word l
    // ---- epilogue: accumulators -> wave-private LDS scratch (32 rows at a time) -> row-contiguous float4 ----
    const int colq = (lane & 15) * 4;
    const int gcol = n0 + wc * 64 + colq;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (kBias && bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (gcol + e < g.N) ? bias[gcol + e] : 0.f;
    }
    const bool full = vec_ok && (gcol + 3 < g.N) && (gcol + 3 < act_cols || gcol >= act_cols);
    // the wave's whole 64 x 64 slab is interior and on one side of act_cols (wave-uniform): the fast path below
    const int wcol0 = n0 + wc * 64, wrow0 = m0 + wr * 64;
    const bool slab_full = vec_ok && (wrow0 + 64 <= g.M) && (wcol0 + 64 <= g.N) && (wcol0 + 64 <= act_cols || wcol0 >= act_cols);
    const bool slab_plain = kNeedH && wcol0 >= act_cols;
    // fast-path addressing: one wave-uniform base per matrix (SGPRs; the wave id is made provably uniform) + a 32-bit
    // per-lane byte offset, so no 64-bit per-lane pointer lives in VGPRs
    const int uwid = __builtin_amdgcn_readfirstlane(wid);
    const long long urow0 = m0 + (uwid >> 1) * 64, ucol0 = n0 + (uwid & 1) * 64;
    const unsigned lrow = lane >> 4;
    char* const Cu = reinterpret_cast<char*>(C) + (urow0 * g.ldc + ucol0) * ec;
    char* const C2u = kNeedD ? reinterpret_cast<char*>(C2) + (urow0 * g.ldc2 + ucol0) * ec : nullptr;
    const char* const Hu = kNeedH ? reinterpret_cast<const char*>(H) + (urow0 * g.ldh + ucol0) * ex : nullptr;
    const char* const Du = kNeedD ? reinterpret_cast<const char*>(D) + (urow0 * g.ldd + ucol0) * ex : nullptr;
    const char* const Au = kNeedAdd ? reinterpret_cast<const char*>(Cadd) + (urow0 * g.ldadd + ucol0) * ex : nullptr;
    const unsigned oC = (lrow * (unsigned)g.ldc + (unsigned)colq) * (unsigned)ec, oC2 = (lrow * (unsigned)g.ldc2 + (unsigned)colq) * (unsigned)ec;
    const unsigned oH = (lrow * (unsigned)g.ldh + (unsigned)colq) * (unsigned)ex, oD = (lrow * (unsigned)g.ldd + (unsigned)colq) * (unsigned)ex;
    const unsigned oA = (lrow * (unsigned)g.ldadd + (unsigned)colq) * (unsigned)ex;
#pragma unroll
    for (int tt = 0; tt < TMN; ++tt) {
        const int tm = tm0 + tt;
        if (lab_skip == 3) {                                           // lab: not even the scratch writes
            if (acc[tt][0][0] == 123.456f) C[0] = acc[tt][1][3];
            continue;
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_LDS + tn * 32 + li] = acc[tt][tn][r];
        if (lab_skip == 1) {
            if (acc[tt][0][0] == 123.456f) C[0] = acc[tt][1][3];      // keeps the accumulators live
        } else if (slab_full) {
            // wave-uniform fast path (every interior tile): straight-line code, no per-lane guards; the auxiliary loads of
            // four row groups are in flight together; the activation math is branch-free (nu_common.h)
            constexpr int NB = kNeedH ? NBH : 4;         // row groups in flight
#pragma unroll
            for (int hb = 0; hb < 8 / NB; ++hb) {
                f32x4 v4[NB], h4[NB], d4[NB], c4v[NB];
#pragma unroll
                for (int ii = 0; ii < NB; ++ii) {
                    const int i = hb * NB + ii;
                    const long long roff = (long long)(tm * 32 + i * 4);
                    v4[ii] = *reinterpret_cast<const f32x4*>(&scr[(i * 4 + (lane >> 4)) * EPI_LDS + colq]);
                    if (kNeedH && !slab_plain && !(kMaskR && mwave)) h4[ii] = nt_ld4<H16>(Hu + roff * g.ldh * ex + oH, x16);
                    if (kNeedD && !slab_plain) d4[ii] = nt_ld4<H16>(Du + roff * g.ldd * ex + oD, x16);
                    if (kNeedAdd && !slab_plain) c4v[ii] = nt_ld4<H16>(Au + roff * g.ldadd * ex + oA, x16);
                }
#pragma unroll
                for (int ii = 0; ii < NB; ++ii) {
                    const int i = hb * NB + ii;
                    const long long roff = (long long)((lab_skip == 4 ? 0 : tm * 32) + i * 4);    // lab 4: a small, cache-resident target
                    f32x4 o4, o24;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float h = 0.f, o2 = 0.f;
                        if (kMaskR && mwave) {
                            const int src = (tm * 8 + i) * 4 + e;           // wave-uniform
                            const unsigned lo = __builtin_amdgcn_readlane(mlo, src), hi = __builtin_amdgcn_readlane(mhi, src);
                            h = (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) ? 1.0f : 0.0f;
                        } else if (kNeedH && !slab_plain) h = h4[ii][e];
                        const float v = g.alpha * v4[ii][e];
                        o4[e] = slab_plain ? v : nu_epi_apply<EPI>(v, bv[e], h, (kNeedD && !slab_plain) ? d4[ii][e] : 0.f,
                                                                   (kNeedAdd && !slab_plain) ? c4v[ii][e] : 0.f, o2);
                        o24[e] = o2;
                    }
                    if (lab_skip == 2) {                               // lab: scratch round trip and math, no global store
                        if (o4[0] == 123.456f && o24[1] == 3.f) C[0] = o4[1];
                    } else {
                        nt_st4<H16>(Cu + roff * g.ldc * ec + oC, o4, c16);
                        if (kNeedD) nt_st4<H16>(C2u + roff * g.ldc2 * ec + oC2, o24, c16);
                    }
                    if (kMaskW && mwave) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned long long bits = __ballot(o4[e] > 0.f);
                            const bool mine = lane == (tm * 8 + i) * 4 + e;
                            wlo = mine ? (unsigned)bits : wlo;
                            whi = mine ? (unsigned)(bits >> 32) : whi;
                        }
                    }
                }
            }
        } else if (gcol < zero_to) {
#pragma unroll 4
            for (int i = 0; i < 8; ++i) {
                const int rl = i * 4 + (lane >> 4);
                const int row = m0 + wr * 64 + tm * 32 + rl;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(&scr[rl * EPI_LDS + colq]);
                bool pos[4] = {false, false, false, false};     // ReLU output > 0 (rows past M and non-vector lanes: false)
                if (row < g.M) {
                if (full) {
                    f32x4 h4 = {0.f, 0.f, 0.f, 0.f}, d4 = h4, c4v = h4, o4, o24;
                    const bool plain = gcol >= act_cols;
                    if (kMaskR && mwave) {
                        if (!plain) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int src = (tm * 8 + i) * 4 + e;           // wave-uniform
                                const unsigned lo = __builtin_amdgcn_readlane(mlo, src), hi = __builtin_amdgcn_readlane(mhi, src);
                                h4[e] = (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) ? 1.0f : 0.0f;
                            }
                        }
                    } else if (kNeedH && !plain) h4 = nt_ld4<H16>(reinterpret_cast<const char*>(H) + ((long long)row * g.ldh + gcol) * ex, x16);
                    if (kNeedD && !plain) d4 = nt_ld4<H16>(reinterpret_cast<const char*>(D) + ((long long)row * g.ldd + gcol) * ex, x16);
                    if (kNeedAdd && !plain) c4v = nt_ld4<H16>(reinterpret_cast<const char*>(Cadd) + ((long long)row * g.ldadd + gcol) * ex, x16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float o2;
                        const float v = g.alpha * v4[e];
                        o4[e] = (kNeedH && plain) ? v : nu_epi_apply<EPI>(v, bv[e], h4[e], d4[e], c4v[e], o2);
                        o24[e] = (kNeedH && plain) ? 0.f : o2;
                    }
                    nt_st4<H16>(reinterpret_cast<char*>(C) + ((long long)row * g.ldc + gcol) * ec, o4, c16);
                    if (kNeedD) nt_st4<H16>(reinterpret_cast<char*>(C2) + ((long long)row * g.ldc2 + gcol) * ec, o24, c16);
                    if (kMaskW) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pos[e] = o4[e] > 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = gcol + e;
                        if (col >= zero_to) continue;
                        float out = 0.f, out2 = 0.f;
                        if (col < g.N) {
                            const float v = g.alpha * v4[e];
                            if (kNeedH && col >= act_cols) {
                                out = v;
                            } else {
                                const float h = kNeedH ? nt_ld1<H16>(H, (long long)row * g.ldh + col, x16) : 0.f;
                                const float d = kNeedD ? nt_ld1<H16>(D, (long long)row * g.ldd + col, x16) : 0.f;
                                const float ca = kNeedAdd ? nt_ld1<H16>(Cadd, (long long)row * g.ldadd + col, x16) : 0.f;
                                out = nu_epi_apply<EPI>(v, bv[e], h, d, ca, out2);
                            }
                        }
                        nt_st1<H16>(C, (long long)row * g.ldc + col, out, c16);
                        if (kNeedD) nt_st1<H16>(C2, (long long)row * g.ldc2 + col, out2, c16);
                    }
                }
                }
                if (kMaskW && mwave) {      // every lane of the slab is here (N % 64 == 0 for a writer): ballots are complete
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned long long bits = __ballot(pos[e]);
                        const bool mine = lane == (tm * 8 + i) * 4 + e;
                        wlo = mine ? (unsigned)bits : wlo;
                        whi = mine ? (unsigned)(bits >> 32) : whi;
                    }
                }
            }
        }
    }
    // (a wave that holds one block of a slab owns the words of that block only: lanes 32 tm0 .. 32 tm0 + 31)
    if (kMaskW && mwave && (TMN == 2 || (lane >> 5) == tm0)) mwave[lane] = ((unsigned long long)whi << 32) | wlo;
}

// One LDS buffer (36.9 KB per workgroup) -> 3 workgroups per CU.  The next k-chunk travels global -> registers
// under the MFMAs; only the register -> LDS hand-over sits between two barriers, and the other resident workgroups
// keep the matrix pipe busy meanwhile (measured: +5..12 % over a double-buffered 2-workgroup build on the K = 256
// layers, where the per-tile epilogue is 10-15 % of a tile).  Epilogue in two 32-row halves per wave.
//
// BF16 = true (cfg mlp_dtype 'bf16', BASELINE config 4): same tiles and epilogues, operands rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// Activations and weights stay fp32 in HBM, so this build is bound by streaming them (HBM / L1), not by the
// matrix pipe: 16x the MFMA rate buys ~3x on the K = 256 layers.
//
// PREC = 2 (mlp_dtype 'bf16x6'): fp32-equivalent products on the bf16 pipe.  Each fp32 operand is split EXACTLY into three
// bf16 pieces (x = x1 + x2 + x3, 8 significant bits each); the six partial products of order >= 2^-16 (11, 12, 21, 13, 31,
// 22) are exact in fp32 and are summed smallest first into the fp32 accumulator; the three dropped ones are below 2^-23
// of |x||y|, i.e. at the level of ONE fp32 rounding of the product.  6 x 32 cycles replace 8 x 64 cycles of
// v_mfma_f32_32x32x2_f32 per 16-deep k-step.
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

#ifdef NU_LAB
    int lab_tile = 0;
    const unsigned long long lab_c0 = clock64(), lab_w0 = wall_clock64();
    if (tid == 0 && blockIdx.x < 1024) {
        nu_lab_hwid[blockIdx.x][0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID
        nu_lab_hwid[blockIdx.x][1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // XCC_ID
    }
#endif
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

#ifdef NU_LAB
        if (tid == 0 && blockIdx.x < 1024 && lab_tile < NU_LAB_TILES) nu_lab_trace[blockIdx.x][lab_tile][0] = wall_clock64();
#endif
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

#ifdef NU_LAB
        if (tid == 0 && blockIdx.x < 1024 && lab_tile < NU_LAB_TILES) nu_lab_trace[blockIdx.x][lab_tile][1] = wall_clock64();
        const int lab_skip = nu_lab_skip_epi;
#else
        constexpr int lab_skip = 0;
#endif
        // ---- epilogue ----
        nt_epilogue<EPI, 2>(g, ea, acc, &smem[0][0] + wid * (32 * EPI_LDS), m0, n0, mwave, mlo, mhi, lane, wid, lab_skip);
#ifdef NU_LAB
        if (tid == 0 && blockIdx.x < 1024 && lab_tile < NU_LAB_TILES) nu_lab_trace[blockIdx.x][lab_tile][2] = wall_clock64();
        ++lab_tile;
        if (!has_next && tid == 0 && blockIdx.x == 0) { nu_dbg_clk[0] = clock64() - lab_c0; nu_dbg_clk[1] = wall_clock64() - lab_w0; }
#endif
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
template <int EPI, int TMN>
__global__ __launch_bounds__(256, 2) void gemm_nt2_kernel(NuGemmNT g) {
    constexpr int BM = 64 * TMN;                            // tile rows
    __shared__ __attribute__((aligned(16))) float smem[2 * NT2_STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int z = blockIdx.z;
    const int ntn = (g.N + TBN - 1) / TBN;
    const int mtiles = (g.M + BM - 1) / BM;
    const int nslots = ((mtiles + 7) / 8) * 8 * ntn;       // slot order: see gemm_nt_kernel
    const float* __restrict__ A = g.A + (long long)z * g.sA;
    const float* __restrict__ B = g.B + (long long)z * g.sB;
    const int c4 = tid & 7;
    const int r0 = tid >> 3;
    const int nk = g.K / TBK;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 32 * TMN + li) * NT_LDS + 4 * lh;
    const int b_off = TBM * NT_LDS + (wc * 64 + li) * NT_LDS + 4 * lh;
    const int a0_off = a_off, a1_off = a_off + 32 * NT_LDS, b0_off = b_off, b1_off = b_off + 32 * NT_LDS;   // the four fragments
    const int w_off = r0 * NT_LDS + 4 * c4;                 // this thread's slot of a staged operand row group

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

    // ---- loader: a cursor over the chunks of this workgroup's tiles, in order ----
    int ld_j = j, ld_kt = 0;                                // next chunk to fetch
    const float* ap[4];
    const float* bp[4];
    f32x4 ra4[4], rb4[4];
    auto set_ptrs = [&](int mt_, int nt_) {
#ifdef NU_LAB
        if (nu_lab_small_a) mt_ &= 63;                      // ablation: A stays cache-resident (8 MB window)
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ra = mt_ * BM + r0 + 32 * (i < 2 * TMN ? i : 0);
            ra = ra < g.M ? ra : g.M - 1;
            ap[i] = A + (long long)ra * g.lda + 4 * c4;
            bp[i] = B + (long long)(nt_ * TBN + r0 + 32 * i) * g.ldb + 4 * c4;
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
        if (++ld_kt == nk) {
            ld_kt = 0;
            int m2 = 0, n2 = 0;
            ld_j = next_valid(ld_j + gridDim.x, m2, n2);
            if (ld_j < nslots) set_ptrs(m2, n2);
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

    const NtEpiArgs<EPI> ea = nt_epi_args<EPI>(g, z);
    constexpr bool kMaskR = NtEpiArgs<EPI>::kMaskR;

#ifdef NU_LAB
    int lab_tile = 0;
    const unsigned long long lab_c0 = clock64(), lab_w0 = wall_clock64();
    if (tid == 0 && blockIdx.x < 1024) {
        nu_lab_hwid[blockIdx.x][0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID
        nu_lab_hwid[blockIdx.x][1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // XCC_ID
    }
#endif
    // ---- prologue: chunk 0 -> stage 0, chunk 1 -> registers, first fragments ----
    set_ptrs(mt, nt);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(i);
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) write_piece(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(i);
    advance();
#ifdef NU_LAB
    if (nu_lab_stagger_ticks > 0 && blockIdx.x >= (gridDim.x >> 1)) {      // lab only: phase-shift the second half of the grid
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)nu_lab_stagger_ticks) __builtin_amdgcn_s_sleep(64);
    }
    if (nu_lab_stagger_ticks < 0) {      // lab only: 16 start-up phases across the CUs of an XCD (workgroups b and b + 8 land on one XCD)
        const unsigned long long t0 = wall_clock64(), d = (unsigned long long)(-nu_lab_stagger_ticks) * ((blockIdx.x >> 3) & 15);
        while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(16);
    }
#endif
    __syncthreads();
    Frag F0, F1;
    read_frag(F0, 0, 0);
    int cur = 0;

    while (true) {
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
#ifdef NU_LAB
        if (tid == 0 && blockIdx.x < 1024 && lab_tile < NU_LAB_TILES) nu_lab_trace[blockIdx.x][lab_tile][0] = wall_clock64();
#endif
#define NT2_PIN __builtin_amdgcn_sched_barrier(0);
#define NT2_M(F, e, i, j) if constexpr (i < TMN) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a##i[e], F.b##j[e], acc[i][j], 0, 0, 0); NT2_PIN }
#ifdef NU_LAB
#define NT2_STAMP(I) if (lab_stamp && kt < 8) { nu_lab_stage[kt][I] = clock64(); __builtin_amdgcn_sched_barrier(0); }
        const bool lab_stamp = nu_lab_stamps != 0 && tid == 0 && blockIdx.x == 0 && lab_tile == 2;
#else
#define NT2_STAMP(I)
#endif
        // One chunk = 4 k-groups of 16 MFMAs.  An in-order wave issues nothing while one of its own instructions issues, and the
        // matrix pipe runs dry when that takes longer than the MFMA in front of it executes (64 cycles): a ds_write_b128 costs
        // ~43 issue cycles, a global_load_dwordx4 ~34, a ds_read_b128 ~20 -- ONE of them hides behind an MFMA, two in a row do not
        // (measured with the phase stamps: 4 reads + 8 writes in two gaps cost a lone workgroup 200 of a k-group's 1024 cycles).
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
            NT2_STAMP(0)
            NT2_M(F0, 0, 0, 0) NT2_M(F0, 0, 0, 1) NT2_RD(F1, sc, 1, a0) NT2_M(F0, 0, 1, 0) NT2_M(F0, 0, 1, 1)
            NT2_M(F0, 1, 0, 0) NT2_M(F0, 1, 0, 1) NT2_RD1(F1, sc, 1) NT2_M(F0, 1, 1, 0) NT2_M(F0, 1, 1, 1)
            NT2_M(F0, 2, 0, 0) NT2_M(F0, 2, 0, 1) NT2_RD(F1, sc, 1, b0) NT2_M(F0, 2, 1, 0) NT2_M(F0, 2, 1, 1)
            NT2_M(F0, 3, 0, 0) NT2_M(F0, 3, 0, 1) NT2_RD(F1, sc, 1, b1) NT2_M(F0, 3, 1, 0) NT2_M(F0, 3, 1, 1)
            NT2_STAMP(1)
            NT2_M(F1, 0, 0, 0) NT2_WA(0) NT2_M(F1, 0, 0, 1) NT2_RD(F0, sc, 2, a0) NT2_M(F1, 0, 1, 0) NT2_WB(0) NT2_M(F1, 0, 1, 1)
            NT2_M(F1, 1, 0, 0) NT2_WA(1) NT2_M(F1, 1, 0, 1) NT2_RD1(F0, sc, 2) NT2_M(F1, 1, 1, 0) NT2_WB(1) NT2_M(F1, 1, 1, 1)
            NT2_M(F1, 2, 0, 0) NT2_WA(2) NT2_M(F1, 2, 0, 1) NT2_RD(F0, sc, 2, b0) NT2_M(F1, 2, 1, 0) NT2_WB(2) NT2_M(F1, 2, 1, 1)
            NT2_M(F1, 3, 0, 0) NT2_WA(3) NT2_M(F1, 3, 0, 1) NT2_RD(F0, sc, 2, b1) NT2_M(F1, 3, 1, 0) NT2_WB(3) NT2_M(F1, 3, 1, 1)
            NT2_STAMP(2)
            NT2_M(F0, 0, 0, 0) NT2_LA(0) NT2_M(F0, 0, 0, 1) NT2_RD(F1, sc, 3, a0) NT2_M(F0, 0, 1, 0) NT2_LB(0) NT2_M(F0, 0, 1, 1)
            NT2_M(F0, 1, 0, 0) NT2_LA(1) NT2_M(F0, 1, 0, 1) NT2_RD1(F1, sc, 3) NT2_M(F0, 1, 1, 0) NT2_LB(1) NT2_M(F0, 1, 1, 1)
            NT2_M(F0, 2, 0, 0) NT2_LA(2) NT2_M(F0, 2, 0, 1) NT2_RD(F1, sc, 3, b0) NT2_M(F0, 2, 1, 0) NT2_LB(2) NT2_M(F0, 2, 1, 1)
            NT2_M(F0, 3, 0, 0) NT2_LA(3) NT2_M(F0, 3, 0, 1) NT2_RD(F1, sc, 3, b1) NT2_M(F0, 3, 1, 0) NT2_LB(3) NT2_M(F0, 3, 1, 1)
            NT2_STAMP(4)
            advance();
            __syncthreads();        // the other stage is complete; every wave holds its last fragments of this one
            NT2_STAMP(5)
            // (after the very last chunk the reads below return stale bytes that are never used)
            NT2_M(F1, 0, 0, 0) NT2_M(F1, 0, 0, 1) NT2_RD(F0, so, 0, a0) NT2_M(F1, 0, 1, 0) NT2_M(F1, 0, 1, 1)
            NT2_M(F1, 1, 0, 0) NT2_M(F1, 1, 0, 1) NT2_RD1(F0, so, 0) NT2_M(F1, 1, 1, 0) NT2_M(F1, 1, 1, 1)
            NT2_M(F1, 2, 0, 0) NT2_M(F1, 2, 0, 1) NT2_RD(F0, so, 0, b0) NT2_M(F1, 2, 1, 0) NT2_M(F1, 2, 1, 1)
            NT2_M(F1, 3, 0, 0) NT2_M(F1, 3, 0, 1) NT2_RD(F0, so, 0, b1) NT2_M(F1, 3, 1, 0) NT2_M(F1, 3, 1, 1)
            NT2_STAMP(6)
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
#undef NT2_QUAD
#undef NT2_PIN
#undef NT2_STAMP
#undef NT2_GROUP
#undef NT2_MFMA8
#ifdef NU_LAB
        if (tid == 0 && blockIdx.x < 1024 && lab_tile < NU_LAB_TILES) nu_lab_trace[blockIdx.x][lab_tile][1] = wall_clock64();
        const int lab_skip = nu_lab_skip_epi;
#else
        constexpr int lab_skip = 0;
#endif
        // ---- epilogue: the stage consumed last (cur ^ 1 after the flip) is free until the next chunk's hand-over ----
        nt_epilogue<EPI, 4, false, TMN>(g, ea, acc, &smem[(cur ^ 1) * NT2_STAGE] + wid * (32 * EPI_LDS), lab_skip == 4 ? (int)(blockIdx.x >> 1) * TBM : m0, n0, mwave, mlo, mhi, lane, slab, lab_skip, tm0);
#ifdef NU_LAB
        if (tid == 0 && blockIdx.x < 1024 && lab_tile < NU_LAB_TILES) nu_lab_trace[blockIdx.x][lab_tile][2] = wall_clock64();
        ++lab_tile;
#endif
        int mtn = 0, ntnx = 0;
        const int jn = next_valid(j + gridDim.x, mtn, ntnx);
        if (jn >= nslots) {
#ifdef NU_LAB
            if (tid == 0 && blockIdx.x == 0) { nu_dbg_clk[0] = clock64() - lab_c0; nu_dbg_clk[1] = wall_clock64() - lab_w0; }
#endif
            break;
        }
        __syncthreads();            // every wave is done with the scratch before the next hand-over writes that stage
        j = jn; mt = mtn; nt = ntnx;
    }
}

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
        nt_epilogue<EPI, 4, true>(g, ea, acc, &smem[(cur ^ 1) * NT2_STAGE] + wid * (32 * EPI_LDS), m0, n0, mwave, mlo, mhi, lane, wid, false);
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
        nt_epilogue<EPI, 2, true>(g, ea, acc, smem + wid * (32 * EPI_LDS), m0, n0, mwave, mlo, mhi, lane, wid, false);
        if (!has_next) break;
        __syncthreads();   // every wave is done with the scratch
        store_regs();
        __syncthreads();
        j = jn; mt = mtn; nt = ntnx;
    }
}

int nu_gemm_nt_launch(const NuGemmNT& g, hipStream_t stream) {
    if (g.M <= 0) return NU_OK;
    if (g.N <= 0 || g.K <= 0 || (g.K % TBK) != 0 || g.lda < g.K || g.ldb < g.K) return NU_ERR_ARG;
    if ((g.lda & 3) || (g.ldb & 3)) return NU_ERR_ARG;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15)) return NU_ERR_ARG;
    const int ntn = nu_cdiv(g.N, TBN);
    const long long nslots = (long long)nu_rup(nu_cdiv(g.M, TBM), 8) * ntn;
    if (nslots > 0x7fffffffLL) return NU_ERR_ARG;
    const int groups = g.groups > 0 ? g.groups : 1;
    if (g.mask && (g.epi == NU_EPI_BIAS_RELU || g.epi == NU_EPI_MUL_DRELU || g.epi == NU_EPI_B_RELU)) {
        // the sign-bit path lives in the 16-byte epilogue only: every lane's 4 columns inside N, matrices aligned
        if ((g.N & 3) || (g.ldc & 3) || ((uintptr_t)g.C & 15) || g.mask_nct <= 0) return NU_ERR_ARG;
        if (g.epi == NU_EPI_B_RELU && ((g.ldadd & 3) || ((uintptr_t)g.Cadd & 15))) return NU_ERR_ARG;
        if (g.epi == NU_EPI_BIAS_RELU && ((long long)groups * ntn > g.mask_nct || (g.N & 63))) return NU_ERR_ARG;
        if (g.act_cols > 0 && (g.act_cols & 63)) return NU_ERR_ARG;
    }
    // persistent: NT_WPC workgroups per CU (256 CUs) shared over the groups, a multiple of 8 so the XCD grouping holds
#ifdef NU_LAB
    const int grid_env = nu_lab_grid;
#else
    static const int grid_env = getenv("NU_NT_GRID") ? atoi(getenv("NU_NT_GRID")) : 0;
#endif
    const int prec = g.bf16 & 3;
    if (prec == 3 || ((g.bf16 & ~3) && !(prec == 1 && (g.bf16 & NU_GEMM_B16)))) return NU_ERR_ARG;   // storage flags need the bf16-storage kernel
    const int grid_target = grid_env ? grid_env : 256 * (prec == 2 ? 2 : NT_WPC);   // workgroups the build keeps resident
    long long per = nu_rup(nu_cdiv(grid_target, groups), 8);
    if (per > nslots) per = nslots;
    dim3 grid((unsigned)per, 1, groups), block(256);
    static const bool v1_env = getenv("NU_NT_V1") && atoi(getenv("NU_NT_V1")) != 0;   // development switch: first-generation fp32 kernel
    bool v1 = v1_env;
#ifdef NU_LAB
    v1 = nu_lab_v1 != 0;
#endif
    if (prec == 1 && (g.bf16 & NU_GEMM_B16)) {          // bf16 storage
        if ((g.ldb & 7) || ((g.bf16 & NU_GEMM_A16) && (g.lda & 7))) return NU_ERR_ARG;
        static const int v16 = getenv("NU_NT16_V") ? atoi(getenv("NU_NT16_V")) : 2;     // development switch: 1 = pipelined, 2 = occupancy
        long long per2 = nu_rup(nu_cdiv(grid_env ? grid_env : (v16 == 2 ? 768 : 512), groups), 8);
        if (per2 > nslots) per2 = nslots;
        dim3 grid2((unsigned)per2, 1, groups);
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
    if (prec == 0 && !v1) {
        // 64-row tiles when the 128-row tiles cannot give every CU its two workgroups (point sets of a few thousand rows)
#ifdef NU_LAB
        const int small_env = nu_lab_small;
#else
        static const int small_env = getenv("NU_NT_SMALL") ? atoi(getenv("NU_NT_SMALL")) : -1;     // development switch: 0 never, 1 always
#endif
        // ... and when they shorten the last round: the persistent grid walks ceil(tiles / 512) rounds, so 1054 tiles of 128 rows
        // (a 67 k-row point set, the outer points of a 512-ray batch) take three rounds for 2.06 rounds of work; as 2108 tiles of
        // 64 rows they take 5 for 4.12.  A 64-row tile costs ~5 % more per FLOP (half the reuse of the weight tile).
        auto round_eff = [&](long long tiles) { return (double)tiles / (double)(nu_cdivl(tiles, 512) * 512); };
        const long long t128 = (long long)nu_cdiv(g.M, TBM) * ntn * groups, t64 = (long long)nu_cdiv(g.M, 64) * ntn * groups;
        const bool small = small_env >= 0 ? small_env != 0 : (t128 < 512 || 0.95 * round_eff(t64) > round_eff(t128));
        const long long nslots2 = small ? (long long)nu_rup(nu_cdiv(g.M, 64), 8) * ntn : nslots;
        if (nslots2 > 0x7fffffffLL) return NU_ERR_ARG;
        long long per2 = nu_rup(nu_cdiv(grid_env ? grid_env : 512, groups), 8);       // two workgroups per CU
        if (per2 > nslots2) per2 = nslots2;
        dim3 grid2((unsigned)per2, 1, groups);
        const NuGemmNT& gs = g;
        switch (g.epi) {
#define NU_CASE2(E) case E: if (small) hipLaunchKernelGGL((gemm_nt2_kernel<E, 1>), grid2, block, 0, stream, gs); \
                            else hipLaunchKernelGGL((gemm_nt2_kernel<E, 2>), grid2, block, 0, stream, gs); break;
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

// ------------------------------------------------------------------------------------------------
// TN kernel (weight gradients): split over the reduced (point) dimension, partial slabs out.
// ------------------------------------------------------------------------------------------------
// The operands are transposed on their way into LDS ([column][k], k contiguous, row stride 36): each
// thread fetches 16 consecutive reduced rows of ONE column (a wave load = 256 contiguous bytes of one row), so the
// inner loop is exactly the NT kernel's (one ds_read_b128 feeds four MFMA k-steps) and nothing consumes a global
// load before the hand-over to LDS -- the loads stay in flight under the 64 MFMAs of the current chunk.
// BIG = operands of 4 GiB or more (64-bit element offsets instead of one uniform base + a 32-bit byte offset).
template <bool BIG, int PREC>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(NuGemmTN g) {
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
__global__ __launch_bounds__(128 * NJ, 2) void gemm_tn2_kernel(NuGemmTN g) {
    constexpr int T = 64 * NJ;                       // tile edge: 256 or 128
    __shared__ __attribute__((aligned(16))) float smem[2][2][T * NT_LDS];     // [stage][A | B][column][k (+4 pad)]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;           // wave tile: rows (N1) 64 wr .. +64, columns (N2) 32 NJ wc .. + 32 NJ
    const int t2 = (g.N2 + T - 1) / T;
    const int n1t = blockIdx.x / t2, n2t = blockIdx.x - n1t * t2;
    const int n1_0 = n1t * T, n2_0 = n2t * T;
    const int split = blockIdx.y;
    const int grp = blockIdx.z;
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

// development aid: occupancy query for the two GEMM kernels (blocks per CU)
extern "C" int nu_debug_occupancy(int which) {
    int n = -1;
    if (which == 0) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (gemm_nt_kernel<NU_EPI_BIAS_SOFTPLUS, 0>), 256, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (gemm_tn_kernel<false, 0>), 256, 0);
    return n;
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
