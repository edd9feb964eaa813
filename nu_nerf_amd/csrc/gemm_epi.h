// gemm_epi.h -- tile constants and the epilogue shared by the NT GEMM kernels (gemm_nt.hip, gemm_nt16.hip): accumulators ->
// wave-private LDS scratch -> row-contiguous 16-byte rows, fused bias / activation / derivative epilogues, ReLU sign-bit words.
#pragma once
#include "gemm.h"
#include <type_traits>

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
    const int ct = g.mask_ct0 + z * ntn + nt;
    return (ea.mask && ct < g.mask_nct) ? ea.mask + ((((long long)mt * g.mask_nct + ct) * 4 + wid) * 64) : nullptr;
}

// NBH: row groups whose auxiliary loads are in flight together in the fast path (register budget of the caller)
// TMN / tm0: the wave holds TMN 32-row blocks of the 64 x 64 slab `wid` of the 128 x 128 tile at (m0, n0), starting at block tm0
// (2 / 0: the whole slab -- the 128-row-tile kernels; 1 / 0 or 1: one block -- the 64-row-tile kernel, where the two waves that
// share a slab's sign-bit words belong to different workgroups).
template <int EPI, int NBH, bool H16 = false, int TMN = 2>
static __device__ __forceinline__ void nt_epilogue(const NuGemmNT& g, const NtEpiArgs<EPI>& ea, f32x16 (&acc)[TMN][2], float* scr,
                                                   int m0, int n0, unsigned long long* mwave, unsigned mlo, unsigned mhi,
                                                   int lane, int wid, int tm0 = 0) {
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
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_LDS + tn * 32 + li] = acc[tt][tn][r];
        if (slab_full) {
            // wave-uniform fast path (every interior tile): straight-line code, no per-lane guards; the auxiliary loads of
            // four row groups are in flight together; the activation math is branch-free (nu_common.h).  The two wave-uniform
            // switches of the slab -- sign-bit words instead of H (UB), columns past act_cols written plain (PL) -- select one of
            // up to three COPIES of the path, so that no element of it carries a branch or a select for them.
            auto fast = [&](auto ub_c, auto pl_c, auto c16_c, auto x16_c) {
                constexpr bool UB = decltype(ub_c)::value, PL = decltype(pl_c)::value;
                // (bf16-storage kernel: the storage widths of this launch's matrices, compile-time in the copy)
                constexpr bool c16 = decltype(c16_c)::value, x16 = decltype(x16_c)::value;
                constexpr int ec = c16 ? 2 : 4, ex = x16 ? 2 : 4;
                constexpr int NB = kNeedH ? NBH : 4;         // row groups in flight
#pragma unroll
                for (int hb = 0; hb < 8 / NB; ++hb) {
                    f32x4 v4[NB], h4[NB], d4[NB], c4v[NB];
#pragma unroll
                    for (int ii = 0; ii < NB; ++ii) {
                        const int i = hb * NB + ii;
                        const long long roff = (long long)(tm * 32 + i * 4);
                        v4[ii] = *reinterpret_cast<const f32x4*>(&scr[(i * 4 + (lane >> 4)) * EPI_LDS + colq]);
                        if (kNeedH && !PL && !UB) h4[ii] = nt_ld4<H16>(Hu + roff * g.ldh * ex + oH, x16);
                        if (kNeedD && !PL) d4[ii] = nt_ld4<H16>(Du + roff * g.ldd * ex + oD, x16);
                        if (kNeedAdd && !PL) c4v[ii] = nt_ld4<H16>(Au + roff * g.ldadd * ex + oA, x16);
                    }
#pragma unroll
                    for (int ii = 0; ii < NB; ++ii) {
                        const int i = hb * NB + ii;
                        const long long roff = (long long)(tm * 32 + i * 4);
                        f32x4 o4, o24;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float o2 = 0.f;
                            const float v = g.alpha * v4[ii][e];
                            if constexpr (PL) {
                                o4[e] = v;
                            } else if constexpr (kMaskR && UB) {
                                // the element's sign bit is bit `lane` of word (tm, i, e): the word, read into an SGPR pair, IS the
                                // select mask of one v_cndmask (h > 0 ? v : 0 of nu_epi_apply, then + Cadd for B_RELU)
                                const int src = (tm * 8 + i) * 4 + e;           // wave-uniform
                                const unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, src) << 32) |
                                                             (unsigned)__builtin_amdgcn_readlane(mlo, src);
                                // (gfx940 family: a VALU read of an SGPR needs two wait states behind the VALU -- here v_readlane -- that
                                // wrote it; the compiler keeps that distance for its own instructions but cannot see into this one)
                                float sel;
                                asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(sel) : "v"(v), "s"(w));
                                o4[e] = kNeedAdd ? sel + c4v[ii][e] : sel;
                            } else {
                                o4[e] = nu_epi_apply<EPI>(v, bv[e], kNeedH ? h4[ii][e] : 0.f, kNeedD ? d4[ii][e] : 0.f,
                                                          kNeedAdd ? c4v[ii][e] : 0.f, o2);
                            }
                            o24[e] = o2;
                        }
                        nt_st4<H16>(Cu + roff * g.ldc * ec + oC, o4, c16);
                        if (kNeedD) nt_st4<H16>(C2u + roff * g.ldc2 * ec + oC2, o24, c16);
                        if constexpr (kMaskW && UB) {          // (UB of a writer: the slab has sign-bit words)
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
            };
            // ... and, in the bf16-storage kernel, the launch's two storage flags (C / C2; H / D / Cadd): a copy per setting, so that
            // no load or store of the path carries the width switch
            auto fast_w = [&](auto ub_c, auto pl_c) {
                constexpr bool kAux = kNeedH || kNeedD || kNeedAdd;
                if constexpr (H16) {
                    if (c16) {
                        if (kAux && x16) fast(ub_c, pl_c, std::true_type{}, std::true_type{});
                        else fast(ub_c, pl_c, std::true_type{}, std::false_type{});
                    } else {
                        if (kAux && x16) fast(ub_c, pl_c, std::false_type{}, std::true_type{});
                        else fast(ub_c, pl_c, std::false_type{}, std::false_type{});
                    }
                } else {
                    fast(ub_c, pl_c, std::false_type{}, std::false_type{});
                }
            };
            if (kNeedH && slab_plain) fast_w(std::false_type{}, std::true_type{});
            else if ((kMaskR || kMaskW) && mwave) fast_w(std::true_type{}, std::false_type{});
            else fast_w(std::false_type{}, std::false_type{});
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
