// mlp_ops.hip -- network-level C entry points: nu_sdf_mlp_{fwd,normal,bwd}, nu_nerfpp_mlp_{fwd,bwd},
// nu_shading_stack_{fwd,bwd}.  Each SEQUENCES the library's kernels for one network pass from C++ -- the launches a host
// binding would otherwise issue one by one (about 500 per training step) -- over caller-provided buffers: the structs of
// include/nu_nerf.h carry the packed layer tables (NuLin) and the activation / gradient buffers (device pointers borrowed
// from the caller, layouts as in DESIGN.md "Data layout").  Nothing is allocated here; split reductions go to the
// caller's arena through NuOpCtx and are finished by nu_ctx_flush.
//
// Reference code replaced (paths relative to /root/reference):
//   SDFNetwork.forward / .gradient and their (double) backward     network/field.py:133-170
//   NeRFNetwork.forward and backward                               network/field.py:265-289
//   AppShadingNetwork.forward and backward                         network/field.py:684-777, :636-682, :371-408
#include "gemm.h"

#define CHK(x) do { const int rc_ = (x); if (rc_ != NU_OK) return rc_; } while (0)

extern "C" int nu_skinny_fwd(const float*, int, int, int, const float*, int, const float*, int, float*, int, hipStream_t);
extern "C" int nu_skinny_fwd_h16(const void*, int, int, int, const float*, int, const float*, int, float*, int, hipStream_t);
extern "C" int nu_skinny_bwd_enqueue_h16(const float*, int, const void*, int, int, int, const float*, int, int, void*, int, int, int,
                                         float*, int, float*, void*, long long, NuReduceDesc*, int*, int, hipStream_t);
extern "C" long long nu_skinny_bwd_workspace_bytes(int, int);
extern "C" int nu_skinny_bwd_enqueue(const float*, int, const float*, int, int, int, const float*, int, int, float*, int, int, int,
                                     float*, int, float*, void*, long long, NuReduceDesc*, int*, int, hipStream_t);
extern "C" long long nu_colsum_workspace_bytes(int);
extern "C" int nu_colsum_enqueue(const float*, int, int, int, float*, int, void*, long long, NuReduceDesc*, int*, int, hipStream_t);
extern "C" int nu_rowscale_dsp(const float*, int, int, int, const float*, float*, int, hipStream_t);
extern "C" long long nu_wgrad_workspace_bytes(int, int, int, int);
extern "C" int nu_wgrad_enqueue(const NuGemmTN*, float*, int, long long, float*, long long, void*, long long, NuReduceDesc*, int*, int,
                                hipStream_t);

static inline int rup_i(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------------------
// context: arithmetic mode + the deferred-reduction arena
// ---------------------------------------------------------------------------------------------------------
extern "C" int nu_op_ctx_size(void) { return (int)sizeof(NuOpCtx); }

static inline int ctx_take(NuOpCtx* c, long long nbytes, int ndesc_needed, hipStream_t stream, float** out, long long* out_bytes) {
    return nu_ctx_take(c, nbytes, ndesc_needed, stream, out, out_bytes);      // (arena, queue and events: gemm_tn.hip)
}

// storage flags of one launch under NuOpCtx.h16 (ignored otherwise): which of A, {C, C2}, {H, D, Cadd} are bf16
#define A16 NU_GEMM_A16
#define C16 NU_GEMM_C16
#define X16 NU_GEMM_X16
typedef unsigned short h16_t;
// address of element `elem_off` of a 16-bit weight table: the bf16 copy (NuOpCtx.h16), or -- arithmetic mode 2 -- the pre-split planes,
// whose layout keeps the fp32 table's offsets with a factor 3 for row offsets that are multiples of 256 (NuGemmNT.B6)
static inline const float* w16(const NuOpCtx* c, const void* tbl, long long elem_off) {
    if (tbl == nullptr) return nullptr;
    return reinterpret_cast<const float*>(static_cast<const h16_t*>(tbl) + (c->prec == 2 ? 3 : 1) * elem_off);
}

static inline void ev_begin(NuOpCtx* c, hipStream_t stream) { nu_ctx_ev_begin(c, stream); }
static inline void ev_end(NuOpCtx* c, hipStream_t stream, double kind, double flops, double bytes) { nu_ctx_ev_end(c, stream, kind, flops, bytes); }

struct NtArgs {
    const float* A; int lda; const float* B; int ldb; int M, N, K; float* C; int ldc; int epi;
    float* C2 = nullptr; int ldc2 = 0; const float* bias = nullptr; const float* H = nullptr; int ldh = 0; const float* D = nullptr;
    int ldd = 0; const float* Cadd = nullptr; int ldadd = 0; int zero_to = 0; int act_cols = 0; int groups = 1;
    long long sA = 0, sB = 0, sC = 0, sBias = 0, sH = 0; unsigned long long* mask = nullptr; int mask_nct = 0;
    const float* B16 = nullptr;     // the bf16 copy of B (h16), or its pre-split planes (mode 2; optional there)
    int st = 0;                     // A16 | C16 | X16 (h16)
    int ktrue = 0;                  // unpadded reduction extent (roofline accounting only)
};
static int nt_fill(const NuOpCtx* c, const NtArgs& a, NuGemmNT& g) {
    const bool h16 = c->h16 != 0;
    if (h16 && (c->prec != 1 || a.B16 == nullptr)) return NU_ERR_ARG;
    g = NuGemmNT{};
    g.A = a.A; g.lda = a.lda; g.B = h16 ? a.B16 : a.B; g.ldb = a.ldb; g.M = a.M; g.N = a.N; g.K = a.K; g.C = a.C; g.ldc = a.ldc; g.C2 = a.C2;
    g.ldc2 = a.ldc2; g.bias = a.bias; g.H = a.H; g.ldh = a.ldh; g.D = a.D; g.ldd = a.ldd; g.Cadd = a.Cadd; g.ldadd = a.ldadd;
    g.zero_to = a.zero_to; g.act_cols = a.act_cols; g.alpha = 1.0f; g.groups = a.groups; g.sA = a.sA; g.sB = a.sB; g.sC = a.sC;
    g.sBias = a.sBias; g.sH = a.sH; g.epi = a.epi; g.bf16 = h16 ? (1 | NU_GEMM_B16 | a.st) : c->prec; g.mask = a.mask;
    g.mask_nct = a.mask ? a.mask_nct : 0;
    if (c->prec == 2 && !h16) g.B6 = a.B16;
    return NU_OK;
}
// algorithmic FLOPs and bytes of one launch at the widths its matrices are stored in (roofline accounting)
static void nt_cost(const NuOpCtx* c, const NtArgs& a, double* flops, double* bytes) {
    const bool h16 = c->h16 != 0;
    const double eA = h16 && (a.st & A16) ? 2 : 4, eB = h16 ? 2 : 4, eC = h16 && (a.st & C16) ? 2 : 4, eX = h16 && (a.st & X16) ? 2 : 4;
    const bool mask_r = a.mask && (a.epi == NU_EPI_MUL_DRELU || a.epi == NU_EPI_B_RELU);      // sign bits replace H
    const double K = a.ktrue ? a.ktrue : a.K, M = a.M, N = a.N;
    *bytes += a.groups * (M * K * eA + N * K * eB + M * N * (eC * (1 + (a.C2 ? 1 : 0)) +
                          eX * ((a.H && !mask_r ? 1 : 0) + (a.D ? 1 : 0) + (a.Cadd ? 1 : 0))));
    *flops += 2.0 * M * N * K * a.groups;
}
static int nt(NuOpCtx* c, const NtArgs& a, hipStream_t stream) {
    if (a.M <= 0) return NU_OK;
    NuGemmNT g;
    CHK(nt_fill(c, a, g));
    ev_begin(c, stream);
    const int rc = nu_gemm_nt_launch(g, stream);
    if (c->ev) {
        double flops = 0, bytes = 0;
        nt_cost(c, a, &flops, &bytes);
        ev_end(c, stream, 0.0, flops, bytes);
    }
    return rc;
}
// n independent problems: one tile list per run of equal epilogue kinds (nu_gemm_nt_batch_launch); timed as ONE launch
static int nt_batch(NuOpCtx* c, const NtArgs* a, int n, hipStream_t stream) {
    NuGemmNT g[16];
    if (n > 16) return NU_ERR_ARG;
    int m = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
        if (a[i].M <= 0) continue;
        CHK(nt_fill(c, a[i], g[m++]));
        nt_cost(c, a[i], &flops, &bytes);
    }
    if (m == 0) return NU_OK;
    ev_begin(c, stream);
    const int rc = nu_gemm_nt_batch_launch(g, m, stream);
    if (c->ev) ev_end(c, stream, 0.0, flops, bytes);
    return rc;
}

// dW[N1, N2] = A0^T B0 (+ A1^T B1), db = column sums of A0.  QUEUED (nu_wgrad_defer): every entry below ends with wgrad_flush, which
// launches the pass's weight gradients together.  st16: NU_TN_*_16 flags (h16)
static int wgrad(NuOpCtx* c, const float* A0, int lda0, const float* B0, int ldb0, int P, int N1, int N2, float* dW, int ldw, float* db,
                 hipStream_t stream, int st16 = 0, const float* A1 = nullptr, int lda1 = 0, const float* B1 = nullptr, int ldb1 = 0,
                 int groups = 1, long long sA0 = 0, long long sB0 = 0, long long sW = 0, long long sDb = 0) {
    if (P <= 0) return NU_OK;
    const int st = c->h16 ? st16 : 0;
    NuGemmTN g = {};
    g.A0 = A0; g.lda0 = lda0; g.B0 = B0; g.ldb0 = ldb0; g.A1 = A1; g.lda1 = lda1; g.B1 = B1; g.ldb1 = ldb1; g.P = P; g.N1 = N1;
    g.N2 = N2; g.S = 1; g.groups = groups; g.sA0 = sA0; g.sB0 = sB0; g.bf16 = c->prec | st;
    const double p = P, n1 = N1, n2 = N2;
    double bytes = p * (n1 * (st & NU_TN_A0_16 ? 2 : 4) + n2 * (st & NU_TN_B0_16 ? 2 : 4)) + 4.0 * n1 * n2;
    if (A1) bytes += p * (n1 * (st & NU_TN_A1_16 ? 2 : 4) + n2 * (st & NU_TN_B1_16 ? 2 : 4));
    return nu_wgrad_defer(c, &g, dW, ldw, sW, db, sDb, 2.0 * p * n1 * n2 * groups * (A1 ? 2 : 1), bytes * groups, stream);
}
static inline int wgrad_flush(NuOpCtx* c, hipStream_t stream) { return nu_wgrad_flush(c, stream); }

// h16: the hidden rows H and the dH written are bf16 under NuOpCtx.h16 (the last hidden layer in front of a skinny head)
static int skinny_bwd(NuOpCtx* c, const float* dy, int ldy, const float* H, int ldh, int P, int K, const float* Ws, int ldw, int NO,
                      float* dH, int lddh, int relu_mask, float* dWs, int lddw, float* db, hipStream_t stream, bool h16 = false) {
    if (P <= 0) return NU_OK;
    float* ws;
    long long nb;
    CHK(ctx_take(c, nu_skinny_bwd_workspace_bytes(K, NO), 2, stream, &ws, &nb));
    if (h16 && c->h16)
        return nu_skinny_bwd_enqueue_h16(dy, ldy, H, ldh, P, K, Ws, ldw, NO, dH, lddh, relu_mask, 0, dWs, lddw, db, ws, nb, c->descs,
                                         &c->ndesc, c->cap, stream);
    return nu_skinny_bwd_enqueue(dy, ldy, H, ldh, P, K, Ws, ldw, NO, dH, lddh, relu_mask, 0, dWs, lddw, db, ws, nb, c->descs, &c->ndesc,
                                 c->cap, stream);
}
static int skinny_fwd(const NuOpCtx* c, const float* H, int ldh, int P, int K, const float* Ws, int ldw, const float* b, int NO, float* out,
                      int ldo, hipStream_t stream, bool h16) {
    if (h16 && c->h16) return nu_skinny_fwd_h16(H, ldh, P, K, Ws, ldw, b, NO, out, ldo, stream);
    return nu_skinny_fwd(H, ldh, P, K, Ws, ldw, b, NO, out, ldo, stream);
}

// ---------------------------------------------------------------------------------------------------------
// SDF network (field.py:133-170): dims 39 -> 256 x3 -> 217 (+39 skip) -> 256 x4 -> 257, Softplus(beta = 100)
// ---------------------------------------------------------------------------------------------------------
extern "C" int nu_sdf_net_size(void) { return (int)sizeof(NuSdfNet); }
extern "C" int nu_sdf_bufs_size(void) { return (int)sizeof(NuSdfBufs); }

// bf16-storage rule (NuOpCtx.h16; include/nu_nerf.h): buffers that an elementwise kernel also touches stay fp32
static inline bool sdfH16(int l) { return l == 1 || l == 2 || l == 3 || l == 5 || l == 6 || l == 7; }   // H[l], Q[l]
static inline bool sdfD16(int l) { return l == 0 || l == 1 || l == 2 || l == 4 || l == 5 || l == 6; }   // D[l], C[l], Aux[l]

extern "C" int nu_sdf_mlp_fwd(NuOpCtx* c, const NuSdfNet* net, const float* X, int x_ld, NuSdfBufs* a, int want_feat,
                              hipStream_t stream) {
    const int P = a->P;
    if (P <= 0) return NU_OK;
    CHK(nu_sdf_embed(X, x_ld, P, a->E, a->U4, want_feat ? a->YX : nullptr, stream));
    const float* src = a->E;
    int lds = 64, K = 64;
    for (int l = 0; l < 8; ++l) {
        const NuLin& L = net->lin[l];
        NtArgs g = {src, lds, L.Wp, L.Kp, P, L.N, K, a->H[l + 1], 256, NU_EPI_BIAS_SOFTPLUS};
        g.bias = L.bias; g.zero_to = L.N; g.ktrue = L.K;
        // the no-gradient form (sampler chain, want_feat == 0) also keeps H[8] in bf16: only the sdf head reads it
        g.B16 = w16(c, L.Wp16, 0); g.st = (sdfH16(l) ? A16 : 0) | ((sdfH16(l + 1) || (l == 7 && !want_feat)) ? C16 : 0);
        CHK(nt(c, g, stream));
        src = a->H[l + 1]; lds = 256; K = 256;
    }
    const NuLin& L8 = net->lin[8];
    if (want_feat) {     // row 0 = sdf (skinny), rows 1..256 = feature
        CHK(nu_skinny_fwd(a->H[8], 256, P, 256, L8.Wp, 256, L8.bias, 1, a->YX, 288, stream));
        NtArgs g = {a->H[8], 256, L8.Wp + 256, 256, P, 256, 256, a->YX + 1, 288, NU_EPI_BIAS_NONE};
        g.bias = L8.bias + 1; g.B16 = c->prec == 2 ? nullptr : w16(c, L8.Wp16, 256);     // (row 1: not a block boundary of the pre-split planes)
        CHK(nt(c, g, stream));
    } else {
        CHK(skinny_fwd(c, a->H[8], 256, P, 256, L8.Wp, 256, L8.bias, 1, a->sdf, 1, stream, true));
    }
    return NU_OK;
}

// reverse sweep: n = d sdf / d x, keeping delta_l for the second-order backward
extern "C" int nu_sdf_mlp_normal(NuOpCtx* c, const NuSdfNet* net, NuSdfBufs* a, hipStream_t stream) {
    const int P = a->P;
    if (P <= 0) return NU_OK;
    const NuLin* ls = net->lin;
    CHK(nu_rowscale_dsp(a->H[8], 256, P, 256, ls[8].Wp, a->D[7], 256, stream));
    for (int l = 7; l > 0; --l) {
        const int Kred = rup_i(ls[l].N, 32);
        NtArgs g = {a->D[l], 256, ls[l].WpT, ls[l].ldT, P, l == 4 ? 256 : ls[l - 1].N, Kred, a->D[l - 1], 256, NU_EPI_MUL_DSP};
        g.H = a->H[l]; g.ldh = 256;
        if (l == 4) g.act_cols = 217;      // columns 217..255 are the skip gradient w.r.t. the embedding: written plain
        else g.zero_to = 256;
        g.B16 = w16(c, ls[l].WpT16, 0); g.st = (sdfD16(l) ? A16 : 0) | (sdfD16(l - 1) ? C16 : 0) | (sdfH16(l) ? X16 : 0);
        CHK(nt(c, g, stream));
    }
    NtArgs g0 = {a->D[0], 256, ls[0].WpT, ls[0].ldT, P, 39, 256, a->G0, 64, NU_EPI_PLAIN};
    g0.zero_to = 64; g0.B16 = w16(c, ls[0].WpT16, 0); g0.st = A16;
    CHK(nt(c, g0, stream));
    return nu_embed_jt(a->E, a->G0, 64, a->D[3] + 217, 256, P, a->n, stream);
}

// backward of (y, n) w.r.t. the parameters (and x when dx != NULL) given dYX [P,288] and nbar [P,3] (NULL: first order)
extern "C" int nu_sdf_mlp_bwd(NuOpCtx* c, const NuSdfNet* net, NuSdfBufs* a, const float* dYX, const float* nbar, float* dx,
                              hipStream_t stream) {
    const int P = a->P;
    if (P <= 0) return NU_OK;
    const NuLin* ls = net->lin;
    const bool second = nbar != nullptr;
    if (second) {
        CHK(nu_embed_j(a->E, nbar, P, a->Q[0], a->Q[4], stream));
        const float* src = a->Q[0];
        int lds = 64, K = 64;
        for (int l = 0; l < 8; ++l) {
            NtArgs g = {src, lds, ls[l].Wp, ls[l].Kp, P, ls[l].N, K, a->Q[l + 1], 256, NU_EPI_Q_SP};
            g.C2 = a->C[l]; g.ldc2 = 256; g.H = a->H[l + 1]; g.ldh = 256; g.D = a->D[l]; g.ldd = 256; g.zero_to = l == 3 ? ls[l].N : 256;
            // Q[l + 1] and C[l] share the output flag, H[l + 1] and D[l] the auxiliary flag: the rule makes each pair one dtype
            g.B16 = w16(c, ls[l].Wp16, 0); g.st = (sdfH16(l) ? A16 : 0) | (sdfH16(l + 1) ? C16 | X16 : 0);
            CHK(nt(c, g, stream));
            src = a->Q[l + 1]; lds = 256; K = 256;
        }
    }
    // B sweep: abar_l (written over C_l when second order)
    float* A[8];
    for (int l = 7; l >= 0; --l) {
        const float* srcA; int lda, K; const float* WT; const float* WT16; int ldT;
        if (l == 7) { srcA = dYX; lda = 288; K = 288; WT = ls[8].WpT; WT16 = w16(c, ls[8].WpT16, 0); ldT = ls[8].ldT; }
        else { srcA = A[l + 1]; lda = 256; K = rup_i(ls[l + 1].N, 32); WT = ls[l + 1].WpT; WT16 = w16(c, ls[l + 1].WpT16, 0); ldT = ls[l + 1].ldT; }
        A[l] = second ? a->C[l] : a->Aux[l];
        NtArgs g = {srcA, lda, WT, ldT, P, ls[l].N, K, A[l], 256, second ? NU_EPI_B_SP : NU_EPI_MUL_DSP};
        g.H = a->H[l + 1]; g.ldh = 256; g.Cadd = second ? a->C[l] : nullptr; g.ldadd = 256;
        if (l == 3 && dx != nullptr) { g.N = 256; g.act_cols = 217; }     // keep the plain skip columns 217..255
        else g.zero_to = 256;
        g.B16 = WT16; g.st = ((l < 7 && sdfD16(l + 1)) ? A16 : 0) | (sdfD16(l) ? C16 | X16 : 0);
        CHK(nt(c, g, stream));
    }
    for (int l = 0; l < 8; ++l) {
        const float* u = l == 0 ? a->E : a->H[l];
        const int ldu = l == 0 ? 64 : 256;
        const int st = (sdfD16(l) ? NU_TN_A0_16 | NU_TN_A1_16 : 0) | (sdfH16(l) ? NU_TN_B0_16 | NU_TN_B1_16 : 0);
        if (second) CHK(wgrad(c, A[l], 256, u, ldu, P, ls[l].N, ls[l].Kp, ls[l].dWp, ls[l].ldd, c->flat + ls[l].db_off, stream, st, a->D[l], 256,
                              a->Q[l], l == 0 ? 64 : 256));
        else CHK(wgrad(c, A[l], 256, u, ldu, P, ls[l].N, ls[l].Kp, ls[l].dWp, ls[l].ldd, c->flat + ls[l].db_off, stream, st));
    }
    CHK(wgrad(c, dYX, 288, a->H[8], 256, P, 257, 256, ls[8].dWp, 256, c->flat + ls[8].db_off, stream));
    if (second) {          // d W8[sdf row] += sum_p q_8: an ACCUMULATING reduction, so the reduction that writes dW8 must be queued first
        CHK(wgrad_flush(c, stream));
        float* ws; long long nb;
        CHK(ctx_take(c, nu_colsum_workspace_bytes(256), 1, stream, &ws, &nb));
        CHK(nu_colsum_enqueue(a->Q[8], 256, P, 256, ls[8].dWp, 1, ws, nb, c->descs, &c->ndesc, c->cap, stream));
    }
    if (dx != nullptr) {
        NtArgs g = {A[0], 256, ls[0].WpT, ls[0].ldT, P, 39, 256, a->dE0, 64, NU_EPI_PLAIN};
        g.zero_to = 64; g.B16 = w16(c, ls[0].WpT16, 0); g.st = A16;
        CHK(nt(c, g, stream));
        CHK(nu_embed_jt2(a->E, a->dE0, 64, A[3] + 217, 256, second ? a->G0 : nullptr, 64, second ? a->D[3] + 217 : nullptr, 256,
                         second ? nbar : nullptr, P, dx, 0, stream));
    }
    return wgrad_flush(c, stream);
}

// ---------------------------------------------------------------------------------------------------------
// make_predictor stacks (field.py:371-408): 3 hidden ReLU layers + a 1..3-wide head
// h16: the hidden activations [0..2], the backward scratch tmp[0], tmp[1] and dH3 are bf16 (the skinny head kernels read / write bf16)
// ---------------------------------------------------------------------------------------------------------
// One make_predictor stack of a pass: layers, input, rows, hidden activations + their sign bits; backward: cotangent of the raw head,
// scratch, optional input gradient.
struct Stack {
    const NuLin* ls; const float* X; int ldx, rows; float* const* Hs; unsigned long long* const* masks;
    const float* dy = nullptr; int ldy = 0, no = 0; float* dH3 = nullptr; float* const* tmp = nullptr; float* dX = nullptr; int lddx = 0, dxc = 0;
};
// The stacks of one pass are independent of each other, so level j of ALL of them is one launch (one tile list, nt_batch): the four
// light predictors of AppShadingNetwork.forward are 3 launches instead of 12.
static int relu_stacks_fwd(NuOpCtx* c, const Stack* st, int n, int nct, hipStream_t stream) {
    NtArgs g[8];
    if (n > 8) return NU_ERR_ARG;
    for (int j = 0; j < 3; ++j) {
        for (int k = 0; k < n; ++k) {
            const NuLin* ls = st[k].ls;
            g[k] = NtArgs{j == 0 ? st[k].X : st[k].Hs[j - 1], j == 0 ? st[k].ldx : 256, ls[j].Wp, ls[j].Kp, st[k].rows, 256, ls[j].Kp, st[k].Hs[j], 256,
                          NU_EPI_BIAS_RELU};
            g[k].bias = ls[j].bias; g[k].mask = st[k].masks[j]; g[k].mask_nct = nct;
            g[k].B16 = w16(c, ls[j].Wp16, 0); g[k].st = (j > 0 ? A16 : 0) | C16;
        }
        CHK(nt_batch(c, g, n, stream));
    }
    return NU_OK;
}
// dH3: gradient w.r.t. the pre-activation of layer 2 (written by the skinny head's backward); tmp[2]: [rows,256] scratch; dX
// (optional): input gradient.  h16: dH3 and tmp[] are bf16.
static int relu_stacks_bwd(NuOpCtx* c, const Stack* st, int n, int nct, hipStream_t stream) {
    NtArgs g[8];
    if (n > 8) return NU_ERR_ARG;
    for (int j = 2; j >= 0; --j) {
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const Stack& s = st[k];
            const NuLin* ls = s.ls;
            const float* dA = j == 2 ? s.dH3 : s.tmp[j];
            const float* u = j == 0 ? s.X : s.Hs[j - 1];
            const int ldu = j == 0 ? s.ldx : 256;
            CHK(wgrad(c, dA, 256, u, ldu, s.rows, 256, ls[j].Kp, ls[j].dWp, ls[j].ldd, c->flat + ls[j].db_off, stream,
                      NU_TN_A0_16 | (j > 0 ? NU_TN_B0_16 : 0)));
            if (j > 0) {
                g[m] = NtArgs{dA, 256, ls[j].WpT, ls[j].ldT, s.rows, 256, 256, s.tmp[j - 1], 256, NU_EPI_MUL_DRELU};
                g[m].H = s.Hs[j - 1]; g[m].ldh = 256; g[m].mask = s.masks[j - 1]; g[m].mask_nct = nct;
                g[m].B16 = w16(c, ls[j].WpT16, 0); g[m].st = A16 | C16 | X16;
                ++m;
            } else if (s.dX != nullptr) {
                g[m] = NtArgs{dA, 256, ls[j].WpT, ls[j].ldT, s.rows, s.dxc, 256, s.dX, s.lddx, NU_EPI_PLAIN};
                g[m].B16 = w16(c, ls[j].WpT16, 0); g[m].st = A16;
                ++m;
            }
        }
        CHK(nt_batch(c, g, m, stream));
    }
    return NU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// NeRF++ (field.py:265-289): 8 x 256 ReLU with the 84-d embedding re-concatenated before layer 5, alpha / feature heads,
// one 283 -> 128 view layer, rgb head
// h16: H[1..4], H[6..8], dA[1..4], dA[6..8] and dH8a are bf16
// ---------------------------------------------------------------------------------------------------------
extern "C" int nu_nerf_net_size(void) { return (int)sizeof(NuNerfNet); }
extern "C" int nu_nerf_bufs_size(void) { return (int)sizeof(NuNerfBufs); }
static inline bool nerfH16(int i) { return i == 1 || i == 2 || i == 3 || i == 4 || i == 6 || i == 7 || i == 8; }
static inline bool nerfdA16(int i) { return nerfH16(i); }

extern "C" int nu_nerfpp_mlp_fwd(NuOpCtx* c, const NuNerfNet* net, const float* pt, int pt_ld, NuNerfBufs* b, hipStream_t stream) {
    const int P = b->P;
    if (P <= 0) return NU_OK;
    CHK(nu_nerf_embed(pt, pt_ld, P, b->H[0], b->H[5], b->V, stream));
    const float* src = b->H[0];
    int lds = 96;
    for (int i = 0; i < 8; ++i) {
        const NuLin& L = net->pts[i];
        const int ldc = i == 4 ? 352 : 256;
        NtArgs g = {src, lds, L.Wp, L.Kp, P, 256, L.Kp, b->H[i + 1], ldc, NU_EPI_BIAS_RELU};
        g.bias = L.bias; g.mask = b->mask[i + 1]; g.mask_nct = 2;
        g.B16 = w16(c, L.Wp16, 0); g.st = (nerfH16(i) ? A16 : 0) | (nerfH16(i + 1) ? C16 : 0);
        CHK(nt(c, g, stream));
        src = b->H[i + 1]; lds = ldc;
    }
    CHK(skinny_fwd(c, b->H[8], 256, P, 256, net->alpha.Wp, 256, net->alpha.bias, 1, b->sig, 1, stream, true));
    NtArgs gf = {b->H[8], 256, net->feat.Wp, 256, P, 256, 256, b->V, 288, NU_EPI_BIAS_NONE};
    gf.bias = net->feat.bias; gf.B16 = w16(c, net->feat.Wp16, 0); gf.st = A16;
    CHK(nt(c, gf, stream));
    NtArgs gv = {b->V, 288, net->view.Wp, 288, P, 128, 288, b->HV, 128, NU_EPI_BIAS_RELU};
    gv.bias = net->view.bias; gv.B16 = w16(c, net->view.Wp16, 0);
    CHK(nt(c, gv, stream));
    return nu_skinny_fwd(b->HV, 128, P, 128, net->rgb.Wp, 128, net->rgb.bias, 3, b->rgb, 4, stream);
}

// raw head cotangents dsig [P], drgb [P,4] -> parameter gradients (and dx, ddir when b->dx != NULL)
extern "C" int nu_nerfpp_mlp_bwd(NuOpCtx* c, const NuNerfNet* net, const float* pt, int pt_ld, NuNerfBufs* b, const float* dsig,
                                 const float* drgb, hipStream_t stream) {
    const int P = b->P;
    if (P <= 0) return NU_OK;
    const bool want_in = b->dx != nullptr;
    const int ldf = want_in ? 288 : 256;
    CHK(skinny_bwd(c, drgb, 4, b->HV, 128, P, 128, net->rgb.Wp, 128, 3, b->dHV, 128, 1, net->rgb.dWp, 128, c->flat + net->rgb.db_off, stream));
    CHK(wgrad(c, b->dHV, 128, b->V, 288, P, 128, 288, net->view.dWp, 288, c->flat + net->view.db_off, stream));
    NtArgs gF = {b->dHV, 128, net->view.WpT, net->view.ldT, P, ldf, 128, b->dF, ldf, NU_EPI_PLAIN};
    gF.B16 = w16(c, net->view.WpT16, 0);
    CHK(nt(c, gF, stream));
    CHK(wgrad(c, b->dF, ldf, b->H[8], 256, P, 256, 256, net->feat.dWp, 256, c->flat + net->feat.db_off, stream, NU_TN_B0_16));
    CHK(skinny_bwd(c, dsig, 1, b->H[8], 256, P, 256, net->alpha.Wp, 256, 1, b->dH8a, 256, 1, net->alpha.dWp, 256, c->flat + net->alpha.db_off, stream,
                   true));
    NtArgs g8 = {b->dF, ldf, net->feat.WpT, net->feat.ldT, P, 256, 256, b->dA[8], 256, NU_EPI_B_RELU};
    g8.H = b->H[8]; g8.ldh = 256; g8.Cadd = b->dH8a; g8.ldadd = 256; g8.mask = b->mask[8]; g8.mask_nct = 2;
    g8.B16 = w16(c, net->feat.WpT16, 0); g8.st = C16 | X16;          // H[8] and dH8a are bf16
    CHK(nt(c, g8, stream));
    const float* dA = b->dA[8];
    int lda = 256;
    int dA_i = 8;                              // index of the buffer dA points at (its storage type follows nerfdA16)
    const float* dskip = nullptr;
    for (int i = 7; i >= 0; --i) {
        const NuLin& L = net->pts[i];
        const int ldu = i == 0 ? 96 : (i == 5 ? 352 : 256);
        CHK(wgrad(c, dA, lda, b->H[i], ldu, P, 256, L.Kp, L.dWp, L.ldd, c->flat + L.db_off, stream,
                  (nerfdA16(dA_i) ? NU_TN_A0_16 : 0) | (nerfH16(i) ? NU_TN_B0_16 : 0)));
        if (i > 0) {
            const int st = (nerfdA16(dA_i) ? A16 : 0) | (nerfdA16(i) ? C16 : 0) | (nerfH16(i) ? X16 : 0);
            if (i == 5 && want_in) {       // columns 256..339 of the layer-5 input: the re-concatenated embedding, plain gradient
                NtArgs g = {dA, lda, L.WpT, L.ldT, P, 340, 256, b->dA[i], 352, NU_EPI_MUL_DRELU};
                g.H = b->H[i]; g.ldh = ldu; g.act_cols = 256; g.zero_to = 352; g.mask = b->mask[i]; g.mask_nct = 2;
                g.B16 = w16(c, L.WpT16, 0); g.st = st;
                CHK(nt(c, g, stream));
                dskip = b->dA[i];
                dA = b->dA[i]; lda = 352;
            } else {
                NtArgs g = {dA, lda, L.WpT, L.ldT, P, 256, 256, b->dA[i], 256, NU_EPI_MUL_DRELU};
                g.H = b->H[i]; g.ldh = ldu; g.mask = b->mask[i]; g.mask_nct = 2;
                g.B16 = w16(c, L.WpT16, 0); g.st = st;
                CHK(nt(c, g, stream));
                dA = b->dA[i]; lda = 256;
            }
            dA_i = i;
        } else if (want_in) {
            NtArgs g = {dA, lda, L.WpT, L.ldT, P, 84, 256, b->dE4, 96, NU_EPI_PLAIN};
            g.zero_to = 96; g.B16 = w16(c, L.WpT16, 0); g.st = nerfdA16(dA_i) ? A16 : 0;
            CHK(nt(c, g, stream));
            CHK(nu_nerf_embed_bwd(pt, pt_ld, b->H[0], b->V, b->dE4, 96, dskip + 256, 352, b->dF + 256, ldf, P, b->dx, b->ddir, stream));
        }
    }
    return wgrad_flush(c, stream);
}

// ---------------------------------------------------------------------------------------------------------
// Shading stack (field.py:684-777): 4 material predictors (batched), the encodings, 4 light predictors, BRDF combine
// h16: M[0..2] and dM[0..2] are bf16
// ---------------------------------------------------------------------------------------------------------
extern "C" int nu_shade_net_size(void) { return (int)sizeof(NuShadeNet); }
extern "C" int nu_shade_bufs_size(void) { return (int)sizeof(NuShadeBufs); }

extern "C" int nu_shading_stack_fwd(NuOpCtx* c, const NuShadeNet* net, NuShadeBufs* s, const float* YX, const float* E, const float* nrm,
                                    const float* pt, const int* idx, float* color_rm, hipStream_t stream) {
    const int P = s->P, R = s->R;
    if (P <= 0) return NU_OK;
    const int rows_ol = 3 * P + R;
    // materials: layer 0 batched (N = 1024), layers 1-2 grouped x4, block-diagonal 6-wide head
    NtArgs m0 = {YX, 288, net->WpM0, 288, P, 1024, 288, s->M[0], 1024, NU_EPI_BIAS_RELU};
    m0.bias = net->bM0; m0.mask = s->maskM[0]; m0.mask_nct = 8; m0.B16 = w16(c, net->WpM0_16, 0); m0.st = C16;
    CHK(nt(c, m0, stream));
    for (int j = 1; j <= 2; ++j) {
        NtArgs g = {s->M[j - 1], 1024, net->WpM[j], 256, P, 256, 256, s->M[j], 1024, NU_EPI_BIAS_RELU};
        g.bias = net->bM[j]; g.groups = 4; g.sA = 256; g.sB = 65536; g.sC = 256; g.sBias = 256; g.mask = s->maskM[j]; g.mask_nct = 8;
        g.B16 = w16(c, net->WpM16[j], 0); g.st = A16 | C16;
        CHK(nt(c, g, stream));
    }
    CHK(skinny_fwd(c, s->M[2], 1024, P, 1024, net->Ws6, 1024, net->b6, 6, s->Mraw, 8, stream, true));
    CHK(nu_shade_encode_fwd(nrm, pt, 8, E, s->Mraw, 8, P, net->sphere, net->ld_ol, net->refrac_dim, net->ld_rl, s->OLin, s->ILin, s->IWin,
                            s->RLin, s->SD, stream));
    if (R > 0) CHK(nu_spec_encode(s->extra_dirs, s->extra_pts, R, s->extra_pts ? net->sphere : 0, s->OLin + (long long)3 * P * net->ld_ol,
                                  net->ld_ol, stream));
    const Stack stacks[4] = {{net->outer_light, s->OLin, net->ld_ol, rows_ol, s->OLh, s->maskOL},
                             {net->inner_light, s->ILin, 128, 2 * P, s->ILh, s->maskIL},
                             {net->inner_weight, s->IWin, 96, P, s->IWh, s->maskIW},
                             {net->refrac_light, s->RLin, net->ld_rl, P, s->RLh, s->maskRL}};
    CHK(relu_stacks_fwd(c, stacks, 4, 2, stream));
    CHK(skinny_fwd(c, s->OLh[2], 256, rows_ol, 256, net->outer_light[3].Wp, 256, net->outer_light[3].bias, 3, s->OLo, 4, stream, true));
    CHK(skinny_fwd(c, s->ILh[2], 256, 2 * P, 256, net->inner_light[3].Wp, 256, net->inner_light[3].bias, 3, s->ILo, 4, stream, true));
    CHK(skinny_fwd(c, s->IWh[2], 256, P, 256, net->inner_weight[3].Wp, 256, net->inner_weight[3].bias, 1, s->IWo, 1, stream, true));
    CHK(skinny_fwd(c, s->RLh[2], 256, P, 256, net->refrac_light[3].Wp, 256, net->refrac_light[3].bias, 3, s->RLo, 4, stream, true));
    return nu_shade_combine_fwd(s->Mraw, 8, s->OLo, s->ILo, s->IWo, s->RLo, s->SD, net->lut, idx, P, net->exp_max, color_rm, s->aux, stream);
}

// dcolor_rm -> parameter gradients, dYX [P,288] (feature / x columns) and dn [P,3].  The caller has already put the cotangents
// of the R per-ray mirror queries into dOLo[3P..] and added any direct cotangent of the occlusion head to dIWo AFTER
// nu_shade_combine_bwd -- so this entry takes `stage`: 0 = combine backward only, 1 = everything after it.
extern "C" int nu_shading_stack_bwd(NuOpCtx* c, const NuShadeNet* net, NuShadeBufs* s, const float* YX, const float* nrm, const float* pt,
                                    const int* idx, const float* dcolor_rm, int stage, hipStream_t stream) {
    const int P = s->P, R = s->R;
    if (P <= 0) return NU_OK;
    const int rows_ol = 3 * P + R;
    if (stage == 0)
        return nu_shade_combine_bwd(s->Mraw, 8, s->OLo, s->ILo, s->IWo, s->RLo, s->SD, net->lut, idx, P, net->exp_max, dcolor_rm, s->dMraw,
                                    s->dOLo, s->dILo, s->dIWo, s->dRLo, s->dNoV, stream);
    const Stack stacks[4] = {
        {net->outer_light, s->OLin, net->ld_ol, rows_ol, s->OLh, s->maskOL, s->dOLo, 4, 3, s->dH3[0], s->tmpOL, s->dOLin, net->ld_ol, net->ld_ol},
        {net->inner_light, s->ILin, 128, 2 * P, s->ILh, s->maskIL, s->dILo, 4, 3, s->dH3[1], s->tmpIL, s->dILin, 128, 128},
        {net->inner_weight, s->IWin, 96, P, s->IWh, s->maskIW, s->dIWo, 1, 1, s->dH3[2], s->tmpIW, nullptr, 0, 0},
        {net->refrac_light, s->RLin, net->ld_rl, P, s->RLh, s->maskRL, s->dRLo, 4, 3, s->dH3[3], s->tmpRL, nullptr, 0, 0}};
    for (const Stack& p : stacks) {
        const NuLin& head = p.ls[3];
        CHK(skinny_bwd(c, p.dy, p.ldy, p.Hs[2], 256, p.rows, 256, head.Wp, 256, p.no, p.dH3, 256, 1, head.dWp, head.ldd, c->flat + head.db_off, stream,
                       true));
    }
    CHK(relu_stacks_bwd(c, stacks, 4, 2, stream));
    CHK(nu_shade_encode_bwd(nrm, pt, 8, s->SD, s->dOLin, net->ld_ol, net->sphere, s->dILin, s->dNoV, P, s->dn, s->dMraw, 8, stream));
    // materials backward
    CHK(skinny_bwd(c, s->dMraw, 8, s->M[2], 1024, P, 1024, net->Ws6, 1024, 6, s->dM[2], 1024, 1, net->dWs6, 1024, c->flat + net->db6_off, stream,
                   true));
    const float* dA = s->dM[2];
    for (int j = 2; j >= 1; --j) {
        const int a16 = 1;                         // dM[2] (skinny head backward), dM[1], dM[0]: bf16
        CHK(wgrad(c, dA, 1024, s->M[j - 1], 1024, P, 256, 256, net->dWpM[j], 256, c->flat + net->dbM_off[j], stream,
                  (a16 ? NU_TN_A0_16 : 0) | NU_TN_B0_16, nullptr, 0, nullptr, 0, 4, 256, 256, 65536, 256));
        NtArgs g = {dA, 1024, net->WpTM[j], 256, P, 256, 256, s->dM[j - 1], 1024, NU_EPI_MUL_DRELU};
        g.H = s->M[j - 1]; g.ldh = 1024; g.groups = 4; g.sA = 256; g.sB = 65536; g.sC = 256; g.sH = 256; g.mask = s->maskM[j - 1]; g.mask_nct = 8;
        g.B16 = w16(c, net->WpTM16[j], 0); g.st = (a16 ? A16 : 0) | C16 | X16;
        CHK(nt(c, g, stream));
        dA = s->dM[j - 1];
    }
    CHK(wgrad(c, dA, 1024, YX, 288, P, 1024, 288, net->dWpM0, 288, c->flat + net->dbM_off[0], stream, NU_TN_A0_16));
    NtArgs gy = {dA, 1024, net->WpTM0, 1024, P, 288, 1024, s->dYX, 288, NU_EPI_PLAIN};
    gy.B16 = w16(c, net->WpTM0_16, 0); gy.st = A16;
    CHK(nt(c, gy, stream));
    return wgrad_flush(c, stream);
}
