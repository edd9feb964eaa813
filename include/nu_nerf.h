/* nu_nerf.h -- C ABI of libnunerf.so: the MI355X (gfx950) hot path of NU-NeRF's stage-1 training step.
 *
 * The reference (jjjkkyz/NU-NeRF) has no FFI: its hot path is eager PyTorch inside
 * network/renderer_zerothick.py and network/field.py.  This header is the boundary a maintainer binds instead
 * (ctypes stub: INTEGRATION.md); each entry names the reference code it replaces (paths relative to the
 * reference repository root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (fp32 unless stated; indices int32; masks uint8),
 *     row-major, contiguous unless a leading dimension (ld*) is given; nothing is allocated or freed inside;
 *     scratch memory comes in through (workspace, workspace_bytes) with a *_workspace_bytes() size query;
 *   - all work is enqueued on `stream` (the caller's current HIP stream); no call synchronises the device;
 *   - return value: 0 on success, negative NU_ERR_* otherwise (never throws);
 *   - "P" counts points, "R" rays, "S" samples per ray.  Activation buffers use padded leading dimensions whose pad
 *     columns are zero (see DESIGN.md "Data layout").
 */
#ifndef NU_NERF_H
#define NU_NERF_H

/* the HIP runtime's own definition, repeated so that plain-C consumers need no HIP headers
 * (an identical typedef redeclaration is legal in C11 and C++) */
typedef struct ihipStream_t* hipStream_t;

#ifdef __cplusplus
extern "C" {
#endif

#define NU_OK 0
#define NU_ERR_ARG (-1)
#define NU_ERR_LAUNCH (-2)
#define NU_ERR_WORKSPACE (-3)

/* ---------------------------------------------------------------------------------------------------------
 * fp32-MFMA GEMMs: the contractions of every `lin(x)` / autograd matmul in field.py:133-150 (SDFNetwork.forward),
 * :158-170 (SDFNetwork.gradient and its double backward), :265-289 (NeRFNetwork.forward), :371-408 (make_predictor).
 * --------------------------------------------------------------------------------------------------------- */
enum NuEpi {                   /* epilogue applied to v = alpha * (A . B^T)[row, col] */
    NU_EPI_BIAS_NONE = 0,      /* C = v + bias[col]                                         (last linear layers) */
    NU_EPI_BIAS_RELU = 1,      /* C = relu(v + bias[col])                                   (field.py:274, :387-391) */
    NU_EPI_BIAS_SOFTPLUS = 2,  /* C = softplus_beta100(v + bias[col])                       (field.py:126-127, :147-148) */
    NU_EPI_MUL_DRELU = 3,      /* C = v * (H > 0)                                           (ReLU backward) */
    NU_EPI_MUL_DSP = 4,        /* C = v * sp'(H),  sp' = 1 - exp(-100 H)                    (Softplus backward) */
    NU_EPI_Q_SP = 5,           /* C = v * sp'(H);  C2 = v * D * 100 * (1 - sp'(H))          (second-order sweep) */
    NU_EPI_B_SP = 6,           /* C = v * sp'(H) + Cadd */
    NU_EPI_PLAIN = 7,          /* C = v */
    NU_EPI_B_RELU = 8,         /* C = v * (H > 0) + Cadd */
    NU_EPI_COUNT = 9
};

#define NU_GEMM_B16 8
#define NU_GEMM_A16 16
#define NU_GEMM_PRESPLIT_ALWAYS 4   /* mode 2 with B6: take the pre-split kernel whatever the tile count (tests; the library otherwise picks) */
#define NU_GEMM_C16 32
#define NU_GEMM_X16 64
typedef struct NuGemmNT {      /* C[M,N] = epi(A[M,K] . B[N,K]^T);  K % 32 == 0, lda/ldb % 4 == 0, A/B 16-byte aligned */
    const float* A; int lda;
    const float* B; int ldb;   /* packed weights: >= ceil128(N) rows, zero padded */
    int M, N, K;
    float* C; int ldc;
    float* C2; int ldc2;
    const float* bias;
    const float* H; int ldh;
    const float* D; int ldd;
    const float* Cadd; int ldadd;
    int zero_to;               /* columns [N, zero_to) of C/C2 are written as 0 */
    int act_cols;              /* derivative epilogues: columns >= act_cols are written as plain v (0: all columns) */
    float alpha;
    int groups;                /* grouped launch; element strides per group follow */
    long long sA, sB, sC, sC2, sBias, sH, sD, sCadd;
    int epi;                   /* enum NuEpi */
    int bf16;                  /* bits 0-1: 0 exact fp32 MFMA (default); 1 bf16 MFMA, fp32 accumulate (operands rounded to bf16);
                                  2 exact 3-way bf16 split of both operands, the six partial products >= 2^-16 (fp32-equivalent).
                                  With mode 1, storage flags (bf16 tensors in HBM, `float*` fields then point to __bf16 data,
                                  leading dimensions stay in ELEMENTS): NU_GEMM_B16 (required: selects the bf16-storage kernel)
                                  B is a bf16 weight table (NuPackDesc.Wp16 / WpT16); NU_GEMM_A16: A is bf16; NU_GEMM_C16: C and
                                  C2 are written as bf16; NU_GEMM_X16: H, D, Cadd are bf16 */
    /* ReLU sign bits (optional): NU_EPI_BIAS_RELU writes, NU_EPI_MUL_DRELU / NU_EPI_B_RELU read them INSTEAD of H -- 2 KB per
     * 128x128 tile in place of 64 KB of activations.  Layout is private to the kernel (wave ballots per 4-row group):
     * cdiv(M,128) * mask_nct * 256 words, mask_nct = column tiles of the activation matrix as the WRITER saw it
     * (groups * cdiv(N,128)); reader and writer must tile the same [M, columns] matrix. */
    unsigned long long* mask; int mask_nct;
    int mask_ct0;              /* first column tile of this problem in the sign-bit matrix: a problem that is one column group of a
                                  wider activation matrix (the four material predictors side by side) when it is launched on its own */
    const void* B6;            /* mode 2 only, optional: B already split into its three bf16 planes by the pack launch (NuPackDesc.planes = 3).
                                  Layout: the table [rows][ldb] (ldb % 16 == 0, rows padded to a multiple of 256) is cut into blocks of 256
                                  rows; a block stores its 16-wide k-groups one after the other, each as [256 rows][hi x16 | mid x16 | lo x16]
                                  (24 576 contiguous bytes -- one LDS stage of the kernel).  bf16 index of element (n, k) of plane p:
                                  ((n >> 8) * (ldb >> 4) + (k >> 4)) * 12288 + (n & 255) * 48 + 16 p + ((k & 15) ^ (n & 8))   [the two halves of a
                                  16-group are swapped in rows with bit 3 set: LDS bank spreading of the kernel's fragment reads].  B6 points at the block the
                                  problem's first row opens (its row offset in the table must be a multiple of 256 -- then 3 x the fp32
                                  table's element offset, and sB applies with the same factor 3).
                                  Selects gemm_nt6_kernel (csrc/gemm_nt6.hip): same results as without it, bit for bit */
} NuGemmNT;

typedef struct NuGemmTN {      /* dW[N1,N2] = A0^T B0 (+ A1^T B1), reduced over P rows in S deterministic splits */
    const float* A0; int lda0; const float* B0; int ldb0;
    const float* A1; int lda1; const float* B1; int ldb1;   /* A1 == NULL: single pair */
    int P, N1, N2;
    float* slab; float* bias_slab;                           /* filled by nu_wgrad from the workspace */
    int S, groups;
    long long sA0, sB0, sA1, sB1, sSlab, sBiasSlab;
    int bf16, pad_;                                          /* bits 0-1 as NuGemmNT.bf16; storage flags (mode 1): NU_TN_A0_16 ... */
} NuGemmTN;
#define NU_TN_A0_16 16
#define NU_TN_B0_16 32
#define NU_TN_A1_16 64
#define NU_TN_B1_16 128

/* One deterministic split reduction: out[n1*ldo + n2] (+)= alpha * sum_{s<S} slab[s*ss + n1*rs + n2], n1 < N1, n2 < N2.
 * Producers (nu_wgrad_enqueue, nu_skinny_bwd_enqueue, nu_colsum_enqueue) append these to a caller-owned HOST array;
 * nu_slab_reduce_batched runs up to NU_REDUCE_MAX of them per launch.  G and blk_begin are filled by the library. */
#define NU_REDUCE_MAX 48
typedef struct NuReduceDesc {
    const float* slab; float* out;
    long long ss;
    int S, N1, N2, rs, ldo, accumulate, G, blk_begin;
    float alpha; int pad_;
} NuReduceDesc;
int nu_reduce_desc_size(void);
int nu_slab_reduce_batched(const NuReduceDesc* descs_host, int n, hipStream_t stream);

int nu_gemm_nt_size(void);      /* sizeof(NuGemmNT) / sizeof(NuGemmTN) as compiled: binding-side ABI check */
int nu_gemm_tn_size(void);
int nu_gemm_nt_ex(const NuGemmNT* g, hipStream_t stream);
/* n independent problems (HOST array; own M, N, K, pointers and epilogue arguments) in as few persistent launches as possible: runs
 * of problems with one epilogue kind share ONE tile list -- the four light predictors of AppShadingNetwork.forward (field.py:636-682),
 * the IoR / thickness pair of stage 2 (field.py:1046-1087).  Results are bit-identical to n calls of nu_gemm_nt_ex. */
#define NU_NT_BATCH_MAX 8
int nu_gemm_nt_batch(const NuGemmNT* problems_host, int n, hipStream_t stream);
long long nu_wgrad_workspace_bytes(int N1, int N2, int S, int groups);
/* the split (number of deterministic partial slabs) the library's own sequencing uses for a weight gradient of this shape */
int nu_wgrad_pick_split(int P, int N1, int N2, int groups, int prec);
/* deferred weight gradient: split GEMM now, reductions appended to descs[*ndesc...] (capacity cap); the workspace must
 * stay untouched until nu_slab_reduce_batched(descs, *ndesc) has been enqueued */
int nu_wgrad_enqueue(const NuGemmTN* g, float* dW, int ldw, long long sW, float* db, long long sDb, void* workspace,
                     long long workspace_bytes, NuReduceDesc* descs, int* ndesc, int cap, hipStream_t stream);
/* dW (ld = ldw, group stride sW) and optionally db[N1] = column sums of A0 (group stride sDb) */
int nu_wgrad(const NuGemmTN* g, float* dW, int ldw, long long sW, float* db, long long sDb, void* workspace,
             long long workspace_bytes, hipStream_t stream);
/* flat-argument variants (tests / stand-alone use) */
int nu_gemm_nt(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc, float* C2,
               int ldc2, const float* bias, const float* H, int ldh, const float* D, int ldd, const float* Cadd,
               int ldadd, int zero_to, float alpha, int epi, hipStream_t stream);
long long nu_gemm_tn_workspace_bytes(int N1, int N2, int S);
int nu_gemm_tn(const float* A0, int lda0, const float* B0, int ldb0, const float* A1, int lda1, const float* B1,
               int ldb1, int P, int N1, int N2, float* C, int ldc, float* bias_out, int S, void* workspace,
               long long workspace_bytes, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Weight packing: nn.utils.weight_norm (W = v * g / ||v||_row, field.py:121-122, :386-393) folded into padded,
 * optionally column-permuted and transposed operand buffers; and its chain rule back to (weight_v, weight_g).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct NuPackDesc {
    const float* v; const float* g; const int* colmap;
    float* Wp; float* WpT; const float* dWp;
    long long dv_off, dg_off;                 /* offsets (floats) into the flat gradient buffer */
    const float* bias; float* bias_p;
    float scale;
    int N, K, Kp, ldT, ldd, row_begin, col_off;
    void* Wp16; void* WpT16;                  /* optional bf16 copies of Wp / WpT (same shapes, same leading dimensions) */
    int planes, pad_;                         /* 1 (or 0): the copies are Wp / WpT rounded to bf16.  3: the exact hi / mid / lo split in the
                                                 layout of NuGemmNT.B6 (the bf16x6 mode's weight tables): Wp16 / WpT16 then point at the START of
                                                 the twin of the whole table, and the layer's place in it is given below */
    int w6_row0, w6_ld, t6_row0, t6_col0, t6_ld, pad2_;   /* planes == 3: first row (and leading dimension) of this layer in the Wp table;
                                                 first row / first column / leading dimension in the WpT table */
} NuPackDesc;
int nu_pack_desc_size(void);
int nu_pack_layers(const void* descs_dev, int ndesc, int total_rows, hipStream_t stream);
int nu_unpack_grads(const void* descs_dev, int ndesc, int total_rows, float* flat_grads, hipStream_t stream);
/* The same for the rows [row0, row0 + nrows) of the table only (rows are numbered through all layers in descriptor order): what one
 * network-level op of stage 2 owns -- its weight-norm gradients land in `flat_grads`, every other slot stays untouched. */
int nu_unpack_grads_range(const void* descs, int ndesc, int row0, int nrows, float* flat_grads, hipStream_t stream);

/* 1..6-wide output heads (field.py:393 final Linear of make_predictor; :260-261 alpha_linear / rgb_linear) */
int nu_skinny_fwd(const float* H, int ldh, int P, int K, const float* Ws, int ldw, const float* b, int NO, float* out,
                  int ldo, hipStream_t stream);
long long nu_skinny_bwd_workspace_bytes(int K, int NO);
int nu_skinny_bwd(const float* dy, int ldy, const float* H, int ldh, int P, int K, const float* Ws, int ldw, int NO,
                  float* dH, int lddh, int relu_mask, int accumulate, float* dWs, int lddw, float* db, void* workspace,
                  long long workspace_bytes, hipStream_t stream);
int nu_skinny_bwd_enqueue(const float* dy, int ldy, const float* H, int ldh, int P, int K, const float* Ws, int ldw,
                          int NO, float* dH, int lddh, int relu_mask, int accumulate, float* dWs, int lddw, float* db,
                          void* workspace, long long workspace_bytes, NuReduceDesc* descs, int* ndesc, int cap,
                          hipStream_t stream);
/* bf16-storage forms (NuOpCtx.h16): the hidden rows H -- and the dH written -- are __bf16; head weights, dy, outputs and every
 * reduction stay fp32 */
int nu_skinny_fwd_h16(const void* H, int ldh, int P, int K, const float* Ws, int ldw, const float* b, int NO, float* out, int ldo,
                      hipStream_t stream);
int nu_skinny_bwd_enqueue_h16(const float* dy, int ldy, const void* H, int ldh, int P, int K, const float* Ws, int ldw, int NO,
                              void* dH, int lddh, int relu_mask, int accumulate, float* dWs, int lddw, float* db, void* workspace,
                              long long workspace_bytes, NuReduceDesc* descs, int* ndesc, int cap, hipStream_t stream);
long long nu_colsum_workspace_bytes(int ncols);
int nu_colsum_enqueue(const float* A, int lda, int P, int ncols, float* out, int accumulate, void* workspace,
                      long long workspace_bytes, NuReduceDesc* descs, int* ndesc, int cap, hipStream_t stream);
int nu_colsum(const float* A, int lda, int P, int ncols, float* out, int accumulate, void* workspace,
              long long workspace_bytes, hipStream_t stream);
/* D = w[col] * softplus'(H): seed of the reverse sweep of SDFNetwork.gradient (field.py:158-170) */
int nu_rowscale_dsp(const float* H, int ldh, int P, int K, const float* w, float* D, int ldd, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Encodings
 * --------------------------------------------------------------------------------------------------------- */
/* get_embedder(6,3) of the SDF input (field.py:14-61, :133-136): E[P,64], skip-concat slot of U4[P,256], x-slot of YX */
int nu_sdf_embed(const float* pt, int pt_ld, int P, float* E, float* U4, float* YX, hipStream_t stream);
/* Generic get_embedder(n_freq <= 10, 3) (network/field.py:14-61) for the widths the SDF path does not use -- the 8-frequency position
 * code and the 2-frequency refraction codes of AppShadingNetwork_SpecInner (field.py:1351-1354): out [P, ldo >= 3 + 6 n_freq];
 * _bwd: g [P, ldg] -> dx [P,3]. */
int nu_embed_n_fwd(const float* x, int P, int n_freq, float* out, int ldo, hipStream_t stream);
int nu_embed_n_bwd(const float* x, const float* g, int ldg, int P, int n_freq, float* dx, hipStream_t stream);
/* n = J_emb^T (G0 + Gs): last step of d sdf / d x (field.py:163-170);  q0 = J_emb nbar: first step of its adjoint */
int nu_embed_jt(const float* E, const float* G0, int ldg0, const float* Gs, int ldgs, int P, float* n, hipStream_t stream);
int nu_embed_j(const float* E, const float* nbar, int P, float* Q0, float* Q4, hipStream_t stream);
/* integrated directional encoding, 72-d (utils/ref_utils.py:84-114) and its gradient */
int nu_ide(const float* dirs, const float* kappa_inv, int P, float* out, int ldo, hipStream_t stream);
int nu_ide_bwd(const float* dirs, const float* kappa_inv, const float* gout, int ldg, int P, float* ddirs,
               float* dkappa, hipStream_t stream);
/* inputs of the four light predictors (field.py:636-682, :686-689, :717) and the backward to normals / roughness */
/* sphere = shader_config.sphere_direction (144-d outer_light input, field.py:641-646, :675-679);
 * refrac_dim = 3 + 6 * refrac_freq (field.py:590-591) */
int nu_shade_encode_fwd(const float* nrm, const float* pt, int pt_ld, const float* E, const float* Mraw, int ldm, int P,
                        int sphere, int ld_ol, int refrac_dim, int ld_rl, float* OLin, float* ILin, float* IWin,
                        float* RLin, float* SD, hipStream_t stream);
int nu_shade_encode_bwd(const float* nrm, const float* pt, int pt_ld, const float* SD, const float* dOLin, int ld_ol,
                        int sphere, const float* dILin, const float* dNoV, int P, float* dn, float* dMraw, int ldm,
                        hipStream_t stream);
/* per-ray mirror query of colour_spec (renderer_zerothick.py:780-781; network/renderer.py:710-725) */
int nu_spec_encode(const float* dirs, const float* x, int R, int sphere, float* out, int ldo, hipStream_t stream);
/* NeRF++ inputs (x/|x|, 1/|x|) L=10 and view -d L=4 (renderer_zerothick.py:687-690; field.py:266-269) */
int nu_nerf_embed(const float* pt, int pt_ld, int P, float* E4, float* U5, float* V, hipStream_t stream);
/* input gradients (stage 2: sample positions depend on the learned IoR, renderer_zerothick.py:1642-1684): d L / d x of the
 * SDF network incl. the second-order term of its normal, and d L / d x, d L / d dir of the NeRF++ inputs */
int nu_embed_jt2(const float* E, const float* dE, int lde, const float* dS, int lds, const float* G0, int ldg0,
                 const float* Gs, int ldgs, const float* nbar, int P, float* dx, int accumulate, hipStream_t stream);
int nu_nerf_embed_bwd(const float* pt, int pt_ld, const float* E4, const float* V, const float* gE, int lde,
                      const float* gS, int lds, const float* gV, int ldv, int P, float* dx, float* ddir,
                      hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * render_core (renderer_zerothick.py:725-820)
 * --------------------------------------------------------------------------------------------------------- */
/* mid-points, section lengths, inner mask |x| <= 1 and its compaction (:730-736, boolean-mask gathers :748-767) */
int nu_partition_count(const float* o, const float* d, const float* z, int R, int S, int* cnt_in, int* off_in,
                       int* totals, hipStream_t stream);
int nu_partition_write(const float* o, const float* d, const float* z, int R, int S, const int* off_in, float* pt_in,
                       int* idx_in, float* pt_out, int* idx_out, unsigned char* inner_rm, hipStream_t stream);
/* compute_sdf_alpha (:657-685) + eikonal term (:769) */
/* also writes max(n.d, 0) into colour channel 3 (the loss_normal integrand of network/renderer.py:693-705) */
int nu_neus_alpha_fwd(const float* YX, int ldy, const float* nrm, const float* pt, const int* idx, int P,
                      const float* variance, float anneal, float* alpha_rm, float* gerr, float* color_rm,
                      hipStream_t stream);
/* dvar (optional): d L / d variance, a sum over all points -- per-block partials go to `workspace`
 * (nu_neus_alpha_bwd_workspace_bytes(P)) and ONE deferred reduction problem is appended to descs (finished by
 * nu_slab_reduce_batched, like every weight gradient): deterministic, and no state inside the library */
long long nu_neus_alpha_bwd_workspace_bytes(int P);
int nu_neus_alpha_bwd(const float* YX, int ldy, const float* nrm, const float* pt, const int* idx, int P,
                      const float* variance, float anneal, const float* dalpha_rm, const float* dgerr,
                      const float* dn_shade, const float* dcolor_rm, float* dYX, int lddy, float* nbar, float* dvar,
                      void* workspace, long long workspace_bytes, NuReduceDesc* descs, int* ndesc, int cap, hipStream_t stream);
/* density_activation + colour activation of compute_density_alpha (:515-516, :691-692) */
int nu_nerf_act_fwd(const float* sigma, int lds, const float* rgb, int ldr, const float* pt, const int* idx, int P,
                    float* alpha_rm, float* color_rm, hipStream_t stream);
int nu_nerf_act_bwd(const float* sigma, int lds, const float* rgb, int ldr, const float* pt, const int* idx, int P,
                    const float* dalpha_rm, const float* dcolor_rm, float* dsigma, int ldds, float* drgb, int lddr,
                    hipStream_t stream);
/* BRDF mix, split-sum LUT lookup (nvdiffrast dr.texture, field.py:719-722), Fresnel, sRGB (field.py:698-740) */
int nu_shade_combine_fwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo,
                         const float* RLo, const float* SD, const float* lut, const int* idx, int P, float exp_max,
                         float* color_rm, float* aux, hipStream_t stream);
int nu_shade_combine_bwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo,
                         const float* RLo, const float* SD, const float* lut, const int* idx, int P, float exp_max,
                         const float* dcolor_rm, float* dMraw, float* dOLo, float* dILo, float* dIWo, float* dRLo,
                         float* dNoV, hipStream_t stream);
/* front-to-back composite incl. the background-only composite (:773-779); colour is [R*S,4] = rgb + one auxiliary
 * channel whose composite goes to aux_sum[R] (normal-orientation loss, network/renderer.py:705) */
int nu_composite_fwd(const float* alpha, const float* color, const unsigned char* inner, int R, int S, float* weights,
                     float* rgb, float* acc, float* rgb_bg, float* aux_sum, hipStream_t stream);
int nu_composite_bwd(const float* alpha, const float* color, const unsigned char* inner, int R, int S,
                     const float* drgb, const float* dacc, const float* drgb_bg, const float* daux, float* dalpha,
                     float* dcolor, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Hierarchical sampler (renderer_zerothick.py:572-612, :525-570; field.py:468-498, :501-554)
 * --------------------------------------------------------------------------------------------------------- */
int nu_sample_coarse(const float* o, const float* d, const float* near, const float* far, const float* lin,
                     const float* lower, const float* upper, const float* u1, const float* u2, int R, int Nc, int Nbg,
                     float* zc, float* zbg, float* X, hipStream_t stream);
int nu_upsample(const float* o, const float* d, const float* z, const float* sdf, int R, int sn, const float* variance,
                float inv_s_cap, int use_variance, const float* uvals, int n_new, float* z_new, float* Xn,
                hipStream_t stream);
int nu_probe_weights(const float* o, const float* d, const float* z, const float* sdf, int R, int sn,
                     const float* variance, const float* uvals, int n_new, float* z_new, float* Xn, float* wsum,
                     hipStream_t stream);
int nu_merge_sorted(const float* z, const float* sdf, int sn, const float* zn, const float* sdfn, int nn, int R,
                    float* zo, float* sdfo, hipStream_t stream);
int nu_concat_cols(const float* A, int a, const float* B, int b, int R, float* out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Mesh closest-hit tracing (stage 2): replaces the OptiX GAS build (network/tracing_optix.py:20-23, :142-146), the
 * launch/IO wrapper (:74-117, :154-158) and the device programs of cuda/triangle.cu:48-99.
 *   rays [N,6] = (origin, direction); hit[N] = 1.0/0.0; idx[N] = face id of the closest hit with tmin < t < tmax, or
 *   10000000 on a miss; ties in t -> lowest face id.  `bvh` is a caller-owned buffer of nu_lbvh_bytes(n_faces) bytes.
 * --------------------------------------------------------------------------------------------------------- */
long long nu_lbvh_bytes(int n_faces);
int nu_lbvh_build(const float* V, int n_verts, const int* F, int n_faces, void* bvh, long long bvh_bytes,
                  hipStream_t stream);
int nu_lbvh_trace(const void* bvh, int n_faces, const float* rays, int N, float tmin, float tmax, float* hit, int* idx,
                  float* t_out, hipStream_t stream);
/* O(N*F) sweep with the same ray/triangle test (cross-check, tiny meshes) */
int nu_brute_trace(const float* V, const int* F, int n_faces, const float* rays, int N, float tmin, float tmax,
                   float* hit, int* idx, float* t_out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Network-level entry points: one call sequences every kernel launch of a network pass from C++.
 *   nu_sdf_mlp_{fwd,normal,bwd}      SDFNetwork.forward / .gradient and their (double) backward   field.py:133-170
 *   nu_nerfpp_mlp_{fwd,bwd}          NeRFNetwork.forward / backward                               field.py:265-289
 *   nu_shading_stack_{fwd,bwd}       AppShadingNetwork.forward / backward                         field.py:684-777, :636-682
 * Buffers are the caller's (device pointers, layouts in DESIGN.md "Data layout in HBM"); NuLin holds one packed layer as
 * nu_pack_layers leaves it.  Split reductions (weight gradients, skinny heads, column sums) go to the arena of NuOpCtx and
 * are finished by nu_ctx_flush (or implicitly when the arena / descriptor list is full).  Bias gradients are written at
 * NuOpCtx.flat + NuLin.db_off (the flat gradient buffer nu_unpack_grads reads).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct NuLin {
    const float* Wp; const float* WpT; float* dWp; const float* bias;
    long long db_off;
    int N, K, Kp, ldT, ldd, pad_;
    const void* Wp16; const void* WpT16;  /* bf16 copies of Wp / WpT (NuPackDesc.Wp16 / WpT16); used when NuOpCtx.h16 */
} NuLin;

/* NuOpCtx.h16 != 0 (with prec == 1): bf16 STORAGE.  Every NT GEMM reads the bf16 weight tables and the hidden activations
 * that only GEMMs touch live in HBM as bf16; the caller allocates exactly these buffers as __bf16 (same shapes and leading
 * dimensions, in elements), everything else stays fp32:
 *   SDF      H[1..3], H[5..7] (and H[8] in the no-gradient form, want_feat == 0);  D[0..2], D[4..6];  Q[1..3], Q[5..7];
 *            C[l] / Aux[l] for l in {0,1,2,4,5,6}
 *   NeRF++   H[1..4], H[6..8];  dA[1..4], dA[6..8];  dH8a
 *   shading  M[0..2], dM[0..2];  hidden [0..2], tmp[0], tmp[1] and dH3 of every light predictor
 * Optional per-launch timing: when `ev` is set, every GEMM launch is bracketed by hipEventRecord on ev[nev], ev[nev + 1]
 * and described in ev_meta[nev / 2] = {kind (0 NT, 1 TN), algorithmic flops, algorithmic bytes} until ev_cap is reached. */
/* One queued weight gradient of a backward pass (nu_wgrad_defer): the problem, where its result goes, and its algorithmic cost
 * (roofline accounting of the launch that ends up carrying it). */
#define NU_WGRAD_QUEUE_MAX 32
typedef struct NuWgradItem {
    NuGemmTN g;
    float* dW; int ldw, pad_; long long sW;
    float* db; long long sDb;
    double flops, bytes;
} NuWgradItem;
int nu_wgrad_item_size(void);

typedef struct NuOpCtx {
    int prec, h16;                        /* NuGemmNT.bf16 arithmetic mode of every GEMM; bf16 storage */
    float* flat;                          /* flat gradient buffer of the current backward */
    float* arena; long long arena_floats; long long arena_off;
    NuReduceDesc* descs; int ndesc, cap;  /* HOST array of deferred reductions */
    void** ev; double* ev_meta; int nev, ev_cap;      /* HOST arrays: hipEvent_t handles, 3 doubles per launch */
    int forked, pad_;                     /* != 0 while ops of this context are in flight on MORE THAN ONE stream: a forced
                                             mid-step flush would reduce slabs another stream still writes and hand their space
                                             out again, so the entries return NU_ERR_WORKSPACE instead (fail closed) */
    NuWgradItem* pend; int npend, pend_cap;   /* HOST array: the weight gradients of the current pass, queued by nu_wgrad_defer and
                                             launched together by nu_wgrad_flush (NULL: every weight gradient launches at once) */
} NuOpCtx;
int nu_op_ctx_size(void);
/* launches the queued weight gradients, then the batched reductions of everything enqueued so far; resets the arena */
int nu_ctx_flush(NuOpCtx* ctx, hipStream_t stream);
/* the reductions only (queued weight gradients stay queued): frees the arena in the middle of a pass */
int nu_ctx_reduce(NuOpCtx* ctx, hipStream_t stream);
/* The weight gradients of a backward pass are independent of each other: a pass queues them and ONE launch per tile class carries
 * the whole queue (splits chosen for the queue as a whole; see csrc/gemm_tn.hip).  Operands must stay valid and unmodified until
 * the flush; every network-level entry below flushes before it returns.  flops / bytes: algorithmic cost for the event record. */
int nu_wgrad_defer(NuOpCtx* ctx, const NuGemmTN* g, float* dW, int ldw, long long sW, float* db, long long sDb, double flops,
                   double bytes, hipStream_t stream);
int nu_wgrad_flush(NuOpCtx* ctx, hipStream_t stream);

typedef struct NuSdfNet { NuLin lin[9]; } NuSdfNet;
typedef struct NuSdfBufs {
    int P, pad_;
    float *E, *U4, *YX, *sdf;             /* E [P,64]; U4 [P,256]; YX [P,288] (want_feat) or sdf [P] */
    float* H[9];                          /* H[1..8] [P,256] post-softplus, H[4] == U4 (inference: two ping-pong buffers) */
    float* D[8]; float* G0; float* n;     /* reverse sweep: delta_l [P,256], G0 [P,64], n [P,3] */
    float* Q[9]; float* C[8];             /* second-order: Q[0] [P,64], Q[1..8] [P,256] (Q[4] doubles as the skip slot), C_l [P,256] */
    float* Aux[8];                        /* abar_l [P,256] of a first-order-only backward */
    float* dE0;                           /* [P,64] (input gradient) */
} NuSdfBufs;
int nu_sdf_net_size(void);
int nu_sdf_bufs_size(void);
int nu_sdf_mlp_fwd(NuOpCtx* ctx, const NuSdfNet* net, const float* X, int x_ld, NuSdfBufs* bufs, int want_feat, hipStream_t stream);
/* The no-gradient forward as ONE fully-fused kernel (csrc/fused_sdf.hip): sdf[P] = SDFNetwork(x)[..., 0] for X [P, x_ld] (x = the
 * first three floats of a row), exact fp32, nothing kept -- embedding, the eight softplus layers and the sdf head with the
 * activations of a 32- or 64-point tile held in LDS and the weights streamed through it.  Bit-identical to nu_sdf_mlp_fwd(...,
 * want_feat = 0).  Serves the hierarchical sampler (renderer_zerothick.py:525-612: 64 + 3 x 16 queries per ray), the occlusion
 * probe (field.py:501-554), extract_fields (field.py:1286-1307) and the stage-2 inner up-sampler (renderer_zerothick.py:1742-1760). */
int nu_sdf_fused_fwd(const NuSdfNet* net, const float* X, int x_ld, int P, float* sdf, hipStream_t stream);
/* the same in the bf16-STORAGE arithmetic (NuOpCtx.h16: bf16 weight tables NuLin.Wp16, activations rounded to bf16 between layers,
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation): bit-identical to nu_sdf_mlp_fwd(..., want_feat = 0) under h16 */
int nu_sdf_fused16_fwd(const NuSdfNet* net, const float* X, int x_ld, int P, float* sdf, hipStream_t stream);
int nu_sdf_mlp_normal(NuOpCtx* ctx, const NuSdfNet* net, NuSdfBufs* bufs, hipStream_t stream);
int nu_sdf_mlp_bwd(NuOpCtx* ctx, const NuSdfNet* net, NuSdfBufs* bufs, const float* dYX, const float* nbar, float* dx,
                   hipStream_t stream);

typedef struct NuNerfNet { NuLin pts[8]; NuLin feat, alpha, view, rgb; } NuNerfNet;
typedef struct NuNerfBufs {
    int P, pad_;
    float* H[9];                          /* H[0] = E4 [P,96]; H[i] = input of layer i ([P,256]; H[5] = U5 [P,352]); H[8] last hidden */
    unsigned long long* mask[9];          /* ReLU sign bits of H[1..8] */
    float *V, *HV, *sig, *rgb;            /* V [P,288] = feature | view embedding; HV [P,128]; raw heads sig [P], rgb [P,4] */
    float *dHV, *dF, *dH8a;               /* backward: [P,128], [P,256|288], [P,256] */
    float* dA[9];                         /* dA[i] = d pre-activation of layer i-1's output, i = 1..8 ([P,256]; dA[5] [P,352] with dx) */
    float *dE4, *dx, *ddir;               /* input gradients (stage 2): [P,96], [P,3], [P,3]; dx == NULL: parameters only */
} NuNerfBufs;
int nu_nerf_net_size(void);
int nu_nerf_bufs_size(void);
int nu_nerfpp_mlp_fwd(NuOpCtx* ctx, const NuNerfNet* net, const float* pt, int pt_ld, NuNerfBufs* bufs, hipStream_t stream);
int nu_nerfpp_mlp_bwd(NuOpCtx* ctx, const NuNerfNet* net, const float* pt, int pt_ld, NuNerfBufs* bufs, const float* dsig,
                      const float* drgb, hipStream_t stream);

typedef struct NuShadeNet {
    const float *WpM0, *WpTM0, *bM0; float* dWpM0;                  /* materials layer 0, 4 predictors side by side (N = 1024) */
    const float* WpM[3]; const float* WpTM[3]; const float* bM[3]; float* dWpM[3];      /* [1], [2]: grouped x4 */
    long long dbM_off[3];
    const float *Ws6, *b6; float* dWs6; long long db6_off;          /* block-diagonal 6-wide head */
    NuLin outer_light[4], inner_light[4], inner_weight[4], refrac_light[4];
    const float* lut;
    float exp_max; int sphere, ld_ol, refrac_dim, ld_rl, pad_;
    const void *WpM0_16, *WpTM0_16; const void* WpM16[3]; const void* WpTM16[3];     /* bf16 copies (NuOpCtx.h16) */
} NuShadeNet;
typedef struct NuShadeBufs {
    int P, R;                             /* inner points; per-ray mirror queries riding along the outer_light batch */
    const float *extra_dirs, *extra_pts;
    float* M[3]; unsigned long long* maskM[3]; float* Mraw;         /* materials hidden [P,1024] x3, raw heads [P,8] */
    float *OLin, *ILin, *IWin, *RLin, *SD;                          /* predictor inputs, shading directions [P,8] */
    float* OLh[3]; float* ILh[3]; float* IWh[3]; float* RLh[3];     /* hidden activations [rows,256] */
    unsigned long long* maskOL[3]; unsigned long long* maskIL[3]; unsigned long long* maskIW[3]; unsigned long long* maskRL[3];
    float *OLo, *ILo, *IWo, *RLo, *aux;                             /* raw heads, aux [P,4] */
    float *dMraw, *dOLo, *dILo, *dIWo, *dRLo, *dNoV;                /* backward of the combine */
    float* dH3[4]; float* tmpOL[2]; float* tmpIL[2]; float* tmpIW[2]; float* tmpRL[2];
    float *dOLin, *dILin, *dn; float* dM[3]; float* dYX;
} NuShadeBufs;
int nu_shade_net_size(void);
int nu_shade_bufs_size(void);
int nu_shading_stack_fwd(NuOpCtx* ctx, const NuShadeNet* net, NuShadeBufs* bufs, const float* YX, const float* E, const float* nrm,
                         const float* pt, const int* idx, float* color_rm, hipStream_t stream);
/* stage 0: combine backward only (fills dMraw, dOLo, dILo, dIWo, dRLo, dNoV); stage 1: everything after it */
int nu_shading_stack_bwd(NuOpCtx* ctx, const NuShadeNet* net, NuShadeBufs* bufs, const float* YX, const float* nrm, const float* pt,
                         const int* idx, const float* dcolor_rm, int stage, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused loss assembly: from the renderer's per-ray outputs to the scalar the trainer back-propagates, for
 * the loss set of the shipped stage-1 configs -- white background + clamp (renderer_zerothick.py:783-787), charbonier RGB
 * loss (:501-513), colour_spec activation (:780-781), NeRFRenderLoss / EikonalLoss / OuterRegLoss / NormalOrientationLoss
 * (network/loss.py:26-48, :194-213) and the trainer's sum of means (train/trainer_zero.py:153-161).
 *   rgb, rgb_bg, spec_raw, gt [R,3]; acc, nrm_sum (optional) [R]; gerr [P] (P = 0: no inner point); cand (optional) u8[R]:
 *   rays that take part in the outer regulariser.  Weights of inactive terms are passed as 0.
 *   terms[6] = {loss_rgb, loss_eikonal, loss_outer_reg, loss_normal, their sum, candidate count}; per-ray outputs
 *   ray_rgb / color_spec [R,3], loss_rgb [R].  nu_loss_bwd reads the upstream gradient from device memory.
 *   point_weight (optional device scalar, NULL = 1): factor on the eikonal mean and its gradient -- under data parallelism
 *   this rank's share n_local * world / sum_r n_r of the mean over all ranks' inner points, so the N > 1 step
 *   runs the same fused assembly as the N = 1 step.
 * --------------------------------------------------------------------------------------------------------- */
long long nu_loss_workspace_bytes(int R, int P);
int nu_loss_fwd(const float* rgb, const float* acc, const float* rgb_bg, const float* spec_raw, const float* gerr,
                const float* nrm_sum, const float* gt, const unsigned char* cand, int R, int P, int white_bg, float exp_max,
                float w_eik, float w_reg, float w_nrm, float* ray_rgb, float* color_spec, float* loss_rgb, float* terms,
                const float* point_weight, void* workspace, long long workspace_bytes, hipStream_t stream);
int nu_loss_bwd(const float* rgb, const float* acc, const float* rgb_bg, const float* spec_raw, const float* gt,
                const unsigned char* cand, const float* ray_rgb, const float* color_spec, const float* loss_rgb,
                const float* terms, const float* upstream, const float* point_weight, int R, int P, int white_bg, float exp_max,
                float w_eik, float w_reg, float w_nrm, float* d_rgb, float* d_acc, float* d_rgb_bg, float* d_spec_raw, float* d_gerr, float* d_nrm,
                hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stage 2 (zero-thickness Stage2Renderer, network/renderer_zerothick.py:1571-2011): kernels of the renderer's own logic.
 * A segment is a set of N rays with S1 nodes x_{n,j} = start_n + v_n * z_{n,j}; its S = S1 - 1 samples sit at the nodes
 * j < S, section length |x_{j+1} - x_j| (the last one repeats its predecessor).
 *   nu_s2_seg_count / nu_s2_seg_write   outer samples (|x| > 1) compacted in (ray, sample) order into 32-byte point records
 *                                       (x, dist, dir) for nu_nerfpp_mlp_fwd; idx = ray-major scatter target (idx_base + n S + j),
 *                                       pos_rm = compact index per (ray, sample) or -1          (:1835-1870)
 *   nu_s2_ddist                         d alpha / d dist of compute_density_alpha               (:1531-1540)
 *   nu_s2_seg_bwd                       cotangents of the point records -> d start, d v, d dirs
 *   nu_s2_composite_fwd / _bwd          linear-RGB composite with the running transmittance     (:1976-1990); colour [N,S,4] sRGB
 *   nu_s2_refract_fwd / _bwd            Snell refraction / total internal reflection per hit ray (:1642-1684)
 * --------------------------------------------------------------------------------------------------------- */
int nu_s2_seg_count(const float* start, const float* v, const float* z, int N, int S1, int* cnt, int* off, int* total, hipStream_t stream);
int nu_s2_seg_write(const float* start, const float* v, const float* z, const float* dirs, int N, int S1, const int* off, int pt_base,
                    int idx_base, float* pt, int* idx, int* pos_rm, hipStream_t stream);
int nu_s2_ddist(const float* sigma, const float* pt, const int* idx, int P, const float* dalpha_rm, float* ddist, hipStream_t stream);
int nu_s2_seg_bwd(const float* start, const float* v, const float* z, int N, int S1, int idx_base, const int* pos_rm, const float* dx,
                  const float* ddist, const float* ddir, float* dstart, float* dv, float* ddirs, hipStream_t stream);
int nu_s2_composite_fwd(const float* alpha, const float* color, const float* Tin, int N, int S, float* out, float* Tout, hipStream_t stream);
int nu_s2_composite_bwd(const float* alpha, const float* color, const float* Tin, int N, int S, const float* dout, const float* dTout,
                        float* dalpha, float* dcolor, float* dTin, hipStream_t stream);
int nu_s2_refract_fwd(const float* d, const float* nrm, const float* ior, const float* point, int M, int outside, unsigned char* flag,
                      float* eta, float* nd, float* ns, hipStream_t stream);
int nu_s2_refract_bwd(const float* d, const float* nrm, const float* ior, int M, int outside, const float* g_nd, const float* g_ns,
                      const float* g_eta, float* dd, float* dn, float* dior, float* dpoint, hipStream_t stream);
/*   nu_s2_shade_encode_fwd / _bwd       inputs of the shading stacks for explicit points / normals / view directions [P,3] (stage 2:
 *                                       AppShadingNetwork, _S2, _SpecInner; field.py:636-682, :828-907, :1399-1445): n^, v^, NoV,
 *                                       r = 2 NoV n^ - v^, rho = sigmoid(Mraw[:,1]);  OLin [3P, ld_ol] IDE rows (+ sphere points when
 *                                       `sphere`), ILin [2P,128] = [pe(x) | IDE], IWin [P,96] = [pe(x) | embed(r,6)], RLin [P, ld_rl]
 *                                       = [embed(x, rf) | embed(v^, rf)] (refrac_freq < 0: none), SD [P,12] = n^, NoV, 1/|n|, rho,
 *                                       1/|v|, 0, r, 0; pe(x) has pos_freq frequencies (6 or 8).  _bwd: cotangents of OLin / ILin /
 *                                       RLin / NoV (each may be NULL) -> d x, d n, d v [P,3] and d Mraw[:,1] [P] -- the stage-1 pair
 *                                       nu_shade_encode_* returns d n and d rho only. */
int nu_s2_shade_encode_fwd(const float* x, const float* nrm, const float* view, const float* Mraw, int ldm, int P, int sphere, int pos_freq,
                           int ld_ol, int refrac_freq, int ld_rl, float* OLin, float* ILin, float* IWin, float* RLin, float* SD,
                           hipStream_t stream);
int nu_s2_shade_encode_bwd(const float* x, const float* nrm, const float* view, const float* SD, int P, int sphere, int pos_freq, int ld_ol,
                           int refrac_freq, int ld_rl, const float* dOLin, const float* dILin, const float* dRLin, const float* dNoV,
                           float* dx, float* dn, float* dv, float* drho_raw, hipStream_t stream);
/*   nu_s2_shell_fwd / _bwd              thin-shell refraction of the NON-zero-thickness stage-2 model (network/renderer.py:1692-2032,
 *                                       Stage2Renderer.ray_trace): per hit ray the two refractions through a shell of learned thickness
 *                                       whose faces are concentric spheres of the local curvature radius.  In: d, raw interpolated
 *                                       normal, hit point [M,3], raw IoR / thickness network outputs and Gaussian curvature [M];
 *                                       `inside`: the ray leaves the object.  Out: refracts / tir_ok flags [M] u8, eta [M] (the first
 *                                       face's ratio), unit normal facing the ray, end point of the incoming segment, next origin, next
 *                                       direction [M,3].  _bwd: cotangents of the four [M,3] outputs (each may be NULL) -> d d, d normal,
 *                                       d point, d ior, d g_k, d thickness (a 12 x 12 forward-mode Jacobian per ray in registers). */
int nu_s2_shell_fwd(const float* d, const float* nraw, const float* p, const float* ior_raw, const float* gk, const float* th_raw, int M,
                    int inside, unsigned char* refracts, unsigned char* tir_ok, float* eta, float* nrm, float* pend, float* ns, float* nd,
                    hipStream_t stream);
int nu_s2_shell_bwd(const float* d, const float* nraw, const float* p, const float* ior_raw, const float* gk, const float* th_raw, int M,
                    int inside, const float* g_nrm, const float* g_pend, const float* g_ns, const float* g_nd, float* g_d, float* g_nraw,
                    float* g_p, float* g_ior, float* g_gk, float* g_th, hipStream_t stream);
/*   nu_s2_hit_fwd / _bwd                differentiable Moeller-Trumbore + vertex-normal interpolation of the rays that hit
 *                                       (Scene.Dintersect, network/DiffRender.py:61-125): face [M] int64 from nu_lbvh_trace, faces [F,3]
 *                                       int64, verts / vnrm [V,3] constants; backward -> d o, d d
 *   nu_s2_far_points / _far_resample    importance pass of the rays that miss the mesh (:1786-1812, no gradient): 192 coarse nodes
 *                                       -> point records; alpha [M,192] -> 64 inverse-CDF samples merged in: zout [M,256] sorted */
int nu_s2_hit_fwd(const float* o, const float* d, const long long* face, const float* verts, const float* vnrm, const long long* faces,
                  int M, float* point, float* nrm, float* t, const float* vcurv /* optional [V]: per-vertex Gaussian curvature */,
                  float* gk /* [M]: its barycentric interpolation (DiffRender.py:116) */, hipStream_t stream);
int nu_s2_hit_bwd(const float* o, const float* d, const long long* face, const float* verts, const float* vnrm, const long long* faces,
                  int M, const float* g_point, const float* g_nrm, const float* g_t, float* g_o, float* g_d, const float* vcurv,
                  const float* g_gk, hipStream_t stream);
int nu_s2_far_points(const float* start, const float* dirs, const float* zo, int M, int S, float* pt, int* idx, hipStream_t stream);
int nu_s2_far_resample(const float* alpha, const float* zo, int M, int S, int n_new, float* zout, hipStream_t stream);
/*   nu_s2_shade_combine_fwd / _bwd      AppShadingNetwork_S2.forward's BRDF mix (field.py:909-1010) on raw head outputs, layouts as
 *                                       nu_shade_combine_*: colour = (diffuse + specular)(1 - T) + F light0 T (x 0 when `internal`),
 *                                       rc [P] = (1 - F) T */
/*   nu_s2_neus_alpha_fwd / _bwd          NeuS alpha of the inner segment on explicit points (renderer_zerothick.py:1897-1915): sdf, unit
 *                                       normal, direction, section length, inv_s [1] on device; backward -> d sdf, d n, d d, d dist and
 *                                       the per-point share of d inv_s */
int nu_s2_neus_alpha_fwd(const float* sdf, const float* nrm, const float* dir, const float* dist, const float* inv_s, float cos_anneal,
                         int P, float* alpha, hipStream_t stream);
int nu_s2_neus_alpha_bwd(const float* sdf, const float* nrm, const float* dir, const float* dist, const float* inv_s, float cos_anneal,
                         int P, const float* g_alpha, float* g_sdf, float* g_nrm, float* g_dir, float* g_dist, float* g_s,
                         hipStream_t stream);
int nu_s2_shade_combine_fwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo, const float* SD,
                            const float* lut, const int* idx, int P, float exp_max, int internal, float* color_rm, float* rc,
                            hipStream_t stream);
int nu_s2_shade_combine_bwd(const float* Mraw, int ldm, const float* OLo, const float* ILo, const float* IWo, const float* SD,
                            const float* lut, const int* idx, int P, float exp_max, int internal, const float* dcolor_rm, const float* d_rc,
                            float* dMraw, float* dOLo, float* dILo, float* dIWo, float* dNoV, hipStream_t stream);


/* ---------------------------------------------------------------------------------------------------------
 * Trainer glue: torch.optim.Adam's update (train/trainer_zero.py:74-85, lr from
 * train/lr_common_manager.py:22-46) over many parameter tensors in one launch per NU_ADAM_MAX tensors.
 * p, g, m (exp_avg), v (exp_avg_sq): contiguous fp32 of n elements; step >= 1 is the count AFTER this update.
 * --------------------------------------------------------------------------------------------------------- */
#define NU_ADAM_MAX 80
typedef struct NuAdamDesc {
    float* p; const float* g; float* m; float* v;
    long long n;
    int blk_begin, pad_;                      /* filled by the library */
} NuAdamDesc;
int nu_adam_desc_size(void);
int nu_adam_step(const NuAdamDesc* descs_host, int n, double lr, double beta1, double beta2, double eps, int step,
                 hipStream_t stream);   /* hyper-parameters in double: 1 - beta and the bias corrections are formed as torch forms them */

#ifdef __cplusplus
}
#endif
#endif /* NU_NERF_H */
