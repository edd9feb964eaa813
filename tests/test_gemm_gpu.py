"""GPU parity of the raw fp32-MFMA GEMM kernels (csrc/gemm.hip) against torch fp32 matmul.

These are floating-point kernels, so the checker here is a plain torch fp32/fp64 reference of the
same op (tolerance 2e-5 relative to the row's |a|.|b| mass: the MFMA is a k-ordered fp32 fma chain).
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _nt(A, B, N, epi, bias=None, H=None, D=None, Cadd=None, zero_to=0, alpha=1.0, ldc=None):
    from nu_nerf_amd import _lib as L
    lib = L.load()
    M, lda = A.shape
    K = B.shape[1]
    ldc = ldc or max(N, zero_to)
    C = torch.full((M, ldc), float("nan"), device=A.device)
    C2 = torch.full((M, ldc), float("nan"), device=A.device) if epi == 5 else None
    rc = lib.nu_gemm_nt(L.ptr(A), lda, L.ptr(B), B.shape[1], M, N, K, L.ptr(C), ldc, L.ptr(C2), ldc,
                        L.ptr(bias), L.ptr(H), H.shape[1] if H is not None else 0,
                        L.ptr(D), D.shape[1] if D is not None else 0,
                        L.ptr(Cadd), Cadd.shape[1] if Cadd is not None else 0,
                        zero_to, ctypes.c_float(alpha), epi, L.stream())
    L.check(rc, "nu_gemm_nt")
    return C, C2


def _packB(W, Kp):
    N, K = W.shape
    Np = (N + 127) // 128 * 128
    B = torch.zeros(Np, Kp, device=W.device)
    B[:N, :K] = W
    return B


@pytest.mark.parametrize("M,N,K", [(1, 1, 32), (127, 3, 64), (300, 217, 256), (1000, 257, 288), (4099, 256, 96)])
def test_nt_bias_epilogues(gpu, M, N, K):
    torch.manual_seed(M * 7 + N)
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(N, K, device=gpu) / K ** 0.5
    b = torch.randn(N, device=gpu)
    B = _packB(W, K)
    ref = (A.double() @ W.double().t() + b.double())
    for epi, fn in [(0, lambda x: x), (1, torch.relu),
                    (2, lambda x: torch.nn.functional.softplus(x, beta=100))]:
        C, _ = _nt(A, B, N, epi, bias=b)
        torch.testing.assert_close(C[:, :N].double(), fn(ref), rtol=2e-5, atol=2e-5)


def test_nt_zero_fill_and_untouched_columns(gpu):
    M, N, K = 200, 217, 64
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(N, K, device=gpu)
    B = _packB(W, K)
    C, _ = _nt(A, B, N, 7, zero_to=240, ldc=256)
    assert torch.all(C[:, N:240] == 0)
    assert torch.isnan(C[:, 240:]).all()  # never written
    torch.testing.assert_close(C[:, :N], A @ W.t(), rtol=2e-5, atol=2e-4)


def test_nt_derivative_epilogues(gpu):
    M, N, K = 777, 256, 256
    torch.manual_seed(3)
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(N, K, device=gpu) / 16
    B = _packB(W, K)
    pre = torch.randn(M, N, device=gpu) * 0.02
    Hsp = torch.nn.functional.softplus(pre, beta=100)
    Hre = torch.relu(pre)
    D = torch.randn(M, N, device=gpu)
    Cadd = torch.randn(M, N, device=gpu)
    v = (A.double() @ W.double().t())
    sp = torch.sigmoid(100 * pre.double())
    C, _ = _nt(A, B, N, 3, H=Hre)
    torch.testing.assert_close(C.double(), v * (pre > 0), rtol=2e-5, atol=2e-5)
    C, _ = _nt(A, B, N, 4, H=Hsp)
    torch.testing.assert_close(C.double(), v * sp, rtol=2e-5, atol=2e-5)
    C, C2 = _nt(A, B, N, 5, H=Hsp, D=D)
    torch.testing.assert_close(C.double(), v * sp, rtol=2e-5, atol=2e-5)
    # D is the stored delta = gbar * sp', so C2 = v * D * beta * (1 - sp')
    torch.testing.assert_close(C2.double(), v * D.double() * 100 * (1 - sp), rtol=2e-4, atol=2e-3)
    C, _ = _nt(A, B, N, 6, H=Hsp, Cadd=Cadd, alpha=0.5)
    torch.testing.assert_close(C.double(), 0.5 * v * sp + Cadd.double(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("P,N1,N2,S", [(1, 1, 1, 1), (1000, 257, 256, 7), (5000, 256, 39, 16), (333, 3, 256, 4), (4000, 256, 256, 8),
                                      (60001, 256, 256, 200), (50000, 256, 512, 100)])          # the last two: 256 x 256-tile kernel
def test_tn_weight_grad(gpu, P, N1, N2, S):
    from nu_nerf_amd import _lib as L
    lib = L.load()
    lib.nu_gemm_tn_workspace_bytes.restype = ctypes.c_longlong
    torch.manual_seed(P)
    lda, ldb = (N1 + 3) // 4 * 4 + 4, (N2 + 3) // 4 * 4
    A0 = torch.randn(P, lda, device=gpu)
    B0 = torch.randn(P, ldb, device=gpu)
    A1 = torch.randn(P, lda, device=gpu)
    B1 = torch.randn(P, ldb, device=gpu)
    wsb = lib.nu_gemm_tn_workspace_bytes(N1, N2, S)
    ws = torch.empty(wsb // 4, device=gpu)
    C = torch.full((N1, N2), float("nan"), device=gpu)
    bo = torch.full((N1,), float("nan"), device=gpu)
    rc = lib.nu_gemm_tn(L.ptr(A0), lda, L.ptr(B0), ldb, L.ptr(A1), lda, L.ptr(B1), ldb, P, N1, N2,
                        L.ptr(C), N2, L.ptr(bo), S, L.ptr(ws), ctypes.c_longlong(wsb), L.stream())
    L.check(rc, "nu_gemm_tn")
    ref = A0[:, :N1].double().t() @ B0[:, :N2].double() + A1[:, :N1].double().t() @ B1[:, :N2].double()
    tol = 3e-5 * P ** 0.5
    torch.testing.assert_close(C.double(), ref, rtol=2e-5, atol=tol)
    torch.testing.assert_close(bo.double(), A0[:, :N1].double().sum(0), rtol=2e-5, atol=tol)


def test_tn_weight_grad_operand_over_4gib(gpu):
    """Operands of 4 GiB or more take the 64-bit-offset build of the TN kernel (32-bit byte offsets elsewhere)."""
    from nu_nerf_amd import _lib as L
    lib = L.load()
    lib.nu_gemm_tn_workspace_bytes.restype = ctypes.c_longlong
    P, ld, N1, N2, S = (1 << 20) + 77, 1024, 96, 130, 64
    X = torch.empty(P, ld, device=gpu)
    X.normal_()
    assert X.numel() * 4 >= 1 << 32
    a_col, b_col = 8, 512                      # two column windows of the same big matrix
    wsb = lib.nu_gemm_tn_workspace_bytes(N1, N2, S)
    ws = torch.empty(wsb // 4, device=gpu)
    C = torch.full((N1, N2), float("nan"), device=gpu)
    bo = torch.full((N1,), float("nan"), device=gpu)
    rc = lib.nu_gemm_tn(ctypes.c_void_p(X.data_ptr() + 4 * a_col), ld, ctypes.c_void_p(X.data_ptr() + 4 * b_col), ld,
                        None, 0, None, 0, P, N1, N2, L.ptr(C), N2, L.ptr(bo), S, L.ptr(ws), ctypes.c_longlong(wsb),
                        L.stream())
    L.check(rc, "nu_gemm_tn")
    ref = torch.zeros(N1, N2, dtype=torch.float64, device=gpu)
    bref = torch.zeros(N1, dtype=torch.float64, device=gpu)
    for lo in range(0, P, 1 << 18):
        a = X[lo:lo + (1 << 18), a_col:a_col + N1].double()
        ref += a.t() @ X[lo:lo + (1 << 18), b_col:b_col + N2].double()
        bref += a.sum(0)
    tol = 3e-5 * P ** 0.5
    torch.testing.assert_close(C.double(), ref, rtol=2e-5, atol=tol)
    torch.testing.assert_close(bo.double(), bref, rtol=2e-5, atol=tol)


def test_relu_sign_bits_replace_the_activation_in_the_backward_epilogues(gpu):
    """BIAS_RELU writes one ballot word per (4-row group, column quad); MUL_DRELU / B_RELU read those instead of H (the H
    pointer handed to them here is poisoned with NaN to prove it).  Also the writer-ungrouped / reader-grouped pairing of the
    batched material predictors, a ragged M, and a plain-column tail (act_cols)."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(5)
    M, K, N = 1000, 96, 512
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(N, K, device=gpu) / K ** 0.5
    bias = torch.randn(N, device=gpu) * 0.3
    Hout = torch.full((M, N), float("nan"), device=gpu)
    nct = N // 128
    mask = torch.zeros(((M + 127) // 128) * nct * 256, dtype=torch.int64, device=gpu)
    g = GemmNT(addr(A), K, addr(_packB(W, K)), K, M, N, K, addr(Hout), N, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 1, 0, mask.data_ptr(), nct, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "fwd")
    ref_h = torch.relu(A.double() @ W.double().t() + bias.double())
    torch.testing.assert_close(Hout.double(), ref_h, rtol=2e-5, atol=2e-5)
    # backward, grouped x2 over the two 256-column halves (the material predictors' pairing), H poisoned
    dA = torch.randn(M, N, device=gpu)
    W2 = torch.randn(2, 256, 256, device=gpu) / 16
    out = torch.full((M, N), float("nan"), device=gpu)
    poison = torch.full((M, N), float("nan"), device=gpu)
    g = GemmNT(addr(dA), N, addr(W2), 256, M, 256, 256, addr(out), N, 0, 0, 0, addr(poison), N, 0, 0, 0, 0, 0, 0, 1.0, 2,
               256, 65536, 256, 0, 0, 256, 0, 0, 3, 0, mask.data_ptr(), nct, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "bwd")
    for z in range(2):
        v = dA[:, 256 * z:256 * (z + 1)].double() @ W2[z].double().t()
        torch.testing.assert_close(out[:, 256 * z:256 * (z + 1)].double(), v * (Hout[:, 256 * z:256 * (z + 1)] > 0), rtol=2e-5, atol=2e-5)
    # B_RELU with Cadd, and MUL_DRELU with plain columns past act_cols (first 256 columns masked, next 84 plain)
    Cadd = torch.randn(M, 256, device=gpu)
    out2 = torch.full((M, 256), float("nan"), device=gpu)
    m256 = torch.zeros(((M + 127) // 128) * 2 * 256, dtype=torch.int64, device=gpu)
    H256 = torch.full((M, 256), float("nan"), device=gpu)
    g = GemmNT(addr(A), K, addr(_packB(W[:256], K)), K, M, 256, K, addr(H256), 256, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 1, 0, m256.data_ptr(), 2, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "fwd256")
    g = GemmNT(addr(dA), N, addr(W2), 256, M, 256, 256, addr(out2), 256, 0, 0, 0, addr(poison), N, 0, 0, addr(Cadd), 256, 0, 0, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 8, 0, m256.data_ptr(), 2, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "b_relu")
    v = dA[:, :256].double() @ W2[0].double().t()
    torch.testing.assert_close(out2.double(), v * (H256 > 0) + Cadd.double(), rtol=2e-5, atol=2e-5)
    W3 = torch.randn(384, 256, device=gpu) / 16
    out3 = torch.full((M, 352), float("nan"), device=gpu)
    g = GemmNT(addr(dA), N, addr(W3), 256, M, 340, 256, addr(out3), 352, 0, 0, 0, addr(poison), N, 0, 0, 0, 0, 352, 256, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 3, 0, m256.data_ptr(), 2, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "plain tail")
    v = dA[:, :256].double() @ W3[:340].double().t()
    want = torch.cat([v[:, :256] * (H256 > 0), v[:, 256:], torch.zeros(M, 12, dtype=torch.float64, device=gpu)], 1)
    torch.testing.assert_close(out3.double(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M", [66048, 65990])
def test_epilogue_fast_path_copies_on_128_row_tiles(gpu, M):
    """The interior slabs of a launch run one of up to three copies of the epilogue's fast path (gemm_epi.h): sign-bit words
    (writer and reader), plain columns past act_cols, auxiliary rows -- here all three in launches big enough for the 128-row tiles
    (>= 512 tiles), with a ragged last row tile in the second case (the slow path next to them).  H is poisoned wherever the sign
    bits must replace it."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(11)
    K = 64
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(256, K, device=gpu) / K ** 0.5
    bias = torch.randn(256, device=gpu) * 0.3
    Hout = torch.full((M, 256), float("nan"), device=gpu)
    mask = torch.zeros(((M + 127) // 128) * 2 * 256, dtype=torch.int64, device=gpu)
    g = GemmNT(addr(A), K, addr(_packB(W, K)), K, M, 256, K, addr(Hout), 256, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 1, 0, mask.data_ptr(), 2, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "writer")
    ref_h = torch.relu(A.double() @ W.double().t() + bias.double())
    torch.testing.assert_close(Hout.double(), ref_h, rtol=2e-5, atol=2e-5)
    pos = Hout > 0
    # reader (sign bits, H poisoned) + plain columns past act_cols in ONE launch: N = 384, the third column tile is plain
    dA = torch.randn(M, 256, device=gpu)
    W3 = torch.randn(384, 256, device=gpu) / 16
    poison = torch.full((M, 256), float("nan"), device=gpu)
    out = torch.full((M, 384), float("nan"), device=gpu)
    g = GemmNT(addr(dA), 256, addr(W3), 256, M, 384, 256, addr(out), 384, 0, 0, 0, addr(poison), 256, 0, 0, 0, 0, 384, 256, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 3, 0, mask.data_ptr(), 2, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "reader + plain")
    v = dA.double() @ W3.double().t()
    torch.testing.assert_close(out.double(), torch.cat([v[:, :256] * pos, v[:, 256:]], 1), rtol=2e-5, atol=2e-5)
    # the same reader WITHOUT sign-bit words: the auxiliary-row copy reads H itself; both give the same bits
    out_h = torch.full((M, 384), float("nan"), device=gpu)
    g = GemmNT(addr(dA), 256, addr(W3), 256, M, 384, 256, addr(out_h), 384, 0, 0, 0, addr(Hout), 256, 0, 0, 0, 0, 384, 256, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "reader from H")
    assert torch.equal(out_h, out)
    # B_RELU on the sign bits with Cadd
    Cadd = torch.randn(M, 256, device=gpu)
    out2 = torch.full((M, 256), float("nan"), device=gpu)
    g = GemmNT(addr(dA), 256, addr(W3), 256, M, 256, 256, addr(out2), 256, 0, 0, 0, addr(poison), 256, 0, 0, addr(Cadd), 256, 0, 0, 1.0, 1,
               0, 0, 0, 0, 0, 0, 0, 0, 8, 0, mask.data_ptr(), 2, 0)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "b_relu")
    torch.testing.assert_close(out2.double(), v[:, :256] * pos + Cadd.double(), rtol=2e-5, atol=2e-5)


def _gemm_struct(A, lda, B, ldb, M, N, K, C, ldc, epi, *, bias=None, H=None, ldh=0, mask=None, nct=0, ct0=0, groups=1, sA=0, sB=0, sC=0,
                 sBias=0, sH=0, zero_to=0, act_cols=0):
    from nu_nerf_amd.engine import GemmNT, addr
    return GemmNT(addr(A), lda, addr(B), ldb, M, N, K, addr(C), ldc, 0, 0, addr(bias), addr(H), ldh, 0, 0, 0, 0, zero_to, act_cols, 1.0, groups,
                  sA, sB, sC, 0, sBias, sH, 0, 0, epi, 0, mask.data_ptr() if mask is not None else 0, nct, ct0)


@pytest.mark.parametrize("rows", [(40000, 23001, 11111, 11111), (3000, 2049, 1, 700), (128, 0, 64, 5000)])
def test_nt_batch_is_bit_identical_to_one_launch_per_problem(gpu, rows):
    """nu_gemm_nt_batch: several problems (own M, K, pointers, sign-bit buffers) in ONE persistent launch -- the four light predictors
    of a pass, a grouped launch expanded into its groups.  Every output bit, zero-filled pad column and sign-bit word must equal what
    one launch per problem writes (a tile's arithmetic does not depend on the launch it belongs to), for both tile heights."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT
    lib = L.load()
    torch.manual_seed(sum(rows))
    Ks = (96, 128, 256, 160)

    def run(batched):
        torch.manual_seed(11)
        outs, keep, probs = [], [], []
        # level 0: BIAS_RELU with sign bits, differing K; one of the problems is a 2-group launch (columns of one 512-wide matrix)
        for i, (M, K) in enumerate(zip(rows, Ks)):
            Mq = max(M, 1)
            grouped = i == 2
            N = 256
            A = torch.randn(Mq, K * (2 if grouped else 1), device=gpu)
            W = torch.randn(2 if grouped else 1, N, K, device=gpu) / K ** 0.5
            bias = torch.randn(2 if grouped else 1, N, device=gpu) * 0.3
            ncol = N * (2 if grouped else 1)
            C = torch.full((Mq, ncol + 32), float("nan"), device=gpu)
            nct = ncol // 128
            mask = torch.zeros(((Mq + 127) // 128) * nct * 256, dtype=torch.int64, device=gpu)
            if grouped:
                g = _gemm_struct(A, 2 * K, W, K, M, N, K, C, ncol + 32, 1, bias=bias, mask=mask, nct=nct, groups=2, sA=K, sB=N * K, sC=N, sBias=N)
            else:
                g = _gemm_struct(A, K, W, K, M, N, K, C, ncol + 32, 1, bias=bias, mask=mask, nct=nct, zero_to=N + 32)
            probs.append(g)
            outs += [C, mask]
            keep += [A, W, bias]
        # backward level: MUL_DRELU reading those sign bits (H poisoned), plain columns past act_cols on one problem
        back = []
        for i, M in enumerate(rows):
            Mq = max(M, 1)
            grouped = i == 2
            ncol = 256 * (2 if grouped else 1)
            dA = torch.randn(Mq, ncol, device=gpu)
            WT = torch.randn(2 if grouped else 1, 384, 256, device=gpu) / 16
            N = 340 if i == 0 else 256
            out = torch.full((Mq, 352 * (2 if grouped else 1)), float("nan"), device=gpu)
            poison = torch.full((Mq, ncol), float("nan"), device=gpu)
            mask = outs[2 * i + 1]
            if grouped:
                g = _gemm_struct(dA, ncol, WT, 256, M, N, 256, out, 704, 3, H=poison, ldh=ncol, mask=mask, nct=4, groups=2, sA=256, sB=384 * 256,
                                 sC=352, sH=256)
            else:
                g = _gemm_struct(dA, ncol, WT, 256, M, N, 256, out, 352, 3, H=poison, ldh=ncol, mask=mask, nct=2, zero_to=352 if i == 0 else 0,
                                 act_cols=256 if i == 0 else 0)
            back.append(g)
            outs.append(out)
            keep += [dA, WT, poison]
        for level in (probs, back):
            if batched:
                arr = (GemmNT * len(level))(*level)
                L.check(lib.nu_gemm_nt_batch(arr, len(level), L.stream()), "nu_gemm_nt_batch")
            else:
                for g in level:
                    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex")
        torch.cuda.synchronize()
        return [o.clone() for o in outs]

    one = run(False)
    many = run(True)
    assert any(torch.isfinite(o.float()).any() for o in one if o.dtype == torch.float32)
    for a, b in zip(one, many):
        if a.dtype == torch.float32:
            assert torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))
        else:
            assert torch.equal(a, b)


def _bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize("M,N,K", [(1, 1, 32), (300, 217, 256), (1000, 257, 288), (4099, 256, 96)])
def test_bf16_nt_matches_bf16_rounded_operands(gpu, M, N, K):
    """mlp_dtype 'bf16': operands are rounded to bf16 (RNE) on load and multiplied exactly (a bf16 x bf16 product is exact
    in fp32), so against fp64 on the rounded operands only the fp32 accumulation order is left."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(M + N)
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(N, K, device=gpu) / K ** 0.5
    B = _packB(W, K)
    bias = torch.randn(N, device=gpu)
    H = torch.rand(M, N, device=gpu) - 0.5
    ref = _bf(A).double() @ _bf(W).double().t()
    for epi, want in ((7, ref), (1, torch.relu(ref + bias.double())), (3, ref * (H > 0))):
        C = torch.full((M, N), float("nan"), device=gpu)
        g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), N, 0, 0, addr(bias), addr(H), N, 0, 0, 0, 0, 0, 0, 1.0, 1,
                   0, 0, 0, 0, 0, 0, 0, 0, epi, 1)
        L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex")
        torch.testing.assert_close(C.double(), want, rtol=2e-5, atol=2e-5 * K ** 0.5)
    # and it really is a different arithmetic from the fp32 path
    exact = A.double() @ W.double().t()
    if M * N > 1000:
        assert (ref - exact).abs().max() > 1e-4


@pytest.mark.parametrize("P,N1,N2,S", [(1000, 257, 256, 7), (5000, 256, 39, 16), (333, 3, 256, 4), (4000, 256, 256, 8)])
def test_bf16_tn_weight_grad(gpu, P, N1, N2, S):
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmTN, addr
    lib = L.load()
    lib.nu_wgrad_workspace_bytes.restype = ctypes.c_longlong
    torch.manual_seed(P)
    lda, ldb = (N1 + 3) // 4 * 4 + 4, (N2 + 3) // 4 * 4
    A0, B0 = torch.randn(P, lda, device=gpu), torch.randn(P, ldb, device=gpu)
    A1, B1 = torch.randn(P, lda, device=gpu), torch.randn(P, ldb, device=gpu)
    wsb = lib.nu_wgrad_workspace_bytes(N1, N2, S, 1)
    ws = torch.empty(wsb // 4, device=gpu)
    C = torch.full((N1, N2), float("nan"), device=gpu)
    bo = torch.full((N1,), float("nan"), device=gpu)
    g = GemmTN(addr(A0), lda, addr(B0), ldb, addr(A1), lda, addr(B1), ldb, P, N1, N2, 0, 0, S, 1, 0, 0, 0, 0, 0, 0, 1, 0)
    L.check(lib.nu_wgrad(ctypes.byref(g), L.ptr(C), N2, ctypes.c_longlong(0), L.ptr(bo), ctypes.c_longlong(0), L.ptr(ws),
                         ctypes.c_longlong(wsb), L.stream()), "nu_wgrad")
    ref = _bf(A0[:, :N1]).double().t() @ _bf(B0[:, :N2]).double() + _bf(A1[:, :N1]).double().t() @ _bf(B1[:, :N2]).double()
    tol = 3e-5 * P ** 0.5
    torch.testing.assert_close(C.double(), ref, rtol=2e-5, atol=tol)
    torch.testing.assert_close(bo.double(), A0[:, :N1].double().sum(0), rtol=2e-5, atol=tol)   # bias sums stay fp32


NU_B16, NU_A16, NU_C16, NU_X16 = 8, 16, 32, 64


@pytest.mark.parametrize("M,N,K,a16", [(1, 1, 32, False), (300, 217, 256, True), (1000, 257, 288, False), (4099, 256, 96, False),
                                       (4099, 256, 96, True), (2048, 256, 1024, True), (513, 128, 64, False), (777, 340, 352, True)])
def test_bf16_storage_nt(gpu, M, N, K, a16):
    """mlp_dtype 'bf16' as the engine runs it: weights (always), A / C / auxiliary matrices (per launch flags) are stored
    as bf16.  bf16 x bf16 products are exact in fp32, so against fp64 on the stored values only the fp32 accumulation order
    and -- for bf16 outputs -- ONE final rounding (rel 2^-8) are left.  Covers K tails (K % 64 == 32), M / N edges, every
    epilogue that takes an auxiliary matrix, fp32 and bf16 output."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(N, K, device=gpu) / K ** 0.5
    B16 = _packB(W, K).bfloat16()
    A_st = A.bfloat16() if a16 else A
    bias = torch.randn(N, device=gpu)
    ldx = (N + 7) // 8 * 8
    H = torch.zeros(M, ldx, device=gpu)
    H[:, :N] = torch.rand(M, N, device=gpu) * 0.02            # softplus(beta = 100) outputs live at this scale
    D = torch.randn(M, ldx, device=gpu)
    Cadd = torch.randn(M, ldx, device=gpu)
    ref = _bf(A).double() @ _bf(W).double().t()
    for x16 in (False, True):
        Hs, Ds, Cs = (H.bfloat16(), D.bfloat16(), Cadd.bfloat16()) if x16 else (H, D, Cadd)
        h, d, ca = Hs.double()[:, :N], Ds.double()[:, :N], Cs.double()[:, :N]
        e = torch.exp(-100.0 * h)
        cases = {7: (ref, None), 0: (ref + bias.double(), None), 1: (torch.relu(ref + bias.double()), None),
                 2: (torch.nn.functional.softplus((ref + bias.double()), beta=100), None), 3: (ref * (h > 0), None),
                 4: (ref * (1 - e), None), 5: (ref * (1 - e), ref * d * 100.0 * e), 6: (ref * (1 - e) + ca, None),
                 8: (ref * (h > 0) + ca, None)}
        for c16 in (False, True):
            for epi, (want, want2) in cases.items():
                if x16 and epi in (7, 0, 1, 2):
                    continue                                   # no auxiliary matrix: covered by the x16 = False pass
                dt = torch.bfloat16 if c16 else torch.float32
                C = torch.full((M, ldx), float("nan"), device=gpu, dtype=dt)
                C2 = torch.full((M, ldx), float("nan"), device=gpu, dtype=dt)
                flags = 1 | NU_B16 | (NU_A16 if a16 else 0) | (NU_C16 if c16 else 0) | (NU_X16 if x16 else 0)
                g = GemmNT(addr(A_st), K, addr(B16), K, M, N, K, addr(C), ldx, addr(C2), ldx, addr(bias), addr(Hs), ldx, addr(Ds), ldx,
                           addr(Cs), ldx, 0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi, flags)
                L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex")
                rt, at = (8e-3, 2e-5 * K ** 0.5 + 1e-6) if c16 else (2e-5, 2e-5 * K ** 0.5)
                torch.testing.assert_close(C[:, :N].double(), want, rtol=rt, atol=at, msg=lambda m: f"epi {epi} c16 {c16} x16 {x16}: {m}")
                if want2 is not None:
                    torch.testing.assert_close(C2[:, :N].double(), want2, rtol=rt, atol=(4e-2 if c16 else 2e-3) * K ** 0.5 / 16)
                assert bool(torch.isnan(C[:, N:].float()).all()) or N == ldx          # nothing written past N


def test_bf16_storage_nt_grouped_and_sign_bits(gpu):
    """Grouped launch (the four material predictors side by side) with bf16 A and C, and the ReLU sign-bit words written by a
    bf16-storage BIAS_RELU launch feeding a bf16-storage MUL_DRELU launch."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(5)
    M = 1500
    A = torch.randn(M, 1024, device=gpu).bfloat16()
    W = (torch.randn(4, 256, 256, device=gpu) / 16).bfloat16()
    bias = torch.randn(1024, device=gpu)
    C = torch.full((M, 1024), float("nan"), device=gpu, dtype=torch.bfloat16)
    nct = 8
    mask = torch.zeros(((M + 127) // 128) * nct * 256, dtype=torch.int64, device=gpu)
    g = GemmNT(addr(A), 1024, addr(W), 256, M, 256, 256, addr(C), 1024, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 4,
               256, 65536, 256, 0, 256, 0, 0, 0, 1, 1 | NU_B16 | NU_A16 | NU_C16, addr(mask), nct)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex")
    want = torch.cat([torch.relu(A[:, 256 * i:256 * i + 256].double() @ W[i].double().t() + bias[256 * i:256 * i + 256].double())
                      for i in range(4)], 1)
    torch.testing.assert_close(C.double(), want, rtol=8e-3, atol=1e-3)
    # backward epilogue reading the sign bits (H pointer poisoned: it must not be read)
    dA = torch.randn(M, 1024, device=gpu).bfloat16()
    WT = (torch.randn(4, 256, 256, device=gpu) / 16).bfloat16()
    poison = torch.full((M, 1024), float("nan"), device=gpu, dtype=torch.bfloat16)
    out = torch.full((M, 1024), float("nan"), device=gpu, dtype=torch.bfloat16)
    g = GemmNT(addr(dA), 1024, addr(WT), 256, M, 256, 256, addr(out), 1024, 0, 0, 0, addr(poison), 1024, 0, 0, 0, 0, 0, 0, 1.0, 4,
               256, 65536, 256, 0, 0, 256, 0, 0, 3, 1 | NU_B16 | NU_A16 | NU_C16 | NU_X16, addr(mask), nct)
    L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex")
    # the sign bits describe the fp32 epilogue value before its rounding to bf16: > 0 there <=> stored bf16 > 0 except for
    # values that round to zero (none at this scale)
    want = torch.cat([(dA[:, 256 * i:256 * i + 256].double() @ WT[i].double().t()) * (C[:, 256 * i:256 * i + 256].double() > 0)
                      for i in range(4)], 1)
    torch.testing.assert_close(out.double(), want, rtol=8e-3, atol=1e-3)


@pytest.mark.parametrize("P,N1,N2,S,flags", [(1000, 257, 256, 7, 16 | 32 | 64 | 128), (5000, 256, 96, 16, 16 | 128), (333, 3, 256, 4, 32 | 64),
                                             (3001, 256, 256, 5, 16 | 32 | 64 | 128), (2000, 512, 256, 3, 32 | 64), (777, 256, 512, 2, 0),
                                             (60001, 256, 256, 200, 16 | 32 | 64 | 128), (50000, 512, 256, 100, 32 | 64)])   # 256-tile kernel
def test_bf16_storage_tn_weight_grad(gpu, P, N1, N2, S, flags):
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmTN, addr
    lib = L.load()
    lib.nu_wgrad_workspace_bytes.restype = ctypes.c_longlong
    torch.manual_seed(P)
    lda, ldb = (N1 + 7) // 8 * 8 + 8, (N2 + 7) // 8 * 8
    ops = [torch.randn(P, ld, device=gpu) for ld in (lda, ldb, lda, ldb)]
    st = [o.bfloat16() if flags & f else o for o, f in zip(ops, (16, 32, 64, 128))]
    wsb = lib.nu_wgrad_workspace_bytes(N1, N2, S, 1)
    ws = torch.empty(wsb // 4, device=gpu)
    C = torch.full((N1, N2), float("nan"), device=gpu)
    bo = torch.full((N1,), float("nan"), device=gpu)
    g = GemmTN(addr(st[0]), lda, addr(st[1]), ldb, addr(st[2]), lda, addr(st[3]), ldb, P, N1, N2, 0, 0, S, 1, 0, 0, 0, 0, 0, 0, 1 | flags, 0)
    L.check(lib.nu_wgrad(ctypes.byref(g), L.ptr(C), N2, ctypes.c_longlong(0), L.ptr(bo), ctypes.c_longlong(0), L.ptr(ws),
                         ctypes.c_longlong(wsb), L.stream()), "nu_wgrad")
    A0, B0, A1, B1 = [_bf(o) for o in ops]
    ref = A0[:, :N1].double().t() @ B0[:, :N2].double() + A1[:, :N1].double().t() @ B1[:, :N2].double()
    tol = 3e-5 * P ** 0.5
    torch.testing.assert_close(C.double(), ref, rtol=2e-5, atol=tol)
    bsrc = st[0].float()[:, :N1].double().sum(0)             # bias sums: of the STORED operand, in fp32
    torch.testing.assert_close(bo.double(), bsrc, rtol=2e-5, atol=tol)


@pytest.mark.parametrize("M,N,K", [(1, 1, 32), (300, 217, 256), (1000, 257, 288), (4099, 256, 96), (2048, 256, 1024)])
def test_bf16x6_nt_is_fp32_equivalent(gpu, M, N, K):
    """mlp_dtype 'bf16x6': exact 3-way bf16 split of both operands, six partial products.  Against fp64 on the UNROUNDED
    operands it must be as accurate as the fp32-MFMA kernel (same tolerance as test_nt_bias_epilogues), and its distance to
    the fp32-MFMA result must stay at the level of fp32 summation noise -- wide dynamic range included."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=gpu) * torch.exp(3 * torch.randn(M, 1, device=gpu))      # rows spanning ~5 decades
    W = torch.randn(N, K, device=gpu) / K ** 0.5
    B = _packB(W, K)
    bias = torch.randn(N, device=gpu)
    ref = A.double() @ W.double().t() + bias.double()
    scale = (A.double().abs() @ W.double().abs().t()) + bias.double().abs()                # |a|.|b| + |bias| mass of every output
    out = {}
    for prec in (0, 2):
        C = torch.full((M, N), float("nan"), device=gpu)
        g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), N, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1,
                   0, 0, 0, 0, 0, 0, 0, 0, 0, prec)
        L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex")
        out[prec] = C.double()
        err = ((C.double() - ref).abs() / (scale + 1e-30)).max()
        assert float(err) < 4e-7 * max(1.0, K ** 0.5 / 4), (prec, float(err))             # ~ a few fp32 ulps of the mass
    rel = ((out[2] - out[0]).abs() / (scale + 1e-30)).max()
    assert float(rel) < 4e-7 * max(1.0, K ** 0.5 / 4), float(rel)


def _p3(B):
    """The pre-split planes of a row-major fp32 table [..., rows, ld] (ld % 16 == 0) in the layout of NuGemmNT.B6: blocks of 256 rows,
    inside a block one 16-wide k-group after the other, each as [256 rows][hi x16 | mid x16 | lo x16]; hi = bf16(w),
    mid = bf16(w - hi), lo = bf16(w - hi - mid) (round-to-nearest-even); the two 8-element halves of a 16-group swapped in rows with
    bit 3 set."""
    ld = B.shape[-1]
    B = B.reshape(-1, ld)
    rows = (B.shape[0] + 255) // 256 * 256
    Bp = torch.zeros(rows, ld, device=B.device)
    Bp[:B.shape[0]] = B
    hi = Bp.bfloat16()
    r1 = Bp - hi.float()
    mid = r1.bfloat16()
    lo = (r1 - mid.float()).bfloat16()
    pl = torch.stack([x.reshape(rows // 256, 256, ld // 16, 16) for x in (hi, mid, lo)], dim=3)     # [blk, row, kg, plane, 16]
    odd = (torch.arange(256, device=B.device) & 8) != 0                                             # rows with bit 3 set: halves swapped
    pl[:, odd] = torch.cat([pl[:, odd][..., 8:], pl[:, odd][..., :8]], dim=-1)
    return pl.permute(0, 2, 1, 3, 4).contiguous().reshape(-1)                                       # [blk, kg, row, plane, 16]


@pytest.mark.parametrize("M,N,K,groups", [(1, 1, 32, 1), (300, 217, 256, 1), (1000, 257, 288, 1), (4099, 256, 96, 1), (2048, 128, 288, 1),
                                          (70001, 256, 256, 1), (5000, 1024, 288, 1), (3333, 256, 256, 4), (20000, 256, 352, 1)])
def test_bf16x6_presplit_weight_planes_give_the_same_bits(gpu, M, N, K, groups):
    """gemm_nt6_kernel (csrc/gemm_nt6.hip: weights pre-split by the pack launch, 128 x 256 tiles, 16-deep pipelined chunks) against
    gemm_nt_kernel<EPI, 2> (operands split in the loop): per accumulator the same MFMAs in the same order -- every epilogue kind,
    grouped launches, sign-bit words, zero-filled pad columns and ragged M / N must come out bit for bit the same."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(M + N + K)
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K * groups, device=gpu) * torch.exp(2 * torch.randn(M, 1, device=gpu))
    W = torch.zeros(groups, Np, K, device=gpu)
    W[:, :N] = torch.randn(groups, N, K, device=gpu) / K ** 0.5
    B6 = _p3(W)
    bias = torch.randn(groups, N, device=gpu)
    ncol = Np * groups
    pre = torch.randn(M, ncol, device=gpu) * 0.02
    Hsp = torch.nn.functional.softplus(pre, beta=100)
    D = torch.randn(M, ncol, device=gpu)
    Cadd = torch.randn(M, ncol, device=gpu)
    nct = ncol // 128
    ldc = ncol + 32
    res = {}
    for use6 in (False, True):
        outs = []
        mask = torch.zeros(((M + 127) // 128) * nct * 256, dtype=torch.int64, device=gpu)
        for epi in (1, 0, 2, 3, 4, 5, 6, 7, 8):
            relu_n = N % 64 == 0                    # the sign-bit path needs N % 64 == 0 (nt_check)
            C = torch.full((M, ldc), float("nan"), device=gpu)
            C2 = torch.full((M, ldc), float("nan"), device=gpu)
            uses_mask = relu_n and epi in (1, 3, 8)
            H = Hsp if epi in (4, 5, 6) else (pre if epi in (3, 8) else None)
            g = GemmNT(addr(A), K * groups, addr(W), K, M, N, K, addr(C), ldc, addr(C2) if epi == 5 else 0, ldc,
                       addr(bias) if epi <= 2 else 0, addr(H) if H is not None else 0, ncol, addr(D) if epi == 5 else 0, ncol,
                       addr(Cadd) if epi in (6, 8) else 0, ncol, Np if groups == 1 and epi != 5 else 0, 0, 1.0, groups,
                       K, Np * K, Np, Np, N, Np, Np, Np, epi, 2 | (4 if use6 else 0), mask.data_ptr() if uses_mask else 0,
                       nct if uses_mask else 0, 0, addr(B6) if use6 else 0)                # 4: NU_GEMM_PRESPLIT_ALWAYS
            L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex epi %d" % epi)
            outs += [C, C2]
        outs.append(mask)
        torch.cuda.synchronize()
        res[use6] = outs
    assert torch.isfinite(res[True][0][:, :N]).all()
    for a, b in zip(res[False], res[True]):
        if a.dtype == torch.float32:
            assert torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))
        else:
            assert torch.equal(a, b)
    # and it is the product it claims to be
    ref = A[:, :K].double() @ W[0, :N].double().t()
    scale = A[:, :K].double().abs() @ W[0, :N].double().abs().t()
    err = ((res[True][14][:, :N].double() - ref).abs() / (scale + 1e-30)).max()          # epilogue 7 (plain), group 0
    assert float(err) < 4e-7 * max(1.0, K ** 0.5 / 4), float(err)


@pytest.mark.parametrize("M,N,K", [(540672, 256, 256), (131072, 1024, 288), (200000, 257, 96)])
def test_bf16x6_presplit_kernel_walks_many_tiles_per_workgroup(gpu, M, N, K):
    """The persistent walk of gemm_nt6_kernel over several tiles per workgroup (hand-over of the next tile's first chunks under the
    last chunks of the current one, both A register sets, the epilogue scratch aliasing a stage): repeated launches give the bits of
    gemm_nt_kernel<EPI, 2>, every time."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmNT, addr
    lib = L.load()
    torch.manual_seed(M + N)
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K, device=gpu)
    W = torch.zeros(Np, K, device=gpu)
    W[:N] = torch.randn(N, K, device=gpu) / K ** 0.5
    B6 = _p3(W)
    bias = torch.randn(N, device=gpu)

    def run(flag, b6):
        C = torch.full((M, Np), float("nan"), device=gpu)
        g = GemmNT(addr(A), K, addr(W), K, M, N, K, addr(C), Np, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1,
                   0, 0, 0, 0, 0, 0, 0, 0, 2, 2 | flag, 0, 0, 0, b6)
        L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nu_gemm_nt_ex")
        return C[:, :N]
    ref = run(0, 0)
    for _ in range(6):
        out = run(4, addr(B6))
        assert torch.equal(out, ref)


@pytest.mark.parametrize("P,N1,N2,S", [(1000, 257, 256, 7), (5000, 256, 39, 16), (70000, 128, 288, 32)])
def test_bf16x6_tn_is_fp32_equivalent(gpu, P, N1, N2, S):
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import GemmTN, addr
    lib = L.load()
    lib.nu_wgrad_workspace_bytes.restype = ctypes.c_longlong
    torch.manual_seed(P)
    lda, ldb = (N1 + 3) // 4 * 4 + 4, (N2 + 3) // 4 * 4
    A0, B0 = torch.randn(P, lda, device=gpu), torch.randn(P, ldb, device=gpu) * torch.exp(2 * torch.randn(P, 1, device=gpu))
    wsb = lib.nu_wgrad_workspace_bytes(N1, N2, S, 1)
    ws = torch.empty(wsb // 4, device=gpu)
    ref = A0[:, :N1].double().t() @ B0[:, :N2].double()
    scale = A0[:, :N1].double().abs().t() @ B0[:, :N2].double().abs()
    for prec in (0, 2):
        C = torch.full((N1, N2), float("nan"), device=gpu)
        bo = torch.full((N1,), float("nan"), device=gpu)
        g = GemmTN(addr(A0), lda, addr(B0), ldb, 0, 0, 0, 0, P, N1, N2, 0, 0, S, 1, 0, 0, 0, 0, 0, 0, prec, 0)
        L.check(lib.nu_wgrad(ctypes.byref(g), L.ptr(C), N2, ctypes.c_longlong(0), L.ptr(bo), ctypes.c_longlong(0), L.ptr(ws),
                             ctypes.c_longlong(wsb), L.stream()), "nu_wgrad")
        err = ((C.double() - ref).abs() / (scale + 1e-30)).max()
        assert float(err) < 2e-6, (prec, float(err))
        torch.testing.assert_close(bo.double(), A0[:, :N1].double().sum(0), rtol=2e-5, atol=3e-5 * P ** 0.5)


def test_gemm_throughput_smoke(gpu):
    """Not a pass/fail perf gate: prints achieved TFLOP/s of the 256x256 layer GEMM."""
    M, N, K = 262144, 256, 256
    A = torch.randn(M, K, device=gpu)
    W = torch.randn(N, K, device=gpu) / 16
    B = _packB(W, K)
    b = torch.zeros(N, device=gpu)
    for _ in range(3):
        _nt(A, B, N, 2, bias=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _nt(A, B, N, 2, bias=b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"\n[gemm_nt 262144x256x256 softplus] {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.1f} TFLOP/s")


@pytest.mark.parametrize("P,K,NO", [(1000, 256, 1), (777, 256, 3), (513, 1024, 6), (300, 128, 3)])
def test_bf16_storage_skinny_heads(gpu, P, K, NO):
    """The 1..6-wide heads on bf16-stored hidden rows (nu_skinny_fwd_h16 / nu_skinny_bwd_enqueue_h16): fp32 arithmetic on the stored
    values; dH comes back as bf16 (one rounding)."""
    from nu_nerf_amd import _lib as L
    from nu_nerf_amd.engine import ReduceDesc, addr
    lib = L.load()
    lib.nu_skinny_bwd_workspace_bytes.restype = ctypes.c_longlong
    torch.manual_seed(P + K)
    H = torch.relu(torch.randn(P, K, device=gpu)).bfloat16()
    W = torch.randn(NO, K, device=gpu) / K ** 0.5
    b = torch.randn(NO, device=gpu)
    out = torch.full((P, 8), float("nan"), device=gpu)
    cp = ctypes.c_void_p
    L.check(lib.nu_skinny_fwd_h16(cp(addr(H)), K, P, K, cp(addr(W)), K, cp(addr(b)), NO, cp(addr(out)), 8, L.stream()), "nu_skinny_fwd_h16")
    ref = H.double() @ W.double().t() + b.double()
    torch.testing.assert_close(out[:, :NO].double(), ref, rtol=2e-5, atol=2e-5)
    dy = torch.randn(P, 8, device=gpu)
    dH = torch.full((P, K), float("nan"), device=gpu, dtype=torch.bfloat16)
    dW = torch.zeros(NO, K, device=gpu)
    db = torch.zeros(NO, device=gpu)
    wsb = lib.nu_skinny_bwd_workspace_bytes(K, NO)
    ws = torch.empty(wsb // 4, device=gpu)
    descs = (ReduceDesc * 4)()
    nd = ctypes.c_int(0)
    L.check(lib.nu_skinny_bwd_enqueue_h16(cp(addr(dy)), 8, cp(addr(H)), K, P, K, cp(addr(W)), K, NO, cp(addr(dH)), K, 1, 0, cp(addr(dW)), K,
                                          cp(addr(db)), cp(addr(ws)), ctypes.c_longlong(wsb), descs, ctypes.byref(nd), 4, L.stream()),
            "nu_skinny_bwd_enqueue_h16")
    L.check(lib.nu_slab_reduce_batched(descs, nd.value, L.stream()), "nu_slab_reduce_batched")
    g = dy[:, :NO].double()
    want_dH = (g @ W.double()) * (H.double() > 0)
    torch.testing.assert_close(dH.double(), want_dH, rtol=8e-3, atol=1e-6)
    torch.testing.assert_close(dW.double(), g.t() @ H.double(), rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(db.double(), g.sum(0), rtol=2e-5, atol=2e-4)
