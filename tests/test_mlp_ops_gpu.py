"""GPU parity of the skinny-head kernels (csrc/mlp.hip) against plain torch: 1..6-wide output heads of the predictor
stacks (field.py:393) and the NeRF++ alpha / rgb heads (field.py:260-261), forward and backward."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,K,NO,ldy", [(1, 256, 1, 1), (1000, 256, 3, 4), (777, 128, 3, 4), (5000, 1024, 6, 8),
                                         (70000, 256, 1, 1), (9, 128, 1, 1)])
@pytest.mark.parametrize("relu_mask,accumulate", [(1, 0), (0, 1)])
def test_skinny_bwd(gpu, P, K, NO, ldy, relu_mask, accumulate):
    from nu_nerf_amd import _lib as L
    lib = L.load()
    lib.nu_skinny_bwd_workspace_bytes.restype = ctypes.c_longlong
    torch.manual_seed(P + K + NO)
    ldh = K + 32
    H = torch.randn(P, ldh, device=gpu)
    dy = torch.randn(P, ldy, device=gpu)
    Ws = torch.randn(NO, K, device=gpu)
    dH0 = torch.randn(P, K, device=gpu)
    dH = dH0.clone()
    dWs = torch.full((NO, K), float("nan"), device=gpu)
    db = torch.full((NO,), float("nan"), device=gpu)
    wsb = lib.nu_skinny_bwd_workspace_bytes(K, NO)
    ws = torch.empty(wsb // 4, device=gpu)
    rc = lib.nu_skinny_bwd(L.ptr(dy), ldy, L.ptr(H), ldh, P, K, L.ptr(Ws), K, NO, L.ptr(dH), K, relu_mask, accumulate,
                           L.ptr(dWs), K, L.ptr(db), L.ptr(ws), ctypes.c_longlong(wsb), L.stream())
    L.check(rc, "nu_skinny_bwd")
    g, h = dy[:, :NO].double(), H[:, :K].double()
    d = g @ Ws.double()
    if relu_mask:
        d = d * (h > 0)
    if accumulate:
        d = d + dH0.double()
    tol = 3e-5 * P ** 0.5
    torch.testing.assert_close(dH.double(), d, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dWs.double(), g.t() @ h, rtol=2e-5, atol=tol)
    torch.testing.assert_close(db.double(), g.sum(0), rtol=2e-5, atol=tol)


@pytest.mark.parametrize("P,K,NO", [(1, 256, 1), (1000, 256, 3), (777, 128, 3), (5000, 1024, 6), (300, 256, 4), (50, 128, 2)])
def test_skinny_fwd(gpu, P, K, NO):
    from nu_nerf_amd import _lib as L
    lib = L.load()
    torch.manual_seed(P + K + NO)
    H = torch.randn(P, K + 4, device=gpu)
    Ws = torch.randn(NO, K, device=gpu)
    b = torch.randn(NO, device=gpu)
    out = torch.full((P, 8), float("nan"), device=gpu)
    L.check(lib.nu_skinny_fwd(L.ptr(H), K + 4, P, K, L.ptr(Ws), K, L.ptr(b), NO, L.ptr(out), 8, L.stream()), "nu_skinny_fwd")
    ref = H[:, :K].double() @ Ws.double().t() + b.double()
    torch.testing.assert_close(out[:, :NO].double(), ref, rtol=2e-5, atol=2e-5 * K ** 0.5)
    assert torch.isnan(out[:, NO:]).all()
