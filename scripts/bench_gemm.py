"""Micro-benchmark of the raw GEMM kernels (development aid): TFLOP/s for the shapes the training step uses."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nu_nerf_amd import _lib as L
from nu_nerf_amd.engine import GemmNT, GemmTN, addr

lib = L.load()
lib.nu_wgrad_workspace_bytes.restype = ctypes.c_longlong
dev = torch.device('cuda:0')


def time_it(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def nt(M, N, K, epi, ldc=None):
    A = torch.randn(M, K, device=dev)
    B = torch.randn((N + 127) // 128 * 128, K, device=dev) / K ** 0.5
    ldc = ldc or N
    C = torch.empty(M, ldc, device=dev); C2 = torch.empty(M, ldc, device=dev)
    H = torch.rand(M, ldc, device=dev) * 0.02; D = torch.randn(M, ldc, device=dev); b = torch.randn(N, device=dev)
    g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), ldc, addr(C2), ldc, addr(b), addr(H), ldc, addr(D), ldc, addr(D), ldc,
               0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi)
    ms = time_it(lambda: L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt"))
    print(f"NT  M={M:7d} N={N:4d} K={K:4d} epi={epi}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TFLOP/s")


def tn(P, N1, N2, pairs, S):
    A0 = torch.randn(P, N1, device=dev); B0 = torch.randn(P, N2, device=dev)
    dW = torch.empty(N1, N2, device=dev); db = torch.empty(N1, device=dev)
    nb = lib.nu_wgrad_workspace_bytes(N1, N2, S, 1)
    ws = torch.empty(nb // 4, device=dev)
    g = GemmTN(addr(A0), N1, addr(B0), N2, addr(A0) if pairs == 2 else 0, N1, addr(B0) if pairs == 2 else 0, N2, P, N1, N2, 0, 0, S, 1, 0, 0, 0, 0, 0, 0)
    ms = time_it(lambda: L.check(lib.nu_wgrad(ctypes.byref(g), ctypes.c_void_p(addr(dW)), N2, ctypes.c_longlong(0), ctypes.c_void_p(addr(db)), ctypes.c_longlong(0), ctypes.c_void_p(addr(ws)), ctypes.c_longlong(nb), L.stream()), "tn"))
    print(f"TN  P={P:7d} N1={N1:4d} N2={N2:4d} pairs={pairs} S={S:4d}: {ms*1e3:8.1f} us  {2.0*P*N1*N2*pairs/ms/1e9:6.1f} TFLOP/s")


if __name__ == "__main__":  # noqa
    for epi in (7, 0, 1, 2, 3, 4, 5, 6):
        nt(262144, 256, 256, epi)
    nt(262144, 256, 1024, 7)
    nt(262144, 256, 64, 2)
    nt(131072, 256, 256, 2)
    nt(524288, 256, 256, 1)
    nt(131072, 1024, 288, 1)
    nt(393216, 256, 96, 1)
    nt(262144, 128, 256, 7)
    for S in (64, 128, 256, 512):
        tn(262144, 256, 256, 1, S)
    tn(131072, 256, 256, 2, 256)
    tn(524288, 256, 256, 1, 256)
    tn(131072, 1024, 288, 1, 32)


def nt_bf16(M, N, K, epi, prec=1):
    A = torch.randn(M, K, device=dev)
    B = torch.randn((N + 127) // 128 * 128, K, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev)
    H = torch.rand(M, N, device=dev) * 0.02; D = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev)
    g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), N, addr(C2), N, addr(b), addr(H), N, addr(D), N, addr(D), N,
               0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi, prec)
    ms = time_it(lambda: L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt"))
    byts = 4.0 * (M * K + M * N * (2 if epi in (3, 4, 6, 8) else 1))
    print(f"NT prec{prec} M={M:7d} N={N:4d} K={K:4d} epi={epi}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TFLOP/s  {byts/ms/1e9:5.2f} TB/s algorithmic")
