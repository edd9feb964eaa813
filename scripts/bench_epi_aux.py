"""Development aid: how much of the derivative epilogues' cost is the auxiliary HBM traffic?  Times MUL_DRELU (epi 3) with a
real H matrix and with every row reading the same 1 KB row (ldh = 0: cache hits), against BIAS_RELU (epi 1)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch
from bench_gemm import lib, L, dev, time_it, GemmNT, addr
for M in (131072, 540000):
    N = K = 256
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) / 16
    C = torch.empty(M, N, device=dev); H = torch.rand(M, N, device=dev) - 0.5; b = torch.randn(N, device=dev)
    H1 = torch.rand(1, N, device=dev) - 0.5
    for name, epi, h, ldh in (("bias_relu", 1, H, N), ("mul_drelu real H", 3, H, N), ("mul_drelu broadcast H", 3, H1, 0),
                              ("mul_dsp real H", 4, H, N), ("mul_dsp broadcast H", 4, H1, 0)):
        g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), N, 0, 0, addr(b), addr(h), ldh, 0, 0, 0, 0, 0, 0, 1.0, 1,
                   0, 0, 0, 0, 0, 0, 0, 0, epi, 0)
        ms = min(time_it(lambda: L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt"), iters=20) for _ in range(3))
        print(f"M={M:7d} {name:24s} {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TFLOP/s")
