"""bf16x6 NT GEMM: the in-loop-split kernel (gemm_nt_kernel<EPI, 2>) against the pre-split-weights kernel (gemm_nt6_kernel) and the
exact fp32-MFMA kernel, lone launches.  TFLOP/s are fp32-equivalent (2 M N K per launch).  Usage: python scripts/bench_nt6.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from nu_nerf_amd import _lib as L  # noqa: E402
from nu_nerf_amd.engine import GemmNT, addr  # noqa: E402
from test_gemm_gpu import _p3  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')


def run(M, N, K, epi, mode, reps=20):
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K, device=dev)
    W = torch.zeros(Np, K, device=dev)
    W[:N] = torch.randn(N, K, device=dev) / K ** 0.5
    B6 = _p3(W)
    bias = torch.randn(N, device=dev)
    H = torch.rand(M, Np, device=dev) * 0.05
    C = torch.empty(M, Np, device=dev)
    prec, b6 = {'fp32': (0, 0), 'split': (2, 0), 'nt6': (2 | 4, addr(B6))}[mode]
    g = GemmNT(addr(A), K, addr(W), K, M, N, K, addr(C), Np, 0, 0, addr(bias) if epi <= 2 else 0, addr(H) if epi in (3, 4) else 0, Np,
               0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi, prec, 0, 0, 0, b6)
    for _ in range(3):
        L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us * 1e-6


for (M, N, K, epi) in [(524288, 256, 256, 1), (524288, 256, 256, 2), (524288, 256, 256, 4), (524288, 256, 256, 7), (114000, 256, 256, 2),
                       (131072, 1024, 288, 1), (524288, 256, 96, 1), (524288, 128, 288, 1), (262144, 256, 1024, 7), (65536, 256, 256, 2),
                       (16384, 256, 256, 2)]:
    row = "NT M=%7d N=%4d K=%4d epi=%d:" % (M, N, K, epi)
    for mode in ('fp32', 'split', 'nt6'):
        us, tf = run(M, N, K, epi, mode)
        row += "  %s %8.1f us %6.1f TF" % (mode, us, tf)
    print(row, flush=True)
