"""Does the fp32 NT kernel wait for its A stream?  The same FLOPs and the same C traffic with A streamed from HBM (one 524 288-row problem)
and with A resident in the caches (32 groups that all read ONE 16 384-row A).  usage: python scripts/bench_nt_a_residency.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nu_nerf_amd import _lib as L  # noqa: E402
from nu_nerf_amd.engine import GemmNT, addr  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')


def run(M, groups, sA, K, epi, reps=20):
    N = 256
    A = torch.randn(M if sA == 0 else M * groups, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    C = torch.empty(M * groups, N, device=dev)
    g = GemmNT(addr(A), K, addr(W), K, M, N, K, addr(C), N, 0, 0, addr(bias) if epi <= 2 else 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, groups,
               sA, 0, M * N, 0, 0, 0, 0, 0, epi, 0)
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
    return us, 2.0 * M * groups * N * K / us * 1e-6


for K in (256, 1024):
    for epi in (7, 1, 2):
        a = run(16384, 32, 16384 * K, K, epi)        # 32 groups, each its own A rows: streamed (= one 524 288-row problem, grouped)
        b = run(16384, 32, 0, K, epi)                # 32 groups reading the same 16 384 rows: A stays in L2 / Infinity Cache
        c = run(524288, 1, 0, K, epi)
        print("K=%4d epi=%d   one problem %7.1f us %6.1f TF | 32 groups, A streamed %7.1f us %6.1f TF | 32 groups, A resident %7.1f us %6.1f TF"
              % (K, epi, c[0], c[1], a[0], a[1], b[0], b[1]), flush=True)
