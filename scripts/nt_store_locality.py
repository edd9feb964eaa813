"""Development aid: is the cost of the NT epilogue's stores a memory-system cost?  32 groups of 8 192 rows (the tile count of one
262 144-row launch); each operand either its own 8-MiB block per group (streams through HBM) or ONE block shared by all groups
(stays in L2 / MALL).  Shared C means the groups overwrite each other -- timing only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nu_nerf_amd import _lib as L
from nu_nerf_amd.engine import GemmNT, addr
lib = L.load(); dev = torch.device('cuda:0')
M, N, K, G = 8192, 256, 256, 32

def time_it(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

A = torch.randn(G * M, K, device=dev); B = torch.randn(N, K, device=dev) / K ** 0.5
C = torch.empty(G * M, N, device=dev); H = torch.rand(G * M, N, device=dev); b = torch.randn(N, device=dev)
for epi in (1, 2, 4):
    for sa, sc in ((1, 1), (1, 0), (0, 1), (0, 0)):
        g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), N, addr(C), N, addr(b), addr(H), N, addr(H), N, addr(H), N,
                   0, 0, 1.0, G, sa * M * K, 0, sc * M * N, sc * M * N, 0, sc * M * N, sc * M * N, sc * M * N, epi)
        us = time_it(lambda: L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt"))
        print(f"epi={epi}  A {'streams' if sa else 'shared '}  C/aux {'streams' if sc else 'shared '}: {us:7.1f} us  {2.0*G*M*N*K/us/1e6:6.1f} TFLOP/s", flush=True)
