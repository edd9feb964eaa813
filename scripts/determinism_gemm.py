"""Development aid: repeated launches of the bf16x6 (mode 2) NT and TN kernels on fixed operands; reports launches whose output bits
differ from the first.  usage: python scripts/determinism_gemm.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nu_nerf_amd import _lib as L  # noqa: E402
from nu_nerf_amd.engine import GemmNT, GemmTN, addr  # noqa: E402

lib = L.load()
lib.nu_wgrad_workspace_bytes.restype = ctypes.c_longlong
dev = torch.device('cuda:0')
torch.manual_seed(0)
for prec in (2, 0):
    for (M, N, K, epi) in [(3840, 256, 256, 2), (3840, 257, 256, 0), (640, 256, 96, 1), (7000, 1024, 288, 1), (3840, 256, 256, 4)]:
        Np = (N + 127) // 128 * 128
        A = torch.randn(M, K, device=dev)
        W = torch.zeros(Np, K, device=dev); W[:N] = torch.randn(N, K, device=dev) / K ** 0.5
        bias = torch.randn(N, device=dev)
        H = torch.rand(M, Np, device=dev) * 0.05
        outs = []
        for r in range(12):
            C = torch.full((M, Np), float('nan'), device=dev)
            g = GemmNT(addr(A), K, addr(W), K, M, N, K, addr(C), Np, 0, 0, addr(bias) if epi <= 2 else 0, addr(H) if epi == 4 else 0, Np,
                       0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi, prec)
            L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")
            outs.append(C[:, :N].clone())
        torch.cuda.synchronize()
        bad = [i for i in range(1, 12) if not torch.equal(outs[0], outs[i])]
        print('NT prec', prec, (M, N, K, epi), 'differing launches:', bad, flush=True)
    for (P, N1, N2, S) in [(3840, 256, 256, 8), (3840, 257, 256, 7), (640, 256, 96, 4), (3840, 256, 39, 16), (3840, 1024, 288, 4), (50000, 256, 256, 64)]:
        lda, ldb = (N1 + 3) // 4 * 4 + 4, (N2 + 3) // 4 * 4
        A0, B0 = torch.randn(P, lda, device=dev), torch.randn(P, ldb, device=dev)
        wsb = lib.nu_wgrad_workspace_bytes(N1, N2, S, 1)
        ws = torch.empty(wsb // 4, device=dev)
        outs = []
        for r in range(12):
            C = torch.full((N1, N2), float('nan'), device=dev)
            bo = torch.full((N1,), float('nan'), device=dev)
            ws.fill_(float(r))            # whatever the slab held before must not matter
            g = GemmTN(addr(A0), lda, addr(B0), ldb, 0, 0, 0, 0, P, N1, N2, 0, 0, S, 1, 0, 0, 0, 0, 0, 0, prec, 0)
            L.check(lib.nu_wgrad(ctypes.byref(g), L.ptr(C), N2, ctypes.c_longlong(0), L.ptr(bo), ctypes.c_longlong(0), L.ptr(ws),
                                 ctypes.c_longlong(wsb), L.stream()), "nu_wgrad")
            outs.append((C.clone(), bo.clone()))
        torch.cuda.synchronize()
        bad = [i for i in range(1, 12) if not (torch.equal(outs[0][0], outs[i][0]) and torch.equal(outs[0][1], outs[i][1]))]
        print('TN prec', prec, (P, N1, N2, S), 'differing launches:', bad, flush=True)
