"""Development aid: the mode-2 (bf16x6) NT and TN kernels launched on one stream while another stream keeps the chip busy with other
GEMM launches; reports launches whose bits differ from a quiet reference launch.  usage: python scripts/determinism_gemm2.py"""
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
side = torch.cuda.Stream(dev)


def nt_desc(M, N, K, epi, prec):
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K, device=dev)
    W = torch.zeros(Np, K, device=dev); W[:N] = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    C = torch.full((M, Np), float('nan'), device=dev)
    g = GemmNT(addr(A), K, addr(W), K, M, N, K, addr(C), Np, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi, prec)
    return g, C, (A, W, bias)


def tn_desc(P, N1, N2, S, prec):
    lda, ldb = (N1 + 3) // 4 * 4 + 4, (N2 + 3) // 4 * 4
    A0, B0 = torch.randn(P, lda, device=dev), torch.randn(P, ldb, device=dev)
    wsb = lib.nu_wgrad_workspace_bytes(N1, N2, S, 1)
    ws = torch.empty(wsb // 4, device=dev)
    C = torch.full((N1, N2), float('nan'), device=dev)
    bo = torch.full((N1,), float('nan'), device=dev)
    g = GemmTN(addr(A0), lda, addr(B0), ldb, 0, 0, 0, 0, P, N1, N2, 0, 0, S, 1, 0, 0, 0, 0, 0, 0, prec, 0)
    return g, C, bo, ws, wsb, (A0, B0)


bg = [nt_desc(8000, 256, 256, 1, p) for p in (0, 2)] + [nt_desc(3000, 256, 96, 2, 2)]
bgt = [tn_desc(8000, 256, 256, 16, p) for p in (0, 2)]
for prec in (2, 0):
    for shape in [(3840, 256, 256, 2), (640, 256, 96, 1), (7000, 1024, 288, 1)]:
        g, C, keep = nt_desc(*shape, prec)
        L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")
        torch.cuda.synchronize()
        ref = C.clone()
        bad = 0
        for r in range(40):
            C.fill_(float('nan'))
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(3):
                    for gb, _, _ in bg:
                        lib.nu_gemm_nt_ex(ctypes.byref(gb), L.stream())
                    for gt, Ct, bt, wst, wsbt, _ in bgt:
                        lib.nu_wgrad(ctypes.byref(gt), L.ptr(Ct), gt.N2, ctypes.c_longlong(0), L.ptr(bt), ctypes.c_longlong(0), L.ptr(wst), ctypes.c_longlong(wsbt), L.stream())
            L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")
            torch.cuda.synchronize()
            bad += 0 if torch.equal(torch.nan_to_num(C, nan=7.0), torch.nan_to_num(ref, nan=7.0)) else 1
        print('NT prec', prec, shape, 'launches differing under concurrency:', bad, 'of 40', flush=True)
    for shape in [(3840, 256, 256, 8), (3840, 1024, 288, 4), (640, 256, 96, 4)]:
        g, C, bo, ws, wsb, keep = tn_desc(*shape, prec)
        L.check(lib.nu_wgrad(ctypes.byref(g), L.ptr(C), g.N2, ctypes.c_longlong(0), L.ptr(bo), ctypes.c_longlong(0), L.ptr(ws), ctypes.c_longlong(wsb), L.stream()), "tn")
        torch.cuda.synchronize()
        ref, refb = C.clone(), bo.clone()
        bad = 0
        for r in range(40):
            C.fill_(float('nan'))
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(3):
                    for gb, _, _ in bg:
                        lib.nu_gemm_nt_ex(ctypes.byref(gb), L.stream())
                    for gt, Ct, bt, wst, wsbt, _ in bgt:
                        lib.nu_wgrad(ctypes.byref(gt), L.ptr(Ct), gt.N2, ctypes.c_longlong(0), L.ptr(bt), ctypes.c_longlong(0), L.ptr(wst), ctypes.c_longlong(wsbt), L.stream())
            L.check(lib.nu_wgrad(ctypes.byref(g), L.ptr(C), g.N2, ctypes.c_longlong(0), L.ptr(bo), ctypes.c_longlong(0), L.ptr(ws), ctypes.c_longlong(wsb), L.stream()), "tn")
            torch.cuda.synchronize()
            bad += 0 if (torch.equal(C, ref) and torch.equal(bo, refb)) else 1
        print('TN prec', prec, shape, 'launches differing under concurrency:', bad, 'of 40', flush=True)
