"""Development aid: a pure-VALU kernel (skinny_bwd_kernel<3, 256>: dH = (dy . Ws) masked by H > 0) launched repeatedly on one stream with
fixed inputs while a second stream runs (a) nothing, (b) exact-fp32 MFMA GEMMs, (c) the bf16x6 split GEMMs, (d) bf16-storage GEMMs.
Reports the launches whose dH differs bit-wise from a quiet reference.  usage: python scripts/determinism_valu_victim.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nu_nerf_amd import _lib as L  # noqa: E402
from nu_nerf_amd.engine import GemmNT, addr  # noqa: E402

lib = L.load()
lib.nu_skinny_bwd_workspace_bytes.restype = ctypes.c_longlong
dev = torch.device('cuda:0')
torch.manual_seed(0)
side = torch.cuda.Stream(dev)
c_p, c_ll = ctypes.c_void_p, ctypes.c_longlong

P, K, NO = 43133, 256, 3
dy = torch.randn(P, 4, device=dev) * 1e-5
H = torch.relu(torch.randn(P, K, device=dev))
Ws = torch.randn(NO, K, device=dev) / 16
dH = torch.empty(P, K, device=dev)
dWs = torch.empty(NO, K, device=dev)
db = torch.empty(NO, device=dev)
wsb = lib.nu_skinny_bwd_workspace_bytes(K, NO)
ws = torch.empty(wsb // 4 + 64, device=dev)


def victim():
    L.check(lib.nu_skinny_bwd(c_p(addr(dy)), 4, c_p(addr(H)), K, P, K, c_p(addr(Ws)), K, NO, c_p(addr(dH)), K, 1, 0, c_p(addr(dWs)), K,
                              c_p(addr(db)), c_p(addr(ws)), c_ll(wsb), L.stream()), "nu_skinny_bwd")


def nt_desc(M, N, Kk, epi, prec):
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, Kk, device=dev)
    W = torch.zeros(Np, Kk, device=dev); W[:N] = torch.randn(N, Kk, device=dev) / Kk ** 0.5
    bias = torch.randn(N, device=dev)
    C = torch.full((M, Np), float('nan'), device=dev)
    g = GemmNT(addr(A), Kk, addr(W), Kk, M, N, Kk, addr(C), Np, 0, 0, addr(bias), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi, prec)
    return g, (A, W, bias, C)


sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gemm_gpu import _p3  # noqa: E402


def nt6_desc(M, N, Kk, epi):
    g, keep = nt_desc(M, N, Kk, epi, 2)
    B6 = _p3(keep[1])
    g.bf16 = 2 | 4
    g.B6 = addr(B6)
    return g, keep + (B6,)


victim()
torch.cuda.synchronize()
ref = dH.clone()
cases = [('nothing', []),
         ('fp32 MFMA GEMMs (gemm_nt2_kernel)', [nt_desc(120000, 256, 256, e, 0) for e in (1, 2, 7)]),
         ('bf16x6 split GEMMs (gemm_nt_kernel<EPI, 2>)', [nt_desc(120000, 256, 256, e, 2) for e in (1, 2, 7)]),
         ('bf16x6 pre-split GEMMs (gemm_nt6_kernel)', [nt6_desc(120000, 256, 256, e) for e in (1, 2, 7)]),
         ('bf16 GEMMs, operands rounded in the loop (gemm_nt_kernel<EPI, 1>)', [nt_desc(120000, 256, 256, e, 1) for e in (1, 2, 7)])]
for name, bg in cases:
    bad, worst, shown = 0, 0.0, False
    for r in range(60):
        dH.fill_(float('nan'))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(2):
                for g, _k in bg:
                    lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream())
        for _ in range(3):
            victim()
        torch.cuda.synchronize()
        if not torch.equal(dH, ref):
            bad += 1
            neq = dH != ref
            worst = max(worst, float(((dH - ref).abs() / (ref.abs() + 1e-30))[neq].max()))
            if not shown:
                shown = True
                rows = neq.any(1).nonzero().flatten()
                cols = neq.any(0).nonzero().flatten()
                r0 = int(rows[0]); c0 = int(neq[r0].nonzero()[0])
                print('      first differing launch: %d elements in %d rows (first rows %s), columns %d..%d; row %d from col %d: ref %s  got %s'
                      % (int(neq.sum()), rows.numel(), rows[:8].tolist(), int(cols.min()), int(cols.max()), r0, c0,
                         ['%.3e' % v for v in ref[r0, c0:c0 + 6].tolist()], ['%.3e' % v for v in dH[r0, c0:c0 + 6].tolist()]), flush=True)
                print('      dy of that row', dy[r0, :3].tolist(), ' mask of those columns', (H[r0, c0:c0 + 6] > 0).tolist())
    print('%-72s victim launches differing: %2d of 60   worst relative element difference %.2e' % (name, bad, worst), flush=True)


# ---- is it these kernels, or the hardware?  (1) the same victim next to torch.mm on bf16 tensors (hipBLASLt / rocBLAS kernels); (2) plain torch
# elementwise kernels as the victim next to this library's bf16x6 GEMMs
a16, b16 = torch.randn(8192, 4096, device=dev, dtype=torch.bfloat16), torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
a32, b32 = torch.randn(4096, 2048, device=dev), torch.randn(2048, 2048, device=dev)
for name, fn in (('torch.mm bf16 (library GEMM)', lambda: torch.mm(a16, b16)), ('torch.mm fp32 (library GEMM)', lambda: torch.mm(a32, b32))):
    bad = 0
    for r in range(60):
        dH.fill_(float('nan'))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):
                fn()
        for _ in range(3):
            victim()
        torch.cuda.synchronize()
        bad += 0 if torch.equal(dH, ref) else 1
    print('%-72s victim launches differing: %2d of 60' % (name, bad), flush=True)
x, y, z = (torch.randn(16 * 1024 * 1024, device=dev) for _ in range(3))
ref_t = torch.addcmul(z, x, y) * 1.5 + torch.sqrt(x.abs())
bgs = [nt_desc(120000, 256, 256, e, 2) for e in (1, 2, 7)]
for name, bg in (('nothing', []), ('bf16x6 split GEMMs', bgs)):
    bad = 0
    for r in range(40):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(2):
                for g, _k in bg:
                    lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream())
        out = torch.addcmul(z, x, y) * 1.5 + torch.sqrt(x.abs())
        torch.cuda.synchronize()
        bad += 0 if torch.equal(out, ref_t) else 1
    print('torch elementwise victim next to %-40s launches differing: %2d of 40' % (name, bad), flush=True)


# ---- (3) the instruction alone: kernels that only issue one kind of MFMA from registers (scripts/mfma_spin.hip; no LDS, no memory traffic)
spin_path = os.path.join(ROOT, 'scripts', 'libmfma_spin.so')
if os.path.exists(spin_path):
    spin = ctypes.CDLL(spin_path)
    sink = torch.zeros(4, device=dev)
    for kind, name in ((0, 'v_mfma_f32_32x32x16_bf16 only'), (1, 'v_mfma_f32_16x16x32_bf16 only'), (2, 'v_mfma_f32_32x32x2_f32 only'),
                       (3, 'v_mfma_f32_32x32x16_f16 only'), (4, 'v_cvt_pk_bf16_f32 only (no MFMA)'), (5, 'LDS b128 traffic only'),
                       (6, 'fp32 -> bf16 conversion -> LDS -> bf16 MFMA (a rounding GEMM loop without the memory stream)')):
        bad = 0
        for r in range(60):
            dH.fill_(float('nan'))
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                spin.mfma_spin(kind, 1024, 20000, c_p(addr(sink)), L.stream())
            for _ in range(3):
                victim()
            torch.cuda.synchronize()
            bad += 0 if torch.equal(dH, ref) else 1
        print('%-72s victim launches differing: %2d of 60' % ('single-instruction loop: ' + name, bad), flush=True)
else:
    print('(scripts/libmfma_spin.so not built: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/libmfma_spin.so scripts/mfma_spin.hip)')
