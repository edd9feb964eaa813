"""Development aid: does running a chain of row-wise layers block by block (so that a block's activations are still in the 256-MB
MALL when the next layer reads them) beat layer by layer over all rows?  8 layers 256 -> 256, bias + ReLU, fp32, P rows."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nu_nerf_amd import _lib as L
from nu_nerf_amd.engine import GemmNT, addr
lib = L.load(); dev = torch.device('cuda:0')
P, N, K, NL = 540672, 256, 256, 8
H = [torch.randn(P, N, device=dev) for _ in range(NL + 1)]
W = [torch.randn(N, K, device=dev) / K ** 0.5 for _ in range(NL)]
b = torch.zeros(N, device=dev)

def chain(rows, epi=1):
    for r0 in range(0, P, rows):
        m = min(rows, P - r0)
        for l in range(NL):
            g = GemmNT(addr(H[l]) + r0 * K * 4, K, addr(W[l]), K, m, N, K, addr(H[l + 1]) + r0 * N * 4, N, 0, 0, addr(b), addr(H[l + 1]) + r0 * N * 4, N,
                       0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi)
            L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")

def time_it(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for epi in (1, 3):
    for rows in (P, 270336, 135168, 65536, 32768, P):
        ms = time_it(lambda: chain(rows, epi))
        print(f"epi={epi} rows per block {rows:7d}: {ms:7.3f} ms per 8-layer chain  {2.0*P*N*K*NL/ms/1e9:6.1f} TFLOP/s", flush=True)
