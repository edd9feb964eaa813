"""Is a row of an NT GEMM the same bits whatever the batch it sits in?  (the sub-batch properties of the parity tests rest on it)
usage: python3 scripts/debug_rowwise_independence.py  -- prints, per arithmetic mode and epilogue, whether rows 0..99 of a 4000-row
launch equal the same rows computed in a 100-row launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nu_nerf_amd import _lib as L
from nu_nerf_amd.engine import GemmNT, addr

lib = L.load()
dev = torch.device('cuda:0')
torch.manual_seed(0)
for prec in (0, 2):
    for (N, K) in ((256, 256), (217, 256), (256, 64), (128, 288)):
        for epi in (7, 1, 2):
            M = 4000
            A = torch.randn(M, K, device=dev)
            B = torch.randn((N + 127) // 128 * 128, K, device=dev) / K ** 0.5
            b = torch.randn(N, device=dev)
            outs = []
            for m in (M, 100, 2049):
                C = torch.zeros(m, N, device=dev)
                g = GemmNT(addr(A), K, addr(B), K, m, N, K, addr(C), N, 0, 0, addr(b), 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi, prec)
                L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")
                torch.cuda.synchronize()
                outs.append(C[:100].clone())
            print(f"prec {prec} N={N} K={K} epi={epi}: rows equal across batch sizes: {torch.equal(outs[0], outs[1])} {torch.equal(outs[0], outs[2])}"
                  f"  max abs diff {float((outs[0] - outs[1]).abs().max()):.3e}")
