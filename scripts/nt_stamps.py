"""Development aid: per-chunk / per-epilogue stamps of one fp32 NT launch (needs profiles/r04/nt2_chunk_stamp_hooks.patch applied:
nu_debug_nt2_stamps).  Saves the raw array under gpurun_out/ for offline analysis."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nu_nerf_amd import _lib as L
from nu_nerf_amd.engine import GemmNT, addr
lib = L.load(); dev = torch.device('cuda:0')
NW = 512 * 16 * 20 + 1024

def run(M, N, K, epi):
    A = torch.randn(M, K, device=dev); B = torch.randn((N + 127) // 128 * 128, K, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev)
    H = torch.rand(M, N, device=dev); D = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev)
    g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), N, addr(C2), N, addr(b), addr(H), N, addr(D), N, addr(D), N,
               0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi)
    assert lib.nu_debug_nt2_stamps(None, 1) == 0
    for _ in range(8):
        L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")
    torch.cuda.synchronize()
    out = np.zeros(NW, dtype=np.uint64)
    assert lib.nu_debug_nt2_stamps(out.ctypes.data_as(ctypes.c_void_p), 0) == 0
    np.save(os.path.join(ROOT, 'gpurun_out', f'nt_chunk_stamps_{M}_{K}_{epi}.npy'), out)
    print('saved', M, K, epi)

for a in ((262144, 256, 256, 1), (262144, 256, 256, 2), (262144, 256, 1024, 7)):
    run(*a)
