import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch
from bench_gemm import nt, time_it, lib, L, dev
print("ABL", os.environ.get("NU_NT_ABL"), "GRID", os.environ.get("NU_NT_GRID"))
nt(524288, 256, 256, 7)
nt(524288, 256, 1024, 7)
if not os.environ.get("NU_NT_ABL"):
    out = torch.zeros(4, device=dev)
    for blocks in (256, 512):
        iters = 20000
        ms = time_it(lambda: lib.nu_debug_mfma_peak(L.ptr(out), blocks, iters, L.stream()), iters=5)
        fl = blocks * 4 * iters * 4 * (32 * 32 * 2 * 2)
        print(f"mfma peak blocks={blocks}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s")
