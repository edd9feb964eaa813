"""Development aid: shader clock a bare fp32-MFMA loop holds (what "peak" means on the box at hand)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch
from bench_gemm import lib, L, dev, time_it
lib.nu_debug_clock_mhz.restype = ctypes.c_double
out = torch.zeros(16, device=dev)
for blocks in (256, 512, 1024):
    it = 20000
    ms = time_it(lambda: lib.nu_debug_mfma_peak(ctypes.c_void_p(out.data_ptr()), blocks, it, L.stream()), iters=5)
    fl = blocks * 4 * it * 4 * 4096.0
    print(f"bare mfma blocks={blocks}: {fl/ms/1e9:6.1f} TFLOP/s  clock {lib.nu_debug_clock_mhz():7.1f} MHz")
