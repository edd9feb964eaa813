"""Development aid: the no-gradient SDF forward, fused kernel (csrc/fused_sdf.hip) vs the layered path, by point count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from nu_nerf_amd.engine import addr
from test_stage1_gpu import make_net

dev = torch.device('cuda:0')
net = make_net(dev)
eng = net.engine()
eng.pack()
FL = 2 * (64 * 256 + 6 * 256 * 256 + 256 * 217 + 256)      # padded MACs x 2 per point


def time_it(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for P in (2048, 8192, 16384, 32768, 65536, 262144):
    X = (torch.rand(P, 3, device=dev) * 2 - 1) * 0.9
    res = {}
    for name, on in (("layered", False), ("fused", True)):
        eng._fused_sdf = on
        res[name] = time_it(lambda: eng.sdf_forward(addr(X), 3, P, keep=False, want_feat=False))
    print(f"P={P:7d}: layered {res['layered']:8.1f} us ({P * FL / res['layered'] / 1e6:6.1f} TFLOP/s)   fused {res['fused']:8.1f} us "
          f"({P * FL / res['fused'] / 1e6:6.1f} TFLOP/s)   NU_FUSED_SDF_TM={os.environ.get('NU_FUSED_SDF_TM', 'rule')}")
