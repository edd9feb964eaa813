"""Report aid: LBVH build time and closest-hit throughput on config 3's stand-in mesh (icosphere, 20480 faces)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nu_nerf_amd.lbvh import LBVH, icosphere
from nu_nerf_amd.synthetic import make_object_rays, make_rays

dev = torch.device('cuda:0')
for sub in (5, 7):
    V, F = icosphere(sub, 0.5)
    Vt, Ft = torch.from_numpy(V).to(dev), torch.from_numpy(F).to(dev)
    bvh = LBVH(Vt, Ft)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bvh = LBVH(Vt, Ft)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 100
    for name, maker in (("object-aimed", make_object_rays), ("camera", make_rays)):
        n = 1 << 20
        r = maker(n, seed=5)
        ray = torch.from_numpy(np.concatenate([r['rays_o'], r['rays_d']], 1).astype(np.float32)).to(dev)
        hit, idx = bvh.intersect(ray)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hit, idx = bvh.intersect(ray)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"faces {F.shape[0]:7d}  build {build_ms:6.2f} ms  {name:13s} rays {n}: {ms:7.3f} ms  {n/ms/1e3:7.1f} Mrays/s  "
              f"hit frac {float(hit.mean()):.3f}  ray I/O {n*32/ms/1e6:6.1f} GB/s")

# the batch sizes a training step traces: rays per wave 64 (full waves) vs the library rule (16 / 32)
V, F = icosphere(5, 0.5)
bvh = LBVH(torch.from_numpy(V).to(dev), torch.from_numpy(F).to(dev))
for n in (1024, 4096, 8192, 16384, 65536):
    for name, maker in (("object-aimed", make_object_rays), ("camera", make_rays)):
        r = maker(n, seed=5)
        ray = torch.from_numpy(np.concatenate([r['rays_o'], r['rays_d'] / np.linalg.norm(r['rays_d'], axis=1, keepdims=True)], 1).astype(np.float32)).to(dev)
        for _ in range(3):
            hit, idx = bvh.intersect(ray)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            hit, idx = bvh.intersect(ray)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"small batch: faces {F.shape[0]}  {name:13s} rays {n:6d}: {us:7.1f} us per trace (NU_LBVH_QUAD={os.environ.get('NU_LBVH_QUAD', 'rule')}, NU_LBVH_RPW={os.environ.get('NU_LBVH_RPW', 'rule')})  {n/us:7.1f} Mrays/s")
