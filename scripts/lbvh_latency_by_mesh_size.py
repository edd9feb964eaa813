import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nu_nerf_amd.lbvh import LBVH, icosphere
from nu_nerf_amd.synthetic import make_object_rays
dev = torch.device('cuda:0')
stats = 'stats' in os.environ.get('NU_NERF_LIB', '')
for sub in (2, 3, 4, 5, 6):
    V, F = icosphere(sub, 0.5)
    bvh = LBVH(torch.from_numpy(V).to(dev), torch.from_numpy(F).to(dev))
    r = make_object_rays(4096, seed=5)
    ray = torch.from_numpy(np.concatenate([r['rays_o'], r['rays_d'] / np.linalg.norm(r['rays_d'], axis=1, keepdims=True)], 1).astype(np.float32)).to(dev)
    if stats:
        hit, idx, t = bvh.intersect(ray, return_t=True)
        s = t.cpu().numpy()
        print(f"faces {F.shape[0]:6d}: records per ray mean {s.mean():.1f} max {s.max():.0f}")
    else:
        for _ in range(3): bvh.intersect(ray)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): bvh.intersect(ray)
        e1.record(); torch.cuda.synchronize()
        print(f"faces {F.shape[0]:6d}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per 4096-ray trace")
