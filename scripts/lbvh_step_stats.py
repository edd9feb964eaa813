"""Development aid: records visited per ray by the closest-hit kernels (needs a library built with -DNU_LBVH_STATS, where t_out
carries the count; see scripts/README.md).  usage: NU_NERF_LIB=.../libnunerf_stats.so [NU_LBVH_QUAD=0] python3 scripts/lbvh_step_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nu_nerf_amd.lbvh import LBVH, icosphere
from nu_nerf_amd.synthetic import make_object_rays, make_rays

dev = torch.device('cuda:0')
V, F = icosphere(5, 0.5)
bvh = LBVH(torch.from_numpy(V).to(dev), torch.from_numpy(F).to(dev))
for name, maker in (("object-aimed", make_object_rays), ("camera", make_rays)):
    r = maker(4096, seed=5)
    ray = torch.from_numpy(np.concatenate([r['rays_o'], r['rays_d'] / np.linalg.norm(r['rays_d'], axis=1, keepdims=True)], 1).astype(np.float32)).to(dev)
    hit, idx, t = bvh.intersect(ray, return_t=True)
    s = t.cpu().numpy()
    print(f"{name:13s} NU_LBVH_QUAD={os.environ.get('NU_LBVH_QUAD', 'rule')}: records visited per ray: mean {s.mean():.1f}  median {np.median(s):.0f}  p90 {np.percentile(s, 90):.0f}  max {s.max():.0f}")
