import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle.lbvh_oracle import brute_force_closest_hit
from nu_nerf_amd.lbvh import LBVH, icosphere
from test_lbvh_gpu import _rays
dev = torch.device('cuda:0')
V, F = icosphere(0, 0.5)
bvh = LBVH(torch.from_numpy(V).to(dev), torch.from_numpy(F).to(dev))
rays = _rays(8192, seed=0)
hit, idx, t = bvh.intersect(torch.from_numpy(rays).to(dev), return_t=True)
ohit, oidx, ot = brute_force_closest_hit(V, F, rays)
t = t.cpu().numpy()
bad = np.nonzero(t.view(np.uint32) != ot.view(np.uint32))[0]
print("mismatch", len(bad), "of", len(t), "idx equal", np.array_equal(idx.cpu().numpy(), oidx))
f32 = np.float32
for i in bad[:6]:
    o, d = rays[i, :3], rays[i, 3:]
    a, b, c = F[oidx[i]]
    v0, v1, v2 = V[a], V[b], V[c]
    e1, e2 = v1 - v0, v2 - v0
    pv = np.cross(d.astype(np.float64), e2.astype(np.float64)); det = e1.astype(np.float64) @ pv
    tv = (o - v0).astype(np.float64); qv = np.cross(tv, e1.astype(np.float64)); t64 = (e2.astype(np.float64) @ qv) / det
    print(i, "gpu", repr(t[i]), "oracle", repr(ot[i]), "f64", t64, "ulps", int(t[i].view(np.int32)) - int(ot[i].view(np.int32)), "d", d, "det", det)
