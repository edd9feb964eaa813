"""Development aid: per-parameter gradient hashes of ONE bench step (see determinism_probe.py)."""
import hashlib
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from nu_nerf_amd.loss import fused_stage1_loss, name2loss, SPHEREPOT_LOSSES  # noqa: E402
from nu_nerf_amd.params import init_stage1_params  # noqa: E402
from nu_nerf_amd.renderer import NeROShapeRenderer  # noqa: E402
from nu_nerf_amd.synthetic import make_rays  # noqa: E402

md = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda:0')
torch.manual_seed(6033)
cfg = bench.build_cfg(R)
cfg['mlp_dtype'] = md
net = NeROShapeRenderer(cfg, training=False)
net.load_param_dict(init_stage1_params(6033))
net = net.to(dev)
losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
pool = {k: torch.from_numpy(v).to(dev) for k, v in make_rays(R, seed=6033).items() if k != 'idxs'}
total, log, out = fused_stage1_loss(net, pool, 20000, losses)
total.backward()
for n, p in sorted(net.named_parameters()):
    if p.grad is not None:
        g = p.grad.detach().cpu()
        print(n, hashlib.sha256(g.numpy().tobytes()).hexdigest()[:10], float(g.double().norm()), bool(torch.isfinite(g).all()))
