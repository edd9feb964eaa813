"""Development aid: per-step hashes of the loss and of every parameter after each optimizer step of the bench workload, to find the
first step at which two runs of a mode diverge.  usage: determinism_probe.py <mlp_dtype> <steps> [rays]"""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from nu_nerf_amd.loss import fused_stage1_loss, name2loss, SPHEREPOT_LOSSES  # noqa: E402
from nu_nerf_amd.params import init_stage1_params  # noqa: E402
from nu_nerf_amd.renderer import NeROShapeRenderer  # noqa: E402
from nu_nerf_amd.synthetic import make_rays  # noqa: E402
from nu_nerf_amd.train_glue import FusedAdam  # noqa: E402

md, steps = sys.argv[1], int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = torch.device('cuda:0')
torch.manual_seed(6033)
cfg = bench.build_cfg(R)
cfg['mlp_dtype'] = md
net = NeROShapeRenderer(cfg, training=False)
net.load_param_dict(init_stage1_params(6033))
net = net.to(dev)
losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
pool = {k: torch.from_numpy(v).to(dev) for k, v in make_rays(R * 8, seed=6033).items() if k != 'idxs'}
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
for i in range(steps):
    b = {k: v[(i % 8) * R:(i % 8 + 1) * R] for k, v in pool.items()}
    opt.zero_grad(set_to_none=True)
    total, log, out = fused_stage1_loss(net, b, 20000 + i, losses)
    total.backward()
    gh = hashlib.sha256()
    for n, p in sorted(net.named_parameters()):
        if p.grad is not None:
            gh.update(p.grad.detach().cpu().numpy().tobytes())
    opt.step()
    print(i, repr(float(total.detach())), hashlib.sha256(out['ray_rgb'].detach().cpu().numpy().tobytes()).hexdigest()[:10], gh.hexdigest()[:10],
          net.engine().last_ctx['P_in'], flush=True)
