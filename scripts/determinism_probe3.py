"""Development aid: the same batch through forward + backward several times on ONE network (no optimizer step); lists the parameters
whose gradient bits differ between repetitions.  usage: determinism_probe3.py <mlp_dtype> [rays] [reps]"""
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from nu_nerf_amd.loss import fused_stage1_loss, name2loss, SPHEREPOT_LOSSES  # noqa: E402
from nu_nerf_amd.params import init_stage1_params  # noqa: E402
from nu_nerf_amd.renderer import NeROShapeRenderer  # noqa: E402
from nu_nerf_amd.synthetic import make_rays  # noqa: E402

md = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device('cuda:0')
cfg = bench.build_cfg(R)
cfg['mlp_dtype'] = md
net = NeROShapeRenderer(cfg, training=False)
net.load_param_dict(init_stage1_params(6033))
net = net.to(dev)
losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
b = {k: torch.from_numpy(v).to(dev)[:R] for k, v in make_rays(R, seed=6033).items() if k != 'idxs'}
rand = (torch.rand(R, 1, device=dev), torch.rand(R, 1, device=dev)) if False else None
ref = None
for i in range(reps):
    torch.manual_seed(1234)
    for p in net.parameters():
        p.grad = None
    total, log, out = fused_stage1_loss(net, b, 20000, losses)
    total.backward()
    torch.cuda.synchronize()
    cur = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    cur['__rgb'] = out['ray_rgb'].detach().clone()
    if i < 2:                      # warm-up (allocator capacity classes, lazy initialisations)
        continue
    if ref is None:
        ref = cur
        continue
    bad = []
    for n in ref:
        if n in cur and not torch.equal(ref[n], cur[n]):
            d = float((ref[n].double() - cur[n].double()).norm() / (ref[n].double().norm() + 1e-30))
            bad.append((n, d))
    import collections
    pre = collections.Counter(n.split('.')[0] + '.' + (n.split('.')[1] if '.' in n else '') for n, _ in bad)
    print('rep', i, 'differing:', len(bad), dict(pre), 'max %.1e' % max([d for _, d in bad] + [0.0]), flush=True)
