"""Development aid: how far ahead of the GPU does the host run?  Per step: time until everything is enqueued vs time until
the GPU is done.  (One mid-step sync exists: the inner-point count.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nu_nerf_amd.renderer import NeROShapeRenderer
from nu_nerf_amd.params import init_stage1_params
from nu_nerf_amd.synthetic import make_rays
from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
from nu_nerf_amd.train_glue import FusedAdam

dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = bench.build_cfg(R)
net = NeROShapeRenderer(cfg, training=False)
net.load_param_dict(init_stage1_params(6033))
net = net.to(dev)
losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
opt = FusedAdam(net.parameters(), lr=1e-3)
pool = {k: torch.from_numpy(v).to(dev) for k, v in make_rays(R * 16, seed=6033).items() if k != 'idxs'}
eng = net.engine()
marks = {}
orig_rf = eng.render_forward


def rf(*a, **k):
    marks['fwd_enter'] = time.perf_counter()
    out = orig_rf(*a, **k)
    marks['fwd_exit'] = time.perf_counter()
    return out


eng.render_forward = rf
rows = []
for it in range(14):
    b = {k: v[(it % 16) * R:((it % 16) + 1) * R] for k, v in pool.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = net.train_step_rays(b, 20000 + it)
    t_f = time.perf_counter()
    total, _ = total_loss(out, losses, 20000 + it)
    total.backward()
    opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it >= 4:
        rows.append((1e3 * (marks['fwd_enter'] - t0), 1e3 * (marks['fwd_exit'] - marks['fwd_enter']), 1e3 * (t_f - t0), 1e3 * (t1 - t_f), 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
a = np.array(rows)
print("ms: sampler-enqueue  render_forward(host incl. count sync)  forward-total  loss+backward+adam enqueue  all enqueued  GPU done")
print(np.round(a.mean(0), 2))
print("host slack at the end of the step (GPU done - all enqueued):", round(float((a[:, 5] - a[:, 4]).mean()), 2), "ms")
