"""Development probe: every comparison of tests/test_stage2_thick_gpu.py printed instead of asserted."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import golden
from test_stage2_thick_gpu import build_thick
from nu_nerf_amd.loss import name2loss, total_loss

gpu = torch.device('cuda:0')
g = golden(os.environ.get("NU_THICK_FIXTURE", "stage2_thick_step6000_r24.npz"))
net, cfg = build_thick(gpu, g)
step = int(g['step'])
batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
if len(sys.argv) > 1 and sys.argv[1] == 'refz':
    # placement forced to the reference's: fractions recovered from the fixture's inner-segment nodes
    P1 = torch.from_numpy(g['path1']).to(gpu)
    def forced(n2, start, dirs, end):
        num = torch.linalg.norm(P1 - P1[:, :1], dim=-1)
        return num / num[:, -1:]
    net._upsample_inner = forced
out = net.train_step_rays(batch, step)
total, log = total_loss(out, [name2loss[n](cfg) for n in cfg['loss']], step)
total.backward()
print("tir", np.array_equal(out['tir_mask'].cpu().numpy(), g['out_tir_mask']))
paths = [p.detach().cpu().numpy() for p in out['_paths']]
for i in range(3):
    d = np.abs(paths[i] - g['path%d' % i])
    print("path", i, paths[i].shape, "max", d.max(), "frac<1e-5", (d < 1e-5).mean(), "rays with any >1e-4:", (d.reshape(d.shape[0], -1).max(1) > 1e-4).sum())
    if i == 1:
        print("  first/last node diff", d[:, 0].max(), d[:, -1].max(), "per-ray max", d.reshape(d.shape[0], -1).max(1))
for i in range(2):
    print("ior", i, np.abs(out['_ior_ratios'][i].detach().cpu().numpy() - g['ior%d' % i]).max(),
          "normal", np.abs(out['_normals'][i].detach().cpu().numpy() - g['normal_mesh%d' % i]).max())
for i in range(3):
    print("dir", i, np.abs(out['_directions'][i].detach().cpu().numpy() - g['dir%d' % i]).max())
rgb = out['ray_rgb'].detach().cpu().numpy()
print("rgb max abs", np.abs(rgb - g['out_ray_rgb']).max(), "rel", (np.abs(rgb - g['out_ray_rgb']) / (np.abs(g['out_ray_rgb']) + 1e-5)).max())
print("std", float(out['std']), float(g['out_std']), "gerr", out['gradient_error'].shape, g['out_gradient_error'].shape)
for k in g:
    if k.startswith('term_'):
        print(k, float(torch.mean(log[k[5:]]).detach()), float(g[k]))
print("total", float(total.detach()), float(g['total_loss']))
named = dict(net.named_parameters())
rows = []
for n, ref in zip([str(n) for n in g['grad_names']], g['grad_norms']):
    gr = named[n].grad
    got = float(gr.double().norm()) if gr is not None else float('nan')
    rows.append((abs(got - ref) / (ref + 1e-30) if ref > 0 else (0.0 if (gr is None or got == 0) else 1.0), n, got, ref))
rows.sort(reverse=True)
for r in rows[:12]:
    print("%.3e %s got %.6e ref %.6e" % r)
names = set(str(n) for n in g['grad_names'])
print("extra grads:", [n for n, p in named.items() if n not in names and p.grad is not None and float(p.grad.abs().sum()) > 0][:5])
