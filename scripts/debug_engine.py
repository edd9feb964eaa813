"""Stage-by-stage GPU-vs-oracle error report (development aid; the assertions live in tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from helpers import parity_params, oracle_cfg, rel_err, golden
from oracle import stage1_oracle as O
from nu_nerf_amd.renderer import NeROShapeRenderer
from nu_nerf_amd.engine import addr
from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss

dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = {'is_nerf': True, 'n_samples': 32, 'n_importance': 32, 'n_bg_samples': 16, 'freeze_inv_s_step': 15000,
       'apply_occ_loss': True, 'occ_loss_step': 15000, 'eikonal_weight': 0.1}
net = NeROShapeRenderer(cfg, training=False)
net.load_param_dict(randomize_for_parity(init_stage1_params(6033), seed=1))
net = net.to(dev)
eng = net.engine()
eng.pack()
torch.cuda.synchronize()
P = parity_params(requires_grad=True)
ocfg = oracle_cfg()

def report(name, a, b):
    print(f"{name:40s} rel {rel_err(a.detach().cpu(), b.detach().cpu()):.3e}  max|d| {float((a.detach().cpu().double()-b.detach().cpu().double()).abs().max()):.3e}")

# ---- pack
W4 = O.wn_weight(P, 'sdf_network.lin4') / 2 ** 0.5
report("pack lin4", eng.sdf[4].Wp[0][:256, :256], W4)
report("pack lin4 T", eng.sdf[4].WpT[0][:256, :256].t(), W4)
# ---- SDF fwd + normal
g = np.random.Generator(np.random.PCG64(5))
pts = torch.from_numpy(g.uniform(-0.8, 0.8, (1000, 3)).astype(np.float32))
y = O.sdf_forward(P, pts); n = O.sdf_gradient(P, pts)
pd = pts.to(dev).contiguous()
a = eng.sdf_forward(addr(pd), 3, 1000, keep=True)
eng.sdf_normal(a)
report("sdf y", a['YX'][:, :257], y)
report("sdf n", a['n'], n)
# ---- SDF backward with cotangents
cy = torch.from_numpy(g.standard_normal((1000, 257)).astype(np.float32)); cn = torch.from_numpy(g.standard_normal((1000, 3)).astype(np.float32))
for v in P.values(): v.grad = None
((y * cy).sum() + (n * cn).sum()).backward()
flat = eng.zeros(eng.n_grad)
dYX = eng.zeros(1000, 288); dYX[:, :257] = cy.to(dev); dYX[:, 257:260] = 7.0
eng.sdf_backward(a, dYX, cn.to(dev).contiguous(), flat)
eng.unpack_grads(flat)
for l in range(9):
    for nm in ('weight_v', 'weight_g', 'bias'):
        k = f'sdf_network.lin{l}.{nm}'
        off, shape = eng.grad_views[k]
        report("sdf grad " + k, flat[off:off + P[k].numel()].view(shape), P[k].grad)

# ---- full step vs oracle on golden inputs
for name in ("train_step0_r48.npz", "train_step20000_r48.npz"):
    gd = golden(name)
    step = int(gd['step'])
    batch = {k: torch.from_numpy(gd[k]).to(dev) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(gd['u1']).to(dev), torch.from_numpy(gd['u2']).to(dev))
    net.zero_grad()
    out = net.train_step_rays(batch, step, rand=rand)
    z = net.sample_ray(batch['rays_o'], torch.nn.functional.normalize(batch['rays_d'], dim=-1), torch.full((48,), 0.8, device=dev), torch.full((48,), 4.5, device=dev), 1.0, rand)
    dz = (z.cpu() - torch.from_numpy(gd['z_vals'])).abs()
    print(name, "z_vals: max", float(dz.max()), "frac>1e-5", float((dz > 1e-5).float().mean()))
    report("ray_rgb", out['ray_rgb'], torch.from_numpy(gd['out_ray_rgb']))
    report("acc", out['acc'], torch.from_numpy(gd['out_acc']))
    report("color_bkgr", out['color_bkgr'], torch.from_numpy(gd['out_color_bkgr']))
    report("color_spec", out['color_spec'], torch.from_numpy(gd['out_color_spec']))
    if out['gradient_error'].shape[0] == gd['out_gradient_error'].shape[0]:
        report("gradient_error", out['gradient_error'], torch.from_numpy(gd['out_gradient_error']))
    else:
        print("gradient_error count", out['gradient_error'].shape, gd['out_gradient_error'].shape)
    losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
    total, log = total_loss(out, losses, step)
    print("total", float(total), "golden", float(gd['total_loss']), {k: float(torch.mean(v)) for k, v in log.items() if k.startswith('loss')})
    total.backward()
    names = [str(n) for n in gd['grad_names']]
    named = dict(net.named_parameters())
    worst = []
    for nme, rn in zip(names, gd['grad_norms']):
        gmine = named[nme].grad
        mine = float(gmine.double().norm()) if gmine is not None else float('nan')
        worst.append((abs(mine - rn) / (rn + 1e-30), nme, mine, rn))
    worst.sort(reverse=True)
    for w in worst[:12]:
        print("   gradnorm relerr %.3e %s mine %.4e ref %.4e" % w)
    for k in gd:
        if k.startswith('grad__') and named[k[6:]].grad is not None:
            report(k, named[k[6:]].grad, torch.from_numpy(gd[k]))
