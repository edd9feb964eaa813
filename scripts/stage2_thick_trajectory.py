import sys, os, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from nu_nerf_amd.params import init_stage1_params, init_stage2_thick_own_params
from nu_nerf_amd.lbvh import icosphere
from nu_nerf_amd.synthetic import make_object_rays
from nu_nerf_amd.loss import name2loss, total_loss
from nu_nerf_amd.train_glue import FusedAdam
from nu_nerf_amd.stage2_thick import Stage2Renderer
dev = torch.device('cuda:0')
s1 = init_stage1_params(6033)
cfg = {'name': 's2', 'network': 'stage2', 'is_nerf': False, 'shader_config': {'sphere_direction': False, 'human_light': False},
       'eikonal_weight': 0.02, 'freeze_inv_s_step': 5000, 'get_mask': False,
       'stage1_cfg': {'is_nerf': False, 'apply_occ_loss': True, 'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'get_mask': False},
       'stage1_mesh_arrays': icosphere(5, 0.5)}
net = Stage2Renderer(cfg, training=False)
net.load_param_dict(init_stage2_thick_own_params(7044, net.color_network_inner.cfg))
net.load_param_dict({'stage1_network.' + k: v for k, v in s1.items()})
net = net.to(dev)
losses = [name2loss[n](cfg) for n in ('eikonal', 'std', 'nerf_render')]
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-3)
R = 1024
pool = {k: torch.from_numpy(v).to(dev) for k, v in make_object_rays(R * 40, seed=6033).items() if k in ('rays_o', 'rays_d', 'rgbs')}
for i in range(36):
    b = {k: v[i * R:(i + 1) * R] for k, v in pool.items()}
    opt.zero_grad(set_to_none=True)
    out = net.train_step_rays(b, 6000 + i)
    total, _ = total_loss(out, losses, 6000 + i)
    total.backward()
    gi = float(sum(p.grad.double().pow(2).sum() for p in net.IORs_pred.parameters() if p.grad is not None) ** 0.5)
    gt = float(sum(p.grad.double().pow(2).sum() for p in net.thickness_pred.parameters() if p.grad is not None) ** 0.5)
    opt.step()
    if i < 6 or i % 5 == 0:
        print(i, "loss %.7f entered %.4f tir_valid %.4f |g ior| %.4e |g thick| %.4e" % (float(total), out['_paths'][1].shape[0] / R if len(out['_paths']) > 1 else 0, float(out['tir_mask'].float().mean()), gi, gt))
