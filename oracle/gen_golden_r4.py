"""Round-4 golden vectors from the REFERENCE itself (build container only; shims and conventions of oracle/gen_golden.py).

Two stage-1 training steps of the reference's own NeROShapeRenderer (network/renderer_zerothick.py) at sizes the earlier
fixtures do not cover:

  train_default_sampling_step20000_r24.npz   the DEFAULT sampling of renderer_zerothick.py:110-117 -- 64 coarse + 64 importance
                                             + 32 background samples per ray -- (every earlier reference step is 32 + 32 + 16),
                                             24 rays, step 20000 (occlusion + outer-regulariser losses on, inv_s trainable)
  train_config0_step0_r256.npz               BASELINE.json configs[0] at its stated size: 256 rays, 32 + 32 + 32 samples, step 0
                                             (init-SDF regulariser on); per-ray outputs, loss terms and gradient norms only

Usage:  python oracle/gen_golden_r4.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import install_shims, to_t, OUT   # noqa: E402

BASE_CFG = {'name': 'golden', 'network': 'shape', 'database_name': 'nerf/spherepot', 'apply_occ_loss': True,
            'occ_loss_step': 15000, 'is_nerf': True, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1,
            'loss': ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'mask', 'outer_reg']}


def run_step(fname, sampling, R, step, ray_seed, full_grads=()):
    from network.renderer_zerothick import NeROShapeRenderer  # reference
    from network.loss import name2loss
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    from nu_nerf_amd.synthetic import make_rays, make_jitter
    cfg = dict(BASE_CFG, **sampling)
    net = NeROShapeRenderer(cfg, training=False)
    print("load_state_dict:", net.load_state_dict(to_t(randomize_for_parity(init_stage1_params(6033), seed=1)), strict=True))
    losses = [name2loss[n](cfg) for n in cfg['loss']]
    n_bg = net.cfg['n_bg_samples']
    rays = make_rays(R, seed=ray_seed)
    o, d, rgbs = (torch.from_numpy(rays[k]) for k in ('rays_o', 'rays_d', 'rgbs'))
    u1, u2 = make_jitter(R, n_bg, seed=ray_seed + 7)
    draws = [torch.from_numpy(u1), torch.from_numpy(u2)]
    real_rand = torch.rand

    def fake_rand(*a, **k):
        t = draws.pop(0)
        shape = list(a[0]) if len(a) == 1 and isinstance(a[0], (list, tuple)) else list(a)
        assert list(t.shape) == shape, (t.shape, shape)
        return t
    torch.rand = fake_rand
    try:
        dn = torch.nn.functional.normalize(d, dim=-1)
        near, far = torch.full((R, 1), 0.8), torch.full((R, 1), 4.5)
        net.zero_grad()
        z = net.sample_ray(o, dn, near, far, 1.0)
        outputs = net.render_core(o, dn, z, torch.zeros(R, 3, 4), cos_anneal_ratio=net.get_anneal_val(step), step=step,
                                  is_train=True, is_nerf=True)
    finally:
        torch.rand = real_rand
    outputs['loss_rgb'] = net.compute_rgb_loss(outputs['ray_rgb'], rgbs)
    log = {}
    for ls in losses:
        log.update(ls(outputs, {}, step))
    total = 0
    for k, v in log.items():
        if k.startswith('loss'):
            total = total + torch.mean(v)
    total.backward()
    res = {'rays_o': rays['rays_o'], 'rays_d': rays['rays_d'], 'rgbs': rays['rgbs'], 'u1': u1, 'u2': u2, 'step': np.asarray(step),
           'z_vals': z.numpy(), 'total_loss': total.detach().numpy(),
           'sampling': np.asarray([net.cfg['n_samples'], net.cfg['n_importance'], n_bg])}
    for k in ('ray_rgb', 'acc', 'color_bkgr', 'color_spec', 'loss_rgb'):
        if k in outputs:
            res['out_' + k] = outputs[k].detach().numpy()
    res['out_gradient_error_mean'] = outputs['gradient_error'].detach().mean().numpy()
    res['n_inner'] = np.asarray(outputs['gradient_error'].numel())
    for k, v in log.items():
        if k.startswith('loss'):
            res['term_' + k] = torch.mean(v).detach().numpy()
    gn = {name: prm.grad for name, prm in net.named_parameters() if prm.grad is not None}
    res['grad_names'] = np.asarray(sorted(gn.keys()))
    res['grad_norms'] = np.asarray([float(gn[k].double().norm()) for k in sorted(gn.keys())])
    for k in full_grads:
        if k in gn:
            res['grad__' + k] = gn[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, fname), **res)
    print(fname, "loss", float(total), {k: float(v) for k, v in res.items() if k.startswith('term_')}, "samples per ray", z.shape[1],
          "inner points", int(res['n_inner']), "of", R * z.shape[1])


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run_step("train_default_sampling_step20000_r24.npz", {}, 24, 20000, ray_seed=600,
             full_grads=('sdf_network.lin8.weight_v', 'color_network.roughness_predictor.6.bias', 'deviation_network.variance',
                         'outer_nerf.rgb_linear.bias'))
    run_step("train_config0_step0_r256.npz", {'n_samples': 32, 'n_importance': 32, 'n_bg_samples': 32}, 256, 0, ray_seed=700)


if __name__ == "__main__":
    main()
