"""Round-2 fixtures generated from the REFERENCE itself (build container only; same shims as oracle/gen_golden.py).

  core_<tag>.npz            per-SAMPLE quantities of the three golden train steps (same seeds as train_<tag>.npz, so the
                            per-ray outputs there and the per-sample ones here belong to one run): the reference's alpha,
                            sampled colour and composite weights (renderer_zerothick.py:748-779), captured from its own
                            methods' return values (compute_sdf_alpha / compute_density_alpha / color_network) and the
                            transmittance torch.cumprod of :773.  With the fixture's z_vals fed to render_core the sampler's
                            last-bit noise is out of the picture and per-sample parity can be held to 1e-4.
  occ_cap_step20000_r48.npz the step-20000 run with occ_loss_max_pn below the number of near-surface points, so the
                            subsample branch (renderer_zerothick.py:708-714) runs; the torch.randperm draw is recorded.
  ray_batch_std.npz         _process_ray_batch / get_human_coordinate_poses of network/renderer.py:346-378 on seeded poses.

Usage:  python oracle/gen_golden_r2.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import install_shims, to_t, OUT   # noqa: E402


def capture_core(net, o, dn, z, step, is_nerf=True):
    """Run the reference's render_core (is_train=True) and record alpha / colour per sample from the return values of the
    reference's own methods, and the transmittance from the first cumprod of the composite."""
    rec = {'cumprod': []}
    real_sdf_alpha, real_density_alpha = net.compute_sdf_alpha, net.compute_density_alpha
    real_cumprod = torch.cumprod

    def sdf_alpha(*a, **k):
        out = real_sdf_alpha(*a, **k)
        rec['alpha_in'] = out[0].detach().clone()
        return out

    def density_alpha(*a, **k):
        out = real_density_alpha(*a, **k)
        rec['alpha_out'], rec['color_out'] = out[0].detach().clone(), out[1].detach().clone()
        return out

    def cumprod(*a, **k):
        out = real_cumprod(*a, **k)
        rec['cumprod'].append(out.detach().clone())
        return out
    hook = net.color_network.register_forward_hook(lambda m, i, out: rec.__setitem__('color_in', out[0].detach().clone()))
    net.compute_sdf_alpha, net.compute_density_alpha, torch.cumprod = sdf_alpha, density_alpha, cumprod
    try:
        R = o.shape[0]
        outputs = net.render_core(o, dn, z, torch.zeros(R, 3, 4), cos_anneal_ratio=net.get_anneal_val(step), step=step,
                                  is_train=True, is_nerf=is_nerf)
    finally:
        net.compute_sdf_alpha, net.compute_density_alpha, torch.cumprod = real_sdf_alpha, real_density_alpha, real_cumprod
        hook.remove()
    # the masks exactly as render_core forms them (:729-736)
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, dists[..., -1:]], -1)
    mid = z + dists * 0.5
    pts = o.unsqueeze(-2) + dn.unsqueeze(-2) * mid.unsqueeze(-1)
    inner = torch.norm(pts, dim=-1) <= 1.0
    alpha = torch.zeros_like(z)
    color = torch.zeros(*z.shape, 3)
    alpha[~inner], color[~inner] = rec['alpha_out'], rec['color_out']
    alpha[inner], color[inner] = rec['alpha_in'], rec['color_in']
    T = rec['cumprod'][0][..., :-1]                 # the composite's transmittance (:773); later cumprods belong to the bkgr pass
    weights = alpha * T
    assert torch.allclose(weights.sum(-1), outputs['acc'].detach(), rtol=0, atol=1e-6)
    return outputs, {'alpha': alpha.numpy(), 'sampled_color': color.numpy(), 'weights': weights.numpy(),
                     'inner_mask': inner.numpy().astype(np.uint8)}


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from network.renderer_zerothick import NeROShapeRenderer  # reference
    from network.loss import name2loss
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    from nu_nerf_amd.synthetic import make_rays, make_jitter

    base_cfg = {'name': 'golden', 'network': 'shape', 'database_name': 'nerf/spherepot', 'apply_occ_loss': True,
                'occ_loss_step': 15000, 'is_nerf': True, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1,
                'loss': ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'mask', 'outer_reg'],
                'n_samples': 32, 'n_importance': 32, 'n_bg_samples': 16}
    params = randomize_for_parity(init_stage1_params(6033), seed=1)

    def make_net(**over):
        cfg = dict(base_cfg)
        cfg.update(over)
        net = NeROShapeRenderer(cfg, training=False)
        net.load_state_dict(to_t(params), strict=True)
        return net, cfg, [name2loss[n](cfg) for n in cfg['loss']]

    def run(net, cfg, losses, R, step, ray_seed, perturb=True, perm_seed=None):
        rays = make_rays(R, seed=ray_seed)
        o, d, rgbs = (torch.from_numpy(rays[k]) for k in ('rays_o', 'rays_d', 'rgbs'))
        u1, u2 = make_jitter(R, cfg['n_bg_samples'], seed=ray_seed + 7)
        draws = [torch.from_numpy(u1), torch.from_numpy(u2)]
        real_rand, real_randperm = torch.rand, torch.randperm
        perms = []

        def fake_rand(*a, **k):
            return draws.pop(0)

        def fake_randperm(n, *a, **k):
            p = torch.from_numpy(np.random.Generator(np.random.PCG64(perm_seed)).permutation(n).astype(np.int64))
            perms.append(p)
            return p
        torch.rand = fake_rand
        if perm_seed is not None:
            torch.randperm = fake_randperm
        try:
            dn = torch.nn.functional.normalize(d, dim=-1)
            near, far = torch.full((R, 1), 0.8), torch.full((R, 1), 4.5)
            net.zero_grad()
            z = net.sample_ray(o, dn, near, far, 1.0 if perturb else 0.0)
            outputs, core = capture_core(net, o, dn, z, step)
        finally:
            torch.rand, torch.randperm = real_rand, real_randperm
        outputs['loss_rgb'] = net.compute_rgb_loss(outputs['ray_rgb'], rgbs)
        log = {}
        for ls in losses:
            log.update(ls(outputs, {}, step))
        total = 0
        for k, v in log.items():
            if k.startswith('loss'):
                total = total + torch.mean(v)
        total.backward()
        return rays, u1, u2, z, outputs, core, log, total, perms

    net, cfg, losses = make_net()
    for tag, R, step, seed, perturb in (("step0_r48", 48, 0, 100, True), ("step20000_r48", 48, 20000, 200, True),
                                        ("step500_r32_noperturb", 32, 500, 300, False)):
        rays, u1, u2, z, outputs, core, log, total, _ = run(net, cfg, losses, R, step, seed, perturb)
        old = dict(np.load(os.path.join(OUT, f"train_{tag}.npz"), allow_pickle=False))
        assert np.array_equal(old['z_vals'], z.numpy()) and np.array_equal(old['out_ray_rgb'], outputs['ray_rgb'].detach().numpy()), tag
        res = dict(core)
        res['z_vals'] = z.numpy()
        res['gradient_error'] = outputs['gradient_error'].detach().numpy()
        np.savez_compressed(os.path.join(OUT, f"core_{tag}.npz"), **res)
        print("core", tag, {k: v.shape for k, v in res.items()}, "inner", int(core['inner_mask'].sum()))

    # ---- occlusion-loss subsample branch ----
    net2, cfg2, losses2 = make_net(occ_loss_max_pn=12)
    rays, u1, u2, z, outputs, core, log, total, perms = run(net2, cfg2, losses2, 48, 20000, 200, True, perm_seed=77)
    assert len(perms) == 1, "the subsample branch did not run: lower occ_loss_max_pn"
    res = {'rays_o': rays['rays_o'], 'rays_d': rays['rays_d'], 'rgbs': rays['rgbs'], 'u1': u1, 'u2': u2, 'step': np.asarray(20000),
           'z_vals': z.numpy(), 'perm': perms[0].numpy(), 'occ_loss_max_pn': np.asarray(12),
           'out_loss_occ': outputs['loss_occ'].detach().numpy(), 'total_loss': total.detach().numpy()}
    gn = {n: p.grad for n, p in net2.named_parameters() if p.grad is not None}
    res['grad_names'] = np.asarray(sorted(gn.keys()))
    res['grad_norms'] = np.asarray([float(gn[k].double().norm()) for k in sorted(gn.keys())])
    for k in ('color_network.inner_weight.0.weight_v', 'color_network.inner_weight.6.bias'):
        res['grad__' + k] = gn[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "occ_cap_step20000_r48.npz"), **res)
    print("occ_cap: candidates", perms[0].numel(), "kept 12, loss_occ", float(outputs['loss_occ']), "total", float(total))

    # ---- real-capture ray construction (network/renderer.py:346-378) ----
    from network.renderer import NeROShapeRenderer as RefStd
    g = np.random.Generator(np.random.PCG64(91))
    n_img, R = 7, 40
    A = g.standard_normal((n_img, 3, 3))
    Q = np.stack([np.linalg.qr(a)[0] for a in A], 0)
    Q[np.linalg.det(Q) < 0, :, 0] *= -1
    t = g.uniform(-2, 2, (n_img, 3, 1))
    poses = np.concatenate([Q, t], -1).astype(np.float32)
    dirs = g.standard_normal((R, 3)).astype(np.float32)
    idxs = g.integers(0, n_img, (R, 1)).astype(np.int64)
    res = {'poses': poses, 'dirs': dirs, 'idxs': idxs}
    for fixed in (False, True):
        std = RefStd({'name': 'rb', 'network': 'shape', 'database_name': 'custom/x/720', 'is_nerf': False, 'fixed_camera': fixed, 'get_mask': False,
                      'shader_config': {'sphere_direction': True, 'human_light': False}}, training=False)
        tag = 'fixed' if fixed else 'free'
        try:
            hp = std.get_human_coordinate_poses(torch.from_numpy(poses.copy()))
        except RuntimeError as e:     # in-place write into an expanded tensor (renderer.py:354-355): one pose at a time is legal
            print("get_human_coordinate_poses on", n_img, "poses raised:", str(e).splitlines()[0], "-> pose by pose")
            hp = torch.cat([std.get_human_coordinate_poses(torch.from_numpy(poses[i:i + 1].copy())) for i in range(n_img)], 0)
        res['human_poses_' + tag] = hp.numpy()
        if not fixed:
            try:
                ro, rd, near, far, hpi = std._process_ray_batch({'dirs': torch.from_numpy(dirs), 'idxs': torch.from_numpy(idxs)},
                                                                torch.from_numpy(poses.copy()))
                res.update(rays_o=ro.numpy(), rays_d=rd.numpy(), near=near.numpy(), far=far.numpy(), human_poses_rays=hpi.numpy())
            except RuntimeError as e:
                print("_process_ray_batch raised:", str(e).splitlines()[0])
                # everything before the human-pose call is plain arithmetic on the inputs: take those lines' results via
                # the method's own pieces
                P = torch.from_numpy(poses.copy())
                ro = (P[:, :, :3].permute(0, 2, 1) @ -P[:, :, 3:])[torch.from_numpy(idxs)[..., 0], :, 0]
                rd = torch.nn.functional.normalize((P[torch.from_numpy(idxs)[..., 0], :, :3].permute(0, 2, 1)
                                                    @ torch.from_numpy(dirs).unsqueeze(-1))[..., 0], dim=-1)
                near, far = std.near_far_from_sphere(ro, rd)
                res.update(rays_o=ro.numpy(), rays_d=rd.numpy(), near=near.numpy(), far=far.numpy(),
                           human_poses_rays=hp.numpy()[idxs[:, 0]])
                res['process_ray_batch_restated'] = np.asarray(1)
    np.savez_compressed(os.path.join(OUT, "ray_batch_std.npz"), **res)
    print("ray_batch_std", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
