"""mlp_dtype 'bf16' (BASELINE config 4: real-capture code path, bf16 MLP GEMMs with fp32 accumulation).

The reference has no bf16 mode, so there is no reference vector to pin this against: the checker is the same step on the
exact-fp32 HIP path (itself pinned against the reference's golden vectors in test_stage1_gpu.py), with the tolerance
SURVEY 8(d) proposes for this config declared here: per-ray RGB within 2e-2 absolute."""
import os
import numpy as np
import pytest
import torch

# bf16 storage exists in the network-level C entries only (the default path); a whole-suite run with NU_PY_SEQ=1 skips this file
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('NU_PY_SEQ') == '1', reason='launch-by-launch sequencing has no bf16-storage mode')]


def _step(gpu, mlp_dtype, R=512, step=20000):
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.params import init_stage1_params
    from nu_nerf_amd.synthetic import make_object_rays, make_jitter
    from nu_nerf_amd.loss import name2loss, total_loss
    cfg = {'name': 'c4', 'network': 'shape', 'database_name': 'synthetic/64', 'is_nerf': False, 'apply_occ_loss': True,
           'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1, 'outer_reg_loss_weight': 0.1,
           'shader_config': {'sphere_direction': True, 'human_light': False, 'light_exp_max': 5.0},
           'n_samples': 64, 'n_importance': 64, 'n_bg_samples': 32, 'mlp_dtype': mlp_dtype}
    net = NeROShapeRenderer(cfg, training=False)
    net.load_param_dict(init_stage1_params(6033, sphere_direction=True))
    net = net.to(gpu)
    rays = make_object_rays(R, seed=77, aim_radius=0.9)
    batch = {k: torch.from_numpy(v).to(gpu) for k, v in rays.items()}
    u1, u2 = make_jitter(R, 32, seed=78)
    out = net.train_step_rays(batch, step, rand=(torch.from_numpy(u1).to(gpu), torch.from_numpy(u2).to(gpu)))
    losses = [name2loss[n](cfg) for n in ('nerf_render', 'eikonal', 'std', 'occ', 'outer_reg')]
    total, _ = total_loss(out, losses, step)
    total.backward()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    return out, float(total.detach()), grads, net


def test_bf16_step_within_declared_tolerance_of_fp32(gpu):
    o32, t32, g32, net32 = _step(gpu, 'fp32')
    o16, t16, g16, net16 = _step(gpu, 'bf16')
    assert net32.engine().bf16 == 0 and net16.engine().bf16 == 1
    rgb32, rgb16 = o32['ray_rgb'].detach(), o16['ray_rgb'].detach()
    assert float((rgb32 - rgb16).abs().max()) > 1e-6          # it really is a different arithmetic
    assert float((rgb32 - rgb16).abs().max()) <= 2e-2, float((rgb32 - rgb16).abs().max())
    assert float((rgb32 - rgb16).abs().mean()) <= 2e-3
    assert abs(t16 - t32) <= 2e-2 * abs(t32)
    # gradients: same set of parameters, close in norm and direction network by network
    assert set(g16) == set(g32)
    for pre in ('sdf_network.', 'outer_nerf.', 'color_network.'):
        a = torch.cat([g32[n].flatten() for n in sorted(g32) if n.startswith(pre)]).double()
        b = torch.cat([g16[n].flatten() for n in sorted(g16) if n.startswith(pre)]).double()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        assert cos > 0.98, (pre, cos)
        assert abs(float(b.norm() / a.norm()) - 1.0) < 0.1, (pre, float(b.norm() / a.norm()))


def test_mlp_dtype_is_validated(gpu):
    from nu_nerf_amd.renderer import NeROShapeRenderer
    net = NeROShapeRenderer({'name': 'x', 'network': 'shape', 'database_name': 'synthetic/8', 'mlp_dtype': 'fp8'},
                            training=False).to(gpu)
    with pytest.raises(ValueError):
        net.engine()


def test_bf16_step_vs_cpu_oracle_declared_tolerance(gpu):
    """BASELINE config 4's arithmetic against the CPU ORACLE (fp32, pinned to the reference): real-capture code path
    (is_nerf False, sphere_direction True), bf16 MLP GEMMs.  Declared tolerance (no reference counterpart exists):
    per-ray RGB within 2e-2 absolute, total loss within 2 %."""
    from nu_nerf_amd.renderer_std import NeROShapeRenderer      # the reference's real-capture configs use network/renderer.py
    from nu_nerf_amd.params import init_stage1_params
    from nu_nerf_amd.synthetic import make_object_rays, make_jitter
    from oracle import stage1_oracle as O
    cfg = {'name': 'c4', 'network': 'shape', 'database_name': 'synthetic/64', 'is_nerf': False, 'apply_occ_loss': True,
           'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1,
           'shader_config': {'sphere_direction': True, 'human_light': False, 'light_exp_max': 5.0},
           'n_samples': 64, 'n_importance': 32, 'n_bg_samples': 16, 'mlp_dtype': 'bf16'}
    arrays = init_stage1_params(6033, sphere_direction=True)
    net = NeROShapeRenderer(cfg, training=False)
    net.load_param_dict(arrays)
    net = net.to(gpu)
    R, step = 96, 20000
    rays = make_object_rays(R, seed=177, aim_radius=0.9)
    u1, u2 = make_jitter(R, 16, seed=178)
    batch = {k: torch.from_numpy(v).to(gpu) for k, v in rays.items()}
    out = net.train_step_rays(batch, step, rand=(torch.from_numpy(u1).to(gpu), torch.from_numpy(u2).to(gpu)))
    assert net.engine().bf16 == 1
    ocfg = dict(O.DEFAULT_CFG)
    ocfg.update(n_samples=64, n_importance=32, n_bg_samples=16, is_nerf=False, sphere_direction=True, light_exp_max=5.0)
    params = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in arrays.items()}
    o, d = torch.from_numpy(rays['rays_o']), torch.nn.functional.normalize(torch.from_numpy(rays['rays_d']), dim=-1)
    near, far = O.near_far_from_sphere(o, d)
    with torch.no_grad():
        z = O.sample_ray(params, ocfg, o, d, near, far, 1.0, (torch.from_numpy(u1), torch.from_numpy(u2)))
        oo = O.render_core(params, ocfg, o, d, z, step, O.get_anneal_val(ocfg, step), False, None, std=True)
    err = (out['ray_rgb'].detach().cpu() - oo['ray_rgb']).abs()
    assert float(err.max()) <= 2e-2, float(err.max())
    assert float(err.mean()) <= 3e-3
    l_hip = float(net.compute_rgb_loss(out['ray_rgb'].detach().cpu(), torch.from_numpy(rays['rgbs'])).mean())
    l_ora = float(O.rgb_loss(oo['ray_rgb'], torch.from_numpy(rays['rgbs'])).mean())
    assert abs(l_hip - l_ora) <= 2e-2 * l_ora


def test_config4_full_size_property_run(gpu):
    """BASELINE configs[3] at its full size: 8192 rays x 160 samples, real-capture path, bf16 GEMMs.  Size-independent
    properties: every ray of a 128-ray sub-batch renders bit-identically alone and inside the 8192-ray batch (rays are
    independent units; no tile-edge artefact at 1.3 M points), outputs are finite and in range, the backward of the full batch
    produces finite gradients for every trained parameter."""
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.params import init_stage1_params
    from nu_nerf_amd.synthetic import make_object_rays
    from nu_nerf_amd.loss import name2loss, total_loss
    cfg = {'name': 'c4', 'network': 'shape', 'database_name': 'synthetic/64', 'is_nerf': False, 'apply_occ_loss': True,
           'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1, 'outer_reg_loss_weight': 0.1,
           'shader_config': {'sphere_direction': True, 'human_light': False, 'light_exp_max': 5.0},
           'n_samples': 64, 'n_importance': 64, 'n_bg_samples': 32, 'mlp_dtype': 'bf16'}
    net = NeROShapeRenderer(cfg, training=False)
    net.load_param_dict(init_stage1_params(6033, sphere_direction=True))
    net = net.to(gpu)
    rays = make_object_rays(8192, seed=277, aim_radius=0.9)
    batch = {k: torch.from_numpy(v).to(gpu) for k, v in rays.items()}
    o, d = batch['rays_o'], torch.nn.functional.normalize(batch['rays_d'], dim=-1)
    near, far = net.near_far_from_sphere(o, d)
    with torch.no_grad():
        full = net.render(o, d, near, far, perturb_overwrite=0, cos_anneal_ratio=0.4, step=20000, is_nerf=False)
        sub = net.render(o[4000:4128], d[4000:4128], near[4000:4128], far[4000:4128], perturb_overwrite=0, cos_anneal_ratio=0.4,
                         step=20000, is_nerf=False)
    torch.testing.assert_close(full['ray_rgb'][4000:4128], sub['ray_rgb'], rtol=0, atol=0)
    assert bool(torch.isfinite(full['ray_rgb']).all()) and float(full['ray_rgb'].min()) >= 0.0 and float(full['ray_rgb'].max()) <= 1.0
    out = net.train_step_rays(batch, 20000)
    total, _ = total_loss(out, [name2loss[n](cfg) for n in ('nerf_render', 'eikonal', 'std', 'occ', 'outer_reg')], 20000)
    total.backward()
    assert bool(torch.isfinite(total))
    n_grad = 0
    for n, p in net.named_parameters():
        if p.grad is not None:
            n_grad += 1
            assert bool(torch.isfinite(p.grad).all()), n
    assert n_grad >= 128
