"""GPU parity of the stage-2 renderer (HIP LBVH + HIP MLP ops with input gradients + torch glue) against the vectors the
reference's own Stage2Renderer produced (oracle/gen_golden_stage2.py): refraction geometry, sample placement, per-ray RGB,
losses and the gradients of every trained parameter incl. the IoR network."""
import numpy as np
import pytest
import torch

from helpers import golden, rel_err

pytestmark = pytest.mark.gpu


def build(gpu):
    from nu_nerf_amd.stage2 import Stage2Renderer
    from nu_nerf_amd.params import init_stage1_params, init_stage2_params, randomize_for_parity
    from nu_nerf_amd.lbvh import icosphere
    s1 = randomize_for_parity(init_stage1_params(6033), seed=1)
    p2 = randomize_for_parity(init_stage2_params(6033, 7044, {'sphere_direction': False}), seed=3)
    for k, v in s1.items():
        p2['stage1_network.' + k] = v
        p2['color_network.stage1_network.' + k] = v
    cfg = {'name': 's2', 'network': 'stage2', 'is_nerf': True, 'shader_config': {'sphere_direction': False, 'human_light': False},
           'eikonal_weight': 0.02, 'freeze_inv_s_step': 5000,
           'stage1_cfg': {'is_nerf': True, 'apply_occ_loss': True, 'occ_loss_step': 15000, 'freeze_inv_s_step': 15000},
           'stage1_mesh_arrays': icosphere(3, 0.5)}
    net = Stage2Renderer(cfg, training=False)
    assert list(net.state_dict().keys()) == list(p2.keys())
    net.load_param_dict(p2)
    return net.to(gpu), cfg


def test_stage2_train_step_vs_reference_golden(gpu):
    from nu_nerf_amd.loss import name2loss, total_loss
    g = golden("stage2_step6000_r24.npz")
    net, cfg = build(gpu)
    assert [str(k) for k in g['state_dict_keys']] == list(net.state_dict().keys())
    step = int(g['step'])
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    out = net.train_step_rays(batch, step)
    total, log = total_loss(out, [name2loss[n](cfg) for n in ('eikonal', 'std', 'nerf_render')], step)
    total.backward()
    assert np.array_equal(out['tir_mask'].cpu().numpy(), g['out_tir_mask'])
    np.testing.assert_allclose(out['_paths'][0].detach().cpu().numpy(), g['path0'], rtol=1e-5, atol=1e-5)
    d1 = np.abs(out['_paths'][1].detach().cpu().numpy() - g['path1'])
    assert (d1 < 1e-5).mean() > 0.95 and d1.max() < 5e-3            # inverse-CDF placement, see test_oracle_golden
    np.testing.assert_allclose(out['_ior_ratios'][0].detach().cpu().numpy(), g['ior0'], rtol=1e-5)
    np.testing.assert_allclose(out['_directions'][1].detach().cpu().numpy(), g['dir1'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out['ray_rgb'].detach().cpu().numpy(), g['out_ray_rgb'], rtol=1e-4, atol=1e-5)
    for k in g:
        if k.startswith('term_'):
            np.testing.assert_allclose(float(torch.mean(log[k[5:]]).detach()), float(g[k]), rtol=5e-4, err_msg=k)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=5e-5)
    named = dict(net.named_parameters())
    names = [str(n) for n in g['grad_names']]
    for n, ref_norm in zip(names, g['grad_norms']):
        assert named[n].grad is not None, n
        assert abs(float(named[n].grad.double().norm()) - ref_norm) <= 5e-3 * ref_norm + 1e-10, (n, float(named[n].grad.norm()), ref_norm)
    for n, p in named.items():
        if n not in names:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n
    for k in g:
        if k.startswith('grad__'):
            assert rel_err(named[k[6:]].grad.cpu(), g[k]) < 2e-2, k


def test_stage2_validation_render_vs_reference_golden(gpu):
    """render(..., is_train=False) -- test_step's per-chunk call (renderer_zerothick.py:1238-1240) -- against the reference's
    outputs on the fixture's rays: RGB, TIR mask and the validation images of the first surface; then the trainer protocol
    (forward({'step'}) / forward({'index','eval','step'})) on the module's own ray store."""
    g = golden("stage2_step6000_r24.npz")
    net, cfg = build(gpu)
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    with torch.no_grad():
        whole = net.render_eval(batch, int(g['step']))
        parts = net.render_eval(batch, int(g['step']), chunk=10)
    assert np.array_equal(whole['tir_mask'].cpu().numpy(), g['eval_tir_mask'])
    for k in ('ray_rgb', 'normal', 'specular_color', 'specular_light', 'specular_ref'):
        np.testing.assert_allclose(whole[k].cpu().numpy(), g['eval_' + k], rtol=1e-4, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(parts[k].cpu().numpy(), whole[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    from nu_nerf_amd.stage2 import name2renderer
    cfg2 = dict(cfg, database_name='synthetic/8192', train_ray_num=256, test_ray_num=512, synthetic_hw=48, downsample_ratio=0.5)
    net2 = name2renderer['stage2'](cfg2, training=True).to(gpu)
    out = net2({'step': 6000})
    assert out['ray_rgb'].shape == (256, 3) and out['loss_rgb'].requires_grad
    with torch.no_grad():
        ev = net2({'index': 1, 'eval': True, 'step': 0})
    assert ev['ray_rgb'].shape == (24, 24, 3) and ev['normal'].shape == (576, 3) and torch.isfinite(ev['ray_rgb']).all()
