"""Pin the stage-2 CPU oracle (oracle/stage2_oracle.py) against vectors produced by the reference's own Stage2Renderer
(oracle/gen_golden_stage2.py: reference code under shims, OptiX scene replaced by the brute-force scene)."""
import numpy as np
import pytest
import torch

from helpers import golden, rel_err
from oracle import stage1_oracle as O
from oracle import stage2_oracle as O2


def stage2_params(requires_grad=True, sphere_direction=False):
    from nu_nerf_amd.params import init_stage1_params, init_stage2_params, randomize_for_parity
    s1 = randomize_for_parity(init_stage1_params(6033, sphere_direction=sphere_direction), seed=1)
    p2 = randomize_for_parity(init_stage2_params(6033, 7044, {'sphere_direction': sphere_direction}), seed=3)
    out = {}
    for k, v in p2.items():
        if k.startswith('color_network.stage1_network.'):
            continue                     # alias of stage1_network.* in the reference's state_dict
        if k.startswith('stage1_network.'):
            v = s1[k[len('stage1_network.'):]]
        t = torch.from_numpy(np.ascontiguousarray(v))
        if requires_grad and not k.endswith('FG_LUT'):
            t.requires_grad_(True)
        out[k] = t
    return out


STAGE2_CFG = dict(O.DEFAULT_CFG, eikonal_weight=0.02, freeze_inv_s_step=5000, sphere_direction=False, refrac_freq=6)
# the two reference-generated fixtures (oracle/gen_golden_stage2.py VARIANTS): configs/stage2/nerf/*.yaml and the real-capture
# combination of configs/stage2/real/eikonal_wineglass.yaml:5-13 (`sphere_direction: true`, eikonal_weight 0.1, freeze_inv_s_step 15000)
VARIANTS = {'nerf': ("stage2_step6000_r24.npz", STAGE2_CFG),
            'real': ("stage2_real_step6000_r24.npz", dict(STAGE2_CFG, eikonal_weight=0.1, freeze_inv_s_step=15000, sphere_direction=True))}


def test_stage2_state_dict_inventory():
    from nu_nerf_amd.params import init_stage2_params
    g = golden("stage2_step6000_r24.npz")
    assert [str(k) for k in g['state_dict_keys']] == list(init_stage2_params().keys())


@pytest.mark.parametrize("variant", ["nerf", "real"])
def test_stage2_train_step_vs_reference(variant):
    from nu_nerf_amd.lbvh import icosphere
    fixture, STAGE2_CFG = VARIANTS[variant]
    g = golden(fixture)
    params = stage2_params(sphere_direction=STAGE2_CFG['sphere_direction'])
    V, Fc = icosphere(3, 0.5)
    scene = O2.BruteScene(V, Fc)
    step = int(g['step'])
    total, terms, out = O2.train_step(params, STAGE2_CFG, scene, torch.from_numpy(g['rays_o']), torch.from_numpy(g['rays_d']),
                                      torch.from_numpy(g['rgbs']), step)
    assert np.array_equal(out['tir_mask'].numpy(), g['out_tir_mask'])
    np.testing.assert_allclose(out['paths'][0].detach().numpy(), g['path0'], rtol=1e-5, atol=1e-5)
    d1 = np.abs(out['paths'][1].detach().numpy() - g['path1'])
    assert (d1 < 1e-5).mean() > 0.97 and d1.max() < 2e-3            # inverse-CDF placement, see test_oracle_golden
    np.testing.assert_allclose(out['ior_ratios'][0].detach().numpy(), g['ior0'], rtol=1e-5)
    np.testing.assert_allclose(out['directions'][1].detach().numpy(), g['dir1'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out['ray_rgb'].detach().numpy(), g['out_ray_rgb'], rtol=1e-4, atol=2e-5)
    for k in g:
        if k.startswith('term_'):
            np.testing.assert_allclose(float(torch.mean(terms[k[5:]]).detach()), float(g[k]), rtol=2e-4, err_msg=k)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=2e-5)
    total.backward()
    names = [str(n) for n in g['grad_names']]
    for n, ref_norm in zip(names, g['grad_norms']):
        assert params[n].grad is not None, n
        assert abs(float(params[n].grad.double().norm()) - ref_norm) <= 3e-3 * ref_norm + 1e-10, (n, float(params[n].grad.norm()), ref_norm)
    for n, v in params.items():
        if n not in names and not n.endswith('FG_LUT'):
            assert v.grad is None or float(v.grad.abs().sum()) == 0.0, n
    for k in g:
        if k.startswith('grad__'):
            assert rel_err(params[k[6:]].grad, g[k]) < 1e-2, k
