"""GPU parity of the HIP hot path (through the C ABI) against the CPU oracle and the golden vectors generated from
the reference.  fp32; the north star asks per-ray RGB/weights within 1e-4 relative -- tolerances are written at
each check.  Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from helpers import golden, oracle_cfg, parity_params, rel_err
from oracle import stage1_oracle as O

pytestmark = pytest.mark.gpu

CFG = {'is_nerf': True, 'n_samples': 32, 'n_importance': 32, 'n_bg_samples': 16, 'freeze_inv_s_step': 15000,
       'apply_occ_loss': True, 'occ_loss_step': 15000, 'eikonal_weight': 0.1}


def make_net(gpu, cfg=CFG, randomized=True):
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    arrays = init_stage1_params(6033)
    if randomized:
        arrays = randomize_for_parity(arrays, seed=1)
    net = NeROShapeRenderer(cfg, training=False)
    net.load_param_dict(arrays)
    return net.to(gpu)


@pytest.fixture(scope="module")
def net(gpu):
    n = make_net(gpu)
    n.engine().pack()
    return n


@pytest.fixture(scope="module")
def ops():
    return golden("ops.npz")


def test_native_library_is_loaded(net):
    import ctypes
    from nu_nerf_amd import _lib
    assert isinstance(_lib.load(), ctypes.CDLL)
    maps = open("/proc/self/maps").read()
    assert "libnunerf.so" in maps


def test_pack_folds_weight_norm(net):
    P = parity_params()
    eng = net.engine()
    for l in (0, 3, 4, 8):
        W = O.wn_weight(P, f'sdf_network.lin{l}') * (2 ** -0.5 if l == 4 else 1.0)
        N, K = W.shape
        assert rel_err(eng.sdf[l].Wp[0][:N, :K].cpu(), W) < 1e-6
        assert rel_err(eng.sdf[l].WpT[0][:K, :N].t().cpu(), W) < 1e-6
        assert float(eng.sdf[l].Wp[0][:, K:].abs().max()) == 0.0 if eng.sdf[l].Kp > K else True


def test_ide_kernel_vs_reference_vector(net, ops, gpu):
    from nu_nerf_amd import _lib as L
    lib = L.load()
    d = torch.from_numpy(ops['ide_dirs']).to(gpu).contiguous()
    k = torch.from_numpy(ops['ide_kappa']).to(gpu).contiguous()
    out = torch.empty(64, 96, device=gpu)
    L.check(lib.nu_ide(L.ptr(d), L.ptr(k), 64, L.ptr(out), 96, L.stream()), "nu_ide")
    np.testing.assert_allclose(out[:, :72].cpu().numpy(), ops['ide_out'], rtol=1e-4, atol=2e-6)
    assert float(out[:, 72:].abs().max()) == 0.0
    # backward against torch autograd of the oracle IDE
    g = torch.randn(64, 72)
    dd = torch.from_numpy(ops['ide_dirs']).requires_grad_(True)
    kk = torch.from_numpy(ops['ide_kappa']).requires_grad_(True)
    (O.ide(dd, kk) * g).sum().backward()
    gd, gk = torch.empty(64, 3, device=gpu), torch.empty(64, device=gpu)
    gg = torch.zeros(64, 96, device=gpu)
    gg[:, :72] = g.to(gpu)
    L.check(lib.nu_ide_bwd(L.ptr(d), L.ptr(k), L.ptr(gg), 96, 64, L.ptr(gd), L.ptr(gk), L.stream()), "nu_ide_bwd")
    assert rel_err(gd.cpu(), dd.grad) < 1e-4 and rel_err(gk.cpu(), kk.grad[:, 0]) < 1e-4


def test_sdf_forward_normal_and_second_order_vs_reference_vectors(net, ops, gpu):
    from nu_nerf_amd.engine import addr
    eng = net.engine()
    pts = torch.from_numpy(ops['sdf_pts']).to(gpu).contiguous()
    P = pts.shape[0]
    a = eng.sdf_forward(addr(pts), 3, P, keep=True)
    n = eng.sdf_normal(a)
    np.testing.assert_allclose(a['YX'][:, :257].cpu().numpy(), ops['sdf_out'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(n.cpu().numpy(), ops['sdf_grad'], rtol=1e-4, atol=2e-6)
    flat = eng.zeros(eng.n_grad)
    dYX = eng.zeros(P, 288)
    dYX[:, :257] = torch.from_numpy(ops['sdf_cot_y']).to(gpu)
    dYX[:, 257:260] = 3.0     # x-slot cotangents must be ignored
    eng.sdf_backward(a, dYX, torch.from_numpy(ops['sdf_cot_n']).to(gpu).contiguous(), flat)
    eng.unpack_grads(flat)
    for l in (0, 3, 4, 8):
        for nm in ('weight_g', 'weight_v', 'bias'):
            off, shape = eng.grad_views[f'sdf_network.lin{l}.{nm}']
            ref = ops[f'sdf_dl_lin{l}_{nm}']
            assert rel_err(flat[off:off + ref.size].view(shape).cpu(), ref) < 1e-4, (l, nm)


def test_sampler_matches_oracle(net, gpu):
    g = golden("train_step0_r48.npz")
    P = parity_params()
    cfg = oracle_cfg()
    o, d = torch.from_numpy(g['rays_o']), torch.nn.functional.normalize(torch.from_numpy(g['rays_d']), dim=-1)
    R = o.shape[0]
    for perturb, rand in ((1.0, (torch.from_numpy(g['u1']), torch.from_numpy(g['u2']))), (0.0, None)):
        zr = O.sample_ray(P, cfg, o, d, torch.full((R, 1), 0.8), torch.full((R, 1), 4.5), perturb, rand)
        rg = None if rand is None else (rand[0].to(gpu), rand[1].to(gpu))
        z = net.sample_ray(o.to(gpu), d.to(gpu), torch.full((R,), 0.8, device=gpu), torch.full((R,), 4.5, device=gpu),
                           perturb, rg).cpu()
        assert z.shape == zr.shape
        assert bool((z[:, 1:] >= z[:, :-1]).all())                       # sortedness incl. the background tail
        dz = (z - zr).abs() / zr.abs().clamp(min=1.0)
        # inverse-CDF placement amplifies last-bit differences of sigmoid(sdf*inv_s) (see test_oracle_golden)
        assert float((dz < 1e-5).float().mean()) >= 0.95 and float(dz.max()) < 5e-3
        np.testing.assert_allclose(z[:, :1].numpy(), zr[:, :1].numpy(), rtol=1e-6)   # first coarse sample: exact formula


@pytest.mark.parametrize("mlp_dtype", ["fp32", "bf16x6"])
@pytest.mark.parametrize("name", ["train_step0_r48.npz", "train_step20000_r48.npz", "train_step500_r32_noperturb.npz"])
def test_full_train_step_vs_reference_golden(gpu, name, mlp_dtype):
    """mlp_dtype 'fp32' = exact fp32 MFMA (the default, the headline).  'bf16x6' = the fp32-equivalent split mode (exact
    3-way bf16 split of both operands, six partial products): held to the SAME reference vectors and the SAME tolerances."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g = golden(name)
    step = int(g['step'])
    cfg = dict(CFG, mlp_dtype=mlp_dtype)
    if 'noperturb' in name:
        cfg['perturb'] = 0.0
    net = make_net(gpu, cfg)
    assert net.engine().bf16 == (2 if mlp_dtype == 'bf16x6' else 0)
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    out = net.train_step_rays(batch, step, rand=rand)
    total, log = total_loss(out, [name2loss[n](cfg) for n in SPHEREPOT_LOSSES], step)
    total.backward()
    np.testing.assert_allclose(out['ray_rgb'].detach().cpu().numpy(), g['out_ray_rgb'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['acc'].detach().cpu().numpy(), g['out_acc'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['color_bkgr'].detach().cpu().numpy(), g['out_color_bkgr'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['color_spec'].detach().cpu().numpy(), g['out_color_spec'], rtol=1e-4, atol=2e-5)
    assert out['gradient_error'].shape == g['out_gradient_error'].shape      # same inner/outer partition
    np.testing.assert_allclose(out['gradient_error'].detach().cpu().numpy(), g['out_gradient_error'], rtol=1e-3, atol=2e-3)
    for k in g:
        if k.startswith('term_'):
            np.testing.assert_allclose(float(torch.mean(log[k[5:]]).detach()), float(g[k]), rtol=2e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=1e-5)
    named = dict(net.named_parameters())
    for n, ref_norm in zip([str(s) for s in g['grad_names']], g['grad_norms']):
        mine = named[n].grad
        assert mine is not None, n
        assert abs(float(mine.double().norm()) - ref_norm) <= 2e-3 * ref_norm + 1e-9, (n, float(mine.norm()), ref_norm)
    for n, p in named.items():          # dead parameters stay without gradient
        if n.startswith(('color_network.iors', 'infinity_far_bkgr')):
            assert p.grad is None
    if step < 15000:
        assert named['deviation_network.variance'].grad is None       # inv_s frozen (freeze_inv_s_step)
    for k in g:
        if k.startswith('grad__') and named[k[6:]].grad is not None:
            tol = 3e-2 if 'inner_weight' in k and step < 15000 else 3e-3
            assert rel_err(named[k[6:]].grad.cpu(), g[k]) < tol, k


def test_full_train_step_at_the_default_sampling_vs_reference_golden(gpu):
    """A step of the reference's own NeROShapeRenderer at its DEFAULT sampling (64 + 64 + 32, renderer_zerothick.py:110-117; fixture
    of oracle/gen_golden_r4.py, step 20000: occlusion + outer-regulariser losses on, inv_s trainable) replayed on the HIP path:
    per-ray outputs 1e-4, total loss 1e-5, every gradient norm 2e-3 -- the tolerances of the 32 + 32 + 16 reference steps."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g = golden("train_default_sampling_step20000_r24.npz")
    assert tuple(int(v) for v in g['sampling']) == (64, 64, 32)
    cfg = dict(CFG, n_samples=64, n_importance=64, n_bg_samples=32)
    step = int(g['step'])
    net = make_net(gpu, cfg)
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    out = net.train_step_rays(batch, step, rand=rand)
    total, log = total_loss(out, [name2loss[n](cfg) for n in SPHEREPOT_LOSSES], step)
    total.backward()
    assert out['gradient_error'].numel() == int(g['n_inner'])             # same inner / outer partition
    np.testing.assert_allclose(out['ray_rgb'].detach().cpu().numpy(), g['out_ray_rgb'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['acc'].detach().cpu().numpy(), g['out_acc'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['color_bkgr'].detach().cpu().numpy(), g['out_color_bkgr'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['color_spec'].detach().cpu().numpy(), g['out_color_spec'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(float(out['gradient_error'].mean()), float(g['out_gradient_error_mean']), rtol=2e-4)
    for k in g:
        if k.startswith('term_'):
            np.testing.assert_allclose(float(torch.mean(log[k[5:]]).detach()), float(g[k]), rtol=2e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=1e-5)
    named = dict(net.named_parameters())
    for n, ref_norm in zip([str(s) for s in g['grad_names']], g['grad_norms']):
        assert named[n].grad is not None, n
        assert abs(float(named[n].grad.double().norm()) - ref_norm) <= 2e-3 * ref_norm + 1e-9, (n, float(named[n].grad.norm()), ref_norm)
    for k in g:
        if k.startswith('grad__'):
            assert rel_err(named[k[6:]].grad.cpu(), g[k]) < 3e-3, k


def test_full_step_vs_oracle_default_sampling_and_init_weights(gpu):
    """Default 64+64+32 sampling, untouched geometric init (zero embedding columns, unit weight_g), step 200:
    exercises the init-SDF regulariser incl. the shell SDF op."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    from nu_nerf_amd.synthetic import make_rays, make_jitter
    from nu_nerf_amd.params import init_stage1_params
    cfg = dict(CFG, n_samples=64, n_importance=64, n_bg_samples=32)
    net = make_net(gpu, cfg, randomized=False)
    with torch.no_grad():   # push the sphere out so that shell points violate the large-radius constraint
        net.sdf_network.lin8.bias[0] = -1.15
    R, step = 40, 200
    rays = make_rays(R, seed=31)
    u1, u2 = make_jitter(R, 32, seed=32)
    batch = {k: torch.from_numpy(rays[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    out = net.train_step_rays(batch, step, rand=(torch.from_numpy(u1).to(gpu), torch.from_numpy(u2).to(gpu)))
    total, log = total_loss(out, [name2loss[n](cfg) for n in SPHEREPOT_LOSSES], step)
    total.backward()
    params = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in init_stage1_params(6033).items()}
    params['sdf_network.lin8.bias'][0] = -1.15
    for k, v in params.items():
        if not k.endswith('FG_LUT'):
            v.requires_grad_(True)
    ocfg = dict(O.DEFAULT_CFG)
    ototal, oterms, oout = O.train_step(params, ocfg, torch.from_numpy(rays['rays_o']), torch.from_numpy(rays['rays_d']),
                                        torch.from_numpy(rays['rgbs']), step, rand=(torch.from_numpy(u1), torch.from_numpy(u2)))
    ototal.backward()
    assert float(oterms['loss_sdf_large']) > 1e-3                       # the branch is really exercised
    np.testing.assert_allclose(float(log['loss_sdf_large'].detach()), float(oterms['loss_sdf_large']), rtol=2e-4)
    np.testing.assert_allclose(float(total.detach()), float(ototal), rtol=2e-5)
    np.testing.assert_allclose(out['ray_rgb'].detach().cpu().numpy(), oout['ray_rgb'].detach().numpy(), rtol=1e-4, atol=1e-5)
    named = dict(net.named_parameters())
    for k in ('sdf_network.lin0.weight_v', 'sdf_network.lin5.weight_g', 'sdf_network.lin8.bias', 'outer_nerf.pts_linears.3.weight',
              'color_network.albedo_predictor.2.weight_v'):
        assert rel_err(named[k].grad.cpu(), params[k].grad) < 5e-3, k


def test_rays_are_independent_units_at_full_batch(gpu):
    """Size-independent property at the BASELINE batch (4096 rays x 160 samples): rendering a sub-batch alone gives the
    same per-ray outputs as rendering it inside the full batch (no cross-ray coupling, no tile-edge artefacts)."""
    from nu_nerf_amd.synthetic import make_rays
    cfg = dict(CFG, n_samples=64, n_importance=64, n_bg_samples=32)
    net = make_net(gpu, cfg)
    rays = make_rays(4096, seed=99)
    o = torch.from_numpy(rays['rays_o']).to(gpu)
    d = torch.nn.functional.normalize(torch.from_numpy(rays['rays_d']).to(gpu), dim=-1)
    near, far = torch.full((4096, 1), 0.8, device=gpu), torch.full((4096, 1), 4.5, device=gpu)
    with torch.no_grad():
        full = net.render(o, d, near, far, perturb_overwrite=0, cos_anneal_ratio=0.4, step=20000, is_nerf=True)
        sub = net.render(o[1000:1100], d[1000:1100], near[:100], far[:100], perturb_overwrite=0, cos_anneal_ratio=0.4,
                         step=20000, is_nerf=True)
    torch.testing.assert_close(full['ray_rgb'][1000:1100], sub['ray_rgb'], rtol=0, atol=0)   # bit-exact: same kernels
    torch.testing.assert_close(full['acc'][1000:1100], sub['acc'], rtol=0, atol=0)
    assert bool(torch.isfinite(full['ray_rgb']).all()) and float(full['acc'].max()) <= 1.0 + 1e-5
    # oracle spot-check on 24 rays of the big batch
    P = parity_params()
    ocfg = dict(O.DEFAULT_CFG)
    oc, dc = o[2000:2024].cpu(), d[2000:2024].cpu()
    z = O.sample_ray(P, ocfg, oc, dc, torch.full((24, 1), 0.8), torch.full((24, 1), 4.5), 0.0)
    oo = O.render_core(P, ocfg, oc, dc, z, 20000, 0.4, True)
    np.testing.assert_allclose(full['ray_rgb'][2000:2024].cpu().numpy(), oo['ray_rgb'].detach().numpy(), rtol=1e-4, atol=2e-5)


def test_edge_cases_all_rays_miss_and_single_ray(gpu):
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    net = make_net(gpu)
    # rays that never enter the unit sphere: the inner set is empty
    o = torch.tensor([[0.0, 0.0, 5.0]] * 3, device=gpu)
    d = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.1], [0.0, 1.0, 0.0], [1.0, 1.0, 0.3]], device=gpu), dim=-1)
    batch = {'rays_o': o, 'rays_d': d, 'rgbs': torch.rand(3, 3, device=gpu)}
    out = net.train_step_rays(batch, 20000)
    assert net.engine().last_ctx['P_in'] == 0
    total, _ = total_loss(out, [name2loss[n](CFG) for n in SPHEREPOT_LOSSES], 20000)
    total.backward()
    assert bool(torch.isfinite(out['ray_rgb']).all())
    assert float(net.sdf_network.lin3.weight_v.grad.abs().max()) == 0.0       # no stale gradients for the unused net
    assert float(net.outer_nerf.pts_linears[2].weight.grad.abs().max()) > 0.0
    # a single ray through the object
    net.zero_grad()
    batch1 = {'rays_o': torch.tensor([[0.0, 0.0, 4.0]], device=gpu), 'rays_d': torch.tensor([[0.0, 0.0, -1.0]], device=gpu),
              'rgbs': torch.rand(1, 3, device=gpu)}
    out1 = net.train_step_rays(batch1, 0)
    assert out1['ray_rgb'].shape == (1, 3) and net.engine().last_ctx['P_in'] > 0
    total_loss(out1, [name2loss[n](CFG) for n in SPHEREPOT_LOSSES], 0)[0].backward()
    assert bool(torch.isfinite(net.sdf_network.lin0.weight_v.grad).all())


def test_standard_renderer_sphere_direction_vs_reference_golden(gpu):
    """nu_nerf_amd.renderer_std (drop-in for network/renderer.py's stage-1 renderer): sphere_direction=True (144-d
    outer_light input), refrac_freq=3, real-capture near/far, loss_normal, candidate-ray colour_spec."""
    from helpers import STD_CFG
    from nu_nerf_amd.renderer_std import NeROShapeRenderer as StdRenderer
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    from nu_nerf_amd.loss import name2loss, total_loss
    g = golden("train_std_step20000_r40.npz")
    cfg = {'is_nerf': False, 'n_samples': 64, 'n_importance': 32, 'n_bg_samples': 16, 'freeze_inv_s_step': 15000,
           'apply_occ_loss': True, 'occ_loss_step': 15000, 'eikonal_weight': 0.05, 'outer_reg_loss_weight': 0.1,
           'shader_config': {'sphere_direction': True, 'human_light': False, 'refrac_freq': 3}}
    net = StdRenderer(cfg, training=False)
    net.load_param_dict(randomize_for_parity(init_stage1_params(6033, sphere_direction=True, refrac_freq=3), seed=1))
    net = net.to(gpu)
    step = int(g['step'])
    o = torch.from_numpy(g['rays_o']).to(gpu)
    d = torch.nn.functional.normalize(torch.from_numpy(g['rays_d']).to(gpu), dim=-1)
    near, far = net.near_far_from_sphere(o, d)
    np.testing.assert_allclose(near.cpu().numpy(), g['near'], rtol=1e-5, atol=1e-6)
    names = ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'outer_reg', 'normal_ori']
    rgbs = torch.from_numpy(g['rgbs']).to(gpu)
    # (1) render_core on the REFERENCE's z_vals: the sampler's last-bit sensitivity is out of the picture -> tight tolerances
    out = net.render_core(o, d, torch.from_numpy(g['z_vals']).to(gpu), None, cos_anneal_ratio=net.get_anneal_val(step), step=step,
                          is_train=True, is_nerf=False)
    out['loss_rgb'] = net.compute_rgb_loss(out['ray_rgb'], rgbs)
    total, log = total_loss(out, [name2loss[n](cfg) for n in names], step)
    total.backward()
    for k in ('ray_rgb', 'acc', 'color_bkgr', 'color_spec', 'loss_normal'):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g['out_' + k], rtol=1e-4, atol=2e-6, err_msg=k)
    for k in g:
        if k.startswith('term_') and k != 'term_loss_occ':      # the occlusion target runs a second inverse-CDF sampler
            np.testing.assert_allclose(float(torch.mean(log[k[5:]]).detach()), float(g[k]), rtol=1e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(float(log['loss_occ'].detach()), float(g['term_loss_occ']), rtol=2e-3)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=2e-5)
    named = dict(net.named_parameters())
    bad = []
    for n, ref_norm in zip([str(s) for s in g['grad_names']], g['grad_norms']):
        assert named[n].grad is not None, n
        err = abs(float(named[n].grad.double().norm()) - ref_norm) / (ref_norm + 1e-12)
        # outer_light's first layers see the 72 + 72 IDE columns; the degree-8 / 16 terms make their weight gradients
        # ill-conditioned in fp32: the CPU oracle itself sits 5e-3 (element-wise) from the reference on outer_light.0.weight_v,
        # with the same worst columns (11, 19, 21-23: the l = 16 terms)
        if err > (2e-3 if ('inner_weight' in n or 'outer_light' in n) else 3e-4):
            bad.append((round(err, 6), n))
    for k in g:
        if k.startswith('grad__'):
            err = rel_err(named[k[6:]].grad.cpu(), g[k])
            if err > (3e-2 if 'outer_light.0' in k else 1.5e-3):     # measured 1.6e-2 / 7e-4 (oracle: 5e-3 / 5e-4)
                bad.append((round(err, 6), k))
    assert not bad, sorted(bad, reverse=True)[:10]
    # (2) the whole entry point with the build's own sampler: z differs from the reference's in a few per cent of the samples
    # (tests/test_stage1_gpu.py::test_sampler_matches_oracle), per-ray outputs do not care, element-wise gradients a little
    net.zero_grad()
    out = net.render(o, d, near, far, None, -1, net.get_anneal_val(step), is_train=True, step=step, is_nerf=False,
                     rand=(torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu)))
    out['loss_rgb'] = net.compute_rgb_loss(out['ray_rgb'], rgbs)
    total, log = total_loss(out, [name2loss[n](cfg) for n in names], step)
    total.backward()
    for k in ('ray_rgb', 'acc', 'color_bkgr', 'color_spec', 'loss_normal'):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g['out_' + k], rtol=2e-4, atol=3e-5, err_msg=k)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=2e-5)
    for n, ref_norm in zip([str(s) for s in g['grad_names']], g['grad_norms']):
        assert abs(float(named[n].grad.double().norm()) - ref_norm) <= 3e-3 * ref_norm + 1e-9, (n, float(named[n].grad.norm()), ref_norm)
    for k in g:
        if k.startswith('grad__'):
            assert rel_err(named[k[6:]].grad.cpu(), g[k]) < 3e-2, k   # element-wise: sensitive to the shifted samples


def test_process_ray_batch_and_human_poses_on_the_device(gpu):
    """Row a2 on the GPU: the same assertions as the CPU test (tests/test_host_logic.py), on `cuda` tensors -- the real-capture
    ray construction is what feeds the device-resident ray store."""
    from helpers import check_ray_batch_against_reference_fixture
    check_ray_batch_against_reference_fixture(gpu)


def test_ray_store_from_loaded_images_on_the_device(gpu):
    """`set_ray_store(imgs_info)` builds the ray store ON the GPU (reference fixture tests/golden/ray_store.npz), train_step
    slices it there (no per-step host -> device copy) for both data conventions, and test_step renders a test image from it."""
    from helpers import check_ray_store_against_reference_fixture
    from nu_nerf_amd.renderer import name2renderer
    info = check_ray_store_against_reference_fixture(gpu)
    small = {'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8, 'train_ray_num': 32, 'test_ray_num': 20}
    # (test_downsample_ratio stays at the reference's default True: test_step blurs and halves the 6 x 5 test image on the device)
    for is_nerf, name in ((True, 'nerf/spherepot'), (False, 'real/bear')):
        net = name2renderer['shape'](dict(small, database_name=name, is_nerf=is_nerf), training=True).to(gpu)
        poses = info['poses'].clone()
        if not is_nerf:       # world -> camera poses looking at the origin from 3 units away, so that the rays meet the unit sphere
            poses[:, :, :3] = torch.eye(3, device=gpu)
            poses[:, :, 3] = torch.tensor([0.0, 0.0, 3.0], device=gpu)
        net.set_ray_store(dict(info, poses=poses), test_imgs_info=dict(info, poses=poses))
        assert all(v.is_cuda for v in net.train_batch.values()) and net.tbn == 90
        out = net({'step': 10})
        assert out['ray_rgb'].shape == (32, 3) and bool(torch.isfinite(out['ray_rgb']).all()) and out['loss_rgb'].requires_grad
        out['loss_rgb'].mean().backward()
        with torch.no_grad():
            ev = net({'index': 1, 'eval': True, 'step': 0})
        from nu_nerf_amd.renderer import imgs_info_downsample
        assert ev['ray_rgb'].shape == (3, 2, 3) and ev['gt_rgb'].shape == (3, 2, 3) and ev['gt_mask'].shape == (3, 2, 1)
        half = imgs_info_downsample({'imgs': info['imgs'][1:2], 'Ks': info['Ks'][1:2]}, 0.5)['imgs'][0]
        np.testing.assert_allclose(ev['gt_rgb'].cpu().numpy(), half.permute(1, 2, 0).cpu().numpy(), rtol=0, atol=1e-7)


@pytest.mark.parametrize("mlp_dtype", ["fp32", "bf16x6", "bf16"])
def test_a_train_step_is_bit_reproducible(gpu, mlp_dtype):
    """The same batch through forward + loss + backward four times on one network: every output and every parameter gradient must
    come out bit for bit the same (no atomics, deterministic split reductions, fixed launch structure).  fp32 overlaps its NeRF++
    chain on a second stream at this size; the bf16-MFMA modes must not (a packed-fp32 VALU kernel running beside those
    GEMMs returns wrong elements now and then -- 'bf16x6' failed this test one repetition in two before it went to one stream)."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g = golden("train_step20000_r48.npz")
    step = int(g['step'])
    cfg = dict(CFG, mlp_dtype=mlp_dtype)
    net = make_net(gpu, cfg)
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
    ref = None
    for rep in range(4):
        for p in net.parameters():
            p.grad = None
        out = net.train_step_rays(batch, step, rand=rand)
        total, _ = total_loss(out, losses, step)
        total.backward()
        torch.cuda.synchronize()
        assert net.engine().last_ctx['two_streams'] == (mlp_dtype == 'fp32')
        cur = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        cur['ray_rgb'] = out['ray_rgb'].detach().clone()
        cur['total'] = total.detach().clone()
        if ref is None:
            ref = cur
            continue
        assert set(cur) == set(ref)
        bad = [n for n in ref if not torch.equal(ref[n], cur[n])]
        assert not bad, (rep, bad[:8])
