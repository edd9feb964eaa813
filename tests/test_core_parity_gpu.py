"""Parity in the north star's own words: "per-ray RGB / WEIGHTS within 1e-4 rel fp32".

The per-sample quantities of the reference (alpha, sampled colour, composite weights) were captured by
oracle/gen_golden_r2.py from the reference's own run of the three golden train steps.  Here the HIP render_core is fed the
REFERENCE's z_vals, so the inverse-CDF sampler's last-bit sensitivity (tests/test_stage1_gpu.py::test_sampler_matches_oracle)
is out of the picture and every per-sample output -- and the gradients -- can be held to 1e-4.
"""
import os
import numpy as np
import pytest
import torch

from helpers import golden, parity_params, rel_err, exact_fp32_only
from oracle import stage1_oracle as O
from test_stage1_gpu import CFG, make_net

pytestmark = pytest.mark.gpu

TAGS = ["step0_r48", "step20000_r48", "step500_r32_noperturb"]
# gradients are sums of ~1e5 fp32 products: two fp32 implementations differ by their summation trees.  The CPU oracle itself sits
# up to 7e-4 from the reference on these norms (tests/test_oracle_golden.py); measured here: <= 2.1e-4 (step 0), < 1e-4 (others)
NORM_TOL, ELEM_TOL = 3e-4, 3e-4


def _inputs(tag, gpu):
    g, c = golden(f"train_{tag}.npz"), golden(f"core_{tag}.npz")
    assert np.array_equal(g['z_vals'], c['z_vals'])
    o = torch.from_numpy(g['rays_o']).to(gpu)
    dn = torch.nn.functional.normalize(torch.from_numpy(g['rays_d']).to(gpu), dim=-1)
    return g, c, o, dn, torch.from_numpy(c['z_vals']).to(gpu)


@pytest.mark.parametrize("tag", TAGS)
def test_per_sample_weights_alpha_colour_at_reference_z(gpu, tag):
    g, c, o, dn, z = _inputs(tag, gpu)
    step = int(g['step'])
    net = make_net(gpu)
    eng = net.engine()
    eng.pack()
    with torch.no_grad():
        out, ctx = eng.render_forward(o, dn, z, net.get_anneal_val(step), want_weights=True)
    R, S = z.shape
    inner = ctx['inner_rm'].view(R, S).cpu().numpy()
    assert np.array_equal(inner, c['inner_mask'])                                     # same partition, point for point
    alpha = ctx['alpha_rm'].view(R, S).cpu().numpy()
    color = ctx['color_rm'].view(R, S, 4)[..., :3].cpu().numpy()
    w = out['weights'].cpu().numpy()
    # rtol 1e-4 as declared.  alpha = (s_prev - s_next + 1e-5) / (s_prev + 1e-5) subtracts two sigmoids near 1, so its ABSOLUTE
    # error floor is a few ulp of 1.0 (1.2e-7 each) however small alpha is: atol 5e-7 (the CPU oracle needs the same)
    np.testing.assert_allclose(alpha, c['alpha'], rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(color, c['sampled_color'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(w, c['weights'], rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(out['gradient_error'].cpu().numpy(), c['gradient_error'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w.sum(-1), g['out_acc'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", TAGS)
def test_train_step_gradients_at_reference_z(gpu, tag):
    """Forward + losses + backward on the reference's z_vals: per-ray outputs and every loss term at 1e-4, all 128 gradient
    norms and the stored full gradients at 3e-4 (were 2e-3 / 3e-3 with the build's own sampler in the loop)."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g, c, o, dn, z = _inputs(tag, gpu)
    step = int(g['step'])
    net = make_net(gpu)
    out = net.render_core(o, dn, z, None, cos_anneal_ratio=net.get_anneal_val(step), step=step, is_train=True, is_nerf=True)
    out['loss_rgb'] = net.compute_rgb_loss(out['ray_rgb'], torch.from_numpy(g['rgbs']).to(gpu))
    total, log = total_loss(out, [name2loss[n](CFG) for n in SPHEREPOT_LOSSES], step)
    total.backward()
    for k in ('ray_rgb', 'acc', 'color_bkgr', 'color_spec'):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g['out_' + k], rtol=1e-4, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(out['gradient_error'].detach().cpu().numpy(), g['out_gradient_error'], rtol=1e-4, atol=1e-6)
    for k in g:
        if k.startswith('term_'):
            np.testing.assert_allclose(float(torch.mean(log[k[5:]]).detach()), float(g[k]), rtol=1e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=1e-5)
    named = dict(net.named_parameters())
    bad = []
    for n, ref_norm in zip([str(s) for s in g['grad_names']], g['grad_norms']):
        # the occlusion target comes from a second inverse-CDF sampler (get_intersection): inner_weight keeps its noise
        tol = 2e-3 if ('inner_weight' in n and step >= 15000) else NORM_TOL
        err = abs(float(named[n].grad.double().norm()) - ref_norm) / (ref_norm + 1e-12)
        if err > tol:
            bad.append((round(err, 7), n))
    for k in g:
        if k.startswith('grad__') and named[k[6:]].grad is not None:
            tol = 2e-3 if ('inner_weight' in k and step >= 15000) else ELEM_TOL
            err = rel_err(named[k[6:]].grad.cpu(), g[k])
            if err > tol:
                bad.append((round(err, 7), k))
    assert not bad, sorted(bad, reverse=True)[:12]


def test_occ_loss_subsample_branch_vs_reference(gpu):
    """More near-surface points than occ_loss_max_pn: the random subsample (renderer_zerothick.py:708-714) with the
    reference's recorded permutation."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g = golden("occ_cap_step20000_r48.npz")
    step = int(g['step'])
    cfg = dict(CFG, occ_loss_max_pn=int(g['occ_loss_max_pn']))
    net = make_net(gpu, cfg)
    o = torch.from_numpy(g['rays_o']).to(gpu)
    dn = torch.nn.functional.normalize(torch.from_numpy(g['rays_d']).to(gpu), dim=-1)
    z = torch.from_numpy(g['z_vals']).to(gpu)
    perm = torch.from_numpy(g['perm']).to(gpu)
    out = net.render_core(o, dn, z, None, cos_anneal_ratio=net.get_anneal_val(step), step=step, is_train=True, is_nerf=True,
                          occ_perm=perm)
    out['loss_rgb'] = net.compute_rgb_loss(out['ray_rgb'], torch.from_numpy(g['rgbs']).to(gpu))
    total, log = total_loss(out, [name2loss[n](cfg) for n in SPHEREPOT_LOSSES], step)
    total.backward()
    np.testing.assert_allclose(float(out['loss_occ'].detach()), float(g['out_loss_occ']), rtol=1e-3)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=1e-4)
    named = dict(net.named_parameters())
    for k in g:
        if k.startswith('grad__'):
            assert rel_err(named[k[6:]].grad.cpu(), g[k]) < 5e-3, (k, rel_err(named[k[6:]].grad.cpu(), g[k]))
    # without the cap the loss differs: the branch really selected a subset
    net2 = make_net(gpu)
    out2 = net2.render_core(o, dn, z, None, cos_anneal_ratio=net2.get_anneal_val(step), step=step, is_train=True, is_nerf=True)
    assert abs(float(out2['loss_occ'].detach()) - float(g['out_loss_occ'])) > 1e-4
    # and through the whole entry point (train_step_rays -> render -> sampler -> render_core) with rand = (u1, u2, perm)
    net3 = make_net(gpu, cfg)
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    out3 = net3.train_step_rays(batch, step, rand=(torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu), perm))
    np.testing.assert_allclose(float(out3['loss_occ'].detach()), float(g['out_loss_occ']), rtol=5e-2)


@exact_fp32_only
def test_full_size_backward_sub_batch_property_and_oracle_spot_check(gpu):
    """BASELINE configs[1] size (4096 rays x 160 samples), BACKWARD: the gradient of a loss that only looks at 100 rays of the
    full batch equals the gradient of the same loss on those 100 rays rendered alone (exercises the 2 GiB slab arena, the
    capacity classes, >= 4 GiB operands and the persistent grids' tails), and three parameter gradients agree with the CPU
    oracle on the same rays and z."""
    from nu_nerf_amd.synthetic import make_rays
    cfg = dict(CFG, n_samples=64, n_importance=64, n_bg_samples=32)
    net = make_net(gpu, cfg)
    rays = make_rays(4096, seed=123)
    o = torch.from_numpy(rays['rays_o']).to(gpu)
    d = torch.nn.functional.normalize(torch.from_numpy(rays['rays_d']).to(gpu), dim=-1)
    rgbs = torch.from_numpy(rays['rgbs']).to(gpu)
    near, far = torch.full((4096,), 0.8, device=gpu), torch.full((4096,), 4.5, device=gpu)
    step, anneal = 20000, 0.4
    net.engine().pack()
    with torch.no_grad():
        z = net.sample_ray(o, d, near, far, 0.0)
    lo, hi = 1500, 1600
    S = z.shape[1]

    def loss_on(o_, d_, z_, rgb_, ray_lo, ray_hi):
        out = net.render_core(o_, d_, z_, None, cos_anneal_ratio=anneal, step=step, is_train=True, is_nerf=True)
        ctx = net.engine().last_ctx
        ray_of_pt = torch.div(ctx['idx_in'][:ctx['P_in']].long(), S, rounding_mode='floor')
        sel = ((ray_of_pt >= ray_lo) & (ray_of_pt < ray_hi)).float()
        l_rgb = net.compute_rgb_loss(out['ray_rgb'], rgb_)[ray_lo:ray_hi].sum() / (ray_hi - ray_lo)
        l_eik = 0.1 * (out['gradient_error'] * sel).sum() / sel.sum().clamp(min=1.0)
        l_bg = 0.5 * ((out['color_bkgr'] - out['color_spec'])[ray_lo:ray_hi] ** 2).mean()
        return l_rgb + l_eik + l_bg, int(sel.sum())

    net.zero_grad()
    total_full, n_full = loss_on(o, d, z, rgbs, lo, hi)
    total_full.backward()
    g_full = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    net.zero_grad()
    total_sub, n_sub = loss_on(o[lo:hi].contiguous(), d[lo:hi].contiguous(), z[lo:hi].contiguous(), rgbs[lo:hi].contiguous(), 0, hi - lo)
    total_sub.backward()
    g_sub = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    assert n_full == n_sub and n_full > 0
    np.testing.assert_allclose(float(total_full.detach()), float(total_sub.detach()), rtol=1e-6)
    assert set(g_full) == set(g_sub)
    for n in g_full:
        # same per-point arithmetic, different summation trees (4096 rays' zero rows interleave the split-K slabs)
        assert rel_err(g_full[n], g_sub[n]) < 2e-5, (n, rel_err(g_full[n], g_sub[n]))
        assert bool(torch.isfinite(g_full[n]).all())
    # oracle spot-check: 24 of those rays, same z, three parameters
    P = parity_params(requires_grad=True)
    ocfg = dict(O.DEFAULT_CFG)
    k0, k1 = lo, lo + 24
    oo = O.render_core(P, ocfg, o[k0:k1].cpu(), d[k0:k1].cpu(), z[k0:k1].cpu(), step, anneal, True)
    ol = (torch.sqrt(((oo['ray_rgb'] - rgbs[k0:k1].cpu()) ** 2).sum(-1) + 1e-3).mean() + 0.1 * oo['gradient_error'].mean()
          + 0.5 * ((oo['color_bkgr'] - oo['color_spec']) ** 2).mean())
    ol.backward()
    net.zero_grad()
    t24, _ = loss_on(o, d, z, rgbs, k0, k1)
    t24.backward()
    np.testing.assert_allclose(float(t24.detach()), float(ol.detach()), rtol=2e-5)
    named = dict(net.named_parameters())
    errs = {n: rel_err(named[n].grad.cpu(), P[n].grad) for n in
            ('sdf_network.lin2.weight_v', 'outer_nerf.pts_linears.6.weight', 'color_network.refrac_light.2.weight_v')}
    print("oracle spot-check (24 rays of the 4096-ray batch), element-wise rel err:", errs)
    assert max(errs.values()) < 1e-3, errs          # 24 rays: few terms per sum, so fp32 ordering noise averages less (measured 4.4e-4)


@pytest.mark.parametrize("std", [False, True])
def test_fused_loss_kernels_equal_the_loss_registry(gpu, std):
    """N1: loss.fused_stage1_loss (csrc/loss.hip) against the eager registry path (network/loss.py semantics) on the same
    forward: total, every log term, per-ray outputs and every parameter gradient."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss, fused_stage1_loss
    if std:
        from helpers import STD_CFG   # noqa: F401
        from nu_nerf_amd.renderer_std import NeROShapeRenderer as R
        from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
        g = golden("train_std_step20000_r40.npz")
        cfg = {'is_nerf': False, 'n_samples': 64, 'n_importance': 32, 'n_bg_samples': 16, 'freeze_inv_s_step': 15000,
               'apply_occ_loss': True, 'occ_loss_step': 15000, 'eikonal_weight': 0.05, 'outer_reg_loss_weight': 0.1,
               'shader_config': {'sphere_direction': True, 'human_light': False, 'refrac_freq': 3}}
        names = ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'outer_reg', 'normal_ori']
        def build():
            n = R(cfg, training=False)
            n.load_param_dict(randomize_for_parity(init_stage1_params(6033, sphere_direction=True, refrac_freq=3), seed=1))
            return n.to(gpu)
    else:
        g = golden("train_step20000_r48.npz")
        cfg, names = CFG, SPHEREPOT_LOSSES
        def build():
            return make_net(gpu)
    step = int(g['step'])
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    losses = [name2loss[n](cfg) for n in names]
    a, b = build(), build()
    out_a = a.train_step_rays(batch, step, rand=rand)
    total_a, log_a = total_loss(out_a, losses, step)
    total_a.backward()
    total_b, log_b, out_b = fused_stage1_loss(b, batch, step, losses, rand=rand)
    total_b.backward()
    np.testing.assert_allclose(float(total_b.detach()), float(total_a.detach()), rtol=2e-6)
    # ... and DIRECTLY against the reference's own numbers for this step (fixture written by the reference's classes): the fused
    # total and every term_* the fixture holds, not only through the registry
    np.testing.assert_allclose(float(total_b.detach()), float(g['total_loss']), rtol=1e-5)
    n_terms = 0
    for k, v in g.items():
        if k.startswith('term_') and k[5:] in log_b:
            np.testing.assert_allclose(float(torch.mean(log_b[k[5:]]).detach()), float(np.mean(v)), rtol=2e-4, atol=1e-7, err_msg=k)
            n_terms += 1
    assert n_terms >= 3, sorted(k for k in g if k.startswith('term_'))
    assert set(k for k in log_a if k.startswith('loss')) == set(k for k in log_b if k.startswith('loss'))
    for k in log_a:
        if k.startswith('loss'):
            np.testing.assert_allclose(float(torch.mean(log_b[k]).detach()), float(torch.mean(log_a[k]).detach()), rtol=3e-6, atol=1e-9, err_msg=k)
    for k in ('ray_rgb', 'color_spec', 'color_bkgr', 'loss_rgb'):
        torch.testing.assert_close(out_b[k].detach(), out_a[k].detach(), rtol=2e-6, atol=2e-7)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            # (the two assemblies round the upstream gradients differently; a whole-suite run with NU_MLP_DTYPE=bf16x6 sees that
            # difference through the split products too: 1.6e-5 on one NeRF++ weight, 5.2e-5 on the SDF's first weight_g)
            tol = 1e-4 if os.environ.get('NU_MLP_DTYPE') == 'bf16x6' else 1e-5
            assert rel_err(q.grad, p.grad) < tol, (n, rel_err(q.grad, p.grad))


def test_material_regularisers_have_a_gradient_path(gpu):
    """network/loss.py:166-192 (`transmission_reg`, `metallic_reg`; `mat_reg` passes its inputs through): 0.1 * mean(y^2) of the
    inner points' transmission weight / metallic.  The renderer hands both out as differentiable outputs; values and the
    gradients they add to the material heads are checked against the CPU oracle, through both loss paths."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss, fused_stage1_loss
    g = golden("train_step20000_r48.npz")
    step = int(g['step'])
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    names = list(SPHEREPOT_LOSSES) + ['mat_reg', 'transmission_reg', 'metallic_reg']
    losses = [name2loss[n](CFG) for n in names]
    P = parity_params(requires_grad=True)
    from helpers import oracle_cfg
    ot, oterms, oout = O.train_step(P, oracle_cfg(), torch.from_numpy(g['rays_o']), torch.from_numpy(g['rays_d']), torch.from_numpy(g['rgbs']), step,
                                    rand=(torch.from_numpy(g['u1']), torch.from_numpy(g['u2'])))
    o_trans = 0.1 * torch.mean(oout['transmission'] ** 2)
    o_metal = 0.1 * torch.mean(oout['metallic'] ** 2)
    (ot + o_trans + o_metal).backward()
    heads = ['color_network.transmisstion_weight.6.weight_v', 'color_network.metallic_predictor.6.weight_v',
             'color_network.transmisstion_weight.0.weight_v', 'sdf_network.lin8.weight_v']
    for fused in (False, True):
        net = make_net(gpu)
        if fused:
            total, log, out = fused_stage1_loss(net, batch, step, losses, rand=rand)
        else:
            out = net.train_step_rays(batch, step, rand=rand)
            total, log = total_loss(out, losses, step)
        total.backward()
        np.testing.assert_allclose(float(log['loss_trans_reg'].detach()), float(o_trans.detach()), rtol=2e-5)
        np.testing.assert_allclose(float(log['loss_metal_reg'].detach()), float(o_metal.detach()), rtol=2e-5)
        np.testing.assert_allclose(float(total.detach()), float((ot + o_trans + o_metal).detach()), rtol=2e-5)
        named = dict(net.named_parameters())
        for n in heads:
            assert rel_err(named[n].grad.cpu(), P[n].grad) < 2e-3, (fused, n, rel_err(named[n].grad.cpu(), P[n].grad))
    # without the regularisers the transmission head's gradient is a different one (the path is live, not a constant)
    net = make_net(gpu)
    out = net.train_step_rays(batch, step, rand=rand)
    total, _ = total_loss(out, [name2loss[n](CFG) for n in SPHEREPOT_LOSSES], step)
    total.backward()
    assert rel_err(dict(net.named_parameters())[heads[0]].grad.cpu(), P[heads[0]].grad) > 1e-2


class _FixedWeightReducer:
    """Stand-in for parallel.GradAllReducer on one process: world 2, this rank holds 70 % of the union's inner points."""
    world = 2

    def point_weight(self, n_local, device):
        return torch.full((1,), 1.4, device=device)

    def count_weights(self, counts, device):          # (inner points, occlusion-loss points, candidate rays): only the first is uneven here
        w = torch.ones(len(counts), device=device)
        w[0] = 1.4
        return w


def test_fused_loss_with_the_data_parallel_point_weight(gpu):
    """under data parallelism the eikonal mean is this rank's share of the mean over all ranks' inner points.  The
    fused loss kernels take that count ratio as a DEVICE scalar (nu_loss_fwd / nu_loss_bwd `point_weight`), so the N > 1 step is
    the N = 1 step: same total, terms and gradients as the registry path on `gradient_error * point_weight`."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss, fused_stage1_loss
    g = golden("train_step20000_r48.npz")
    step = int(g['step'])
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    losses = [name2loss[n](CFG) for n in SPHEREPOT_LOSSES]
    red = _FixedWeightReducer()
    a, b, c = make_net(gpu), make_net(gpu), make_net(gpu)
    out_a = a.train_step_rays(batch, step, rand=rand)
    out_a['gradient_error'] = out_a['gradient_error'] * red.point_weight(None, gpu)
    total_a, log_a = total_loss(out_a, losses, step)
    total_a.backward()
    total_b, log_b, _ = fused_stage1_loss(b, batch, step, losses, rand=rand, reducer=red)
    total_b.backward()
    total_c, log_c, _ = fused_stage1_loss(c, batch, step, losses, rand=rand)
    np.testing.assert_allclose(float(total_b.detach()), float(total_a.detach()), rtol=2e-6)
    np.testing.assert_allclose(float(log_b['loss_eikonal'].detach()), float(torch.mean(log_a['loss_eikonal']).detach()), rtol=3e-6)
    np.testing.assert_allclose(float(log_b['loss_eikonal'].detach()), 1.4 * float(log_c['loss_eikonal'].detach()), rtol=3e-6)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            assert rel_err(q.grad, p.grad) < 1e-5, (n, rel_err(q.grad, p.grad))


class _TwoShardReducer:
    """Stands in for parallel.GradAllReducer with world = 2 inside ONE process (no torch.distributed, no child process): the counts a
    real all-reduce would sum are known because both shards are rendered here; phase 1 records each shard's counts, phase 2 hands out
    n_local * world / sum n."""
    world = 2

    def __init__(self):
        self.rec, self.totals, self.shard = {}, None, 0

    def count_weights(self, counts, device):
        n = torch.stack([c.reshape(-1)[0].to(device=device, dtype=torch.float32) if torch.is_tensor(c) else torch.tensor(float(c), device=device)
                         for c in counts])
        if self.totals is None:
            self.rec[self.shard] = n
            return torch.ones_like(n)
        return torch.where(self.totals > 0, n * 2 / torch.clamp(self.totals, min=1.0), torch.ones_like(n))


@exact_fp32_only
@pytest.mark.parametrize("fused", [True, False])
def test_two_ray_shards_average_to_the_union_batch_gradient(gpu, fused):
    """SURVEY 8(e) on the real kernels: a 2 x 48-ray batch rendered as two disjoint shards one after the other, each loss assembled
    with the count ratios of parallel.dp_weight_outputs (inner points -> eikonal, transmission / metallic regularisers; the
    occlusion-loss points), the two gradient sets averaged -- what the all-reduce does -- against the gradient of the 96-ray union
    batch rendered at once.  Step 20000: occlusion + outer-regulariser losses on, inv_s trainable.  Per-ray terms average exactly
    (equal ray counts); the subset means are exact through the weights; what is left is summation order: every gradient within 2e-5
    of its norm.  Under sharding the 2048-point occlusion cap applies per shard (here 96 rays stay far below it, so the test is
    exact; with the cap active N shards use up to N x 2048 points -- stated in parallel.dp_weight_outputs and DESIGN section 7), and
    the init-SDF normalisers (first 1000 steps only) are per-shard ratios -- not exercised at this step."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss, fused_stage1_loss
    from nu_nerf_amd.parallel import dp_weight_outputs
    g = golden("train_step20000_r48.npz")
    h = golden("train_step0_r48.npz")
    step = 20000
    cat = lambda k: torch.from_numpy(np.concatenate([g[k], h[k]], 0)).to(gpu)
    batch = {k: cat(k) for k in ('rays_o', 'rays_d', 'rgbs')}
    u1, u2 = cat('u1'), cat('u2')
    loss_names = SPHEREPOT_LOSSES + ['transmission_reg', 'metallic_reg']
    losses = [name2loss[n](CFG) for n in loss_names]

    def run(sl, reducer):
        net = make_net(gpu)
        b = {k: v[sl].contiguous() for k, v in batch.items()}
        rand = (u1[sl].contiguous(), u2[sl].contiguous())
        if fused:
            total, _, _ = fused_stage1_loss(net, b, step, losses, rand=rand, reducer=reducer)
        else:
            out = net.train_step_rays(b, step, rand=rand)
            if reducer is not None:
                dp_weight_outputs(out, reducer, net)
            total, _ = total_loss(out, losses, step)
        total.backward()
        torch.cuda.synchronize()
        return float(total.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, net

    whole_loss, whole, net_w = run(slice(0, 96), None)
    assert 0 < net_w._n_occ < 2048 and net_w.engine().last_ctx['P_in'] > 0
    red = _TwoShardReducer()
    shards = (slice(0, 48), slice(48, 96))
    for i, sl in enumerate(shards):                     # phase 1: the counts each rank would contribute
        red.shard = i
        run(sl, red)
    red.totals = red.rec[0] + red.rec[1]
    assert float(red.rec[0][0]) != float(red.rec[1][0])                  # the shards really differ in their inner-point counts
    assert abs(float(red.totals[0]) - net_w.engine().last_ctx['P_in']) < 0.5 and abs(float(red.totals[1]) - net_w._n_occ) < 0.5
    parts = []
    for i, sl in enumerate(shards):                     # phase 2: the weighted shard steps
        red.shard = i
        parts.append(run(sl, red))
    avg_loss = 0.5 * (parts[0][0] + parts[1][0])
    assert abs(avg_loss - whole_loss) <= 2e-6 * abs(whole_loss), (avg_loss, whole_loss)
    assert set(parts[0][1]) == set(parts[1][1]) == set(whole)
    worst = 0.0
    for n, ref in whole.items():
        got = 0.5 * (parts[0][1][n] + parts[1][1][n])
        err = float((got - ref).double().norm() / (ref.double().norm() + 1e-30))
        worst = max(worst, err)
        assert err <= 2e-5, (n, err)
    # and the weights matter: without them the eikonal / occlusion shares are wrong by the count ratio
    red_off = None
    plain = [run(sl, red_off)[1] for sl in shards]
    n0 = 'sdf_network.lin4.weight_v'
    off = float((0.5 * (plain[0][n0] + plain[1][n0]) - whole[n0]).double().norm() / whole[n0].double().norm())
    assert off > 10 * max(worst, 1e-7), (off, worst)


@exact_fp32_only
@pytest.mark.parametrize("py_seq", [False, True])
def test_full_arena_flushes_in_one_stream_mode_and_fails_closed_when_forked(gpu, monkeypatch, py_seq):
    """The split-reduction arena is shared by everything an engine enqueues.  When it (or the descriptor table) fills in the
    middle of a pass, the pending slabs are reduced on the calling stream and their space is handed out again -- correct on one
    stream (same gradients, bit for bit), a race while the NeRF++ chain runs on the side stream (engine._fork: the other stream
    may still be writing those slabs).  The forked engine must therefore fail closed (NuOpCtx.forked -> NU_ERR_WORKSPACE), in
    both sequencing paths."""
    from nu_nerf_amd._lib import NuNerfLibraryError
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g = golden("train_step20000_r48.npz")
    step = int(g['step'])
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    losses = [name2loss[n](CFG) for n in SPHEREPOT_LOSSES]

    def run(two_stream, arena_floats=None):
        if arena_floats is None:
            monkeypatch.delenv('NU_ARENA_FLOATS', raising=False)
        else:
            monkeypatch.setenv('NU_ARENA_FLOATS', str(int(arena_floats)))
        net = make_net(gpu)
        eng = net.engine()
        eng.py_seq = py_seq
        eng._TWO_STREAM_SAMPLES = (1 << 30) if two_stream else 0
        peak, flushes = [0], [0]
        real_flush = eng.flush_reductions

        def flush():
            peak[0] = max(peak[0], int(eng._ctx.arena_off))
            flushes[0] += 1
            real_flush()
        eng.flush_reductions = flush
        torch.manual_seed(11)
        out = net.train_step_rays(batch, step, rand=rand)
        assert eng.last_ctx['two_streams'] == two_stream
        total, _ = total_loss(out, losses, step)
        total.backward()
        torch.cuda.synchronize()
        return ({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, peak[0], flushes[0],
                int(eng._arena.numel()))

    ref, peak, flushes0, full = run(False)
    assert 0 < peak < full
    small = max(peak // 3, 1 << 16) // 64 * 64
    got, _, flushes1, numel = run(False, small)
    assert numel == small < peak                                       # a step's slabs do not fit: the arena filled mid-pass
    assert set(got) == set(ref)
    for n in ref:
        assert torch.equal(got[n], ref[n]), n
    forked, _, _, _ = run(True)                                        # default arena, two streams: the same gradients
    for n in ref:
        assert torch.equal(forked[n], ref[n]), n
    with pytest.raises(NuNerfLibraryError):
        run(True, small)
    torch.cuda.synchronize()


@exact_fp32_only
def test_network_level_c_entries_equal_launch_by_launch_sequencing(gpu):
    """nu_sdf_mlp_*, nu_nerfpp_mlp_*, nu_shading_stack_* sequence the same kernels in the same order as the
    launch-by-launch Python path (engine.py_seq): outputs and every gradient are bit-identical."""
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss
    g = golden("train_step20000_r48.npz")
    step = int(g['step'])
    batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(gpu), torch.from_numpy(g['u2']).to(gpu))
    res = []
    for py_seq in (False, True):
        net = make_net(gpu)
        net.engine().py_seq = py_seq
        out = net.train_step_rays(batch, step, rand=rand)
        total, _ = total_loss(out, [name2loss[n](CFG) for n in SPHEREPOT_LOSSES], step)
        total.backward()
        res.append((out['ray_rgb'].detach().clone(), float(total.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert set(res[0][2]) == set(res[1][2])
    for n in res[0][2]:
        assert torch.equal(res[0][2][n], res[1][2][n]), n


@pytest.mark.parametrize("mlp_dtype", ["fp32", pytest.param("bf16", marks=pytest.mark.skipif(
    os.environ.get('NU_PY_SEQ') == '1', reason='launch-by-launch sequencing has no bf16-storage mode'))])
def test_gradients_are_bitwise_reproducible(gpu, mlp_dtype):
    """No float atomics anywhere on the path: two evaluations of one step give bit-identical outputs and gradients for every
    parameter, incl. d variance (a sum over all inner points: per-block partials added in index order by the block that finishes
    last -- it used to be one atomicAdd per block, an ulp of run-to-run noise that 60 optimizer steps amplified to 1e-5 in the loss).
    Large enough that the variance sum spans hundreds of blocks."""
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.params import init_stage1_params
    from nu_nerf_amd.synthetic import make_object_rays
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, fused_stage1_loss
    cfg = {'name': 'det', 'network': 'shape', 'database_name': 'synthetic/64', 'is_nerf': True, 'apply_occ_loss': True,
           'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1, 'outer_reg_loss_weight': 0.1,
           'n_samples': 64, 'n_importance': 64, 'n_bg_samples': 32, 'mlp_dtype': mlp_dtype}
    net = NeROShapeRenderer(cfg, training=False)
    net.load_param_dict(init_stage1_params(6033))
    net = net.to(gpu)
    losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
    rays = make_object_rays(2048, seed=901, aim_radius=0.9)
    batch = {k: torch.from_numpy(v).to(gpu) for k, v in rays.items()}
    runs = []
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        torch.manual_seed(5)                                   # the occlusion subsample and the jitter draw from torch's generator
        total, log, out = fused_stage1_loss(net, batch, 20000, losses)
        total.backward()
        runs.append((out['ray_rgb'].detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    assert net.engine().last_ctx['P_in'] > 256 * 200
    assert 'deviation_network.variance' in runs[0][1]
    for rgb, grads in runs[1:]:
        assert torch.equal(rgb, runs[0][0])
        assert set(grads) == set(runs[0][1])
        for n, g in grads.items():
            assert torch.equal(g, runs[0][1][n]), n
