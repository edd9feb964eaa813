"""Generate tests/golden/*.npz from the REFERENCE itself (runs only in the build container).

The reference (/root/reference, read-only) is imported under the shims of SURVEY.md section 8(c):
stub packages for absent third-party modules, numpy.math, `.cuda()` -> identity, `device=` dropped
from factory calls, and nvdiffrast's `dr.texture` replaced by its bilinear/clamp definition.  The
reference's own classes are constructed, OUR seed-reproducible weights (nu_nerf_amd/params.py) are
loaded into them with `load_state_dict(strict=True)`, and the hot-path functions are run on seeded
synthetic rays.  Only inputs/outputs are stored -- no reference code, no weights (tests rebuild
the weights from the same seed).  /root/reference never travels to the GPU box.

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
import importlib.abc
import importlib.machinery
import math
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

_MISSING = ["nvdiffrast", "mcubes", "cv2", "h5py", "plyfile", "skimage", "transforms3d", "open3d", "trimesh",
            "optix", "cupy", "pymesh", "pymeshlab", "imageio", "tensorboardX", "lpips"]


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Stub(self.__name__ + "." + name)
        m.__path__ = []
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        raise RuntimeError(f"stubbed module {self.__name__} called")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_shims():
    sys.meta_path.insert(0, _Finder())
    np.math = math
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    def _drop_device(fn):
        def wrapped(*a, **k):
            k.pop("device", None)
            return fn(*a, **k)
        return wrapped
    for name in ("zeros", "zeros_like", "ones", "ones_like", "randperm", "linspace", "tensor", "arange"):
        setattr(torch, name, _drop_device(getattr(torch, name)))

    import nvdiffrast.torch as dr  # the stub

    def texture(tex, uv, filter_mode="linear", boundary_mode="clamp"):
        # dr.texture semantics: texel centres at (i+0.5)/N, bilinear, clamp-to-edge
        grid = uv * 2.0 - 1.0
        out = torch.nn.functional.grid_sample(tex.permute(0, 3, 1, 2), grid, mode="bilinear",
                                              padding_mode="border", align_corners=False)
        return out.permute(0, 2, 3, 1)
    dr.texture = texture
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)


def to_t(params):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from network.renderer_zerothick import NeROShapeRenderer  # reference
    import network.field as rfield
    from utils.ref_utils import generate_ide_fn
    from utils.raw_utils import linear_to_srgb as ref_srgb
    from network.loss import name2loss
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    from nu_nerf_amd.synthetic import make_rays, make_jitter

    os.makedirs(OUT, exist_ok=True)
    cfg = {'name': 'golden', 'network': 'shape', 'database_name': 'nerf/spherepot', 'apply_occ_loss': True,
           'occ_loss_step': 15000, 'is_nerf': True, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1,
           'loss': ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'mask', 'outer_reg'],
           'n_samples': 32, 'n_importance': 32, 'n_bg_samples': 16}
    net = NeROShapeRenderer(cfg, training=False)
    params = randomize_for_parity(init_stage1_params(6033), seed=1)
    missing = net.load_state_dict(to_t(params), strict=True)
    print("load_state_dict:", missing)
    losses = [name2loss[n](cfg) for n in cfg['loss']]

    # ---------------- small op-level vectors ----------------
    g = np.random.Generator(np.random.PCG64(11))
    vec = {}
    x = torch.from_numpy(g.uniform(-1, 1, (64, 3)).astype(np.float32))
    emb, _ = rfield.get_embedder(6, 3)
    vec['embed6_in'], vec['embed6_out'] = x.numpy(), emb(x).numpy()
    dirs = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((64, 3)).astype(np.float32)), dim=-1)
    kap = torch.from_numpy(g.uniform(0, 1, (64, 1)).astype(np.float32))
    ide = generate_ide_fn(5)
    vec['ide_dirs'], vec['ide_kappa'], vec['ide_out'] = dirs.numpy(), kap.numpy(), ide(dirs, kap).numpy()
    lin = torch.from_numpy(np.concatenate([g.uniform(0, 0.01, 32), g.uniform(0, 3, 32)]).astype(np.float32))
    vec['srgb_in'], vec['srgb_out'] = lin.numpy(), ref_srgb(lin).numpy()
    bins = torch.sort(torch.from_numpy(g.uniform(0, 4, (16, 33)).astype(np.float32)), -1)[0]
    w = torch.from_numpy(g.uniform(0, 1, (16, 32)).astype(np.float32) ** 4)
    vec['pdf_bins'], vec['pdf_w'] = bins.numpy(), w.numpy()
    vec['pdf_out'] = rfield.sample_pdf(bins, w, 8, det=True).numpy()
    # SDF network: forward, gradient, and a second-order parameter gradient (eikonal-like)
    pts = torch.from_numpy(g.uniform(-0.9, 0.9, (48, 3)).astype(np.float32))
    out = net.sdf_network(pts)
    vec['sdf_pts'], vec['sdf_out'] = pts.numpy(), out.detach().numpy()
    grad = net.sdf_network.gradient(pts.clone())
    vec['sdf_grad'] = grad.detach().numpy()
    cw = torch.from_numpy(g.standard_normal((48, 257)).astype(np.float32))
    cn = torch.from_numpy(g.standard_normal((48, 3)).astype(np.float32))
    net.zero_grad()
    ((net.sdf_network(pts) * cw).sum() + (grad * cn).sum()).backward()
    vec['sdf_cot_y'], vec['sdf_cot_n'] = cw.numpy(), cn.numpy()
    for l in (0, 3, 4, 8):
        for nm in ('weight_g', 'weight_v', 'bias'):
            vec[f'sdf_dl_lin{l}_{nm}'] = getattr(getattr(net.sdf_network, f'lin{l}'), nm).grad.numpy().copy()
    # NeRF++
    p4 = torch.from_numpy(g.uniform(-1, 1, (40, 4)).astype(np.float32))
    vd = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((40, 3)).astype(np.float32)), dim=-1)
    sig, rgb = net.outer_nerf(p4, vd)
    vec['nerf_p4'], vec['nerf_vd'], vec['nerf_sigma'], vec['nerf_rgb'] = p4.numpy(), vd.numpy(), sig.detach().numpy(), rgb.detach().numpy()
    # shading network
    feats = torch.from_numpy((0.3 * g.standard_normal((40, 256))).astype(np.float32))
    nrm = torch.from_numpy(g.standard_normal((40, 3)).astype(np.float32))
    spts = torch.from_numpy(g.uniform(-0.6, 0.6, (40, 3)).astype(np.float32))
    col, occ = net.color_network(spts, nrm, vd, feats, None, step=0)
    vec['shade_pts'], vec['shade_nrm'], vec['shade_view'], vec['shade_feats'] = spts.numpy(), nrm.numpy(), vd.numpy(), feats.numpy()
    vec['shade_color'] = col.detach().numpy()
    for k in ('reflective', 'occ_prob', 'transmission_weight', 'metallic'):
        vec['shade_' + k] = occ[k].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **vec)
    print("ops.npz", {k: v.shape for k, v in vec.items()})

    # ---------------- full train steps ----------------
    def run_step(tag, R, step, ray_seed, perturb=True):
        rays = make_rays(R, seed=ray_seed)
        o, d, rgbs = (torch.from_numpy(rays[k]) for k in ('rays_o', 'rays_d', 'rgbs'))
        u1, u2 = make_jitter(R, cfg['n_bg_samples'], seed=ray_seed + 7)
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
            z = net.sample_ray(o, dn, near, far, 1.0 if perturb else 0.0)
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
        res = {'rays_o': rays['rays_o'], 'rays_d': rays['rays_d'], 'rgbs': rays['rgbs'], 'u1': u1, 'u2': u2,
               'step': np.asarray(step), 'z_vals': z.numpy(), 'total_loss': total.detach().numpy()}
        for k in ('ray_rgb', 'acc', 'color_bkgr', 'color_spec', 'gradient_error', 'std', 'transmission', 'metallic',
                  'loss_occ', 'loss_rgb'):
            if k in outputs:
                res['out_' + k] = outputs[k].detach().numpy()
        for k, v in log.items():
            if k.startswith('loss'):
                res['term_' + k] = torch.mean(v).detach().numpy()
        gn = {}
        for name, prm in net.named_parameters():
            if prm.grad is not None:
                gn[name] = prm.grad
        res['grad_names'] = np.asarray(sorted(gn.keys()))
        res['grad_norms'] = np.asarray([float(gn[k].double().norm()) for k in sorted(gn.keys())])
        # a few full gradients for element-wise comparison
        for k in ('sdf_network.lin0.weight_v', 'sdf_network.lin8.weight_v', 'sdf_network.lin4.weight_g',
                  'outer_nerf.pts_linears.5.weight', 'outer_nerf.rgb_linear.bias',
                  'color_network.albedo_predictor.0.weight_v', 'color_network.outer_light.6.weight_v',
                  'color_network.inner_weight.0.weight_v', 'color_network.roughness_predictor.6.bias',
                  'deviation_network.variance'):
            if k in gn:
                res['grad__' + k] = gn[k].numpy().copy()
        np.savez_compressed(os.path.join(OUT, f"train_{tag}.npz"), **res)
        print(tag, "loss", float(total), {k: float(v) for k, v in res.items() if k.startswith('term_')},
              "inner frac", float(outputs['gradient_error'].numel()) / (R * z.shape[1]))

    run_step("step0_r48", 48, 0, ray_seed=100)
    run_step("step20000_r48", 48, 20000, ray_seed=200)
    run_step("step500_r32_noperturb", 32, 500, ray_seed=300, perturb=False)


def main_std():
    """Vectors from the reference's NON-zero-thickness stage-1 renderer (network/renderer.py) with
    sphere_direction = True and refrac_freq = 3 (configs/shape/real/real_bottle.yaml), real-capture near/far."""
    from network.renderer import NeROShapeRenderer as RefStd   # reference
    from network.loss import name2loss
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    from nu_nerf_amd.synthetic import make_rays, make_jitter
    cfg = {'name': 'golden_std', 'network': 'shape', 'database_name': 'custom/x/720', 'apply_occ_loss': True,
           'occ_loss_step': 15000, 'is_nerf': False, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.05, 'get_mask': False,
           'shader_config': {'sphere_direction': True, 'human_light': False, 'refrac_freq': 3},
           'loss': ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'outer_reg', 'normal_ori'],
           'outer_reg_loss_weight': 0.1, 'n_samples': 64, 'n_importance': 32, 'n_bg_samples': 16}
    net = RefStd(cfg, training=False)
    params = randomize_for_parity(init_stage1_params(6033, sphere_direction=True, refrac_freq=3), seed=1)
    print("load_state_dict (std):", net.load_state_dict(to_t(params), strict=True))
    losses = [name2loss[n](cfg) for n in cfg['loss']]
    R, step, ray_seed = 40, 20000, 400
    rays = make_rays(R, seed=ray_seed)
    # real captures look at the unit sphere from nearby: pull the origins in so near/far_from_sphere is exercised
    o = torch.from_numpy(rays['rays_o']) * 0.6
    d = torch.from_numpy(rays['rays_d'])
    rgbs = torch.from_numpy(rays['rgbs'])
    u1, u2 = make_jitter(R, cfg['n_bg_samples'], seed=ray_seed + 7)
    draws = [torch.from_numpy(u1), torch.from_numpy(u2)]
    real_rand = torch.rand
    torch.rand = lambda *a, **k: draws.pop(0)
    try:
        dn = torch.nn.functional.normalize(d, dim=-1)
        near, far = net.near_far_from_sphere(o, dn)
        net.zero_grad()
        z = net.sample_ray(o, dn, near, far, 1.0)
        outputs = net.render_core(o, dn, z, torch.zeros(R, 3, 4), cos_anneal_ratio=net.get_anneal_val(step), step=step,
                                  is_train=True, is_nerf=False)
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
    res = {'rays_o': o.numpy(), 'rays_d': rays['rays_d'], 'rgbs': rays['rgbs'], 'u1': u1, 'u2': u2, 'step': np.asarray(step),
           'z_vals': z.numpy(), 'near': near.numpy(), 'far': far.numpy(), 'total_loss': total.detach().numpy()}
    for k in ('ray_rgb', 'acc', 'color_bkgr', 'color_spec', 'gradient_error', 'loss_normal', 'loss_occ', 'loss_rgb'):
        res['out_' + k] = outputs[k].detach().numpy()
    for k, v in log.items():
        if k.startswith('loss'):
            res['term_' + k] = torch.mean(v).detach().numpy()
    gn = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    res['grad_names'] = np.asarray(sorted(gn.keys()))
    res['grad_norms'] = np.asarray([float(gn[k].double().norm()) for k in sorted(gn.keys())])
    for k in ('sdf_network.lin0.weight_v', 'sdf_network.lin8.weight_v', 'color_network.outer_light.0.weight_v',
              'color_network.refrac_light.0.weight_v', 'color_network.roughness_predictor.6.bias',
              'deviation_network.variance'):
        res['grad__' + k] = gn[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "train_std_step20000_r40.npz"), **res)
    print("std loss", float(total), {k: float(v) for k, v in res.items() if k.startswith('term_')},
          "candidates", int(outputs['color_spec'].shape[0]), "of", R)


if __name__ == "__main__":
    if "--std-only" in sys.argv:
        install_shims()
        torch.manual_seed(0)
        torch.set_num_threads(8)
        main_std()
    else:
        main()
        main_std()
