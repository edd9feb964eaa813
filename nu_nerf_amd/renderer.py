"""Drop-in boundary: `name2renderer[cfg['network']](cfg)` -> nn.Module with `forward(data) -> dict`.

Mirrors the reference's renderer protocol (network/renderer_zerothick.py:89-864, registry :2057-2060;
called from train/trainer_zero.py:58,152): same constructor signature `(cfg, training=True)`, same
`default_cfg` keys, same `state_dict()` names (legacy weight-norm `weight_g`/`weight_v`), same
`render(...)` signature and output-dict keys.  The arithmetic runs in the HIP library through
`Stage1Engine`; modules here only own the parameters.  Device-agnostic (no `cuda:0` literals, no
default-tensor-type flips) so one process per GPU works.
"""
import math
import os

import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .engine import Stage1Engine
from .params import init_stage1_params, predictor_dims
from . import synthetic


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's state_dict layout
# ------------------------------------------------------------------------------------------------
class WNLinear(nn.Module):
    """Parameters of `nn.utils.weight_norm(nn.Linear(in, out))`: bias, weight_g [out,1], weight_v [out,in]
    (registration order of the legacy hook; reference field.py:121-122, :386-393)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.weight_g = nn.Parameter(torch.ones(out_dim, 1))
        self.weight_v = nn.Parameter(torch.zeros(out_dim, in_dim))


class PlainLinear(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(out_dim, in_dim))
        self.bias = nn.Parameter(torch.zeros(out_dim))


class SDFNetwork(nn.Module):
    """Parameter layout of reference SDFNetwork (field.py:64-124): lin0..lin8, dims 39-256x3-217-256x4-257."""

    def __init__(self):
        super().__init__()
        dims = [39] + [256] * 8 + [257]
        for l in range(9):
            out_dim = dims[l + 1] - dims[0] if l + 1 == 4 else dims[l + 1]
            setattr(self, f"lin{l}", WNLinear(dims[l], out_dim))
        object.__setattr__(self, '_owner', None)    # weakref to the renderer whose HIP engine evaluates this network

    # evaluation surface of the reference module (field.py:133-170), used by its scripts (extract_mesh_stage1.py:30,
    # train_valid / relight tooling): all three run the HIP MLP of the owning renderer, with gradients to the parameters
    def _nets(self):
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("SDFNetwork is a parameter holder: evaluate it through its NeROShapeRenderer")
        from .nets import Stage1Nets
        eng = owner.engine()
        eng.pack()
        return Stage1Nets(eng, owner._named())

    def forward(self, x):
        shp = x.shape[:-1]
        y, _ = self._nets().sdf(x.reshape(-1, 3).contiguous())
        return y.reshape(*shp, 257)

    def sdf(self, x):
        return self.forward(x)[..., :1]

    def gradient(self, x):
        shp = x.shape[:-1]
        _, n = self._nets().sdf(x.reshape(-1, 3).contiguous())
        return n.reshape(*shp, 3)


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.variance = nn.Parameter(torch.tensor(float(init_val)))


class NeRFNetwork(nn.Module):
    """Parameter layout of reference NeRFNetwork(D=8, W=256, d_in=4, multires=10, multires_view=4) (field.py:246-261)."""

    def __init__(self):
        super().__init__()
        self.pts_linears = nn.ModuleList([PlainLinear(84, 256)] +
                                         [PlainLinear(256 + 84 if i == 4 else 256, 256) for i in range(7)])
        self.views_linears = nn.ModuleList([PlainLinear(27 + 256, 128)])
        self.feature_linear = PlainLinear(256, 256)
        self.alpha_linear = PlainLinear(256, 1)
        self.rgb_linear = PlainLinear(128, 3)


def _predictor(in_dim, out_dim):
    """make_predictor layout: Sequential indices 0,2,4,6 hold the weight-normed linears (field.py:386-395)."""
    return nn.Sequential(WNLinear(in_dim, 256), nn.ReLU(), WNLinear(256, 256), nn.ReLU(), WNLinear(256, 256), nn.ReLU(),
                         WNLinear(256, out_dim), nn.Identity())


def imgs_info_downsample(imgs_info, ratio):
    """renderer_zerothick.py:71-87 for tensors on any device: every image is blurred with the Gaussian of
    utils/base_utils.py:131-137 (sigma = 1 / (3 ratio), odd kernel size from OpenCV's rule, BORDER_REFLECT101) and resized to
    (int(ratio h), int(ratio w)) bilinearly with pixel centres at half-integers (cv2.INTER_LINEAR on float images), and the
    intrinsics are scaled by diag(dw / w, dh / h, 1).  'depths' / 'masks', when the caller's test store carries them, are
    resized like the reference resizes gt_depth / gt_mask (nearest, :402-408).  OpenCV is not available offline: restated from
    its documented definitions, parity with cv2's own output unpinned (it only affects the ground-truth side of validation)."""
    import math
    imgs = imgs_info['imgs'].float()
    b, c, h, w = imgs.shape
    dh, dw = int(ratio * h), int(ratio * w)
    sigma = (1.0 / ratio) / 3.0
    ksize = int(math.ceil(((sigma - 0.8) / 0.3 + 1) * 2 + 1))
    ksize = ksize + 1 if ksize % 2 == 0 else ksize
    out = imgs
    if ksize > 1:
        x = torch.arange(ksize, dtype=torch.float32, device=imgs.device) - (ksize - 1) * 0.5
        k1 = torch.exp(-(x * x) / (2.0 * sigma * sigma))
        k1 = k1 / k1.sum()
        pad = ksize // 2
        out = F.pad(out, (pad, pad, pad, pad), mode='reflect')                       # BORDER_REFLECT101
        out = F.conv2d(out, k1.view(1, 1, 1, ksize).expand(c, 1, 1, ksize), groups=c)
        out = F.conv2d(out, k1.view(1, 1, ksize, 1).expand(c, 1, ksize, 1), groups=c)
    out = F.interpolate(out, size=(dh, dw), mode='bilinear', align_corners=False)
    res = {k: v for k, v in imgs_info.items()}
    res['imgs'] = out
    scale = torch.diag(torch.tensor([dw / w, dh / h, 1.0], dtype=torch.float32, device=imgs_info['Ks'].device))
    res['Ks'] = scale[None] @ imgs_info['Ks'].float()
    for k in ('depths', 'masks'):
        if k in imgs_info:
            v = imgs_info[k]
            v4 = v.reshape(b, 1, h, w).float()
            res[k] = F.interpolate(v4, size=(dh, dw), mode='nearest').reshape(v.shape[:-2] + (dh, dw)).to(v.dtype)
    return res


class AppShadingNetwork(nn.Module):
    default_cfg = {'human_light': False, 'sphere_direction': False, 'light_pos_freq': 6, 'inner_init': -0.95,
                   'roughness_init': 0.0, 'metallic_init': 0.0, 'light_exp_max': 3.0, 'refrac_freq': 6}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        if self.cfg['human_light']:
            raise NotImplementedError("human_light=True is outside the stage-1 hot path of the supported configs")
        for name, k, n_out, _ in predictor_dims(self.cfg['sphere_direction'], self.cfg['refrac_freq'], self.cfg['light_pos_freq']):
            setattr(self, name, _predictor(k, n_out))
        self.register_buffer('FG_LUT', torch.zeros(1, 256, 256, 2))


class InfOutNetwork(nn.Module):
    """Constructed by the reference but never evaluated (field.py:1020-1042); kept for state_dict compatibility."""

    def __init__(self):
        super().__init__()
        self.module0 = nn.Sequential(WNLinear(63, 256), nn.ReLU(), WNLinear(256, 256), nn.ReLU(), WNLinear(256, 256),
                                     nn.ReLU(), WNLinear(256, 256), nn.ReLU(), WNLinear(256, 3), nn.ReLU())


# ------------------------------------------------------------------------------------------------
# autograd wrappers around the engine
# ------------------------------------------------------------------------------------------------
class _RenderCoreFn(torch.autograd.Function):
    """render_core as ONE differentiable op: forward + hand-derived backward, both sequences of HIP kernels."""

    @staticmethod
    def forward(ctx, engine, o, d, z, anneal, train_inv_s, names, spec_pts, *params):
        out, c = engine.render_forward(o, d, z, anneal, spec_pts=spec_pts)
        ctx.engine, ctx.c, ctx.names, ctx.train_inv_s = engine, c, names, train_inv_s
        ctx.set_materialize_grads(False)
        dev = o.device
        P_in = c['P_in']
        if P_in > 0:
            gerr, spec, occ, sdf_in = out['gradient_error'], out['spec_raw'].clone(), out['occ_raw'].clone(), out['sdf_in'].clone()
            # transmission weight and metallic of the inner points (post-sigmoid; renderer_zerothick.py:800-802): differentiable
            # outputs, so that the TransmissionRegLoss / MetallicRegLoss of the registry (network/loss.py:166-192) train the heads
            trans, metal = out['aux'][:, 1:2].clone(), out['aux'][:, 2:3].clone()
        else:
            gerr = torch.zeros(0, device=dev)
            spec, occ, sdf_in = torch.zeros(o.shape[0], 3, device=dev), torch.zeros(0, device=dev), torch.zeros(0, device=dev)
            trans = metal = torch.zeros(0, 1, device=dev)
        return out['rgb'], out['acc'], out['rgb_bg'], gerr, spec, occ, sdf_in, out['nrm_sum'], trans, metal

    @staticmethod
    def backward(ctx, d_rgb, d_acc, d_rgb_bg, d_gerr, d_spec, d_occ, d_sdf, d_nrm, d_trans, d_metal):
        eng, c = ctx.engine, ctx.c
        R = c['R']
        dev = c['alpha_rm'].device
        if d_rgb is None:
            d_rgb = torch.zeros(R, 3, device=dev)
        flat = eng.render_backward(c, d_rgb, d_acc, d_rgb_bg, d_gerr, d_spec, d_occ, d_sdf, train_inv_s=ctx.train_inv_s,
                                   d_nrm_sum=d_nrm, d_trans=d_trans, d_metal=d_metal)
        grads = []
        for n in ctx.names:
            off, shape = eng.grad_views[n]
            if n == 'deviation_network.variance' and not ctx.train_inv_s:
                grads.append(None)
                continue
            grads.append(flat[off:off + eng.grad_numel[n]].view(shape))
        ctx.c = None
        return (None,) * 8 + tuple(grads)


class _SdfValueFn(torch.autograd.Function):
    """sdf(x) for a set of points with gradients to the SDF parameters only (init-SDF regulariser on the shell
    1 < |x| < 1.2; renderer_zerothick.py:804-807)."""

    @staticmethod
    def forward(ctx, engine, pts, names, *params):
        from .engine import addr
        P = pts.shape[0]
        a = engine.sdf_forward(addr(pts), 3, P, keep=True, want_feat=True)
        ctx.engine, ctx.a, ctx.names = engine, a, names
        return a['YX'][:, 0].clone()

    @staticmethod
    def backward(ctx, d_sdf):
        eng, a = ctx.engine, ctx.a
        P = a['P']
        flat = eng.zeros(eng.n_grad)
        dYX = eng.zeros(P, 288)
        dYX[:, 0] = d_sdf
        eng.sdf_backward(a, dYX, None, flat)
        eng.unpack_grads(flat)
        grads = []
        for n in ctx.names:
            off, shape = eng.grad_views[n]
            grads.append(flat[off:off + eng.grad_numel[n]].view(shape))
        ctx.a = None
        return (None, None, None) + tuple(grads)


def linear_to_srgb(x):
    """utils/raw_utils.py:5-11 (used here only on [R,3]-sized per-ray tensors)."""
    eps = torch.finfo(torch.float32).eps
    return torch.where(x <= 0.0031308, 323 / 25 * x, (211 * torch.clamp(x, min=eps) ** (5 / 12) - 11) / 200)


# ------------------------------------------------------------------------------------------------
# the renderer module
# ------------------------------------------------------------------------------------------------
class NeROShapeRenderer(nn.Module):
    default_cfg = {
        'std_net': 'default', 'std_act': 'exp', 'inv_s_init': 0.3, 'freeze_inv_s_step': None,
        'sdf_net': 'default', 'sdf_activation': 'none', 'sdf_bias': 0.5, 'sdf_n_layers': 8, 'sdf_freq': 6,
        'sdf_d_out': 257, 'geometry_init': True,
        'shader_config': {},
        'n_samples': 64, 'n_bg_samples': 32, 'inf_far': 1000.0, 'n_importance': 64, 'up_sample_steps': 4,
        'perturb': 1.0, 'anneal_end': 50000, 'train_ray_num': 512, 'test_ray_num': 1024,
        'clip_sample_variance': True, 'is_nerf': False,
        'database_name': 'nerf_synthetic/lego/black_800',
        'test_downsample_ratio': True, 'downsample_ratio': 0.5, 'val_geometry': False,
        'rgb_loss': 'charbonier', 'apply_occ_loss': True, 'occ_loss_step': 20000, 'occ_loss_max_pn': 2048,
        'occ_sdf_thresh': 0.01,
        'fixed_camera': False,
    }

    candidate_rays = False      # the outer regulariser takes every ray (renderer_zerothick.py:780-781); renderer_std: candidate rays

    def __init__(self, cfg, training=True):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        c = self.cfg
        if (c['sdf_n_layers'], c['sdf_freq'], c['sdf_d_out'], c['std_act'], c['sdf_activation']) != (8, 6, 257, 'exp', 'none'):
            raise NotImplementedError("only the default SDF/variance architecture is built (every shipped config uses it)")
        self.is_nerf = c['is_nerf']
        self.sdf_network = SDFNetwork()
        object.__setattr__(self.sdf_network, '_owner', weakref.ref(self))
        self.deviation_network = SingleVarianceNetwork(c['inv_s_init'])
        self.outer_nerf = NeRFNetwork()
        self.color_network = AppShadingNetwork(c['shader_config'])
        self.infinity_far_bkgr = InfOutNetwork()
        self._engine = None
        self._init_parameters()
        if training:
            self._init_dataset()

    # ---- parameters -------------------------------------------------------------------------
    def _init_parameters(self):
        """Reference initial distributions (geometric SDF init etc., see params.py), seeded from torch's RNG so
        `torch.manual_seed` (train/trainer_zero.py:96) makes construction reproducible."""
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        init = init_stage1_params(seed, sphere_direction=self.color_network.cfg['sphere_direction'],
                                  sdf_bias=self.cfg['sdf_bias'], inv_s_init=self.cfg['inv_s_init'],
                                  refrac_freq=self.color_network.cfg['refrac_freq'])
        self.load_param_dict(init)

    def load_param_dict(self, arrays):
        sd = self.state_dict()
        with torch.no_grad():
            for k, v in arrays.items():
                sd[k].copy_(torch.as_tensor(np.asarray(v)).reshape(sd[k].shape))

    def _named(self):
        d = dict(self.named_parameters())
        d['color_network.FG_LUT'] = self.color_network.FG_LUT
        return d

    def engine(self):
        dev = self.deviation_network.variance.device
        if self._engine is None or self._engine.dev != dev:
            ecfg = dict(self.cfg)
            ecfg.update(self.color_network.cfg)
            self._engine = Stage1Engine(self._named(), dev, ecfg)
            self._grad_names = [n for n in self._engine.grad_views.keys()]
            named = self._named()
            self._grad_params = [named[n] for n in self._grad_names]
            self._sdf_names = [n for n in self._grad_names if n.startswith('sdf_network')]
            self._sdf_params = [named[n] for n in self._sdf_names]
        return self._engine

    def _apply(self, fn, *a, **k):
        self._engine = None  # parameters may move: rebuild the packed tables lazily
        return super()._apply(fn, *a, **k)

    # ---- data ----------------------------------------------------------------------------------
    def _init_dataset(self):
        """Ray-batch store (renderer_zerothick.py:167-190).  `database_name: synthetic/<n_rays>` builds the seeded synthetic
        pool.  For the reference's image databases (nerf/..., real/...) LOADING the images is outside this build (SURVEY.md
        row 14: dataset/database.py needs the dataset files): the module is constructed without a store and the caller hands
        the loaded images over with `set_ray_store(train_imgs_info[, test_imgs_info])` -- the second half of the reference's
        _init_dataset -- before the first train_step."""
        name = self.cfg['database_name']
        self.train_batch_i = 0
        self._batch_dev = None
        self.test_imgs_info = None
        if not name.startswith('synthetic'):
            self.train_batch, self.train_poses, self.tbn = None, None, 0
            return
        parts = name.split('/')
        n = int(parts[1]) if len(parts) > 1 else 1 << 20
        rays = synthetic.make_rays(n, seed=int(self.cfg.get('ray_seed', 6033)))
        self.train_batch = {k: torch.from_numpy(v) for k, v in rays.items()}
        self.train_poses = None
        self.tbn = n

    @staticmethod
    def _construct_ray_batch(imgs_info, device=None):
        """Real captures (renderer_zerothick.py:199-220): per-pixel camera-space directions K^-1 [x + 0.5, y + 0.5, 1], colours
        and image indices, pixel-major per image; built on `device` (default: where the images are).
        -> (ray_batch {'dirs','rgbs','idxs'}, poses, n_rays, h, w)."""
        imgs = imgs_info['imgs']
        device = imgs.device if device is None else torch.device(device)
        imn, _, h, w = imgs.shape
        ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing='ij')
        coords = torch.stack([xs, ys], -1).float().reshape(1, h * w, 2).repeat(imn, 1, 1)
        coords = torch.cat([coords + 0.5, torch.ones(imn, h * w, 1, dtype=torch.float32, device=device)], 2)
        dirs = coords @ torch.inverse(imgs_info['Ks'].to(device)).permute(0, 2, 1)
        rgbs = imgs.to(device).permute(0, 2, 3, 1).reshape(imn, h * w, 3)
        idxs = torch.arange(imn, dtype=torch.int64, device=device)[:, None, None].repeat(1, h * w, 1)
        rn = imn * h * w
        batch = {'dirs': dirs.float().reshape(rn, 3), 'rgbs': rgbs.float().reshape(rn, 3), 'idxs': idxs.reshape(rn, 1)}
        return batch, imgs_info['poses'], rn, h, w

    @staticmethod
    def _construct_nerf_ray_batch(imgs_info, device=None, is_train=True):
        """NeRF-synthetic data (renderer_zerothick.py:222-254): ONE intrinsic matrix (Ks[0]), d = R [ (i - cx) / fx, -(j - cy) / fy,
        -1 ], o = camera centre, world space; masks for the training set.  Vectorised over the images and built on `device`.
        -> (ray_batch {'rgbs','idxs','rays_o','rays_d'[,'masks']}, poses, n_rays, h, w)."""
        imgs = imgs_info['imgs']
        device = imgs.device if device is None else torch.device(device)
        imn, _, h, w = imgs.shape
        j, i = torch.meshgrid(torch.linspace(0, h - 1, h, device=device), torch.linspace(0, w - 1, w, device=device), indexing='ij')
        K = imgs_info['Ks'][0].to(device)
        dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)       # h, w, 3
        poses = imgs_info['poses']
        P = poses.to(device).float()
        # rays_d[n, p, a] = sum_b dirs[p, b] * R_n[a, b]  (the reference's per-image sum(dirs[..., None, :] * R, -1))
        rays_d = torch.sum(dirs.reshape(1, h * w, 1, 3) * P[:, None, :3, :3], -1)
        rays_o = P[:, None, :3, 3].expand(imn, h * w, 3)
        rn = imn * h * w
        batch = {'rgbs': imgs.to(device).permute(0, 2, 3, 1).reshape(rn, 3).float(),
                 'idxs': torch.arange(imn, dtype=torch.int64, device=device)[:, None, None].repeat(1, h * w, 1).reshape(rn, 1),
                 'rays_o': rays_o.reshape(rn, 3).float().contiguous(), 'rays_d': rays_d.reshape(rn, 3).float().contiguous()}
        if is_train:
            batch['masks'] = imgs_info['masks'].to(device).reshape(rn).float()
        return batch, poses, rn, h, w

    def set_ray_store(self, train_imgs_info, test_imgs_info=None, device=None):
        """The second half of the reference's _init_dataset (renderer_zerothick.py:173-190) for a caller that has loaded the
        images itself: `imgs_info` = {'imgs' [n,3,h,w] in [0,1], 'Ks' [n,3,3], 'poses' [n,3,4] (world -> camera for real captures,
        camera -> world for NeRF-synthetic data, as in the reference), 'masks' [n,1,h,w] (NeRF-synthetic only)} as numpy arrays or
        tensors.  The ray store is built ON the module's device (or `device`) and stays there: train_step slices it, no per-step
        host -> device copy.  test_imgs_info (optional; may carry 'depths' [n,h,w]) serves test_step."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        as_t = lambda info: {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v).to(dev)
                             for k, v in info.items()}
        info = as_t(train_imgs_info)
        build = self._construct_nerf_ray_batch if self.is_nerf else self._construct_ray_batch
        self.train_batch, poses, self.tbn, _, _ = build(info, dev)
        self.train_poses = poses.float().to(dev)
        self.train_num = int(info['imgs'].shape[0])
        self.test_imgs_info = as_t(test_imgs_info) if test_imgs_info is not None else None
        self._batch_dev = dev
        self._shuffle_train_batch()

    def _shuffle_train_batch(self):
        self.train_batch_i = 0
        dev = self._batch_dev
        idx = torch.randperm(self.tbn, device=dev)
        for k, v in self.train_batch.items():
            self.train_batch[k] = v[idx]

    # ---- reference helpers ------------------------------------------------------------------------
    def get_anneal_val(self, step):
        if self.cfg['anneal_end'] < 0:
            return 1.0
        return float(np.min([1.0, step / self.cfg['anneal_end']]))

    @staticmethod
    def near_far_from_sphere(rays_o, rays_d):
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return torch.clamp(mid - 1.0, min=1e-3), mid + 1.0

    def get_human_coordinate_poses(self, poses):
        """renderer_zerothick.py:329-345 / renderer.py:329-345 (only consumed by human_light, which is off in every config)."""
        pn = poses.shape[0]
        cam_cen = (-poses[:, :, :3].permute(0, 2, 1) @ poses[:, :, 3:])[..., 0]
        if not self.cfg.get('fixed_camera', False):
            cam_cen = cam_cen.clone()
            cam_cen[..., 2] = 0
        Y = torch.zeros(pn, 3, device=poses.device)
        Y[:, 2] = -1.0
        Z = poses[:, 2, :3].clone()
        Z[:, 2] = 0
        Z = F.normalize(Z, dim=-1)
        X = torch.cross(Y, Z, dim=-1)
        R = torch.stack([X, Y, Z], 1)
        t = -R @ cam_cen[:, :, None]
        return torch.cat([R, t], -1)

    def _process_ray_batch(self, ray_batch, poses):
        """Real-capture rays from camera poses: o = -R^T t, d = normalize(R^T dirs), near/far from the unit sphere
        (renderer_zerothick.py:347-361 / renderer.py:347-361)."""
        rays_d = ray_batch['dirs']
        idxs = ray_batch['idxs'][..., 0]
        rays_o = (poses[:, :, :3].permute(0, 2, 1) @ -poses[:, :, 3:])[idxs, :, 0]
        rays_d = (poses[idxs, :, :3].permute(0, 2, 1) @ rays_d.unsqueeze(-1))[..., 0]
        rays_d = F.normalize(rays_d, dim=-1)
        near, far = self.near_far_from_sphere(rays_o, rays_d)
        return rays_o, rays_d, near, far, self.get_human_coordinate_poses(poses)[idxs]

    def _process_nerf_ray_batch(self, ray_batch, poses=None):
        rays_d = F.normalize(ray_batch['rays_d'], dim=-1)
        rays_o = ray_batch['rays_o']
        n = rays_o.shape[0]
        near = torch.full((n, 1), 0.8, device=rays_o.device)
        far = torch.full((n, 1), 4.5, device=rays_o.device)
        return rays_o, rays_d, near, far, None

    def compute_rgb_loss(self, rgb_pr, rgb_gt):
        kind = self.cfg['rgb_loss']
        if kind == 'l2':
            return torch.sum((rgb_pr - rgb_gt) ** 2, -1)
        if kind == 'l1':
            return torch.sum(F.l1_loss(rgb_pr, rgb_gt, reduction='none'), -1)
        if kind == 'smooth_l1':
            return torch.sum(F.smooth_l1_loss(rgb_pr, rgb_gt, reduction='none', beta=0.25), -1)
        if kind == 'charbonier':
            return torch.sqrt(torch.sum((rgb_gt - rgb_pr) ** 2, dim=-1) + 0.001)
        raise NotImplementedError

    # ---- hot path -----------------------------------------------------------------------------------
    def sample_ray(self, rays_o, rays_d, near, far, perturb, rand=None):
        eng = self.engine()
        u1, u2 = rand[:2] if rand is not None else (None, None)
        return eng.sample_ray(rays_o.contiguous(), rays_d.contiguous(), near.reshape(-1).contiguous(),
                              far.reshape(-1).contiguous(), perturb, u1, u2)

    def render(self, rays_o, rays_d, near, far, human_poses=None, perturb_overwrite=-1, cos_anneal_ratio=0.0,
               is_train=True, step=None, is_nerf=False, rand=None, fused=False):
        """Same contract as the reference `render` (renderer_zerothick.py:614-634).  `rand` optionally injects the
        sampler's two uniform draws, and as an optional third element the permutation of the occlusion-loss subsample
        (renderer_zerothick.py:710) -- parity tests."""
        perturb = self.cfg['perturb']
        if perturb_overwrite >= 0:
            perturb = perturb_overwrite
        eng = self.engine()
        eng.pack()
        with torch.no_grad():
            z_vals = self.sample_ray(rays_o, rays_d, near, far, perturb, rand)
        return self.render_core(rays_o, rays_d, z_vals, human_poses, cos_anneal_ratio=cos_anneal_ratio, step=step,
                                is_train=is_train, is_nerf=is_nerf, _packed=True,
                                occ_perm=rand[2] if rand is not None and len(rand) > 2 else None, fused=fused)

    def render_core(self, rays_o, rays_d, z_vals, human_poses=None, cos_anneal_ratio=0.0, step=None, is_train=True,
                    is_nerf=False, _packed=False, occ_perm=None, fused=False):
        eng = self.engine()
        if not _packed:
            eng.pack()
        cfg = self.cfg
        frozen = cfg['freeze_inv_s_step'] is not None and step < cfg['freeze_inv_s_step']
        spec_pts, cand = self._spec_query_points(rays_o, rays_d, z_vals)
        early = cfg['apply_occ_loss'] and step >= cfg['occ_loss_step'] and os.environ.get('NU_OCC_IDX_EARLY', '1') == '1'
        eng.occ_sdf_thresh = float(cfg['occ_sdf_thresh']) if early else None
        try:
            rgb, acc, rgb_bg, gerr, spec_raw, occ_raw, sdf_in, nrm_sum, trans, metal = _RenderCoreFn.apply(
                eng, rays_o, rays_d, z_vals, float(cos_anneal_ratio), not frozen, self._grad_names, spec_pts, *self._grad_params)
        finally:
            eng.occ_sdf_thresh = None
        exp_max = eng.exp_max
        if fused and is_train:
            # loss.fused_stage1_loss finishes the step in the HIP loss kernels: white background, clamp, colour_spec activation
            # and every per-ray loss integrand are formed there, not in eager torch ops
            outputs = {'gradient_error': gerr if gerr.numel() else torch.zeros(1, device=rgb.device), 'acc': acc,
                       '_raw': dict(rgb=rgb, acc=acc, rgb_bg=rgb_bg, spec_raw=spec_raw, gerr=gerr, nrm_sum=nrm_sum, cand=cand,
                                    is_nerf=bool(is_nerf), exp_max=float(exp_max))}
        else:
            color = rgb + (1. - acc[..., None]) if is_nerf else rgb
            color_spec = linear_to_srgb(torch.exp(torch.clamp(spec_raw, max=exp_max)))
            outputs = {
                'ray_rgb': torch.clamp(color, min=0.0, max=1.0),
                'gradient_error': gerr if gerr.numel() else torch.zeros(1, device=rgb.device),
                'acc': acc,
                'color_bkgr': rgb_bg if cand is None else rgb_bg[cand],
                'color_spec': color_spec if cand is None else color_spec[cand],
            }
        self._last_cand = cand                      # candidate-ray mask of the real-capture outer regulariser (None: every ray takes part)
        self._extra_outputs(outputs, nrm_sum)
        var = self.deviation_network.variance
        inv_s = torch.exp(var * 10.0).clip(1e-6, 1e6)
        # std = mean(1 / inv_s) over the inner points = 1 / inv_s (one value); the gradient to the variance stays attached
        # once inv_s is trainable (renderer_zerothick.py:664-668, :795-798), so StdRecorder's optional loss_std trains it
        if frozen:
            inv_s = inv_s.detach()
        outputs['std'] = (1.0 / inv_s) if gerr.numel() else torch.zeros(1, device=rgb.device)
        c = eng.last_ctx
        P_in = c['P_in']
        if P_in > 0:       # (renderer_zerothick.py:800-802; with a gradient path to the material heads: network/loss.py:166-192)
            outputs['transmission'] = trans
            outputs['metallic'] = metal
        if step < 1000:
            outputs['sdf_pts'], outputs['sdf_vals'] = self._init_reg_points(eng, c, sdf_in)
        if cfg['apply_occ_loss']:
            if P_in > 0:
                outputs['loss_occ'] = self.compute_occ_loss(eng, c, occ_raw, step, perm=occ_perm)
            else:
                outputs['loss_occ'] = torch.zeros(1, device=rgb.device)
        if not is_train:
            from .validation import composite_weights, compute_validation_info
            outputs.update(compute_validation_info(self, z_vals, rays_o, rays_d, composite_weights(eng, c), step))
        return outputs

    def _spec_query_points(self, rays_o, rays_d, z_vals):
        """Zero-thickness renderer: colour_spec is queried per ray from the direction alone
        (renderer_zerothick.py:780-781)."""
        return None, None

    def _extra_outputs(self, outputs, nrm_sum):
        pass

    def _init_reg_points(self, eng, c, sdf_in):
        """Points with |x| < 1.2 and their SDF (renderer_zerothick.py:804-807): the inner set reuses the main pass;
        the shell 1 < |x| < 1.2 of the outer set is evaluated separately (gradients to the SDF parameters)."""
        P_in, P_out = c['P_in'], c['P_out']
        dev = c['alpha_rm'].device
        pts = [c['pt_in'][:P_in, :3]] if P_in > 0 else []
        vals = [sdf_in] if P_in > 0 else []
        if P_out > 0:
            xo = c['pt_out'][:P_out, :3]
            sel = torch.nonzero(torch.norm(xo, dim=-1) < 1.2)[:, 0]
            if sel.numel() > 0:
                shell = xo[sel].contiguous()
                pts.append(shell)
                vals.append(_SdfValueFn.apply(eng, shell, self._sdf_names, *self._sdf_params))
        if not pts:
            return torch.zeros(0, 3, device=dev), torch.zeros(0, device=dev)
        return torch.cat(pts, 0), torch.cat(vals, 0)

    def compute_occ_loss(self, eng, c, occ_raw, step, perm=None):
        """Occlusion loss (renderer_zerothick.py:695-723): L1 between the predicted occlusion probability and the hit
        probability of the reflected ray traced through the SDF (no grad)."""
        cfg = self.cfg
        dev = occ_raw.device
        self._n_occ = 0
        if step < cfg['occ_loss_step']:
            return torch.zeros(1, device=dev)
        P_in = c['P_in']
        with torch.no_grad():
            pt = c['pt_in'][:P_in]
            x, dirs = pt[:, :3], pt[:, 4:7]
            a, SD = c['sdf'], c['shade']['SD']
            sdf, n = a['YX'][:, 0], a['n']
            idx = c.get('occ_idx')            # taken inside the forward when the renderer announced the loss (engine.occ_sdf_thresh)
            if idx is None:
                mask = (torch.norm(x, dim=-1) < 0.999) & (torch.sum(n * dirs, -1) < 0) & (torch.abs(sdf) < cfg['occ_sdf_thresh'])
                idx = torch.nonzero(mask)[:, 0]
            self._n_occ = min(int(idx.numel()), int(cfg['occ_loss_max_pn']))   # points in the mean below (data parallelism: parallel.dp_weight_outputs)
            if idx.numel() > cfg['occ_loss_max_pn']:
                if perm is None:
                    perm = torch.randperm(idx.numel(), device=dev)
                idx = torch.sort(idx[perm[:cfg['occ_loss_max_pn']]])[0]
            if idx.numel() == 0:
                return torch.zeros(1, device=dev)
            nh, nov = SD[idx, :3], SD[idx, 3:4]
            vh = F.normalize(-dirs[idx], dim=-1)
            refl = nov * nh * 2 - vh
            occ_gt = eng.occ_probe(x[idx], refl)
        occ_prob = occ_raw[idx] * 0.5 + 0.5
        return F.l1_loss(occ_prob[:, None], occ_gt[:, None])

    def forward(self, data):
        """trainer protocol (renderer_zerothick.py:822-844): {'step'} -> train_step; {'index','eval','step'} (the
        ValidationEvaluator's call, train/train_valid.py:25-29, first made at step 0) -> test_step."""
        is_train = 'eval' not in data
        step = data['step']
        if not is_train:
            index = data['index']
            index = int(index.reshape(-1)[0]) if torch.is_tensor(index) else int(np.asarray(index).reshape(-1)[0])
            return self.test_step(index, step)
        return self.train_step(step)

    def _test_batch_from_store(self, index, dev):
        """Rays, size and ground truth of test image `index` of the image store (renderer_zerothick.py:397-410): the slice of
        test_imgs_info, down-sampled when cfg says so, through the ray construction of the data convention."""
        info = {k: v[index:index + 1] for k, v in self.test_imgs_info.items()}
        if self.cfg['test_downsample_ratio'] and float(self.cfg['downsample_ratio']) != 1.0:
            info = imgs_info_downsample({k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in info.items()},
                                        float(self.cfg['downsample_ratio']))
        if self.is_nerf:
            batch, poses, rn, h, w = self._construct_nerf_ray_batch(info, dev, is_train=False)
        else:
            batch, poses, rn, h, w = self._construct_ray_batch(info, dev)
            ro, rd, _, _, _ = self._process_ray_batch(batch, poses.float().to(dev))
            batch = {'rays_o': ro, 'rays_d': rd, 'rgbs': batch['rgbs']}
        batch = {k: v for k, v in batch.items() if k in ('rays_o', 'rays_d', 'rgbs')}
        depth = info['depths'][0].reshape(h, w, 1).float().cpu() if 'depths' in info else torch.zeros(h, w, 1)
        mask = (info['masks'][0].reshape(h, w, 1) > 0).to(torch.int32).cpu() if 'masks' in info else torch.zeros(h, w, 1, dtype=torch.int32)
        return batch, h, w, depth, mask

    def test_step(self, index, step):
        """Full-image validation render of camera `index` (renderer_zerothick.py:397-445): chunks of cfg['test_ray_num'] rays, no
        jitter, cos_anneal 0, is_train=False; same output keys and image shapes as the reference.  The rays come from the
        test images handed to set_ray_store (their 'depths' / 'masks' give gt_depth / gt_mask when present) or, for the synthetic
        pool, from every pixel of the (down-sampled) synthetic camera (no depth maps: zeros)."""
        if getattr(self, 'train_batch', None) is None:
            raise RuntimeError("test_step needs the module's ray store: construct with training=True (and, for an image "
                               "database, hand the loaded images over with set_ray_store)")
        from .validation import render_eval
        dev = self.deviation_network.variance.device
        if self.test_imgs_info is not None:
            batch, h, w, depth, mask = self._test_batch_from_store(index, dev)
            outputs = render_eval(self, batch, step)
        else:
            hw = int(self.cfg.get('synthetic_hw', 800))
            ratio = float(self.cfg['downsample_ratio']) if self.cfg['test_downsample_ratio'] else 1.0
            rays, h, w = synthetic.make_image_rays(index, hw=hw, seed=int(self.cfg.get('ray_seed', 6033)), downsample=ratio)
            batch = {k: torch.from_numpy(v).to(dev) for k, v in rays.items()}
            outputs = render_eval(self, batch, step)
            depth, mask = torch.zeros(h, w, 1), torch.zeros(h, w, 1, dtype=torch.int32)
        outputs['gt_rgb'] = batch['rgbs'].reshape(h, w, 3)
        outputs['ray_rgb'] = outputs['ray_rgb'].reshape(h, w, 3)
        outputs['gt_depth'] = depth
        outputs['gt_mask'] = mask
        self.zero_grad()
        return outputs

    def train_step(self, step):
        """renderer_zerothick.py:447-466 on the module's device-resident ray store."""
        if getattr(self, 'train_batch', None) is None:
            raise RuntimeError(f"database '{self.cfg['database_name']}': no ray store yet -- loading image databases is outside this "
                               "build; hand the loaded images over with set_ray_store(train_imgs_info[, test_imgs_info]), or use "
                               "'synthetic/<n_rays>', or call train_step_rays() with your own rays")
        rn = self.cfg['train_ray_num']
        dev = self.deviation_network.variance.device
        if self._batch_dev != dev:
            self.train_batch = {k: v.to(dev) for k, v in self.train_batch.items()}
            if self.train_poses is not None:
                self.train_poses = self.train_poses.to(dev)
            self._batch_dev = dev
            self._shuffle_train_batch()
        batch = {k: v[self.train_batch_i:self.train_batch_i + rn] for k, v in self.train_batch.items()}
        self.train_batch_i += rn
        if self.train_batch_i + rn >= self.tbn:
            self._shuffle_train_batch()
        if 'dirs' in batch:          # real captures: world-space rays from the camera poses (renderer_zerothick.py:347-361)
            return self.train_step_rays(batch, step, poses=self.train_poses)
        return self.train_step_rays(batch, step)

    def train_step_rays(self, batch, step, rand=None, fused=False, poses=None):
        """One training forward on an explicit ray batch {'rays_o','rays_d','rgbs'} -- or {'dirs','idxs','rgbs'} with the
        camera `poses` of a real capture -- (renderer_zerothick.py:447-466).
        fused=True leaves the colour finishing and the RGB loss to loss.fused_stage1_loss (HIP loss kernels)."""
        if 'dirs' in batch:
            rays_o, rays_d, near, far, hp = self._process_ray_batch(batch, poses)
        else:
            rays_o, rays_d, near, far, hp = self._process_nerf_ray_batch(batch)
            if not self.is_nerf:    # real captures: near / far bracket the unit sphere (renderer_zerothick.py:320-327, :357)
                near, far = self.near_far_from_sphere(rays_o, rays_d)
        outputs = self.render(rays_o, rays_d, near, far, hp, -1, self.get_anneal_val(step), is_train=True, step=step,
                              is_nerf=self.is_nerf, rand=rand, fused=fused)
        if not fused:
            outputs['loss_rgb'] = self.compute_rgb_loss(outputs['ray_rgb'], batch['rgbs'])
        return outputs


name2renderer = {'shape': NeROShapeRenderer}
