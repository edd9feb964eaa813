"""Inference-only reuse of the hot-path kernels: validation images and SDF grids.

  compute_validation_info   renderer_zerothick.py:636-655   depth, normal, material / light images, traced occlusion
  render_eval               renderer_zerothick.py:397-445   test_step's chunked loop over explicit rays (no image database)
  extract_fields            field.py:1286-1307              dense SDF grid for mesh extraction, batched through the HIP MLP
  extract_geometry          field.py:1310-1317              marching cubes needs PyMCubes (not in this image): gated
"""
import numpy as np
import torch
import torch.nn.functional as F

from .engine import addr
from .nets import Stage1Nets
from .shading_glue import shade


def composite_weights(eng, ctx):
    """The per-sample composite weights of the last render_forward (the training path never materialises them)."""
    from . import _lib as L
    from .engine import c_p
    R, S = ctx['R'], ctx['S']
    w = eng.empty(R, S)
    rgb, acc, bg, ns = eng.empty(R, 3), eng.empty(R), eng.empty(R, 3), eng.empty(R)
    L.check(eng.lib.nu_composite_fwd(c_p(addr(ctx['alpha_rm'])), c_p(addr(ctx['color_rm'])), c_p(addr(ctx['inner_rm'])), R, S,
                                     c_p(addr(w)), c_p(addr(rgb)), c_p(addr(acc)), c_p(addr(bg)), c_p(addr(ns)), eng.stream()),
            "nu_composite_fwd")
    return w


@torch.no_grad()
def compute_validation_info(renderer, z_vals, rays_o, rays_d, weights, step):
    eng = renderer.engine()
    nets = Stage1Nets(eng, renderer._named())
    depth = torch.sum(weights * z_vals, -1, keepdim=True)
    points = (depth * rays_d + rays_o).contiguous()
    with torch.enable_grad():               # the network ops are autograd Functions; nothing is back-propagated
        y, grads = nets.sdf(points)
    inner = (torch.norm(points, dim=-1, keepdim=True) <= 1.0).float()
    out = {'depth': depth, 'normal': ((F.normalize(grads, dim=-1) + 1.0) * 0.5) * inner}
    scfg = renderer.color_network.cfg
    _, occ_info, inter = shade(nets, scfg, renderer.color_network.FG_LUT, points, grads, -F.normalize(rays_d, dim=-1),
                               y[:, 1:], inter_results=True)
    # get_intersection(sn0=128, sn1=9): points with |x| >= 0.999 keep zero weights (field.py:533-553)
    inside = torch.nonzero(torch.norm(points, dim=-1) < 0.999)[:, 0]
    occ_gt = torch.zeros(points.shape[0], 1, device=points.device)
    if inside.numel() > 0:
        occ_gt[inside, 0] = eng.occ_probe(points[inside], occ_info['reflective'][inside].contiguous(), sn0=128, sn1=9)
    out['occ_prob_gt'] = occ_gt
    for k, v in inter.items():
        out[k] = v * inner
    return out


_EVAL_KEYS = ['ray_rgb', 'gradient_error', 'normal', 'depth', 'diffuse_albedo', 'diffuse_light', 'diffuse_color',
              'refraction_light', 'specular_albedo', 'specular_light', 'specular_color', 'specular_ref',
              'transmission_weight', 'roughness', 'occ_prob', 'indirect_light', 'occ_prob_gt']


@torch.no_grad()
def render_eval(renderer, batch, step, chunk=None):
    """test_step's ray loop (renderer_zerothick.py:421-431) over an explicit ray batch {'rays_o','rays_d'[,'rgbs']}:
    chunks of cfg['test_ray_num'] rays, no jitter, cos_anneal 0, is_train=False."""
    trn = int(chunk or renderer.cfg['test_ray_num'])
    is_nerf = renderer.cfg['is_nerf']
    outs = {k: [] for k in _EVAL_KEYS}
    n = batch['rays_o'].shape[0]
    for ri in range(0, n, trn):
        cur = {k: v[ri:ri + trn] for k, v in batch.items()}
        rays_o, rays_d, near, far, hp = renderer._process_nerf_ray_batch(cur)
        if not is_nerf:         # real captures: near / far bracket the unit sphere (renderer_zerothick.py:357)
            near, far = renderer.near_far_from_sphere(rays_o, rays_d)
        o = renderer.render(rays_o, rays_d, near, far, hp, 0, 0, is_train=False, step=step, is_nerf=is_nerf)
        for k in _EVAL_KEYS:
            outs[k].append(o[k].detach())
    outs = {k: torch.cat(v, 0) for k, v in outs.items()}
    if 'rgbs' in batch:
        outs['loss_rgb'] = renderer.compute_rgb_loss(outs['ray_rgb'], batch['rgbs'])
    return outs


@torch.no_grad()
def extract_fields(bound_min, bound_max, resolution, query_func, batch_size=64, outside_val=1.0):
    """Dense grid of query_func (normally lambda x: -sdf(x)) on [bound_min, bound_max]^3, values outside the unit
    sphere replaced by outside_val (field.py:1286-1307).  Same block order and linspace grid as the reference."""
    dev = bound_min.device if torch.is_tensor(bound_min) else None
    N = batch_size
    X = torch.linspace(float(bound_min[0]), float(bound_max[0]), resolution).split(N)
    Y = torch.linspace(float(bound_min[1]), float(bound_max[1]), resolution).split(N)
    Z = torch.linspace(float(bound_min[2]), float(bound_max[2]), resolution).split(N)
    u = np.zeros([resolution, resolution, resolution], dtype=np.float32)
    for xi, xs in enumerate(X):
        for yi, ys in enumerate(Y):
            for zi, zs in enumerate(Z):
                xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing='ij')
                pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
                if dev is not None:
                    pts = pts.to(dev)
                val = query_func(pts).detach().reshape(-1).clone()
                val[torch.norm(pts, dim=-1) >= 1.0] = outside_val
                u[xi * N: xi * N + len(xs), yi * N: yi * N + len(ys), zi * N: zi * N + len(zs)] = \
                    val.reshape(len(xs), len(ys), len(zs)).cpu().numpy()
    return u


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, outside_val=1.0):
    try:
        import mcubes
    except ImportError as e:   # PyMCubes is a pip dependency of the reference (README), absent offline
        raise ImportError("extract_geometry needs PyMCubes for marching cubes; extract_fields() gives the SDF grid") from e
    u = extract_fields(bound_min, bound_max, resolution, query_func, outside_val=outside_val)
    vertices, triangles = mcubes.marching_cubes(u, threshold)
    b_max, b_min = bound_max.detach().cpu().numpy(), bound_min.detach().cpu().numpy()
    return vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :], triangles
