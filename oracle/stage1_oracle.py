"""CPU oracle for NU-NeRF's stage-1 training step (TEST INFRASTRUCTURE -- not product code).

A from-scratch restatement, in plain fp32 PyTorch-on-CPU, of the algorithm in the reference's
`network/renderer_zerothick.py` (NeROShapeRenderer), `network/field.py`, `utils/ref_utils.py`,
`utils/raw_utils.py`, `network/loss.py` and the loss assembly of `train/trainer_zero.py`.
Every function cites the reference file:line it follows (paths relative to /root/reference).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module, and only as the checker / the reported CPU baseline.  The product path (nu_nerf_amd/) never
does and has no CPU fallback.

Pinning: the reference has no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, imported under shims in the build
container by `oracle/gen_golden.py`; the resulting vectors live in `tests/golden/*.npz` and
`tests/test_oracle_golden.py` replays them.  Two pieces stay "parity unpinned": nvdiffrast's
`dr.texture` (absent third-party CUDA; restated as bilinear/clamp) and OptiX (stage 2).

Functional style: networks are dicts `{reference state_dict name: tensor}`; gradients come from
torch autograd on those tensors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# configuration defaults (renderer_zerothick.py:90-137, field.py:558-567, spherepot.yaml)
# --------------------------------------------------------------------------------------------
DEFAULT_CFG = {
    'n_samples': 64, 'n_importance': 64, 'n_bg_samples': 32, 'up_sample_steps': 4,
    'perturb': 1.0, 'anneal_end': 50000, 'clip_sample_variance': True, 'freeze_inv_s_step': 15000,
    'is_nerf': True, 'rgb_loss': 'charbonier', 'apply_occ_loss': True, 'occ_loss_step': 15000,
    'occ_loss_max_pn': 2048, 'occ_sdf_thresh': 0.01, 'eikonal_weight': 0.1,
    'sphere_direction': False, 'light_exp_max': 3.0, 'outer_reg_loss_weight': 0.5,
}


# --------------------------------------------------------------------------------------------
# encodings
# --------------------------------------------------------------------------------------------
def embed(x, n_freq):
    """NeRF positional encoding [x, sin(2^k x), cos(2^k x)]_k  (field.py:14-61)."""
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def _ide_tables(deg_view=5):
    """(m, l) list and polynomial coefficient matrix of the IDE (ref_utils.py:7-79)."""
    ml = [(m, 2 ** i) for i in range(deg_view) for m in range(2 ** i + 1)]
    l_max = 2 ** (deg_view - 1)
    mat = np.zeros((l_max + 1, len(ml)))

    def gen_binom(a, k):
        return np.prod(a - np.arange(k)) / math.factorial(k)

    def legendre_coeff(l, m, k):
        return ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m)
                * gen_binom(0.5 * (l + k + m - 1.0), l))

    def sh_coeff(l, m, k):
        return np.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * np.pi * math.factorial(l + m))) \
            * legendre_coeff(l, m, k)

    for i, (m, l) in enumerate(ml):
        for k in range(l - m + 1):
            mat[k, i] = sh_coeff(l, m, k)
    return np.asarray(ml, np.float32), mat.astype(np.float32)


_IDE_ML, _IDE_MAT = _ide_tables(5)


def ide(xyz, kappa_inv):
    """Integrated directional encoding, 72-d (ref_utils.py:84-114): complex spherical harmonics
    (x+iy)^m * P(z), attenuated by exp(-l(l+1)/2 * kappa_inv); output [Re(36), Im(36)]."""
    x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
    mat = torch.from_numpy(_IDE_MAT)
    ml = torch.from_numpy(_IDE_ML)
    zpow = torch.cat([z ** i for i in range(mat.shape[0])], -1)
    xy = torch.complex(x, y)
    xypow = torch.cat([xy ** m for m in ml[:, 0]], -1)
    harm = xypow * (zpow @ mat)
    sigma = 0.5 * ml[:, 1] * (ml[:, 1] + 1)
    val = harm * torch.exp(-sigma * kappa_inv)
    return torch.cat([val.real, val.imag], -1)


def linear_to_srgb(x):
    """raw_utils.py:5-11."""
    eps = torch.finfo(torch.float32).eps
    lo = 323 / 25 * x
    hi = (211 * torch.clamp(x, min=eps) ** (5 / 12) - 11) / 200
    return torch.where(x <= 0.0031308, lo, hi)


# --------------------------------------------------------------------------------------------
# networks (functional, parameters by reference state_dict name)
# --------------------------------------------------------------------------------------------
def wn_weight(params, prefix):
    """Legacy nn.utils.weight_norm (dim=0): W = g * v / ||v||_row  (field.py:121-122)."""
    g, v = params[prefix + '.weight_g'], params[prefix + '.weight_v']
    return v * (g / v.norm(dim=1, keepdim=True))


def sdf_forward(params, x, prefix='sdf_network'):
    """SDFNetwork.forward: 39-d embedding, 9 weight-normed linears, Softplus(beta=100), skip concat
    before layer 4 scaled by 1/sqrt(2)  (field.py:133-150).  Returns [..., 257]."""
    e = embed(x, 6)
    h = e
    for l in range(9):
        if l == 4:
            h = torch.cat([h, e], -1) / math.sqrt(2)
        h = F.linear(h, wn_weight(params, f'{prefix}.lin{l}'), params[f'{prefix}.lin{l}.bias'])
        if l < 8:
            h = F.softplus(h, beta=100)
    return h


def sdf_gradient(params, x, prefix='sdf_network'):
    """d sdf / d x with the graph kept for second-order terms (field.py:158-170)."""
    x = x.detach().requires_grad_(True)
    with torch.enable_grad():
        y = sdf_forward(params, x, prefix)[..., :1]
        (g,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)
    return g


def sdf_gradient_wrt(params, x, prefix='sdf_network'):
    """Same as sdf_gradient but keeps x's own graph (stage 2: x depends on the IoR network), as the reference's
    `gradient()` does when handed a non-leaf tensor (field.py:158-170)."""
    if not x.requires_grad:
        x = x.detach().requires_grad_(True)
    with torch.enable_grad():
        y = sdf_forward(params, x, prefix)[..., :1]
        (g,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)
    return g


def inv_s_value(params):
    """SingleVarianceNetwork: exp(10 * variance)  (field.py:197-199)."""
    return torch.exp(params['deviation_network.variance'] * 10.0)


def nerf_forward(params, pts4, views, prefix='outer_nerf'):
    """NeRF++ MLP: 8x256 ReLU, input re-concatenated AFTER layer 4, density head, one 283->128 view
    layer, rgb head  (field.py:265-289).  Returns (sigma[...,1], rgb_raw[...,3])."""
    e = embed(pts4, 10)
    ev = embed(views, 4)
    h = e
    for i in range(8):
        h = F.relu(F.linear(h, params[f'{prefix}.pts_linears.{i}.weight'], params[f'{prefix}.pts_linears.{i}.bias']))
        if i == 4:
            h = torch.cat([e, h], -1)
    sigma = F.linear(h, params[f'{prefix}.alpha_linear.weight'], params[f'{prefix}.alpha_linear.bias'])
    feat = F.linear(h, params[f'{prefix}.feature_linear.weight'], params[f'{prefix}.feature_linear.bias'])
    h = torch.cat([feat, ev], -1)
    h = F.relu(F.linear(h, params[f'{prefix}.views_linears.0.weight'], params[f'{prefix}.views_linears.0.bias']))
    rgb = F.linear(h, params[f'{prefix}.rgb_linear.weight'], params[f'{prefix}.rgb_linear.bias'])
    return sigma, rgb


def predictor(params, prefix, x, act, exp_max=0.0):
    """make_predictor: 4 weight-normed linears (sequential indices 0,2,4,6), ReLU x3, final
    activation sigmoid / exp(min(.,exp_max)) / none  (field.py:371-408, :312-318)."""
    h = x
    for j, idx in enumerate((0, 2, 4, 6)):
        h = F.linear(h, wn_weight(params, f'{prefix}.{idx}'), params[f'{prefix}.{idx}.bias'])
        if j < 3:
            h = F.relu(h)
    if act == 'sigmoid':
        return torch.sigmoid(h)
    if act == 'exp':
        return torch.exp(torch.clamp(h, max=exp_max))
    return h


def lut_bilinear_clamp(lut, uv):
    """nvdiffrast dr.texture(filter='linear', boundary='clamp') restated (field.py:719-722):
    texel centres at (i+0.5)/N, clamped bilinear.  lut [H, W, C], uv [P, 2] = (u -> W, v -> H)."""
    H, W, _ = lut.shape
    fx = torch.clamp(uv[:, 0] * W - 0.5, 0.0, W - 1.0)
    fy = torch.clamp(uv[:, 1] * H - 0.5, 0.0, H - 1.0)
    x0 = torch.floor(fx).long()
    y0 = torch.floor(fy).long()
    x1 = torch.clamp(x0 + 1, max=W - 1)
    y1 = torch.clamp(y0 + 1, max=H - 1)
    tx = (fx - x0.float())[:, None]
    ty = (fy - y0.float())[:, None]
    top = lut[y0, x0] * (1 - tx) + lut[y0, x1] * tx
    bot = lut[y1, x0] * (1 - tx) + lut[y1, x1] * tx
    return top * (1 - ty) + bot * ty


def offset_points_to_sphere(points):
    """field.py:447-455."""
    n = torch.norm(points, dim=-1, keepdim=True)
    return torch.where(n > 0.999, points / n * 0.999, points)


def sphere_exit_distance(pts, dirs):
    """Distance along dirs to the unit sphere from inside it (field.py:458-464)."""
    b = torch.sum(pts * dirs, -1, keepdim=True)
    c = torch.sum(pts ** 2, -1, keepdim=True)
    disc = b ** 2 - c + 1
    return -b + torch.sqrt(disc + 1e-6)


def shading_forward(params, cfg, points, normals, view_dirs, feats, prefix='color_network'):
    """AppShadingNetwork.forward (field.py:684-777) incl. predict_specular_lights (:636-667) and
    predict_diffuse_lights (:669-682); human_light disabled (default).  Returns (srgb color [P,3],
    occ_info dict)."""
    exp_max = cfg['light_exp_max']
    sd = cfg['sphere_direction']
    n = F.normalize(normals, dim=-1)
    v = F.normalize(view_dirs, dim=-1)
    nov = torch.sum(n * v, -1, keepdim=True)
    refl = nov * n * 2 - v

    fx = torch.cat([feats, points], -1)
    metallic = predictor(params, f'{prefix}.metallic_predictor', fx, 'sigmoid')
    rough = predictor(params, f'{prefix}.roughness_predictor', fx, 'sigmoid')
    albedo = predictor(params, f'{prefix}.albedo_predictor', fx, 'sigmoid')
    trans = predictor(params, f'{prefix}.transmisstion_weight', fx, 'sigmoid')

    def outer(enc_dir, rough_for_sph, direction):
        if sd:
            sp = offset_points_to_sphere(points)
            sp = F.normalize(sp + direction * sphere_exit_distance(sp, direction), dim=-1)
            enc_dir = torch.cat([enc_dir, ide(sp, rough_for_sph)], -1)
        return predictor(params, f'{prefix}.outer_light', enc_dir, 'exp', exp_max)

    ones = torch.ones_like(rough)
    diffuse_light = outer(ide(n, ones), ones, n)
    diffuse_color = (1 - metallic) * albedo * diffuse_light
    spec_albedo = 0.04 * (1 - metallic) + metallic * albedo

    enc_r = ide(refl, rough)
    enc_r0 = ide(refl, torch.zeros_like(rough))
    pe = embed(points, 6)
    direct = outer(enc_r, rough, refl)
    direct0 = outer(enc_r0, rough, refl)
    indirect = predictor(params, f'{prefix}.inner_light', torch.cat([pe, enc_r], -1), 'exp', exp_max)
    indirect0 = predictor(params, f'{prefix}.inner_light', torch.cat([pe, enc_r0], -1), 'exp', exp_max)
    occ = predictor(params, f'{prefix}.inner_weight',
                    torch.cat([pe.detach(), embed(refl, 6).detach()], -1), 'none')
    occ = occ * 0.5 + 0.5
    occ_c = torch.clamp(occ, 0.0, 1.0)
    light = indirect * occ_c + direct * (1 - occ_c)
    light0 = indirect0 * occ_c + direct0 * (1 - occ_c)

    t = torch.clamp(1 - nov, 0.0, 1.0)
    fres = torch.clamp(0.04 + 0.96 * t * t * t * t * t, 0.0, 1.0)
    rf = cfg.get('refrac_freq', 6)                                  # field.py:590-591
    refrac = predictor(params, f'{prefix}.refrac_light', torch.cat([embed(points, rf), embed(v, rf)], -1),
                       'exp', exp_max)
    uv = torch.cat([torch.clamp(nov, 0.0, 1.0), torch.clamp(rough, 0.0, 1.0)], -1)
    fg = lut_bilinear_clamp(params[f'{prefix}.FG_LUT'][0], uv)
    spec_color = (spec_albedo * fg[:, 0:1] + fg[:, 1:2]) * light
    color = (diffuse_color + spec_color) * (1 - trans) + (fres * light0 + (1 - fres) * refrac) * trans
    occ_info = {'reflective': refl, 'occ_prob': occ, 'transmission_weight': trans, 'metallic': metallic,
                'roughness': rough, 'albedo': albedo}
    return linear_to_srgb(color), occ_info


# --------------------------------------------------------------------------------------------
# sampler
# --------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, n, det=True, u=None):
    """Inverse-CDF sampling (field.py:468-498)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if u is None:
        assert det
        u = torch.linspace(0.5 / n, 1.0 - 0.5 / n, steps=n).expand(list(cdf.shape[:-1]) + [n])
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def _excl_cumprod_weights(alpha):
    """w_j = alpha_j * prod_{i<j} (1 - alpha_i + 1e-7)  (renderer_zerothick.py:550-551, :773-774)."""
    ones = torch.ones_like(alpha[..., :1])
    return alpha * torch.cumprod(torch.cat([ones, 1. - alpha + 1e-7], -1), -1)[..., :-1]


def upsample(rays_o, rays_d, z, sdf, n_imp, inv_s):
    """One NeuS up-sampling round at fixed inv_s (renderer_zerothick.py:525-554)."""
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
    r = torch.linalg.norm(pts, dim=-1)
    inside = (r[:, :-1] < 1.0) | (r[:, 1:] < 1.0)
    s0, s1 = sdf[:, :-1], sdf[:, 1:]
    z0, z1 = z[:, :-1], z[:, 1:]
    mid = (s0 + s1) * 0.5
    cos = (s1 - s0) / (z1 - z0 + 1e-5)
    prev = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], -1)
    cos = torch.minimum(prev, cos).clip(-1e3, 0.0) * inside
    dz = z1 - z0
    p = torch.sigmoid((mid - cos * dz * 0.5) * inv_s)
    q = torch.sigmoid((mid + cos * dz * 0.5) * inv_s)
    alpha = (p - q + 1e-5) / (p + 1e-5)
    return sample_pdf(z, _excl_cumprod_weights(alpha), n_imp, det=True).detach()


def cat_z_vals(params, rays_o, rays_d, z, z_new, sdf, last):
    """Merge new samples, evaluating their SDF unless this is the last round (:556-570)."""
    zc = torch.cat([z, z_new], -1)
    zs, index = torch.sort(zc, dim=-1)
    if not last:
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_new[..., None]
        s_new = sdf_forward(params, pts.reshape(-1, 3))[..., 0].reshape(z_new.shape)
        sdf = torch.gather(torch.cat([sdf, s_new], -1), -1, index)
    return zs, sdf


def sample_ray(params, cfg, rays_o, rays_d, near, far, perturb, rand=None):
    """Hierarchical sampler (renderer_zerothick.py:572-612).  `rand` = (U[R,1], U[R,n_bg]) replaces
    the two torch.rand draws so that a device implementation can be compared sample for sample."""
    nc, nbg, ni, steps = cfg['n_samples'], cfg['n_bg_samples'], cfg['n_importance'], cfg['up_sample_steps']
    R = rays_o.shape[0]
    z = near + (far - near) * torch.linspace(0.0, 1.0, nc)[None, :]
    zo = torch.linspace(1e-3, 1.0 - 1.0 / (nbg + 1.0), nbg)
    if perturb > 0:
        u1, u2 = rand if rand is not None else (torch.rand([R, 1]), torch.rand([R, nbg]))
        z = z + (u1 - 0.5) * 2.0 / nc
        mids = 0.5 * (zo[1:] + zo[:-1])
        upper = torch.cat([mids, zo[-1:]], -1)
        lower = torch.cat([zo[:1], mids], -1)
        zo = lower[None, :] + (upper - lower)[None, :] * u2
    zo = far / torch.flip(zo, dims=[-1]) + 1.0 / nbg
    if zo.dim() == 1:
        zo = zo[None, :].expand(R, nbg)
    with torch.no_grad():
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
        sdf = sdf_forward(params, pts.reshape(-1, 3))[..., 0].reshape(R, nc)
        for i in range(steps):
            if cfg['clip_sample_variance']:
                s = torch.clamp(inv_s_value(params), max=64 * 2 ** i)
            else:
                s = torch.tensor(64.0 * 2 ** i)
            z_new = upsample(rays_o, rays_d, z, sdf, ni // steps, s)
            z, sdf = cat_z_vals(params, rays_o, rays_d, z, z_new, sdf, last=(i + 1 == steps))
    return torch.cat([z, zo], -1)


# --------------------------------------------------------------------------------------------
# alpha + render core
# --------------------------------------------------------------------------------------------
def compute_sdf_alpha(params, cfg, points, dists, dirs, cos_anneal, step):
    """NeuS alpha with annealed cosine (renderer_zerothick.py:657-685)."""
    out = sdf_forward(params, points)
    sdf, feats = out[..., 0], out[..., 1:]
    grads = sdf_gradient(params, points)
    s = inv_s_value(params).clip(1e-6, 1e6)
    if cfg['freeze_inv_s_step'] is not None and step < cfg['freeze_inv_s_step']:
        s = s.detach()
    cos = (dirs * grads).sum(-1)
    it = -(F.relu(-cos * 0.5 + 0.5) * (1.0 - cos_anneal) + F.relu(-cos) * cos_anneal)
    p = torch.sigmoid((sdf - it * dists * 0.5) * s)
    q = torch.sigmoid((sdf + it * dists * 0.5) * s)
    alpha = ((p - q + 1e-5) / (p + 1e-5)).clip(0.0, 1.0)
    return alpha, grads, feats, s, sdf


def compute_density_alpha(params, points, dists, dirs):
    """NeRF++ background alpha/colour (renderer_zerothick.py:687-693, :515-516)."""
    n = torch.norm(points, dim=-1, keepdim=True)
    sigma, rgb = nerf_forward(params, torch.cat([points / n, 1.0 / n], -1), dirs)
    alpha = 1.0 - torch.exp(-F.softplus(sigma[..., 0]) * dists)
    return alpha, linear_to_srgb(torch.exp(torch.clamp(rgb, max=5.0)))


def get_weights(params, z, origins, dirs):
    """field.py:501-521 with sdf_fun = SDFNetwork.sdf, inv_fun = SingleVarianceNetwork."""
    pts = z[..., None] * dirs[:, None, :] + origins[:, None, :]
    s = inv_s_value(params)
    sdf = sdf_forward(params, pts.reshape(-1, 3))[..., 0].reshape(z.shape)
    s0, s1, z0, z1 = sdf[:, :-1], sdf[:, 1:], z[:, :-1], z[:, 1:]
    mid = (s0 + s1) * 0.5
    cos = (s1 - s0) / (z1 - z0 + 1e-5)
    surf = cos < 0
    cos = torch.clamp(cos, max=0)
    dz = z1 - z0
    p = torch.sigmoid((mid - cos * dz * 0.5) * s)
    q = torch.sigmoid((mid + cos * dz * 0.5) * s)
    alpha = (p - q + 1e-5) / (p + 1e-5) * surf.float()
    w = _excl_cumprod_weights(alpha)
    mid = torch.where(surf, mid, -torch.ones_like(mid))
    return w, mid


def get_intersection(params, pts, dirs, sn0=64, sn1=16):
    """Secondary-ray hit probabilities inside the unit sphere (field.py:524-554)."""
    inside = torch.norm(pts, dim=-1) < 0.999
    pn = pts.shape[0]
    hz, hw, hs = torch.zeros(pn, sn1 - 1), torch.zeros(pn, sn1 - 1), -torch.ones(pn, sn1 - 1)
    if inside.any():
        p, d = pts[inside], dirs[inside]
        dmax = sphere_exit_distance(p, d)
        with torch.no_grad():
            z = dmax * torch.linspace(0, 1, sn0)[None, :]
            w, _ = get_weights(params, z, p, d)
            z2 = sample_pdf(z, w, sn1, True)
            w2, mid2 = get_weights(params, z2, p, d)
            zm = (z2[:, 1:] + z2[:, :-1]) * 0.5
        hz[inside], hw[inside], hs[inside] = zm, w2, mid2
    return hz, hw, hs


def compute_occ_loss(params, cfg, occ_info, points, sdf, grads, dirs, step, perm=None):
    """renderer_zerothick.py:695-723.  `perm` replaces the CUDA randperm when the candidate set
    exceeds occ_loss_max_pn."""
    if step < cfg['occ_loss_step']:
        return torch.zeros(1)
    mask = (torch.norm(points, dim=-1) < 0.999) & (torch.sum(grads * dirs, -1) < 0) & \
           (torch.abs(sdf) < cfg['occ_sdf_thresh'])
    if int(mask.sum()) > cfg['occ_loss_max_pn']:
        idx = torch.nonzero(mask)[:, 0]
        perm = torch.randperm(idx.shape[0]) if perm is None else perm
        keep = idx[perm[:cfg['occ_loss_max_pn']]]
        mask = torch.zeros_like(mask)
        mask[keep] = True
    if mask.any():
        _, prob, _ = get_intersection(params, points[mask], occ_info['reflective'][mask], 64, 16)
        return F.l1_loss(occ_info['occ_prob'][mask], prob.sum(-1, keepdim=True))
    return torch.zeros(1)


def render_core(params, cfg, rays_o, rays_d, z_vals, step, cos_anneal=0.0, is_nerf=True, occ_perm=None, std=False):
    """Stage-1 render_core for training (renderer_zerothick.py:725-820).  std=True follows the non-zero-thickness
    file instead (network/renderer.py:738-859): loss_normal, candidate-ray colour_spec / colour_bkgr."""
    R, S = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, dists[..., -1:]], -1)
    mid = z_vals + dists * 0.5
    points = rays_o[:, None, :] + rays_d[:, None, :] * mid[..., None]
    inner = torch.norm(points, dim=-1) <= 1.0
    outer = ~inner
    dirs = F.normalize(rays_d[:, None, :].expand(R, S, 3), dim=-1)

    alpha = torch.zeros(R, S)
    color = torch.zeros(R, S, 3)
    if outer.any():
        a, c = compute_density_alpha(params, points[outer], dists[outer], -dirs[outer])
        alpha = alpha.masked_scatter(outer, a) if False else alpha.index_put((outer,), a)
        color = color.index_put((outer,), c)
    alpha_bg, color_bg = alpha, color

    out = {}
    normal_dir = torch.zeros(R, S, 1)
    if inner.any():
        a, grads, feats, s, sdf = compute_sdf_alpha(params, cfg, points[inner], dists[inner], dirs[inner],
                                                    cos_anneal, step)
        gfl = torch.zeros(R, S, 3).index_put((inner,), grads)
        normal_dir = torch.clamp(torch.sum(gfl * dirs, dim=-1, keepdim=True), min=0.0)
        c, occ_info = shading_forward(params, cfg, points[inner], grads, -dirs[inner], feats)
        alpha = alpha.index_put((inner,), a)
        color = color.index_put((inner,), c)
        out['gradient_error'] = (torch.linalg.norm(grads, dim=-1) - 1.0) ** 2
        out['std'] = torch.mean(1 / s)
        out['transmission'] = occ_info['transmission_weight']
        out['metallic'] = occ_info['metallic']
        out['normal_raw'] = grads
        out['sdf'] = sdf
        out['inv_s'] = s
    else:
        out['gradient_error'] = torch.zeros(1)
        out['std'] = torch.zeros(1)

    w = _excl_cumprod_weights(alpha)
    rgb = (color * w[..., None]).sum(1)
    w_bg = _excl_cumprod_weights(alpha_bg)
    out['color_bkgr'] = (color_bg * w_bg[..., None]).sum(1)
    if not std:
        enc = ide(dirs[:, 0, :], torch.zeros(R, 1))
        out['color_spec'] = linear_to_srgb(predictor(params, 'color_network.outer_light', enc, 'exp',
                                                     cfg['light_exp_max']))
    else:
        # renderer.py:705-725
        out['loss_normal'] = (normal_dir * w[..., None]).sum(dim=1)
        pc, dc = points[:, 64, :], dirs[:, 0, :]
        cand = torch.norm(pc, dim=-1) <= 1.0
        pf, df = pc[cand], dc[cand]
        enc = ide(df, torch.zeros(df.shape[0], 1))
        if cfg['sphere_direction']:
            sp = offset_points_to_sphere(pf)
            sp = F.normalize(sp + df * sphere_exit_distance(sp, df), dim=-1)
            enc = torch.cat([enc, ide(sp, torch.zeros(sp.shape[0], 1))], -1)
        out['color_spec'] = linear_to_srgb(predictor(params, 'color_network.outer_light', enc, 'exp',
                                                     cfg['light_exp_max']))
        out['color_bkgr'] = out['color_bkgr'][cand]
    acc = w.sum(-1)
    if is_nerf:
        rgb = rgb + (1. - acc[..., None])
    out['ray_rgb'] = torch.clamp(rgb, 0.0, 1.0)
    out['acc'] = acc
    out['weights'] = w
    out['alpha'] = alpha
    out['sampled_color'] = color
    out['inner_mask'] = inner

    if step < 1000:
        m = torch.norm(points, dim=-1) < 1.2
        out['sdf_pts'] = points[m]
        out['sdf_vals'] = sdf_forward(params, points[m])[..., 0]
    if cfg['apply_occ_loss']:
        if inner.any():
            out['loss_occ'] = compute_occ_loss(params, cfg, occ_info, points[inner], sdf, grads, dirs[inner],
                                               step, occ_perm)
        else:
            out['loss_occ'] = torch.zeros(1)
    return out


def rgb_loss(pr, gt, kind='charbonier'):
    """renderer_zerothick.py:501-513."""
    if kind == 'charbonier':
        return torch.sqrt(torch.sum((gt - pr) ** 2, -1) + 0.001)
    if kind == 'l2':
        return torch.sum((pr - gt) ** 2, -1)
    if kind == 'l1':
        return torch.sum(torch.abs(pr - gt), -1)
    raise NotImplementedError(kind)


def init_sdf_reg(sdf_pts, sdf_vals, step):
    """InitSDFRegLoss (loss.py:115-149)."""
    norm = torch.norm(sdf_pts, dim=-1)
    small = norm < 0.1
    if small.any():
        sl = torch.mean(torch.clamp(sdf_vals[small] - (norm[small] - 0.1), min=0.0))
        sl = torch.sum(sl) / (torch.sum(sl > 1e-5) + 1e-3)
    else:
        sl = torch.zeros(1)
    large = norm > 1.05
    if large.any():
        ll = torch.clamp((norm[large] - 1.05) - sdf_vals[large], min=0.0)
        ll = torch.sum(ll) / (torch.sum(ll > 1e-5) + 1e-3)
    else:
        ll = torch.zeros(1)
    w = (np.cos((step / 1000) * np.pi) + 1) / 2
    return ll * w, sl * w


def assemble_losses(out, cfg, step):
    """Loss dict of the Spherepot config and its total (loss.py; trainer_zero.py:153-161):
    total = sum of means of every entry whose key starts with 'loss'."""
    terms = {'loss_rgb': out['loss_rgb'], 'loss_eikonal': out['gradient_error'] * cfg['eikonal_weight']}
    if 'sdf_vals' in out and step < 1000:
        terms['loss_sdf_large'], terms['loss_sdf_small'] = init_sdf_reg(out['sdf_pts'], out['sdf_vals'], step)
    if 'loss_occ' in out:
        terms['loss_occ'] = torch.mean(out['loss_occ']).reshape(1)
    if step >= 15000:
        terms['loss_outer_reg'] = F.mse_loss(out['color_bkgr'].flatten(), out['color_spec'].flatten()) \
            * cfg['outer_reg_loss_weight']
    if 'loss_normal' in out and cfg.get('normal_ori', False):
        terms['loss_normal'] = torch.mean(out['loss_normal']).reshape(1)
    if 'loss_mask' in out:
        terms['loss_mask'] = out['loss_mask'].reshape(1) * 0.01
    total = 0
    for v in terms.values():
        total = total + torch.mean(v)
    return total, terms


def get_anneal_val(cfg, step):
    """renderer_zerothick.py:313-317."""
    return 1.0 if cfg['anneal_end'] < 0 else float(min(1.0, step / cfg['anneal_end']))


def near_far_from_sphere(rays_o, rays_d):
    """renderer.py:320-327."""
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return torch.clamp(mid - 1.0, min=1e-3), mid + 1.0


def train_step_std(params, cfg, rays_o, rays_d, rgb_gt, step, rand=None, real=True):
    """One forward of the non-zero-thickness stage-1 renderer (network/renderer.py:465-479, :347-361): real captures
    take near/far from the unit sphere and composite no white background."""
    rays_d = F.normalize(rays_d, dim=-1)
    R = rays_o.shape[0]
    if real:
        nr, fr = near_far_from_sphere(rays_o, rays_d)
    else:
        nr, fr = torch.full((R, 1), 0.8), torch.full((R, 1), 4.5)
    z = sample_ray(params, cfg, rays_o, rays_d, nr, fr, cfg['perturb'], rand)
    out = render_core(params, cfg, rays_o, rays_d, z, step, get_anneal_val(cfg, step), not real, None, std=True)
    out['z_vals'] = z
    out['loss_rgb'] = rgb_loss(out['ray_rgb'], rgb_gt, cfg['rgb_loss'])
    total, terms = assemble_losses(out, cfg, step)
    return total, terms, out


def train_step(params, cfg, rays_o, rays_d, rgb_gt, step, rand=None, occ_perm=None, near=0.8, far=4.5):
    """One stage-1 forward incl. loss (renderer_zerothick.py:447-466, :363-374, :614-634).
    Returns (total loss, loss terms, outputs)."""
    rays_d = F.normalize(rays_d, dim=-1)
    R = rays_o.shape[0]
    nr, fr = torch.full((R, 1), near), torch.full((R, 1), far)
    z = sample_ray(params, cfg, rays_o, rays_d, nr, fr, cfg['perturb'], rand)
    out = render_core(params, cfg, rays_o, rays_d, z, step, get_anneal_val(cfg, step), cfg['is_nerf'], occ_perm)
    out['z_vals'] = z
    out['loss_rgb'] = rgb_loss(out['ray_rgb'], rgb_gt, cfg['rgb_loss'])
    total, terms = assemble_losses(out, cfg, step)
    return total, terms, out


def warmup_cos_lr(step, end_warm=5000, end_iter=300000, lr=5e-4, alpha=0.05):
    """WarmUpCosLR (train/lr_common_manager.py:22-46)."""
    if step < end_warm:
        f = step / end_warm
    else:
        prog = (step - end_warm) / (end_iter - end_warm)
        f = (np.cos(np.pi * prog) + 1.0) * 0.5 * (1 - alpha) + alpha
    return lr * f
