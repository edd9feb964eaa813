"""CPU oracle for NU-NeRF's STAGE-2 training step, zero-thickness variant (TEST INFRASTRUCTURE).

Restates `Stage2Renderer.ray_trace` / `render_core` / `train_step` of network/renderer_zerothick.py (:1571-1828,
:1835-2011, :1259-1275), `AppShadingNetwork_S2` (network/field.py:786-1016), `IoRNetwork` (field.py:1046-1065) and the
differentiable re-intersection of network/DiffRender.py:61-125, :539-549 in plain fp32 PyTorch on the CPU.  The mesh
closest-hit query is the brute-force oracle of oracle/lbvh_oracle.py (OptiX cannot run here).  Parameters are a dict with
the reference's Stage2Renderer.state_dict() names.  Pinned by tests/golden/stage2_*.npz, generated from the reference's own
Stage2Renderer under shims with the same brute-force scene (oracle/gen_golden_stage2.py).

Reference quirks restated on purpose (they decide parity): the segment-1 up-sampler evaluates radii / new SDF samples at
o + d*z with z in [0,1] although z is a FRACTION of the segment (renderer_zerothick.py:1748-1758, :1334-1364); the surface
shader's `light_exp_max` is the stage-1 network's (3.0), not AppShadingNetwork_S2's own default.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import stage1_oracle as O
from .lbvh_oracle import brute_force_closest_hit

S1 = 'stage1_network.'


def srgb_to_linear(s):
    """utils/raw_utils.py:20-26."""
    eps = torch.finfo(torch.float32).eps
    return torch.where(s <= 0.04045, 25 / 323 * s, torch.clamp((200 * s + 11) / 211, min=eps) ** (12 / 5))


class BruteScene:
    """Mesh + angle-weighted vertex normals + Dintersect (DiffRender.py:342-360, :539-549, :61-125)."""

    def __init__(self, V, Fc):
        self.Vn, self.Fn = np.asarray(V, np.float32), np.asarray(Fc, np.int32)
        self.vertices = torch.from_numpy(self.Vn)
        self.faces = torch.from_numpy(self.Fn.astype(np.int64))
        tri = self.vertices[self.faces]
        u, v, w = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], tri[:, 2] - tri[:, 1]
        fn = torch.cross(u, v, dim=1)
        fn = fn / fn.norm(dim=1, keepdim=True)
        u, v, w = (x / x.norm(dim=1, keepdim=True) for x in (u, v, w))
        a0 = torch.acos(torch.clamp((u * v).sum(1), -1, 1))
        a1 = torch.acos(torch.clamp((-u * w).sum(1), -1, 1))
        ang = torch.stack([a0, a1, math.pi - a0 - a1], 1)
        vn = torch.zeros_like(self.vertices)
        vn.index_add_(0, self.faces.reshape(-1), (ang[:, :, None] * fn[:, None, :]).reshape(-1, 3))
        self.normals = vn / vn.norm(dim=1, keepdim=True)

    def Dintersect(self, origin, direction):
        rays = torch.cat([origin, direction], 1).detach().numpy().astype(np.float32)
        hit, idx, _ = brute_force_closest_hit(self.Vn, self.Fn, rays)
        hitted = torch.from_numpy(hit > 0)
        f = self.faces[torch.from_numpy(idx.astype(np.int64))[hitted]]
        o, d = origin[hitted], direction[hitted]
        tri, nrm = self.vertices[f], self.normals[f]
        e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
        pvec = torch.cross(d, e2, dim=1)
        inv_det = 1 / (e1 * pvec).sum(1)
        tvec = o - tri[:, 0]
        u = (tvec * pvec).sum(1) * inv_det
        qvec = torch.cross(tvec, e1, dim=1)
        v = (d * qvec).sum(1) * inv_det
        t = (e2 * qvec).sum(1) * inv_det
        n = (1 - u - v)[:, None] * nrm[:, 0] + u[:, None] * nrm[:, 1] + v[:, None] * nrm[:, 2]
        n = n / n.norm(dim=1, keepdim=True)
        return o + t[:, None] * d, n, hitted


def ior_net(params, prefix, x):
    """IoRNetwork: pos-enc(6) -> 39-256-256-256-1, ReLU after the first two linears only, sigmoid (field.py:1046-1065)."""
    h = O.embed(x, 6)
    for idx, relu in ((0, True), (2, True), (4, False), (5, False)):
        h = F.linear(h, O.wn_weight(params, f'{prefix}.module0.{idx}'), params[f'{prefix}.module0.{idx}.bias'])
        if relu:
            h = F.relu(h)
    return torch.sigmoid(h)


def density_alpha(params, pts, dists, dirs, prefix=S1 + 'outer_nerf'):
    """compute_density_alpha with the stage-1 NeRF++ (renderer_zerothick.py:1531-1540)."""
    n = torch.norm(pts, dim=-1, keepdim=True)
    sigma, rgb = O.nerf_forward(params, torch.cat([pts / n, 1.0 / n], -1), dirs, prefix=prefix)
    alpha = 1.0 - torch.exp(-F.softplus(sigma[..., 0]) * dists)
    return alpha, O.linear_to_srgb(torch.exp(torch.clamp(rgb, max=5.0)))


def _cumprod_excl(alpha):
    ones = torch.ones_like(alpha[..., :1])
    return torch.cumprod(torch.cat([ones, 1. - alpha + 1e-7], -1), -1)


def ray_trace(params, cfg, scene, rays_o, rays_d):
    """Up to 3 refraction bounces + per-segment sample placement (renderer_zerothick.py:1571-1828)."""
    next_start, next_dir = rays_o, rays_d
    starts, directions = [rays_o], [rays_d]
    intersections, converges, infinity_bkgr, ior_ratios, gradient_mesh, tirs = [], [], [], [], [], []
    outside = True
    for i in range(3):
        N = next_start.shape[0]
        tir = torch.ones(N, 1, dtype=torch.bool)
        point, n, hit = scene.Dintersect(next_start, next_dir)
        converged = hit.reshape(-1, 1)
        normal = F.normalize(n, dim=-1) if outside else -F.normalize(n, dim=-1)
        infinity_bkgr.append(~converged)
        mask = converged.flatten()
        cos_i = torch.sum(normal * -next_dir[mask], dim=-1, keepdim=True)
        sin2_i = 1 - cos_i * cos_i
        ratio = 1 / (ior_net(params, 'IORs_pred', point.reshape(-1, 3)).reshape(-1, 1) * 1.0 + 1)
        if not outside:
            ratio = 1 / ratio
        refr = ~(ratio * ratio * sin2_i > 0.999)                    # [H,1]: not totally reflected
        converged_out = converged.clone()
        converged_out[mask] = refr
        tir[mask] = refr.detach()
        tirs.append(tir)
        sel = refr.flatten()
        ratio = ratio[sel]
        sin2_t = sin2_i[sel] * ratio * ratio
        nd = ratio * next_dir[converged_out.flatten()] + (ratio * cos_i[sel] - torch.sqrt(1 - sin2_t)) * normal[sel]
        ns = point[sel] + nd * 1e-5
        nd = nd / (torch.linalg.norm(nd, dim=-1, keepdim=True) + 0.0001)
        gm = normal[sel]
        next_dir, next_start = nd, ns
        directions.append(nd)
        starts.append(ns)
        converges.append(converged_out)
        intersections.append(point)
        if torch.all(~converged_out):
            break
        gradient_mesh.append(gm)
        ior_ratios.append(ratio)
        outside = not outside
    for i in range(len(tirs) - 1, 0, -1):
        t = tirs[i - 1]
        m = converges[i - 1].flatten()
        t[m] = t[m] & tirs[i]
    paths = []
    var_in = params['deviation_network_inner.variance']
    for k in range(len(converges)):
        start = starts[k].reshape(-1, 3)
        dk = directions[k]
        end = start + dk * 4.5
        hitk = ~infinity_bkgr[k].flatten()
        if k != 1:
            z = torch.linspace(0, 1, 256)
            sv = start[:, None, :] + (end - start)[:, None, :] * z[None, :, None]
        else:
            zb, zn = torch.linspace(0, 1, 128), torch.linspace(0, 1, 64)
            sv = start[:, None, :] + (end - start)[:, None, :] * zb[None, :, None]
        if hitk.any():
            end = end.clone()
            end[hitk] = intersections[k]
            sh, eh = start[hitk], end[hitk]
            if k != 1:
                sv = sv.clone()
                sv[hitk] = sh[:, None, :] + (eh - sh)[:, None, :] * z[None, :, None]
            else:
                pts = sh[:, None, :] + (eh - sh)[:, None, :] * zn[None, :, None]
                with torch.no_grad():
                    zz = zn[None, :].expand(pts.shape[0], 64)
                    sdf = O.sdf_forward(params, pts.reshape(-1, 3), prefix='sdf_network_inner')[..., 0].reshape(-1, 64)
                    ro, rd = sh.detach(), dk[hitk].detach()
                    for it in range(2):
                        s = torch.clamp(torch.exp(var_in * 10.0), max=64 * 2 ** it)
                        newz = O.upsample(ro, rd, zz, sdf, 32, s)
                        zc = torch.cat([zz, newz], -1)
                        zz2, index = torch.sort(zc, dim=-1)
                        if it == 0:
                            pn = ro[:, None, :] + rd[:, None, :] * newz[..., None]
                            s_new = O.sdf_forward(params, pn.reshape(-1, 3), prefix='sdf_network_inner')[..., 0].reshape(newz.shape)
                            sdf = torch.gather(torch.cat([sdf, s_new], -1), -1, index)
                        zz = zz2
                sv = sv.clone()
                sv[hitk] = sh[:, None, :] + (eh - sh)[:, None, :] * zz[..., None]
        if (~hitk).any() and k != 1:
            miss = ~hitk
            zo = torch.linspace(0.1, 64.0, 192)
            sm, dm = start[miss], dk[miss]
            with torch.no_grad():
                pts = sm[:, None, :] + dm[:, None, :] * zo[None, :, None]
                zo2 = zo[None, :].expand(pts.shape[0], 192)
                dists = zo2[..., 1:] - zo2[..., :-1]
                dists = torch.cat([dists, dists[..., -1:]], -1)
                alpha, _ = density_alpha(params, pts, dists, -dm[:, None, :].expand(-1, 192, 3))
                w = alpha * _cumprod_excl(alpha)[:, :-1]
                newz = O.sample_pdf(zo2, w[:, :-1], 64, det=True)
                zo2 = torch.sort(torch.cat([zo2, newz], -1), dim=-1)[0]
            sv = sv.clone()
            sv[miss] = sm[:, None, :] + dm[:, None, :] * zo2[..., None]
        paths.append(sv)
    return paths, converges, directions, ior_ratios, infinity_bkgr, gradient_mesh, tirs[0]


def shading_s2(params, cfg, points, normals, view_dirs, feats, is_internal):
    """AppShadingNetwork_S2.forward on [H,3] inputs (field.py:909-1010) with the stage-1 predictors; `sphere_direction`
    (field.py:831-836, :900-904): the direction code is followed by the code of the point where the query direction leaves the unit
    sphere -- at the roughness of the query also for the mirror query -- (the 144-d outer_light input)."""
    p = S1 + 'color_network'
    exp_max = cfg['light_exp_max']
    n = F.normalize(normals, dim=-1)
    v = F.normalize(view_dirs, dim=-1)
    nov = torch.sum(n * v, -1, keepdim=True)
    refl = nov * n * 2 - v
    fx = torch.cat([feats, points], -1)
    metallic = O.predictor(params, f'{p}.metallic_predictor', fx, 'sigmoid')
    rough = O.predictor(params, f'{p}.roughness_predictor', fx, 'sigmoid')
    albedo = O.predictor(params, f'{p}.albedo_predictor', fx, 'sigmoid')
    trans = O.predictor(params, f'{p}.transmisstion_weight', fx, 'sigmoid')
    ones = torch.ones_like(rough)

    def outer(enc_dir, rough_for_sph, direction):
        if cfg.get('sphere_direction', False):
            sp = O.offset_points_to_sphere(points)
            sp = F.normalize(sp + direction * O.sphere_exit_distance(sp, direction), dim=-1)
            enc_dir = torch.cat([enc_dir, O.ide(sp, rough_for_sph)], -1)
        return O.predictor(params, f'{p}.outer_light', enc_dir, 'exp', exp_max)

    diffuse_light = outer(O.ide(n, ones), ones, n)
    diffuse_color = (1 - metallic) * albedo * diffuse_light
    spec_albedo = 0.04 * (1 - metallic) + metallic * albedo
    enc_r, enc_r0 = O.ide(refl, rough), O.ide(refl, torch.zeros_like(rough))
    pe = O.embed(points, 6)
    direct = outer(enc_r, rough, refl)
    direct0 = outer(enc_r0, rough, refl)
    indirect = O.predictor(params, f'{p}.inner_light', torch.cat([pe, enc_r], -1), 'exp', exp_max)
    indirect0 = O.predictor(params, f'{p}.inner_light', torch.cat([pe, enc_r0], -1), 'exp', exp_max)
    occ = O.predictor(params, f'{p}.inner_weight', torch.cat([pe.detach(), O.embed(refl, 6).detach()], -1), 'none') * 0.5 + 0.5
    occ_c = torch.clamp(occ, 0.0, 1.0)
    light = indirect * occ_c + direct * (1 - occ_c)
    light0 = indirect0 * occ_c + direct0 * (1 - occ_c)
    t = torch.clamp(1 - nov, 0.0, 1.0)
    fres = torch.clamp(0.04 + 0.96 * t * t * t * t * t, 0.0, 1.0)
    uv = torch.cat([torch.clamp(nov, 0.0, 1.0), torch.clamp(rough, 0.0, 1.0)], -1)
    fg = O.lut_bilinear_clamp(params[f'{p}.FG_LUT'][0], uv)
    spec_color = (spec_albedo * fg[:, 0:1] + fg[:, 1:2]) * light
    color = (diffuse_color + spec_color) * (1 - trans) + (fres * light0) * trans
    if is_internal:
        color = color * 0
    return O.linear_to_srgb(color), (1 - fres) * trans


def render_core(params, cfg, paths, converges, directions, gradient_mesh, ior_ratios, step, cos_anneal):
    """Multi-segment composite in linear RGB with a running transmittance (renderer_zerothick.py:1835-2011, training)."""
    N0 = converges[0].shape[0]
    T = torch.ones(N0, 3)
    colors = []
    out = {'gradient_error': torch.zeros(1), 'std': torch.zeros(1)}
    icfg = dict(cfg)
    for i in range(len(paths)):
        cp, cd, cc = paths[i], directions[i], converges[i].flatten()
        N = cp.shape[0]
        color_now = torch.zeros(N, 3)
        pfn = cp[:, :-1, :]
        dists = torch.linalg.norm(pfn[:, 1:] - pfn[:, :-1], dim=-1)
        dists = torch.cat([dists, dists[..., -1:]], -1)
        ns = pfn.shape[1]
        p_neus = cp[cc][:, -1, :]
        inner = torch.norm(pfn, dim=-1) <= 1.0
        outer = ~inner
        dirs = cd[:, None, :].expand(N, ns, 3)
        alpha, col = torch.zeros(N, ns), torch.zeros(N, ns, 3)
        if outer.any():
            a, c = density_alpha(params, pfn[outer], dists[outer], -dirs[outer])
            alpha = alpha.index_put((outer,), a)
            col = col.index_put((outer,), c)
        if i == 1 and inner.any():
            pin, din, dsin = pfn[inner], dirs[inner], dists[inner]
            y = O.sdf_forward(params, pin, prefix='sdf_network_inner')
            sdf, feats = y[..., 0], y[..., 1:]
            grads = O.sdf_gradient_wrt(params, pin, prefix='sdf_network_inner')
            s = torch.exp(params['deviation_network_inner.variance'] * 10.0).clip(1e-6, 1e6)
            if cfg['freeze_inv_s_step'] is not None and step < cfg['freeze_inv_s_step']:
                s = s.detach()
            cosv = (din * grads).sum(-1)
            it = -(F.relu(-cosv * 0.5 + 0.5) * (1.0 - cos_anneal) + F.relu(-cosv) * cos_anneal)
            pc = torch.sigmoid((sdf - it * dsin * 0.5) * s)
            nc = torch.sigmoid((sdf + it * dsin * 0.5) * s)
            a = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
            c, _ = O.shading_forward(params, icfg, pin, grads, -din, feats, prefix='color_network_inner')
            alpha = alpha.index_put((inner,), a)
            col = col.index_put((inner,), c)
            out['std'] = torch.mean(1 / s)
            out['gradient_error'] = (torch.linalg.norm(grads, dim=-1) - 1.0) ** 2
        have_hit = p_neus.numel() > 0
        if have_hit:
            y = O.sdf_forward(params, p_neus, prefix=S1 + 'sdf_network')
            col_sdf, refr_coeff = shading_s2(params, cfg, p_neus, gradient_mesh[i], -cd[cc], y[..., 1:], i % 2 != 0)
        col = srgb_to_linear(col)
        cp_ = _cumprod_excl(alpha)
        w = alpha * cp_[:, :-1]
        color_now = color_now + (col * w[..., None]).sum(dim=1) * T
        T = T * cp_[:, -1:]
        if have_hit:
            add = torch.zeros_like(color_now).index_put((cc,), srgb_to_linear(col_sdf) * T[cc])
            color_now = color_now + add
            T = T[cc] * refr_coeff
            colors.append(color_now)
        else:
            colors.append(color_now)
            break
    for i in range(len(colors) - 1, 0, -1):
        m = converges[i - 1].flatten()
        colors[i - 1] = colors[i - 1] + torch.zeros_like(colors[i - 1]).index_put((m,), colors[i])
    out['ray_rgb'] = torch.clamp(O.linear_to_srgb(colors[0]), 0.0, 1.0)
    return out


def train_step(params, cfg, scene, rays_o, rays_d, rgb_gt, step):
    """renderer_zerothick.py:1259-1275 + loss assembly ('eikonal', 'std', 'nerf_render'; trainer_zero.py:153-161)."""
    rays_d = F.normalize(rays_d, dim=-1)
    paths, conv, dirs, iors, inf_b, gmesh, tir = ray_trace(params, cfg, scene, rays_o, rays_d)
    out = render_core(params, cfg, paths, conv, dirs, gmesh, iors, step, O.get_anneal_val(cfg, step))
    out['tir_mask'] = tir
    tm = tir.float()
    out['loss_rgb'] = O.rgb_loss(out['ray_rgb'] * tm, rgb_gt * tm, cfg['rgb_loss'])
    terms = {'loss_eikonal': out['gradient_error'] * cfg['eikonal_weight'], 'loss_rgb': out['loss_rgb']}
    total = 0
    for v in terms.values():
        total = total + torch.mean(v)
    out.update(paths=paths, converges=conv, directions=dirs, ior_ratios=iors)
    return total, terms, out


# ------------------------------------------------------------------------------------------------------------------
# Non-zero-thickness model (network/renderer.py, Stage2Renderer.ray_trace): the thin-shell refraction of the rays that hit
# ------------------------------------------------------------------------------------------------------------------
def _unit_eps(x):
    return x / (torch.linalg.norm(x, dim=-1, keepdim=True) + 0.0001)


def shell_refraction(d, n_raw, point, ior_raw, gk, th_raw, inside):
    """renderer.py:1650-2032 for the rays of one bounce that hit the mesh, restated row-wise (the reference's masked writes
    `x[mask] = y[mask]` become `torch.where`): d [M,3] incoming directions, n_raw [M,3] the interpolated (unnormalised) normal,
    point [M,3], ior_raw / th_raw [M] the PRE-sigmoid outputs of IORs_pred / thickness_pred, gk [M] the interpolated Gaussian
    curvature.  Returns dict(refracts, tir_ok [M] bool, eta [M], normal, end, next_start, next_dir [M,3]); rows of rays that do
    not refract hold end = point and zeros for next_*.  Differentiable (torch autograd) -- the checker of nu_s2_shell_*."""
    normal = F.normalize(n_raw, dim=-1)                                           # :1650 / :1659
    if inside:
        normal = -normal
    cos_i = torch.sum(normal * -d, dim=-1, keepdim=True)                          # :1692
    sin2_i = 1 - cos_i * cos_i
    r = 1 / (torch.sigmoid(ior_raw)[:, None] * 1.0 + 0.6)                         # :1727-1728
    inner = torch.full_like(r, 1 / 1.0001)                                        # :1734 (the inner-IoR network is multiplied by 0)
    ro = inner / r                                                                # :1739
    th = torch.sigmoid(th_raw)[:, None] * 0.01                                    # :1741-1742
    if inside:                                                                    # :1748-1751
        r, ro = 1 / ro, 1 / r
    refr = ~(r * r * sin2_i > 0.999)                                              # :1762
    tir = refr.clone()                                                            # :1767
    sin2_t = sin2_i * r * r                                                       # :1774
    g = gk[:, None]
    R = torch.nan_to_num(1 / torch.sqrt(torch.clamp(torch.abs(g), min=0.000001)), 0.1)   # :1792-1793
    cos_t = torch.sqrt(torch.clamp(1 - sin2_t, min=0.0001))                       # :1800 / :1865
    positive = (g <= 0) if inside else (g >= 0)                                   # :1802 / :1866

    def chord(c):                                                                 # :1816-1819 / :1833-1836 and their twins
        d2 = torch.where(positive, c * c - 2 * R * th + th * th, c * c + 2 * R * th + th * th)
        return torch.abs(c - torch.sqrt(torch.clamp(d2, min=0.0001)))
    if not inside:
        d_in = _unit_eps(r * d + (r * cos_i - torch.sqrt(torch.clamp(1 - sin2_t, min=0.0001))) * normal)   # :1811-1812
        pm, nm = point, normal
    else:
        length = chord(R * cos_i)                                                 # :1884-1887 / :1917-1920
        center = torch.where(positive, point - normal * R, point + normal * R)    # :1908 / :1923
        pm = point - length * d                                                   # :1910 / :1925
        nm = _unit_eps(torch.where(positive, pm - center, center - pm))           # :1912 / :1926, :1931
        cos_im = torch.sum(nm * -d, dim=-1, keepdim=True)                         # :1932
        x = (1 - cos_im * cos_im) * r * r
        tir = tir & ~(x > 0.999)                                                  # :1936
        d_in = _unit_eps(r * d + (r * cos_im - torch.sqrt(torch.clamp(1 - torch.clamp(x, max=0.999), min=0.0001))) * nm)   # :1934-1939
    length = chord(R * cos_t)                                                     # :1816-1819 / :1943-1946
    center = torch.where(positive, pm - nm * R, pm + nm * R)                      # :1822 / :1839 / :1949 / :1979
    next_start = pm + d_in * (length + 0.001)                                     # :1824 / :1951
    n_after = _unit_eps(torch.where(positive, next_start - center, center - next_start))   # :1826-1827 / :1842-1843
    cos_i2 = torch.sum(n_after * -d_in, dim=-1, keepdim=True)                     # :1853 / :1995
    x2 = (1 - cos_i2 * cos_i2) * ro * ro
    tir = tir & ~(x2 > 0.999)                                                     # :1855 / :2005
    next_dir = _unit_eps(ro * d_in + (ro * cos_i2 - torch.sqrt(torch.clamp(1 - torch.clamp(x2, max=0.999), min=0.0001))) * n_after)
    z = torch.zeros_like(point)
    return dict(refracts=refr[:, 0], tir_ok=tir[:, 0], eta=r[:, 0], normal=normal, end=torch.where(refr, pm, point),
                next_start=torch.where(refr, next_start, z), next_dir=torch.where(refr, next_dir, z))
