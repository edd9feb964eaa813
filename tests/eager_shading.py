"""Test-side checker of the shading kernel pairs: AppShadingNetwork(_S2).forward (network/field.py:636-777, :909-1010) written out
with SEPARATE encoding ops and torch glue over the product's network ops -- what nu_s2_shade_encode_* / nu_(s2_)shade_combine_*
fuse.  Not part of the product package."""
import torch
import torch.nn.functional as F

from nu_nerf_amd import torch_glue as G
from nu_nerf_amd.shading_glue import lights, sphere_point


def embed_eager(x, n_freq):
    """get_embedder(n_freq, d) (network/field.py:14-61) in plain torch ops, any device / dtype: the float64 reference of the tests."""
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def raw_lights(nets, points, n, refl, rough, sphere=False, pos_freq=6):
    """Raw (pre-activation) heads of the light predictors, row-batched as `lights` does: outer_light [3P,3], inner_light [2P,3],
    inner_weight [P,1]."""
    P = points.shape[0]
    one, zero = torch.ones_like(rough), torch.zeros_like(rough)
    enc = torch.cat([G.ide(n, one), G.ide(refl, rough), G.ide(refl, zero)], 0)
    if sphere:
        sn, sr = sphere_point(points, n), sphere_point(points, refl)
        enc_ol = torch.cat([enc, torch.cat([G.ide(sn, one), G.ide(sr, rough), G.ide(sr, rough)], 0)], -1)
    else:
        enc_ol = enc
    ol = nets.predictor('outer_light', enc_ol)
    pe = G.embed(points, pos_freq)
    il = nets.predictor('inner_light', torch.cat([torch.cat([pe, enc[P:2 * P]], -1), torch.cat([pe, enc[2 * P:]], -1)], 0))
    iw = nets.predictor('inner_weight', torch.cat([pe.detach(), G.embed(refl, 6).detach()], -1))
    return ol, il, iw



def shade_eager(nets, scfg, lut, points, normals, view_dirs, feats, s2=False, is_internal=False, aux=None, fused_combine=False):
    """fused_combine=True: separate encodings, BRDF mix on the kernel pair (checks nu_s2_shade_encode_*).
    fused_combine=False: everything eager (checks nu_(s2_)shade_combine_*)."""
    exp_max = scfg['light_exp_max']
    rl_max = scfg.get('refrac_exp_max', exp_max)
    pos_freq = int(scfg.get('light_pos_freq', 6))
    sphere = bool(scfg.get('sphere_direction', False))
    n, v = F.normalize(normals, dim=-1), F.normalize(view_dirs, dim=-1)
    nov = torch.sum(n * v, -1, keepdim=True)
    refl = nov * n * 2 - v
    if fused_combine:
        from nu_nerf_amd import stage2_ops as O
        m_raw = nets.materials(feats, points)
        rough = torch.sigmoid(m_raw[:, 1:2])
        ol, il, iw = raw_lights(nets, points, n, refl, rough, sphere, pos_freq)
        rl = None
        if not s2:
            rf = scfg.get('refrac_freq', 6)
            rl = nets.predictor('refrac_light', torch.cat([G.embed(points, rf), G.embed(v, rf)], -1))
            if rl_max < exp_max:
                rl = torch.clamp(rl, max=rl_max)
        if aux is not None:
            aux.update(occ_raw=iw, reflective=refl)
        color, rc = O.shade_combine(nets.eng, m_raw, ol, il, iw, rl, nov, lut, exp_max, s2=s2, internal=is_internal)
        return color, (rc if s2 else None)
    m = torch.sigmoid(nets.materials(feats, points))
    metallic, rough, albedo, trans = m[:, 0:1], m[:, 1:2], m[:, 2:5], m[:, 5:6]
    diffuse_light, light, light0 = lights(nets, exp_max, points, n, refl, rough, sphere, pos_freq=pos_freq)
    t = torch.clamp(1 - nov, 0.0, 1.0)
    fres = torch.clamp(0.04 + 0.96 * t * t * t * t * t, 0.0, 1.0)
    fg = G.lut_bilinear_clamp(lut[0], torch.cat([torch.clamp(nov, 0.0, 1.0), torch.clamp(rough, 0.0, 1.0)], -1))
    spec_albedo = 0.04 * (1 - metallic) + metallic * albedo
    base = ((1 - metallic) * albedo * diffuse_light + (spec_albedo * fg[:, 0:1] + fg[:, 1:2]) * light) * (1 - trans)
    if s2:
        color = base + (fres * light0) * trans
        if is_internal:
            color = color * 0
        return G.linear_to_srgb(color), (1 - fres) * trans
    rf = scfg.get('refrac_freq', 6)
    refrac = torch.exp(torch.clamp(nets.predictor('refrac_light', torch.cat([G.embed(points, rf), G.embed(v, rf)], -1)),
                                   max=min(exp_max, rl_max)))
    return G.linear_to_srgb(base + (fres * light0 + (1 - fres) * refrac) * trans), None
