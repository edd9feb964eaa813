"""Surface shading assembled from the HIP network ops (nets.Stage1Nets): ONE path for rendering -- stack inputs
(nu_s2_shade_encode_*), the predictor / material stacks, the BRDF mix (nu_(s2_)shade_combine_*) -- and the validation
images of `inter_results=True` (the intermediate terms of the same formulas as O(points) torch glue over the same network
ops; only test_step renders them).

Used where the fused stage-1 shading kernels do not apply: stage 2 (input gradients needed, ragged per-bounce point
sets) and validation rendering (intermediate images wanted).  Every MLP GEMM runs in libnunerf.so.  The eager formulation that
checks the kernel pairs in the parity tests lives with the tests (tests/eager_shading.py), not here.
Reference: AppShadingNetwork.forward / predict_specular_lights / predict_diffuse_lights (network/field.py:636-777),
AppShadingNetwork_S2.forward (field.py:909-1010)."""
import torch
import torch.nn.functional as F

from . import torch_glue as G


def offset_points_to_sphere(points):
    """field.py:447-455"""
    nrm = torch.norm(points, dim=-1, keepdim=True)
    return torch.where(nrm > 0.999, points / nrm * 0.999, points)


def get_sphere_intersection(pts, dirs):
    """field.py:458-464"""
    dtx = torch.sum(pts * dirs, dim=-1, keepdim=True)
    xtx = torch.sum(pts ** 2, dim=-1, keepdim=True)
    return -dtx + torch.sqrt(dtx ** 2 - xtx + 1 + 1e-6)


def sphere_point(points, dirs):
    """Where the ray (point, dir) leaves the unit sphere (field.py:641-643): the extra outer_light input of
    sphere_direction=True."""
    sp = offset_points_to_sphere(points)
    return F.normalize(sp + dirs * get_sphere_intersection(sp, dirs), dim=-1)


def lights(nets, exp_max, points, n, refl, rough, sphere=False, detail=False, pos_freq=6):
    """The three outer_light and two inner_light queries + the occlusion weight (field.py:636-682), row-batched."""
    P = points.shape[0]
    one, zero = torch.ones_like(rough), torch.zeros_like(rough)
    enc = torch.cat([G.ide(n, one), G.ide(refl, rough), G.ide(refl, zero)], 0)
    if sphere:
        sn, sr = sphere_point(points, n), sphere_point(points, refl)
        # field.py:643-646: the sphere point of BOTH specular queries is encoded with the point's roughness
        enc_ol = torch.cat([enc, torch.cat([G.ide(sn, one), G.ide(sr, rough), G.ide(sr, rough)], 0)], -1)
    else:
        enc_ol = enc
    lo = torch.exp(torch.clamp(nets.predictor('outer_light', enc_ol), max=exp_max))
    pe = G.embed(points, pos_freq)
    li = torch.exp(torch.clamp(nets.predictor('inner_light', torch.cat([torch.cat([pe, enc[P:2 * P]], -1),
                                                                          torch.cat([pe, enc[2 * P:]], -1)], 0)), max=exp_max))
    occ = nets.predictor('inner_weight', torch.cat([pe.detach(), G.embed(refl, 6).detach()], -1)) * 0.5 + 0.5
    occ_c = torch.clamp(occ, 0.0, 1.0)
    light = li[:P] * occ_c + lo[P:2 * P] * (1 - occ_c)
    light0 = li[P:] * occ_c + lo[2 * P:] * (1 - occ_c)
    if detail:
        return lo[:P], light, light0, occ, li[:P] * occ_c
    return lo[:P], light, light0


def shade(nets, scfg, lut, points, normals, view_dirs, feats, s2=False, is_internal=False, inter_results=False, aux=None):
    """AppShadingNetwork.forward (field.py:684-777) or, with s2=True, AppShadingNetwork_S2.forward (field.py:909-1010): materials ->
    ONE kernel for every stack's padded input rows (n^, v^, NoV, r, IDE / position / refraction codes, sphere points; gradients to
    points, normals, view directions and the roughness logit) -> the four stacks -> the BRDF mix as one kernel pair on the raw
    heads.  inter_results=True (validation images of test_step) returns the intermediate terms as well."""
    if not points.is_cuda:
        from ._lib import NuNerfLibraryError
        raise NuNerfLibraryError("shade needs CUDA(HIP) tensors: there is no CPU fallback for the product path")
    if inter_results:
        return _shade_with_images(nets, scfg, lut, points, normals, view_dirs, feats, s2, is_internal)
    exp_max = scfg['light_exp_max']
    rl_max = scfg.get('refrac_exp_max', exp_max)
    pos_freq = int(scfg.get('light_pos_freq', 6))
    sphere = bool(scfg.get('sphere_direction', False))
    if points.shape[0] == 0:
        if aux is not None:
            aux.update(occ_raw=points.new_zeros(0, 1), reflective=points.new_zeros(0, 3))
        return points.new_zeros(0, 3), (points.new_zeros(0, 1) if s2 else None)
    from . import stage2_ops as O
    m_raw = nets.materials(feats, points)
    rf = -1 if s2 else int(scfg.get('refrac_freq', 6))
    OL, IL, IW, RL, nov1, SD = O.shade_encode(nets.eng, points, normals, view_dirs, m_raw, sphere, pos_freq, rf)
    # the stacks of one shading call are independent of each other: ONE op, level j of all of them in one launch (nets.StacksFn)
    rl = None
    if s2:
        ol, il, iw = nets.predictors(('outer_light', 'inner_light', 'inner_weight'), (OL, IL, IW))
    else:
        ol, il, iw, rl = nets.predictors(('outer_light', 'inner_light', 'inner_weight', 'refrac_light'), (OL, IL, IW, RL))
        if rl_max < exp_max:     # AppShadingNetwork_SpecInner's refrac_light caps at exp(-0.2) (field.py:1373): clamp the raw head,
            rl = torch.clamp(rl, max=rl_max)     # the kernel's own min(., exp_max) is then the identity
    if aux is not None:          # what the occlusion probe of the caller needs (occ_info of field.py:1533-1537)
        aux.update(occ_raw=iw, reflective=SD[:, 8:11])
    color, rc = O.shade_combine(nets.eng, m_raw, ol, il, iw, rl, nov1[:, None], lut, exp_max, s2=s2, internal=is_internal)
    return color, (rc if s2 else None)


def _shade_with_images(nets, scfg, lut, points, normals, view_dirs, feats, s2, is_internal):
    """The validation images (field.py:747-770, :984-1003): the shading formulas term by term on the outputs of the network ops."""
    exp_max = scfg['light_exp_max']
    rl_max = scfg.get('refrac_exp_max', exp_max)
    pos_freq = int(scfg.get('light_pos_freq', 6))
    sphere = bool(scfg.get('sphere_direction', False))
    n, v = F.normalize(normals, dim=-1), F.normalize(view_dirs, dim=-1)
    nov = torch.sum(n * v, -1, keepdim=True)
    refl = nov * n * 2 - v
    m = torch.sigmoid(nets.materials(feats, points))
    metallic, rough, albedo, trans = m[:, 0:1], m[:, 1:2], m[:, 2:5], m[:, 5:6]
    diffuse_light, light, light0, occ, indirect = lights(nets, exp_max, points, n, refl, rough, sphere, detail=True, pos_freq=pos_freq)
    t = torch.clamp(1 - nov, 0.0, 1.0)
    fres = torch.clamp(0.04 + 0.96 * t * t * t * t * t, 0.0, 1.0)
    fg = G.lut_bilinear_clamp(lut[0], torch.cat([torch.clamp(nov, 0.0, 1.0), torch.clamp(rough, 0.0, 1.0)], -1))
    diffuse_albedo = (1 - metallic) * albedo
    spec_albedo = 0.04 * (1 - metallic) + metallic * albedo
    spec_ref = spec_albedo * fg[:, 0:1] + fg[:, 1:2]
    diffuse_color = diffuse_albedo * diffuse_light
    spec_color = spec_ref * light
    base = (diffuse_color + spec_color) * (1 - trans)
    if s2:
        color = base + (fres * light0) * trans
        if is_internal:
            color = color * 0
        c01 = lambda x: torch.clamp(x, 0.0, 1.0)       # the validation images of the first surface (field.py:984-1003)
        inter = {'specular_ref': c01(spec_ref), 'specular_light': c01(G.linear_to_srgb(light0)),
                 'specular_color': c01(G.linear_to_srgb(spec_color) * (1 - trans) + fres * light0 * trans)}
        return G.linear_to_srgb(color), (1 - fres) * trans, inter
    rf = scfg.get('refrac_freq', 6)
    refrac = torch.exp(torch.clamp(nets.predictor('refrac_light', torch.cat([G.embed(points, rf), G.embed(v, rf)], -1)),
                                   max=min(exp_max, rl_max)))
    color = G.linear_to_srgb(base + (fres * light0 + (1 - fres) * refrac) * trans)
    c01 = lambda x: torch.clamp(x, 0.0, 1.0)
    spec_color_srgb = G.linear_to_srgb(spec_color)
    inter = {   # field.py:747-770 (specular_color mixes the sRGB specular term with a linear one, as the reference does)
        'specular_albedo': spec_albedo,
        'specular_ref': c01(spec_ref),
        'specular_light': c01(G.linear_to_srgb(light0)),
        'specular_color': c01(spec_color_srgb * (1 - trans) + fres * light0 * trans),
        'diffuse_albedo': diffuse_albedo,
        'diffuse_light': c01(G.linear_to_srgb(diffuse_light)),
        'diffuse_color': c01(G.linear_to_srgb(diffuse_color)),
        'metallic': metallic, 'transmission_weight': trans, 'roughness': rough,
        'occ_prob': c01(occ), 'indirect_light': indirect,
        'refraction_light': c01(G.linear_to_srgb((1 - fres) * refrac * trans)),
        'reflection_weight': fres,
    }
    occ_info = {'reflective': refl, 'occ_prob': occ, 'transmission_weight': trans, 'metallic': metallic}
    return color, occ_info, inter
