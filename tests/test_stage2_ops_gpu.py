"""Op-level parity of the stage-2 HIP kernels (csrc/stage2.hip via nu_nerf_amd/stage2_ops.py), forward AND backward, against
the eager torch formulation of the same reference lines (renderer_zerothick.py:1531-1540, :1642-1684, :1835-1870, :1976-1990).
The end-to-end check against the reference's own golden step is tests/test_stage2_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

from helpers import cumprod_excl, sample_pdf_det

pytestmark = pytest.mark.gpu


def _eng(gpu):
    from test_stage2_gpu import build
    net, _ = build(gpu)
    n1, n2 = net.nets()
    n1.eng.pack()
    return net, n1


def test_segment_composite_matches_torch(gpu):
    from nu_nerf_amd import stage2_ops as O
    from nu_nerf_amd import torch_glue as G
    net, n1 = _eng(gpu)
    torch.manual_seed(3)
    for N, S in ((37, 255), (5, 127), (64, 1), (9, 70)):
        alpha = (torch.rand(N, S, device=gpu) ** 4).requires_grad_(True)
        alpha.data[0, : S // 2] = 0.0
        col = torch.rand(N, S, 4, device=gpu)
        col.data[1] *= 0.03                                    # the linear branch of the sRGB curve
        col.requires_grad_(True)
        T = torch.rand(N, 3, device=gpu).requires_grad_(True)
        out, Tend = O.segment_composite(n1.eng, alpha, col, T)
        g1, g2 = torch.randn_like(out), torch.randn_like(Tend)
        ga, gc, gT = torch.autograd.grad((out * g1).sum() + (Tend * g2).sum(), (alpha, col, T))
        lin = G.srgb_to_linear(col[..., :3])
        cpx = cumprod_excl(alpha)
        w = alpha * cpx[:, :-1]
        ref, Tref = (lin * w[..., None]).sum(1) * T, T * cpx[:, -1:]
        ra, rc, rT = torch.autograd.grad((ref * g1).sum() + (Tref * g2).sum(), (alpha, col, T))
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(Tend, Tref, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(ga, ra, rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(gc[..., :3], rc[..., :3], rtol=1e-4, atol=1e-6)
        assert float(gc[..., 3].abs().max()) == 0.0
        torch.testing.assert_close(gT, rT, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("outside", [True, False])
def test_refract_matches_torch(gpu, outside):
    from nu_nerf_amd import stage2_ops as O
    net, n1 = _eng(gpu)
    torch.manual_seed(5 + int(outside))
    M = 300
    d = F.normalize(torch.randn(M, 3, device=gpu), dim=-1).requires_grad_(True)
    nrm = F.normalize(-d.detach() + 0.9 * torch.randn(M, 3, device=gpu), dim=-1).requires_grad_(True)
    ior = torch.rand(M, device=gpu).requires_grad_(True)
    point = torch.randn(M, 3, device=gpu).requires_grad_(True)
    refr, eta, nd, ns = O.refract(n1.eng, d, nrm, ior, point, outside)
    cos_i = torch.sum(nrm * -d, dim=-1, keepdim=True)
    sin2_i = 1 - cos_i * cos_i
    ratio = 1 / (ior[:, None] * 1.0 + 1)
    if not outside:
        ratio = 1 / ratio
    r_ref = ~(ratio * ratio * sin2_i > 0.999)
    assert torch.equal(refr, r_ref.flatten())
    if not outside:
        assert int((~refr).sum()) > 0                         # total internal reflection really occurs leaving the object
    sel = r_ref.flatten()
    sin2_t = sin2_i[sel] * ratio[sel] * ratio[sel]
    t = ratio[sel] * d[sel] + (ratio[sel] * cos_i[sel] - torch.sqrt(1 - sin2_t)) * nrm[sel]
    ns_ref = point[sel] + t * 1e-5
    nd_ref = t / (torch.linalg.norm(t, dim=-1, keepdim=True) + 0.0001)
    torch.testing.assert_close(nd[sel], nd_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ns[sel], ns_ref, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(eta[sel], ratio[sel].flatten(), rtol=1e-6, atol=0)
    assert float(nd.detach()[~sel].abs().max() if (~sel).any() else 0.0) == 0.0
    g1, g2, g3 = torch.randn_like(nd_ref), torch.randn_like(ns_ref), torch.randn(int(sel.sum()), device=gpu)
    got = torch.autograd.grad((nd[sel] * g1).sum() + (ns[sel] * g2).sum() + (eta[sel] * g3).sum(), (d, nrm, ior, point))
    want = torch.autograd.grad((nd_ref * g1).sum() + (ns_ref * g2).sum() + (ratio[sel].flatten() * g3).sum(), (d, nrm, ior, point))
    for a, b, name in zip(got, want, ('d', 'nrm', 'ior', 'point')):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5, msg=lambda m, name=name: f"{name}: {m}")


def test_outer_segments_match_the_eager_formulation(gpu):
    """Two segments with different sample counts, rays that straddle the unit sphere: alpha / colour and the gradients w.r.t.
    start, v, dirs and every NeRF++ parameter against boolean-mask indexing + the network op + torch activations."""
    from nu_nerf_amd import stage2_ops as O
    net, n1 = _eng(gpu)
    torch.manual_seed(11)
    segs = []
    for N, S1 in ((50, 256), (17, 128), (0, 256)):
        start = (F.normalize(torch.randn(N, 3, device=gpu), dim=-1) * (1.6 + torch.rand(N, 1, device=gpu))).requires_grad_(True)
        v = (-start.detach() * (0.5 + torch.rand(N, 1, device=gpu)) + 0.3 * torch.randn(N, 3, device=gpu)).requires_grad_(True)
        z = torch.sort(torch.rand(N, S1, device=gpu), dim=-1)[0]
        dirs = F.normalize(v.detach() + 0.01 * torch.randn(N, 3, device=gpu), dim=-1).requires_grad_(True)
        segs.append((start, v, z, dirs))
    res = O.outer_segments(n1, segs)
    gs = [(torch.randn_like(a), torch.randn_like(c[..., :3])) for a, c in res]
    loss = sum((a * ga).sum() + (c[..., :3] * gc).sum() for (a, c), (ga, gc) in zip(res, gs))
    inputs = [t for s in segs for t in (s[0], s[1], s[3])]
    params = list(n1.nerf_params)
    got = torch.autograd.grad(loss, inputs + params, allow_unused=True)
    # eager formulation (what nu_nerf_amd/stage2.py did before these kernels)
    ref_loss = 0.0
    n_outer = 0
    for (start, v, z, dirs), (a_hip, c_hip), (ga, gc) in zip(segs, res, gs):
        N = start.shape[0]
        if N == 0:
            assert a_hip.numel() == 0
            continue
        cp = start[:, None, :] + v[:, None, :] * z[..., None]
        pfn = cp[:, :-1, :]
        dists = torch.linalg.norm(pfn[:, 1:] - pfn[:, :-1], dim=-1)
        dists = torch.cat([dists, dists[..., -1:]], -1)
        ns = pfn.shape[1]
        outer = ~(torch.norm(pfn, dim=-1) <= 1.0)
        n_outer += int(outer.sum())
        assert 0 < int(outer.sum()) < outer.numel()
        dd = dirs[:, None, :].expand(N, ns, 3)
        a, c = net._density_alpha(n1, pfn[outer], dists[outer], dd[outer])
        alpha = torch.zeros(N, ns, device=gpu).index_put((outer,), a)
        col = torch.zeros(N, ns, 3, device=gpu).index_put((outer,), c)
        torch.testing.assert_close(a_hip, alpha, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(c_hip[..., :3], col, rtol=1e-5, atol=1e-6)
        ref_loss = ref_loss + (alpha * ga).sum() + (col * gc).sum()
    want = torch.autograd.grad(ref_loss, inputs + params, allow_unused=True)
    for k, (a, b) in enumerate(zip(got, want)):
        if b is None:
            assert a is None or a.numel() == 0 or float(a.abs().max()) == 0.0
            continue
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-7, (k, float((a - b).abs().max()), scale)


def test_hit_matches_the_torch_dintersect(gpu):
    """nu_s2_hit_fwd/_bwd against the eager Moeller-Trumbore + normal interpolation (lbvh.dintersect), values and d o, d d."""
    from nu_nerf_amd import stage2_ops as O
    from nu_nerf_amd.lbvh import dintersect
    net, n1 = _eng(gpu)
    scene = net.scene
    torch.manual_seed(21)
    R = 500
    o = (F.normalize(torch.randn(R, 3, device=gpu), dim=-1) * 3.0)
    d = F.normalize(-o + 0.15 * torch.randn(R, 3, device=gpu), dim=-1)
    fi, hitted = scene.intersect(o, d)
    assert int(hitted.sum()) > 100
    oh, dh = o[hitted].clone().requires_grad_(True), d[hitted].clone().requires_grad_(True)
    f = fi[hitted]
    point, nrm, t = O.hit(n1.eng, scene, oh, dh, f)
    tri = scene.faces[f]
    u, v, t_ref, n_ref = dintersect(oh, dh, scene.vertices[tri], scene.normals[tri])
    p_ref = oh + t_ref[:, None] * dh
    torch.testing.assert_close(point, p_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(nrm, n_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(t, t_ref, rtol=1e-5, atol=1e-6)
    g1, g2, g3 = torch.randn_like(point), torch.randn_like(nrm), torch.randn_like(t)
    got = torch.autograd.grad((point * g1).sum() + (nrm * g2).sum() + (t * g3).sum(), (oh, dh))
    want = torch.autograd.grad((p_ref * g1).sum() + (n_ref * g2).sum() + (t_ref * g3).sum(), (oh, dh))
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=5e-4, atol=5e-5)


def test_far_importance_nodes_match_the_eager_formulation(gpu):
    from nu_nerf_amd import stage2_ops as O
    from nu_nerf_amd import torch_glue as G
    net, n1 = _eng(gpu)
    torch.manual_seed(31)
    M = 41
    sm = F.normalize(torch.randn(M, 3, device=gpu), dim=-1) * 4.0
    dm = F.normalize(torch.randn(M, 3, device=gpu), dim=-1)
    z = O.far_importance_nodes(n1.eng, sm, dm)
    with torch.no_grad():
        zo = torch.linspace(0.1, 64.0, 192, device=gpu)
        pts = sm[:, None, :] + dm[:, None, :] * zo[None, :, None]
        zo2 = zo[None, :].expand(M, 192)
        dists = zo2[..., 1:] - zo2[..., :-1]
        dists = torch.cat([dists, dists[..., -1:]], -1)
        alpha, _ = net._density_alpha(n1, pts.reshape(-1, 3), dists.reshape(-1), dm[:, None, :].expand(-1, 192, 3).reshape(-1, 3))
        alpha = alpha.reshape(M, 192)
        w = alpha * cumprod_excl(alpha)[:, :-1]
        newz = sample_pdf_det(zo2.contiguous(), w[:, :-1], 64)
        ref = torch.sort(torch.cat([zo2, newz], -1), dim=-1)[0]
    assert z.shape == ref.shape and bool((z[:, 1:] >= z[:, :-1]).all())
    # the 192 coarse nodes are reproduced exactly; the 64 inverse-CDF samples to the rounding of the running sums
    torch.testing.assert_close(z, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("s2,internal", [(True, False), (True, True), (False, False)])
def test_fused_brdf_mix_matches_the_eager_shading(gpu, s2, internal):
    """The BRDF mix on nu_(s2_)shade_combine_* against the eager torch formulation of AppShadingNetwork(_S2).forward (the test-side
    checker tests/eager_shading.py): colour, (1 - F) T, and the gradients w.r.t. points, normals, view directions, features and every
    parameter."""
    from eager_shading import shade_eager
    net, n1 = _eng(gpu)
    s1c = net.stage1_network.color_network
    torch.manual_seed(41)
    P = 257
    base = [F.normalize(torch.randn(P, 3, device=gpu), dim=-1) * 0.5, F.normalize(torch.randn(P, 3, device=gpu), dim=-1),
            F.normalize(torch.randn(P, 3, device=gpu), dim=-1), 0.3 * torch.randn(P, 256, device=gpu)]
    gcol, grc = torch.randn(P, 3, device=gpu), torch.randn(P, 1, device=gpu)
    params = [p for p in net.stage1_network.color_network.parameters() if p.requires_grad]
    res = {}
    for fused in (True, False):      # the stacks' inputs from the same separate ops on both sides: this test is about the mix
        n1.begin_pass()
        ins = [t.clone().requires_grad_(True) for t in base]
        color, rc = shade_eager(n1, s1c.cfg, s1c.FG_LUT, ins[0], ins[1], ins[2], ins[3], s2=s2, is_internal=internal, fused_combine=fused)
        loss = (color * gcol).sum() + ((rc * grc).sum() if s2 else 0.0)
        grads = torch.autograd.grad(loss, ins + params, allow_unused=True)
        res[fused] = (color.detach(), rc.detach() if s2 else None, grads)
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    if s2:
        torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-5, atol=1e-6)
        if internal:
            assert float(res[True][0].abs().max()) == 0.0
    n_checked = 0
    for a, b in zip(res[True][2], res[False][2]):
        if b is None or float(b.abs().max()) == 0.0:
            assert a is None or float(a.abs().max()) <= 1e-12
            continue
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 3e-4 * scale + 1e-7, (float((a - b).abs().max()), scale)
        n_checked += 1
    assert n_checked >= (4 if not internal else 3)


def test_neus_alpha_matches_torch(gpu):
    from nu_nerf_amd import stage2_ops as O
    net, n1 = _eng(gpu)
    torch.manual_seed(51)
    P = 3000
    sdf = (0.02 * torch.randn(P, device=gpu)).requires_grad_(True)
    nrm = (F.normalize(torch.randn(P, 3, device=gpu), dim=-1) * (0.8 + 0.4 * torch.rand(P, 1, device=gpu))).requires_grad_(True)
    dirs = F.normalize(torch.randn(P, 3, device=gpu), dim=-1).requires_grad_(True)
    dist = (0.002 + 0.01 * torch.rand(P, device=gpu)).requires_grad_(True)
    s = torch.tensor(64.0, device=gpu, requires_grad=True)
    for ca in (0.0, 0.35, 1.0):
        a = O.neus_alpha(n1.eng, sdf, nrm, dirs, dist, s, ca)
        cosv = (dirs * nrm).sum(-1)
        it = -(F.relu(-cosv * 0.5 + 0.5) * (1.0 - ca) + F.relu(-cosv) * ca)
        pc = torch.sigmoid((sdf - it * dist * 0.5) * s)
        nc = torch.sigmoid((sdf + it * dist * 0.5) * s)
        ref = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
        torch.testing.assert_close(a, ref, rtol=1e-5, atol=2e-7)
        assert 0.05 < float((ref > 1e-3).float().mean()) < 1.0
        g = torch.randn(P, device=gpu)
        got = torch.autograd.grad((a * g).sum(), (sdf, nrm, dirs, dist, s))
        want = torch.autograd.grad((ref * g).sum(), (sdf, nrm, dirs, dist, s))
        for x, y, name in zip(got, want, ('sdf', 'nrm', 'dirs', 'dist', 's')):
            scale = float(y.abs().max()) + 1e-12
            assert float((x - y).abs().max()) <= 3e-4 * scale + 1e-8, (name, ca, float((x - y).abs().max()), scale)


def test_vertex_gaussian_curvature_and_its_interpolation_at_hits(gpu):
    """N3 groundwork (network/DiffRender.py:116, :360): per-vertex Gaussian curvature of the mesh -- PyMesh's attribute in the
    reference, angle defect / vertex area here, pinned analytically on spheres (K = 1 / r^2) and on the clip to [-10, 10] -- and
    its barycentric interpolation at the hit points with the gradient w.r.t. the ray."""
    from nu_nerf_amd import stage2_ops as O
    from nu_nerf_amd.lbvh import Scene, icosphere
    net, n1 = _eng(gpu)
    for r, sub in ((0.5, 3), (2.0, 4)):
        V, Fc = icosphere(sub, r)
        sc = Scene(torch.from_numpy(V).to(gpu), torch.from_numpy(Fc).to(gpu))
        k = sc.gaussian_curvatures
        assert k.shape == (V.shape[0], 1)
        # the 12 valence-5 vertices of the icosphere carry the angle-defect estimator's known bias (their barycentric area is
        # too small): everywhere else within 2 %, there within 20 %, and the area-weighted mean is exact (Gauss-Bonnet)
        rel = (k * r ** 2 - 1.0).abs().flatten()
        assert int((rel > 2e-2).sum()) <= 12 and float(rel.max()) < 0.2
        tri = sc.vertices[sc.faces]
        area = 0.5 * torch.linalg.norm(torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1), dim=1)
        v_area = torch.zeros(V.shape[0], device=gpu).index_add_(0, sc.faces.reshape(-1), (area / 3)[:, None].expand(-1, 3).reshape(-1))
        assert abs(float((k.flatten() * v_area).sum()) / (4 * 3.141592653589793) - 1.0) < 1e-3
    V, Fc = icosphere(2, 0.1)                                   # K = 100: clipped like the reference
    sc = Scene(torch.from_numpy(V).to(gpu), torch.from_numpy(Fc).to(gpu))
    assert float(sc.gaussian_curvatures.min()) == 10.0 and float(sc.gaussian_curvatures.max()) == 10.0
    # interpolation at hits: a made-up per-vertex field so that the values vary across a face
    scene = net.scene
    scene.gaussian_curvatures = (scene.vertices[:, :1] * 3.0 + scene.vertices[:, 1:2]).contiguous()
    torch.manual_seed(61)
    R = 400
    o = F.normalize(torch.randn(R, 3, device=gpu), dim=-1) * 3.0
    d = F.normalize(-o + 0.1 * torch.randn(R, 3, device=gpu), dim=-1)
    fi, hitted = scene.intersect(o, d)
    oh, dh = o[hitted].clone().requires_grad_(True), d[hitted].clone().requires_grad_(True)
    f = fi[hitted]
    point, nrm, t, gk = O.hit(n1.eng, scene, oh, dh, f, curvature=True)
    # the field is linear in position and the hit point lies in the face's plane: the interpolation reproduces it there
    want = point[:, :1] * 3.0 + point[:, 1:2]
    torch.testing.assert_close(gk[:, None], want, rtol=1e-4, atol=1e-5)
    g = torch.randn_like(gk)
    got = torch.autograd.grad((gk * g).sum(), (oh, dh), retain_graph=True)
    ref = torch.autograd.grad((want[:, 0] * g).sum(), (oh, dh))
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("inside", [False, True])
def test_shell_refraction_matches_the_oracle_restatement(gpu, inside):
    """nu_s2_shell_* (the non-zero-thickness model's two refractions through the shell, network/renderer.py:1692-2032) against the
    oracle's row-wise torch restatement in float64: flags, every output and the gradients w.r.t. all 12 inputs per ray.  The
    inputs cover both curvature signs, shells thicker than the curvature radius allows (the clamped square roots) and total
    internal reflection at each of the faces."""
    from nu_nerf_amd import stage2_ops as O
    from oracle.stage2_oracle import shell_refraction
    net, n1 = _eng(gpu)
    torch.manual_seed(11 + int(inside))
    M = 600
    d = F.normalize(torch.randn(M, 3), dim=-1)
    n_raw = F.normalize(-d + 0.8 * torch.randn(M, 3), dim=-1) * (0.5 + torch.rand(M, 1))
    if inside:
        n_raw = -n_raw                                             # the mesh normal points out of the object; the op flips it
    point = 0.5 * torch.randn(M, 3)
    ior_raw, th_raw = 2.0 * torch.randn(M), 2.0 * torch.randn(M)
    gk = torch.where(torch.rand(M) < 0.5, 1.0, -1.0) * torch.exp(1.5 * torch.randn(M))     # |g_k| from 1e-2 to 1e2, both signs
    ins = [t.to(gpu).requires_grad_(True) for t in (d, n_raw, point, ior_raw, gk, th_raw)]
    refr, ok, eta, nrm, pend, ns, nd = O.shell_refract(n1.eng, *ins, inside)
    ref_in = [t.double().requires_grad_(True) for t in (d, n_raw, point, ior_raw, gk, th_raw)]
    ref = shell_refraction(*ref_in, inside)
    # rays within float rounding of a threshold may flip; they are excluded (and must be rare)
    same = (refr.cpu() == ref['refracts']) & (ok.cpu() == ref['tir_ok'])
    assert int((~same).sum()) <= 2
    assert 0 < int(ref['refracts'].sum()) < M and int((ref['refracts'] & ~ref['tir_ok']).sum()) > 0
    sel = same & ref['refracts']
    for got, key in ((nrm, 'normal'), (pend, 'end'), (ns, 'next_start'), (nd, 'next_dir')):
        torch.testing.assert_close(got.detach().cpu()[sel].double(), ref[key].detach()[sel], rtol=2e-4, atol=2e-5, msg=lambda m, key=key: f"{key}: {m}")
    torch.testing.assert_close(eta.cpu()[same].double(), ref['eta'].detach()[same], rtol=1e-5, atol=0)
    lost = same & ~ref['refracts']
    assert float(nd.detach().cpu()[lost].abs().max()) == 0.0 and torch.equal(pend.detach().cpu()[lost], point[lost])
    gs = [torch.randn(M, 3) * sel[:, None] for _ in range(4)]
    got = torch.autograd.grad(sum((o * g.to(gpu)).sum() for o, g in zip((nrm, pend, ns, nd), gs)), ins)
    want = torch.autograd.grad(sum((ref[k] * g.double()).sum() for k, g in zip(('normal', 'end', 'next_start', 'next_dir'), gs)), ref_in)
    for a, b, name in zip(got, want, ('d', 'n_raw', 'point', 'ior_raw', 'gk', 'th_raw')):
        a, b = a.cpu().double(), b
        a = a * sel.reshape(-1, *([1] * (a.dim() - 1)))
        scale = float(b.abs().max())
        # the chord length R cos_t - sqrt((R cos_t)^2 -+ 2 R th + th^2) cancels in fp32 for flat surfaces (large R): its
        # derivatives w.r.t. curvature and thickness carry that rounding, the fp64 checker does not
        tol = 3e-3 if name in ('gk', 'th_raw') else 2e-4
        assert float((a - b).abs().max()) <= tol * scale + 1e-6, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("n_freq", [2, 8, 10])
def test_generic_embedding_matches_the_eager_formula(gpu, n_freq):
    """nu_embed_n_* (get_embedder(n_freq, 3), field.py:14-61) forward and backward against sin / cos in float64."""
    from nu_nerf_amd import torch_glue as G
    torch.manual_seed(n_freq)
    x = (1.5 * torch.randn(777, 3, device=gpu)).requires_grad_(True)
    out = G._EmbedNFn.apply(x, n_freq)
    xr = x.detach().double().requires_grad_(True)
    cols = [xr]
    for k in range(n_freq):
        cols += [torch.sin(xr * 2.0 ** k), torch.cos(xr * 2.0 ** k)]
    ref = torch.cat(cols, -1)
    assert out.shape == ref.shape == (777, 3 + 6 * n_freq)
    # the argument 2^k x is exact in fp32; sinf / cosf of an argument up to ~2^10 carry ~1e-6 absolute error
    torch.testing.assert_close(out.detach().double(), ref.detach(), rtol=0, atol=5e-6)
    gcot = torch.randn_like(out)
    (gx,) = torch.autograd.grad((out * gcot).sum(), x)
    (rx,) = torch.autograd.grad((ref * gcot.double()).sum(), xr)
    assert float((gx.double() - rx).abs().max()) <= 2e-5 * float(rx.abs().max())


@pytest.mark.parametrize("which", ["surface_s2_sphere", "inner_specinner_sphere", "inner_plain"])
def test_fused_shade_encode_matches_the_separate_encoding_ops(gpu, which):
    """nu_s2_shade_encode_* (one kernel for every stack's input rows, gradients w.r.t. points / normals / view directions /
    roughness) against the separate encoding ops + torch glue it replaces, through shade(): colour, (1 - F) T and the gradients
    w.r.t. every input and parameter.  Cases: the surface shader with sphere directions, AppShadingNetwork_SpecInner (8 position /
    2 refraction frequencies, sphere directions), the plain inner shader (6 / 6, no sphere directions); points inside and outside
    radius 0.999 (the sphere-point offset branch), normals and view directions of arbitrary length."""
    from nu_nerf_amd import shading_glue as SG
    from eager_shading import shade_eager
    if which == "inner_plain":
        net, _ = _eng(gpu)
        nets, ccfg, lut, s2 = net.nets()[1], net.color_network_inner.cfg, net.color_network_inner.FG_LUT, False
        params = [p for p in net.color_network_inner.parameters() if p.requires_grad]
    else:
        from helpers import golden
        from test_stage2_thick_gpu import build_thick
        net, _ = build_thick(gpu, golden("stage2_thick_step6000_r24.npz"))
        n1, n2 = net.nets()
        if which == "surface_s2_sphere":
            s1c = net.stage1_network.color_network
            nets, ccfg, lut, s2 = n1, s1c.cfg, s1c.FG_LUT, True
            params = [p for p in s1c.parameters() if p.requires_grad]
        else:
            nets, ccfg, lut, s2 = n2, net.color_network_inner.cfg, net.color_network_inner.FG_LUT, False
            params = [p for p in net.color_network_inner.parameters() if p.requires_grad]
    nets.eng.pack()
    torch.manual_seed(77)
    P = 301
    x = F.normalize(torch.randn(P, 3, device=gpu), dim=-1) * (0.2 + 1.0 * torch.rand(P, 1, device=gpu))     # |x| from 0.2 to 1.2
    base = [x, torch.randn(P, 3, device=gpu) * 1.7, torch.randn(P, 3, device=gpu) * 0.6, 0.3 * torch.randn(P, 256, device=gpu)]
    gcol, grc = torch.randn(P, 3, device=gpu), torch.randn(P, 1, device=gpu)
    res = {}
    for fused in (True, False):
        nets.begin_pass()
        ins = [t.clone().requires_grad_(True) for t in base]
        aux = {}
        if fused:       # the product path
            color, rc = SG.shade(nets, ccfg, lut, ins[0], ins[1], ins[2], ins[3], s2=s2, aux=aux)
        else:           # the separate encoding ops + torch glue it replaces (tests/eager_shading.py)
            color, rc = shade_eager(nets, ccfg, lut, ins[0], ins[1], ins[2], ins[3], s2=s2, aux=aux, fused_combine=True)
        loss = (color * gcol).sum() + ((rc * grc).sum() if s2 else 0.0) + (aux['occ_raw'] * grc).sum()
        grads = torch.autograd.grad(loss, ins + params, allow_unused=True)
        res[fused] = (color.detach(), rc.detach() if s2 else None, aux['reflective'].detach(), grads)
    # Integration-level tolerances: the two paths round the reflected direction differently (kernel FMA vs torch ops, 1e-7), and
    # the degree-16 terms of the directional encoding amplify that by ~1e4 in fp32 on either side (their Legendre coefficients
    # reach 1e5); the tight checks of the kernel pair itself are in test_shade_encode_kernels_vs_float64.
    torch.testing.assert_close(res[True][0], res[False][0], rtol=5e-4, atol=2e-4)
    torch.testing.assert_close(res[True][2], res[False][2], rtol=1e-5, atol=1e-6)
    if s2:
        torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-5, atol=1e-6)
    n_checked = 0
    for i, (a, b) in enumerate(zip(res[True][3], res[False][3])):
        if b is None or float(b.abs().max()) == 0.0:
            assert a is None or float(a.abs().max()) <= 1e-12, i
            continue
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-1 * scale + 1e-7, (i, float((a - b).abs().max()), scale)       # wiring errors are O(1)
        assert float((a - b).norm()) <= 2e-2 * float(b.norm()) + 1e-7, (i, float((a - b).norm()), float(b.norm()))
        n_checked += 1
    assert n_checked >= 10


@pytest.mark.parametrize("sphere,pos_freq,rf", [(False, 6, 6), (True, 6, -1), (True, 8, 2)])
def test_shade_encode_kernels_vs_float64(gpu, sphere, pos_freq, rf):
    """nu_s2_shade_encode_fwd / _bwd against the same formulas in float64 torch on the CPU (field.py:636-682, :828-907): every
    stack's input rows, NoV, the reflected direction, and the gradients w.r.t. points, normals, view directions and the roughness
    logit for random cotangents.  Points on both sides of radius 0.999 (the sphere-point offset)."""
    import types
    from nu_nerf_amd import stage2_ops as O
    from nu_nerf_amd import shading_glue as SG
    from nu_nerf_amd import torch_glue as G
    net, n1 = _eng(gpu)
    eng = types.SimpleNamespace(lib=n1.eng.lib, stream=n1.eng.stream, ld_ol=160 if sphere else 96, ld_rl=max(32, -(-2 * (3 + 6 * max(rf, 0)) // 32) * 32))
    torch.manual_seed(3 + pos_freq)
    P = 211
    x = F.normalize(torch.randn(P, 3), dim=-1) * (0.2 + 1.0 * torch.rand(P, 1))
    nrm, view, m_raw = torch.randn(P, 3) * 1.7, torch.randn(P, 3) * 0.6, torch.randn(P, 6)
    ins = [t.to(gpu).requires_grad_(True) for t in (x, nrm, view, m_raw)]
    OL, IL, IW, RL, nov, SD = O.shade_encode(eng, *ins, sphere, pos_freq, rf)
    # ---- float64 reference ----
    xd, nd, vd, md = (t.double().requires_grad_(True) for t in (x, nrm, view, m_raw))
    from oracle.stage1_oracle import _IDE_ML, _IDE_MAT        # (m, l) list and coefficient matrix of the IDE (ref_utils.py:7-79)
    mt, lt, mat = torch.from_numpy(_IDE_ML[:, 0]).long(), torch.from_numpy(_IDE_ML[:, 1]), torch.from_numpy(_IDE_MAT)

    def ide64(d, kinv):
        xx, yy, zz = d[..., 0:1], d[..., 1:2], d[..., 2:3]
        zp = torch.cat([torch.ones_like(zz)] + [zz ** i for i in range(1, 17)], -1)
        re, im = [torch.ones_like(xx)], [torch.zeros_like(xx)]
        for _ in range(16):
            re.append(re[-1] * xx - im[-1] * yy)
            im.append(re[-2] * yy + im[-1] * xx)
        re, im = torch.cat(re, -1)[..., mt], torch.cat(im, -1)[..., mt]
        att = torch.exp(-0.5 * lt.double() * (lt.double() + 1) * kinv)
        poly = zp @ mat.double()
        return torch.cat([re * poly * att, im * poly * att], -1)
    n, v = F.normalize(nd, dim=-1), F.normalize(vd, dim=-1)
    nov_r = torch.sum(n * v, -1, keepdim=True)
    refl = nov_r * n * 2 - v
    rho = torch.sigmoid(md[:, 1:2])
    one, zero = torch.ones_like(rho), torch.zeros_like(rho)
    enc = [ide64(n, one), ide64(refl, rho), ide64(refl, zero)]
    if sphere:
        sn, sr = SG.sphere_point(xd, n), SG.sphere_point(xd, refl)
        enc = [torch.cat([enc[0], ide64(sn, one)], -1), torch.cat([enc[1], ide64(sr, rho)], -1), torch.cat([enc[2], ide64(sr, rho)], -1)]
    OL_r = torch.cat(enc, 0)
    from eager_shading import embed_eager
    pe = embed_eager(xd, pos_freq)
    IL_r = torch.cat([torch.cat([pe, ide64(refl, rho)], -1), torch.cat([pe, ide64(refl, zero)], -1)], 0)
    IW_r = torch.cat([pe, embed_eager(refl, 6)], -1)
    RL_r = torch.cat([embed_eager(xd, rf), embed_eager(v, rf)], -1) if rf >= 0 else None
    # fp32 Horner evaluation of a term's polynomial in z carries eps * sum_k |c_k| of absolute error: the degree-16 terms have
    # coefficient sums up to 1e5, so their columns are compared (and, below, differentiated) with that conditioning in mind
    kappa = mat.double().abs().sum(0)                                       # [36] per term
    kap72 = torch.cat([kappa, kappa])
    well = kap72 < 300.0                                                    # terms up to degree 8
    tol72 = 2e-5 + 4e-7 * kap72

    def col_tol(ncols, ide_starts):
        t = torch.full((ncols,), 2e-5, dtype=torch.float64)
        for s0 in ide_starts:
            t[s0:s0 + 72] = tol72
        return t
    pe_dim = 3 + 6 * pos_freq
    pairs = [(OL, OL_r, col_tol(OL_r.shape[1], [0, 72] if sphere else [0])), (IL, IL_r, col_tol(IL_r.shape[1], [pe_dim])),
             (IW, IW_r, col_tol(IW_r.shape[1], []))] + ([(RL, RL_r, col_tol(RL_r.shape[1], []))] if rf >= 0 else [])
    for got, ref, tol in pairs:
        k = ref.shape[1]
        err = (got.detach().cpu()[:, :k].double() - ref.detach()).abs()
        assert bool((err <= tol[None, :] + 1e-4 * ref.detach().abs()).all()), float((err / tol[None, :]).max())
        assert float(got.detach()[:, k:].abs().max() if got.shape[1] > k else 0.0) == 0.0           # zero padding
    torch.testing.assert_close(nov.detach().cpu().double(), nov_r.detach()[:, 0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(SD.cpu()[:, 8:11].double(), refl.detach(), rtol=1e-5, atol=1e-6)
    # ---- gradients for random cotangents (IWin carries none) ----
    # (cotangents only on the well-conditioned encoding columns: the degree-16 terms run through the same code with other table rows)
    c_ol, c_il = torch.randn(OL_r.shape), torch.randn(IL_r.shape)
    for s0 in ([0, 72] if sphere else [0]):
        c_ol[:, s0:s0 + 72] *= well
    c_il[:, pe_dim:pe_dim + 72] *= well
    cots = [c_ol, c_il, torch.randn(P)] + ([torch.randn(RL_r.shape)] if rf >= 0 else [])
    outs = [OL, IL, nov] + ([RL] if rf >= 0 else [])
    refs = [OL_r, IL_r, nov_r[:, 0]] + ([RL_r] if rf >= 0 else [])
    loss = sum((o[..., :c.shape[-1]] * c.to(gpu)).sum() if o.dim() == 2 else (o * c.to(gpu)).sum() for o, c in zip(outs, cots))
    got = torch.autograd.grad(loss, ins)
    want = torch.autograd.grad(sum((r * c.double()).sum() for r, c in zip(refs, cots)), (xd, nd, vd, md))
    for a, b, name in zip(got, want, ('x', 'normal', 'view', 'm_raw')):
        err, scale = float((a.cpu().double() - b).abs().max()), float(b.abs().max())
        assert err <= 2e-4 * scale + 1e-6, (name, err, scale)
