"""Differentiable stage-2 ops on the HIP kernels of csrc/stage2.hip (forward + hand-derived backward, WITH input gradients).

  OuterSegments   all |x| > 1 samples of every ray segment through the stage-1 NeRF++ in ONE network pass:
                  per-segment sample bookkeeping (nu_s2_seg_count / _write: node positions, section lengths, inner/outer split,
                  compaction in (ray, sample) order), nu_nerfpp_mlp_fwd + fused activation scattered ray-major, and on the way
                  back d alpha / d dist, the network backward with input gradients and nu_s2_seg_bwd (-> d start, d v, d dirs)
                  (renderer_zerothick.py:1835-1870, :1531-1540)
  SegmentComposite  linear-RGB composite of one segment with the running transmittance (renderer_zerothick.py:1976-1990)
  Refract         Snell refraction / total internal reflection of the rays that hit the mesh (renderer_zerothick.py:1642-1684)
  ShellRefract    the two refractions through the shell of the non-zero-thickness model (renderer.py:1692-2032)

A segment is (start [N,3], v [N,3], z [N,S1]): nodes x_j = start + v * z_j, z without gradient (csrc/stage2.hip).
"""
import ctypes

import torch

from . import _lib as L
from .engine import addr
from .nets import _token_grad

c_p = ctypes.c_void_p


class _OuterSegmentsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, names, nseg, token, *args):
        segs = [tuple(t.detach().contiguous() for t in args[4 * i:4 * i + 4]) for i in range(nseg)]   # start, v, z, dirs
        lib, S_ = eng.lib, eng.stream()
        dev = eng.dev
        dims = [(s[0].shape[0], s[2].shape[1]) for s in segs]                     # (N, S1)
        rows = [n * (s1 - 1) for n, s1 in dims]
        row_base = [sum(rows[:i]) for i in range(nseg)]
        tot_rows = sum(rows)
        alpha_all = eng.zeros(max(tot_rows, 1))
        color_all = eng.zeros(max(tot_rows, 1), 4)
        pos = torch.empty(max(tot_rows, 1), dtype=torch.int32, device=dev)
        totals = torch.zeros(nseg, dtype=torch.int32, device=dev)
        offs = []
        for i, ((st, v, z, d), (n, s1)) in enumerate(zip(segs, dims)):
            cnt, off = torch.empty(max(n, 1), dtype=torch.int32, device=dev), torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            offs.append(off)
            if n > 0:
                L.check(lib.nu_s2_seg_count(c_p(addr(st)), c_p(addr(v)), c_p(addr(z)), n, s1, c_p(addr(cnt)), c_p(addr(off)),
                                            c_p(addr(totals, i)), S_), "nu_s2_seg_count")
        P_seg = [int(x) for x in totals.tolist()]          # the one device -> host read of the pass (buffer sizes)
        P = sum(P_seg)
        pt = eng.empty(max(P, 1), 8)
        idx = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
        pbase = 0
        for i, ((st, v, z, d), (n, s1)) in enumerate(zip(segs, dims)):
            if n > 0:
                L.check(lib.nu_s2_seg_write(c_p(addr(st)), c_p(addr(v)), c_p(addr(z)), c_p(addr(d)), n, s1, c_p(addr(offs[i])), pbase,
                                            row_base[i], c_p(addr(pt)), c_p(addr(idx)), c_p(addr(pos)), S_), "nu_s2_seg_write")
            pbase += P_seg[i]
        b = eng.nerf_forward(pt[:P], idx, P, alpha_all, color_all) if P > 0 else None
        ctx.eng, ctx.names, ctx.segs, ctx.dims, ctx.row_base = eng, names, segs, dims, row_base
        ctx.b, ctx.pt, ctx.idx, ctx.pos, ctx.P = b, pt, idx, pos, P
        ctx.set_materialize_grads(False)
        return alpha_all, color_all

    @staticmethod
    def backward(ctx, dalpha, dcolor):
        eng, segs, dims = ctx.eng, ctx.segs, ctx.dims
        lib, S_ = eng.lib, eng.stream()
        P, pt, idx = ctx.P, ctx.pt, ctx.idx
        nseg = len(segs)
        flat = eng.zeros(eng.n_grad)
        out = []
        if P > 0:
            tot_rows = sum(n * (s1 - 1) for n, s1 in dims)
            da = dalpha.contiguous() if dalpha is not None else eng.zeros(tot_rows)
            dc = dcolor.contiguous() if dcolor is not None else eng.zeros(tot_rows, 4)
            ddist, dx, dd = eng.empty(P), eng.empty(P, 3), eng.empty(P, 3)
            L.check(lib.nu_s2_ddist(c_p(addr(ctx.b['sig'])), c_p(addr(pt)), c_p(addr(idx)), P, c_p(addr(da)), c_p(addr(ddist)), S_),
                    "nu_s2_ddist")
            eng.nerf_backward(ctx.b, pt[:P], idx, da, dc, flat, dx=dx, ddir=dd)
            eng.unpack_grads(flat, eng.nerf_all)
        for i, ((st, v, z, d), (n, s1)) in enumerate(zip(segs, dims)):
            gs, gv, gd = torch.zeros_like(st), torch.zeros_like(v), torch.zeros_like(d)
            if n > 0 and P > 0:
                L.check(lib.nu_s2_seg_bwd(c_p(addr(st)), c_p(addr(v)), c_p(addr(z)), n, s1, ctx.row_base[i], c_p(addr(ctx.pos)),
                                          c_p(addr(dx)), c_p(addr(ddist)), c_p(addr(dd)), c_p(addr(gs)), c_p(addr(gv)), c_p(addr(gd)), S_),
                        "nu_s2_seg_bwd")
            out += [gs, gv, None, gd]
        ctx.b = None
        return (None, None, None, _token_grad(eng, flat, ctx.names)) + tuple(out)


def outer_segments(nets, segs):
    """segs: list of (start [N,3], v [N,3], z [N,S1], dirs [N,3]).  Returns per segment (alpha [N,S], colour [N,S,4] sRGB): the
    NeRF++ density / colour of the samples outside the unit sphere, zero elsewhere."""
    flat_in = []
    for s in segs:
        flat_in += list(s)
    alpha_all, color_all = _OuterSegmentsFn.apply(nets.eng, nets.nerf_names, len(segs), nets.token(), *flat_in)
    out, base = [], 0
    for st, v, z, d in segs:
        n, s = st.shape[0], z.shape[1] - 1
        out.append((alpha_all[base:base + n * s].view(n, s), color_all[base:base + n * s].view(n, s, 4)))
        base += n * s
    return out


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, alpha, color, T):
        alpha, color, T = alpha.detach().contiguous(), color.detach().contiguous(), T.detach().contiguous()
        N, S = alpha.shape
        out, Tout = torch.empty(N, 3, device=alpha.device), torch.empty(N, 3, device=alpha.device)
        if N > 0:
            L.check(eng.lib.nu_s2_composite_fwd(c_p(addr(alpha)), c_p(addr(color)), c_p(addr(T)), N, S, c_p(addr(out)), c_p(addr(Tout)),
                                                eng.stream()), "nu_s2_composite_fwd")
        ctx.eng = eng
        ctx.save_for_backward(alpha, color, T)
        ctx.set_materialize_grads(False)
        return out, Tout

    @staticmethod
    def backward(ctx, dout, dTout):
        alpha, color, T = ctx.saved_tensors
        N, S = alpha.shape
        dalpha, dcolor, dT = torch.empty_like(alpha), torch.empty_like(color), torch.empty_like(T)
        # named locals: a contiguous copy made inside the argument list dies as soon as addr() returns, and the second copy
        # could then be handed the first one's block (engine.render_backward has the same rule)
        dout_c = dout.contiguous() if dout is not None else None
        dTout_c = dTout.contiguous() if dTout is not None else None
        if N > 0:
            L.check(ctx.eng.lib.nu_s2_composite_bwd(c_p(addr(alpha)), c_p(addr(color)), c_p(addr(T)), N, S,
                                                    c_p(addr(dout_c)), c_p(addr(dTout_c)),
                                                    c_p(addr(dalpha)), c_p(addr(dcolor)), c_p(addr(dT)), ctx.eng.stream()),
                    "nu_s2_composite_bwd")
        return None, dalpha, dcolor, dT


def segment_composite(eng, alpha, color4, T):
    """alpha [N,S], colour [N,S,4] (sRGB in channels 0..2), T [N,3] -> (T * sum_j w_j lin(c_j), T * prod_j (1 - alpha_j + 1e-7))."""
    return _CompositeFn.apply(eng, alpha, color4, T)


class _RefractFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, d, nrm, ior, point, outside):
        d, nrm, ior, point = (t.detach().contiguous() for t in (d, nrm, ior, point))
        M = d.shape[0]
        dev = d.device
        flag = torch.empty(M, dtype=torch.uint8, device=dev)             # (the kernels write every row of every output)
        eta, nd, ns = torch.empty(M, device=dev), torch.empty(M, 3, device=dev), torch.empty(M, 3, device=dev)
        L.check(eng.lib.nu_s2_refract_fwd(c_p(addr(d)), c_p(addr(nrm)), c_p(addr(ior)), c_p(addr(point)), M, 1 if outside else 0,
                                          c_p(addr(flag)), c_p(addr(eta)), c_p(addr(nd)), c_p(addr(ns)), eng.stream()), "nu_s2_refract_fwd")
        ctx.eng, ctx.outside = eng, outside
        ctx.save_for_backward(d, nrm, ior)
        ctx.mark_non_differentiable(flag)
        ctx.set_materialize_grads(False)
        return flag, eta, nd, ns

    @staticmethod
    def backward(ctx, _gflag, g_eta, g_nd, g_ns):
        d, nrm, ior = ctx.saved_tensors
        M = d.shape[0]
        dd, dn, dior, dpoint = torch.empty_like(d), torch.empty_like(nrm), torch.empty_like(ior), torch.empty_like(d)
        cg = lambda t: t.contiguous() if t is not None else None
        g_eta, g_nd, g_ns = cg(g_eta), cg(g_nd), cg(g_ns)
        L.check(ctx.eng.lib.nu_s2_refract_bwd(c_p(addr(d)), c_p(addr(nrm)), c_p(addr(ior)), M, 1 if ctx.outside else 0, c_p(addr(g_nd)),
                                              c_p(addr(g_ns)), c_p(addr(g_eta)), c_p(addr(dd)), c_p(addr(dn)), c_p(addr(dior)),
                                              c_p(addr(dpoint)), ctx.eng.stream()), "nu_s2_refract_bwd")
        return None, dd, dn, dior, dpoint, None


def refract(eng, d, nrm, ior, point, outside):
    """Rays that hit the mesh: d [M,3] incoming directions, nrm [M,3] unit normals facing them, ior [M] network output,
    point [M,3].  -> (refracts [M] bool, eta [M], next direction [M,3], next origin [M,3]); rows of totally reflected rays are 0."""
    flag, eta, nd, ns = _RefractFn.apply(eng, d, nrm, ior, point, outside)
    return flag.bool(), eta, nd, ns


class _ShellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, d, nraw, point, ior_raw, gk, th_raw, inside):
        d, nraw, point, ior_raw, gk, th_raw = (t.detach().contiguous() for t in (d, nraw, point, ior_raw, gk, th_raw))
        M, dev = d.shape[0], d.device
        refr, ok = (torch.empty(M, dtype=torch.uint8, device=dev) for _ in range(2))   # (the kernels write every row of every output)
        eta = torch.empty(M, device=dev)
        nrm, pend, ns, nd = (torch.empty(M, 3, device=dev) for _ in range(4))
        L.check(eng.lib.nu_s2_shell_fwd(c_p(addr(d)), c_p(addr(nraw)), c_p(addr(point)), c_p(addr(ior_raw)), c_p(addr(gk)), c_p(addr(th_raw)),
                                        M, 1 if inside else 0, c_p(addr(refr)), c_p(addr(ok)), c_p(addr(eta)), c_p(addr(nrm)),
                                        c_p(addr(pend)), c_p(addr(ns)), c_p(addr(nd)), eng.stream()), "nu_s2_shell_fwd")
        ctx.eng, ctx.inside = eng, inside
        ctx.save_for_backward(d, nraw, point, ior_raw, gk, th_raw)
        ctx.mark_non_differentiable(refr, ok, eta)
        ctx.set_materialize_grads(False)
        return refr, ok, eta, nrm, pend, ns, nd

    @staticmethod
    def backward(ctx, _g0, _g1, _g2, g_nrm, g_pend, g_ns, g_nd):
        d, nraw, point, ior_raw, gk, th_raw = ctx.saved_tensors
        M = d.shape[0]
        g_d, g_n, g_p = (torch.empty_like(d) for _ in range(3))
        g_i, g_k, g_t = (torch.empty_like(gk) for _ in range(3))
        cg = lambda t: t.contiguous() if t is not None else None
        g_nrm, g_pend, g_ns, g_nd = cg(g_nrm), cg(g_pend), cg(g_ns), cg(g_nd)
        L.check(ctx.eng.lib.nu_s2_shell_bwd(c_p(addr(d)), c_p(addr(nraw)), c_p(addr(point)), c_p(addr(ior_raw)), c_p(addr(gk)),
                                            c_p(addr(th_raw)), M, 1 if ctx.inside else 0, c_p(addr(g_nrm)), c_p(addr(g_pend)),
                                            c_p(addr(g_ns)), c_p(addr(g_nd)), c_p(addr(g_d)), c_p(addr(g_n)), c_p(addr(g_p)), c_p(addr(g_i)),
                                            c_p(addr(g_k)), c_p(addr(g_t)), ctx.eng.stream()), "nu_s2_shell_bwd")
        return None, g_d, g_n, g_p, g_i, g_k, g_t, None


def shell_refract(eng, d, n_raw, point, ior_raw, gk, th_raw, inside):
    """Thin-shell refraction of the rays that hit the mesh (non-zero-thickness model, network/renderer.py:1692-2032): d [M,3]
    incoming directions, n_raw [M,3] the interpolated normal as the hit op returns it, point [M,3], ior_raw / th_raw [M] the raw
    IoR / thickness network outputs, gk [M] interpolated Gaussian curvature.  -> (refracts, tir_ok [M] bool, eta [M], unit normal
    facing the ray, end point of the incoming segment, next origin, next direction [M,3])."""
    refr, ok, eta, nrm, pend, ns, nd = _ShellFn.apply(eng, d, n_raw, point, ior_raw, gk, th_raw, bool(inside))
    return refr.bool(), ok.bool(), eta, nrm, pend, ns, nd


class _HitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, o, d, face, verts, vnrm, faces, vcurv):
        o, d = o.detach().contiguous(), d.detach().contiguous()
        M = o.shape[0]
        point, nrm, t = torch.empty_like(o), torch.empty_like(o), torch.empty(M, device=o.device)
        gk = torch.empty(M, device=o.device) if vcurv is not None else torch.zeros(M, device=o.device)
        L.check(eng.lib.nu_s2_hit_fwd(c_p(addr(o)), c_p(addr(d)), c_p(addr(face)), c_p(addr(verts)), c_p(addr(vnrm)), c_p(addr(faces)), M,
                                      c_p(addr(point)), c_p(addr(nrm)), c_p(addr(t)), c_p(addr(vcurv)), c_p(addr(gk)), eng.stream()),
                "nu_s2_hit_fwd")
        ctx.eng, ctx.consts = eng, (face, verts, vnrm, faces, vcurv)
        ctx.save_for_backward(o, d)
        ctx.set_materialize_grads(False)
        return point, nrm, t, gk

    @staticmethod
    def backward(ctx, g_point, g_nrm, g_t, g_gk):
        o, d = ctx.saved_tensors
        face, verts, vnrm, faces, vcurv = ctx.consts
        M = o.shape[0]
        g_o, g_d = torch.empty_like(o), torch.empty_like(d)
        cg = lambda t: t.contiguous() if t is not None else None
        g_point, g_nrm, g_t, g_gk = cg(g_point), cg(g_nrm), cg(g_t), cg(g_gk)
        L.check(ctx.eng.lib.nu_s2_hit_bwd(c_p(addr(o)), c_p(addr(d)), c_p(addr(face)), c_p(addr(verts)), c_p(addr(vnrm)), c_p(addr(faces)), M,
                                          c_p(addr(g_point)), c_p(addr(g_nrm)), c_p(addr(g_t)), c_p(addr(g_o)), c_p(addr(g_d)),
                                          c_p(addr(vcurv)), c_p(addr(g_gk if vcurv is not None else None)), ctx.eng.stream()), "nu_s2_hit_bwd")
        return None, g_o, g_d, None, None, None, None, None


def hit(eng, scene, o, d, face, curvature=False):
    """Differentiable intersection of the rays (o, d) [M,3] with the faces `face` [M] the LBVH found: (point, unit normal, t),
    plus the interpolated per-vertex Gaussian curvature g_k [M] with curvature=True (DiffRender.py:116, the non-zero-thickness
    stage-2 model's input)."""
    vcurv = scene.gaussian_curvatures.reshape(-1).contiguous() if curvature else None
    point, nrm, t, gk = _HitFn.apply(eng, o, d, face.contiguous(), scene.vertices, scene.normals, scene.faces, vcurv)
    return (point, nrm, t, gk) if curvature else (point, nrm, t)


def far_importance_nodes(eng, start, dirs):
    """Rays that miss the mesh (no gradient): 192 coarse nodes z in [0.1, 64] -> NeRF++ density -> 64 inverse-CDF samples merged
    in.  Returns z [M, 256], sorted (renderer_zerothick.py:1786-1812)."""
    M, dev = start.shape[0], start.device
    zo = torch.linspace(0.1, 64.0, 192, device=dev)
    zout = torch.empty(M, 256, device=dev)
    if M == 0:
        return zout
    lib, S_ = eng.lib, eng.stream()
    start, dirs = start.detach().contiguous(), dirs.detach().contiguous()
    P = M * 192
    pt, idx = eng.empty(P, 8), torch.empty(P, dtype=torch.int32, device=dev)
    L.check(lib.nu_s2_far_points(c_p(addr(start)), c_p(addr(dirs)), c_p(addr(zo)), M, 192, c_p(addr(pt)), c_p(addr(idx)), S_),
            "nu_s2_far_points")
    alpha, color = eng.empty(P), eng.empty(P, 4)
    eng.nerf_forward(pt, idx, P, alpha, color)
    L.check(lib.nu_s2_far_resample(c_p(addr(alpha)), c_p(addr(zo)), M, 192, 64, c_p(addr(zout)), S_), "nu_s2_far_resample")
    return zout


class _ShadeEncodeFn(torch.autograd.Function):
    """Inputs of the shading stacks from explicit points / normals / view directions (nu_s2_shade_encode_*), with gradients
    w.r.t. all of them and the roughness logit."""

    @staticmethod
    def forward(ctx, eng, x, nrm, view, m_raw, sphere, pos_freq, refrac_freq, ld_ol, ld_rl):
        x, nrm, view, m = (t.detach().contiguous() for t in (x, nrm, view, m_raw))
        P, dev = x.shape[0], x.device
        OL, IL, IW = torch.empty(3 * P, ld_ol, device=dev), torch.empty(2 * P, 128, device=dev), torch.empty(P, 96, device=dev)
        RL = torch.empty(P, ld_rl, device=dev) if refrac_freq >= 0 else None
        SD = torch.empty(P, 12, device=dev)
        L.check(eng.lib.nu_s2_shade_encode_fwd(c_p(addr(x)), c_p(addr(nrm)), c_p(addr(view)), c_p(addr(m)), m.shape[1], P, 1 if sphere else 0,
                                               pos_freq, ld_ol, refrac_freq, ld_rl, c_p(addr(OL)), c_p(addr(IL)), c_p(addr(IW)), c_p(addr(RL)),
                                               c_p(addr(SD)), eng.stream()), "nu_s2_shade_encode_fwd")
        ctx.eng, ctx.k = eng, (sphere, pos_freq, refrac_freq, ld_ol, ld_rl, m.shape[1])
        ctx.save_for_backward(x, nrm, view, SD)
        ctx.mark_non_differentiable(IW, SD)
        ctx.set_materialize_grads(False)
        if RL is None:
            RL = torch.empty(0, device=dev)
            ctx.mark_non_differentiable(RL)
        return OL, IL, IW, RL, SD[:, 3].clone(), SD

    @staticmethod
    def backward(ctx, dOL, dIL, _dIW, dRL, dnov, _dSD):
        x, nrm, view, SD = ctx.saved_tensors
        sphere, pos_freq, refrac_freq, ld_ol, ld_rl, mcols = ctx.k
        P = x.shape[0]
        dx, dn, dv = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        dm = torch.zeros(P, mcols, device=x.device)
        drho = torch.empty(P, device=x.device)
        cg = lambda t: t.contiguous() if t is not None else None
        dOL, dIL, dnov = cg(dOL), cg(dIL), cg(dnov)
        dRL = cg(dRL) if refrac_freq >= 0 else None
        L.check(ctx.eng.lib.nu_s2_shade_encode_bwd(c_p(addr(x)), c_p(addr(nrm)), c_p(addr(view)), c_p(addr(SD)), P, 1 if sphere else 0,
                                                   pos_freq, ld_ol, refrac_freq, ld_rl, c_p(addr(dOL)), c_p(addr(dIL)), c_p(addr(dRL)),
                                                   c_p(addr(dnov)), c_p(addr(dx)), c_p(addr(dn)), c_p(addr(dv)), c_p(addr(drho)),
                                                   ctx.eng.stream()), "nu_s2_shade_encode_bwd")
        dm[:, 1] = drho
        return None, dx, dn, dv, dm, None, None, None, None, None


def shade_encode(eng, x, nrm, view, m_raw, sphere, pos_freq, refrac_freq):
    """-> (OLin [3P, ld_ol], ILin [2P,128], IWin [P,96] (no gradient), RLin [P, ld_rl] or None, NoV [P], SD [P,12]): the padded input
    rows of the outer_light / inner_light / inner_weight / refrac_light stacks of `eng` (include/nu_nerf.h nu_s2_shade_encode_*)."""
    OL, IL, IW, RL, nov, SD = _ShadeEncodeFn.apply(eng, x, nrm, view, m_raw, bool(sphere), int(pos_freq), int(refrac_freq), eng.ld_ol,
                                                   eng.ld_rl)
    return OL, IL, IW, (RL if refrac_freq >= 0 else None), nov, SD


class _ShadeCombineFn(torch.autograd.Function):
    """The BRDF mix on raw head outputs (nu_shade_combine_* for AppShadingNetwork.forward, nu_s2_shade_combine_* for
    AppShadingNetwork_S2.forward): sigmoids of the material heads, exp(min(., exp_max)) of the light heads, occlusion mix, Schlick
    Fresnel, split-sum LUT (bilinear, clamp), sRGB."""

    @staticmethod
    def forward(ctx, eng, m_raw, ol_raw, il_raw, iw_raw, rl_raw, nov, lut, exp_max, s2, internal):
        P, dev = m_raw.shape[0], m_raw.device
        pad = lambda t, rows, w: torch.cat([t.detach(), torch.zeros(rows, w - t.shape[1], device=dev)], 1).contiguous()
        Mraw, OLo, ILo = pad(m_raw, P, 8), pad(ol_raw, 3 * P, 4), pad(il_raw, 2 * P, 4)
        IWo = iw_raw.detach().reshape(P).contiguous()
        RLo = None if s2 else pad(rl_raw, P, 4)
        SD = torch.zeros(P, 8, device=dev)
        SD[:, 3] = nov.detach().reshape(P)
        idx = torch.arange(P, dtype=torch.int32, device=dev)
        color = torch.empty(P, 4, device=dev)
        rc = torch.empty(P, device=dev)
        lib, S_ = eng.lib, eng.stream()
        if s2:
            L.check(lib.nu_s2_shade_combine_fwd(c_p(addr(Mraw)), 8, c_p(addr(OLo)), c_p(addr(ILo)), c_p(addr(IWo)), c_p(addr(SD)),
                                                c_p(addr(lut)), c_p(addr(idx)), P, ctypes.c_float(exp_max), 1 if internal else 0,
                                                c_p(addr(color)), c_p(addr(rc)), S_), "nu_s2_shade_combine_fwd")
        else:
            L.check(lib.nu_shade_combine_fwd(c_p(addr(Mraw)), 8, c_p(addr(OLo)), c_p(addr(ILo)), c_p(addr(IWo)), c_p(addr(RLo)),
                                             c_p(addr(SD)), c_p(addr(lut)), c_p(addr(idx)), P, ctypes.c_float(exp_max), c_p(addr(color)),
                                             c_p(0), S_), "nu_shade_combine_fwd")
        ctx.eng, ctx.cfg, ctx.bufs = eng, (exp_max, s2, internal), (Mraw, OLo, ILo, IWo, RLo, SD, idx, lut)
        ctx.set_materialize_grads(False)
        return color[:, :3].contiguous(), rc[:, None]

    @staticmethod
    def backward(ctx, dcolor, d_rc):
        eng = ctx.eng
        exp_max, s2, internal = ctx.cfg
        Mraw, OLo, ILo, IWo, RLo, SD, idx, lut = ctx.bufs
        P, dev = Mraw.shape[0], Mraw.device
        dc4 = torch.zeros(P, 4, device=dev)
        if dcolor is not None:
            dc4[:, :3] = dcolor
        dMraw, dOLo, dILo, dIWo, dNoV = torch.zeros_like(Mraw), torch.zeros_like(OLo), torch.zeros_like(ILo), torch.zeros_like(IWo), torch.zeros(P, device=dev)
        lib, S_ = eng.lib, eng.stream()
        if s2:
            g_rc = d_rc.reshape(P).contiguous() if d_rc is not None else None
            L.check(lib.nu_s2_shade_combine_bwd(c_p(addr(Mraw)), 8, c_p(addr(OLo)), c_p(addr(ILo)), c_p(addr(IWo)), c_p(addr(SD)),
                                                c_p(addr(lut)), c_p(addr(idx)), P, ctypes.c_float(exp_max), 1 if internal else 0,
                                                c_p(addr(dc4)), c_p(addr(g_rc)), c_p(addr(dMraw)), c_p(addr(dOLo)), c_p(addr(dILo)),
                                                c_p(addr(dIWo)), c_p(addr(dNoV)), S_), "nu_s2_shade_combine_bwd")
            dRL = None
        else:
            dRLo = torch.zeros_like(RLo)
            L.check(lib.nu_shade_combine_bwd(c_p(addr(Mraw)), 8, c_p(addr(OLo)), c_p(addr(ILo)), c_p(addr(IWo)), c_p(addr(RLo)),
                                             c_p(addr(SD)), c_p(addr(lut)), c_p(addr(idx)), P, ctypes.c_float(exp_max), c_p(addr(dc4)),
                                             c_p(addr(dMraw)), c_p(addr(dOLo)), c_p(addr(dILo)), c_p(addr(dIWo)), c_p(addr(dRLo)),
                                             c_p(addr(dNoV)), S_), "nu_shade_combine_bwd")
            dRL = dRLo[:, :3]
        return None, dMraw[:, :6], dOLo[:, :3], dILo[:, :3], dIWo[:, None], dRL, dNoV[:, None], None, None, None, None


def shade_combine(eng, m_raw, ol_raw, il_raw, iw_raw, rl_raw, nov, lut, exp_max, s2=False, internal=False):
    """-> (sRGB colour [P,3], (1 - F) T [P,1] (meaningful for s2)).  Raw heads: materials [P,6], outer_light [3P,3] (diffuse |
    specular at the point's roughness | mirror), inner_light [2P,3], inner_weight [P,1], refrac_light [P,3] (stage-1 form only)."""
    return _ShadeCombineFn.apply(eng, m_raw, ol_raw, il_raw, iw_raw, rl_raw, nov, lut.contiguous(), float(exp_max), bool(s2), bool(internal))


class _NeusAlphaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, sdf, nrm, dirs, dist, inv_s, ca):
        sdf, nrm, dirs, dist = (t.detach().contiguous() for t in (sdf, nrm, dirs, dist))
        s1 = inv_s.detach().reshape(1).contiguous()
        P = sdf.shape[0]
        alpha = torch.empty(P, device=sdf.device)
        L.check(eng.lib.nu_s2_neus_alpha_fwd(c_p(addr(sdf)), c_p(addr(nrm)), c_p(addr(dirs)), c_p(addr(dist)), c_p(addr(s1)),
                                             ctypes.c_float(ca), P, c_p(addr(alpha)), eng.stream()), "nu_s2_neus_alpha_fwd")
        ctx.eng, ctx.ca, ctx.s_shape = eng, ca, inv_s.shape
        ctx.save_for_backward(sdf, nrm, dirs, dist, s1)
        return alpha

    @staticmethod
    def backward(ctx, g):
        sdf, nrm, dirs, dist, s1 = ctx.saved_tensors
        P = sdf.shape[0]
        g_sdf, g_n, g_d, g_dist, g_s = torch.empty_like(sdf), torch.empty_like(nrm), torch.empty_like(dirs), torch.empty_like(dist), torch.empty_like(sdf)
        g_c = g.contiguous()
        L.check(ctx.eng.lib.nu_s2_neus_alpha_bwd(c_p(addr(sdf)), c_p(addr(nrm)), c_p(addr(dirs)), c_p(addr(dist)), c_p(addr(s1)),
                                                 ctypes.c_float(ctx.ca), P, c_p(addr(g_c)), c_p(addr(g_sdf)), c_p(addr(g_n)),
                                                 c_p(addr(g_d)), c_p(addr(g_dist)), c_p(addr(g_s)), ctx.eng.stream()), "nu_s2_neus_alpha_bwd")
        return None, g_sdf, g_n, g_d, g_dist, g_s.sum().reshape(ctx.s_shape), None


def neus_alpha(eng, sdf, nrm, dirs, dist, inv_s, cos_anneal):
    """NeuS alpha on explicit points [P]: sdf [P], normals [P,3] (the SDF gradient), directions [P,3], section lengths [P], inv_s
    (scalar tensor)."""
    if sdf.shape[0] == 0:
        return sdf.new_zeros(0)
    return _NeusAlphaFn.apply(eng, sdf, nrm, dirs, dist, inv_s, float(cos_anneal))
