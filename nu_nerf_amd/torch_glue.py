"""Small differentiable torch helpers used by the FIRST stage-2 implementation (nu_nerf_amd/stage2.py).

Stage 2's ragged per-bounce bookkeeping and its element-wise glue (encodings, BRDF mix, segment composites) run as torch
ops on the GPU in this round; every MLP contraction goes through the HIP library (nu_nerf_amd/nets.py).  Fusing this glue
into HIP kernels with hand-derived input gradients -- as stage 1 already does -- is the next step for this path.
Formulas cite the reference (paths relative to its repository root).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .params import load_fg_lut  # noqa: F401


class _EmbedFn(torch.autograd.Function):
    """get_embedder(n_freq <= 6, 3) on the HIP kernels: forward nu_sdf_embed, backward J_emb^T (nu_embed_jt)."""

    @staticmethod
    def forward(ctx, x, n_freq):
        from . import _lib as L
        lib = L.load()
        x = x.detach().contiguous()
        P = x.shape[0]
        E = torch.empty(P, 64, device=x.device)
        L.check(lib.nu_sdf_embed(L.ptr(x), 3, P, L.ptr(E), None, None, L.stream()), "nu_sdf_embed")
        ctx.E, ctx.nc = E, 3 + 6 * n_freq
        return E[:, :ctx.nc].clone()

    @staticmethod
    def backward(ctx, dout):
        from . import _lib as L
        lib = L.load()
        E = ctx.E
        P = E.shape[0]
        g = torch.zeros(P, 64, device=E.device)
        g[:, :ctx.nc] = dout
        dx = torch.empty(P, 3, device=E.device)
        L.check(lib.nu_embed_jt(L.ptr(E), L.ptr(g), 64, None, 0, P, L.ptr(dx), L.stream()), "nu_embed_jt")
        return dx, None


class _IdeFn(torch.autograd.Function):
    """72-d integrated directional encoding on the HIP kernels (nu_ide / nu_ide_bwd)."""

    @staticmethod
    def forward(ctx, xyz, kappa_inv):
        from . import _lib as L
        lib = L.load()
        d = xyz.detach().contiguous()
        k = kappa_inv.detach().reshape(-1).contiguous()
        P = d.shape[0]
        out = torch.empty(P, 72, device=d.device)
        L.check(lib.nu_ide(L.ptr(d), L.ptr(k), P, L.ptr(out), 72, L.stream()), "nu_ide")
        ctx.d, ctx.k, ctx.kshape = d, k, kappa_inv.shape
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        lib = L.load()
        d, k = ctx.d, ctx.k
        P = d.shape[0]
        g = g.contiguous()
        dd, dk = torch.empty(P, 3, device=d.device), torch.empty(P, device=d.device)
        L.check(lib.nu_ide_bwd(L.ptr(d), L.ptr(k), L.ptr(g), 72, P, L.ptr(dd), L.ptr(dk), L.stream()), "nu_ide_bwd")
        return dd, dk.reshape(ctx.kshape)


class _EmbedNFn(torch.autograd.Function):
    """get_embedder(n_freq <= 10, 3) on the generic HIP kernel pair (nu_embed_n_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, n_freq):
        from . import _lib as L
        x = x.detach().contiguous()
        P, nc = x.shape[0], 3 + 6 * n_freq
        out = torch.empty(P, nc, device=x.device)
        L.check(L.load().nu_embed_n_fwd(L.ptr(x), P, n_freq, L.ptr(out), nc, L.stream()), "nu_embed_n_fwd")
        ctx.x, ctx.n_freq = x, n_freq
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _lib as L
        x = ctx.x
        g = dout.contiguous()
        dx = torch.empty_like(x)
        L.check(L.load().nu_embed_n_bwd(L.ptr(x), L.ptr(g), g.shape[1], x.shape[0], ctx.n_freq, L.ptr(dx), L.stream()), "nu_embed_n_bwd")
        return dx, None


def embed(x, n_freq):
    """network/field.py:14-61.  CUDA tensors [P,3] run on the HIP kernels (6 frequencies and fewer through the SDF path's
    embedding, up to 10 through the generic pair)."""
    if x.is_cuda and x.dim() == 2 and x.shape[1] == 3 and x.shape[0] > 0:
        if n_freq <= 6:
            return _EmbedFn.apply(x, n_freq)
        if n_freq <= 10:
            return _EmbedNFn.apply(x, n_freq)
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


_IDE_CACHE = {}


def _ide_tables(device):
    if device not in _IDE_CACHE:
        ml = [(m, 2 ** i) for i in range(5) for m in range(2 ** i + 1)]
        mat = np.zeros((17, len(ml)))
        for i, (m, l) in enumerate(ml):
            for k in range(l - m + 1):
                binom = np.prod(0.5 * (l + k + m - 1.0) - np.arange(l)) / math.factorial(l)
                leg = (-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) * binom
                mat[k, i] = math.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * math.pi * math.factorial(l + m))) * leg
        _IDE_CACHE[device] = (torch.tensor([m for m, _ in ml], device=device), torch.tensor([float(l) for _, l in ml], device=device),
                              torch.from_numpy(mat.astype(np.float32)).to(device))
    return _IDE_CACHE[device]


def ide(xyz, kappa_inv):
    """utils/ref_utils.py:84-114 in real arithmetic: (x+iy)^m by repeated multiplication, polynomial in z, vMF attenuation.
    CUDA tensors [P,3] / [P,1] run on the HIP kernels."""
    if xyz.is_cuda and xyz.dim() == 2 and xyz.shape[0] > 0:
        return _IdeFn.apply(xyz, kappa_inv.expand(xyz.shape[0], 1))
    m, l, mat = _ide_tables(xyz.device)
    x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
    zp = torch.cat([torch.ones_like(z)] + [z ** i for i in range(1, 17)], -1)
    re, im = [torch.ones_like(x)], [torch.zeros_like(x)]
    for _ in range(16):
        re.append(re[-1] * x - im[-1] * y)
        im.append(re[-2] * y + im[-1] * x)
    re, im = torch.cat(re, -1)[..., m], torch.cat(im, -1)[..., m]
    poly = zp @ mat
    att = torch.exp(-0.5 * l * (l + 1) * kappa_inv)
    return torch.cat([re * poly * att, im * poly * att], -1)


def linear_to_srgb(x):
    eps = torch.finfo(torch.float32).eps
    return torch.where(x <= 0.0031308, 323 / 25 * x, (211 * torch.clamp(x, min=eps) ** (5 / 12) - 11) / 200)


def srgb_to_linear(s):
    eps = torch.finfo(torch.float32).eps
    return torch.where(s <= 0.04045, 25 / 323 * s, torch.clamp((200 * s + 11) / 211, min=eps) ** (12 / 5))


def lut_bilinear_clamp(lut, uv):
    """dr.texture(filter='linear', boundary='clamp') (field.py:719-722): lut [H,W,C], uv [P,2]."""
    H, W, _ = lut.shape
    fx = torch.clamp(uv[:, 0] * W - 0.5, 0.0, W - 1.0)
    fy = torch.clamp(uv[:, 1] * H - 0.5, 0.0, H - 1.0)
    x0, y0 = torch.floor(fx).long(), torch.floor(fy).long()
    x1, y1 = torch.clamp(x0 + 1, max=W - 1), torch.clamp(y0 + 1, max=H - 1)
    tx, ty = (fx - x0.float())[:, None], (fy - y0.float())[:, None]
    top = lut[y0, x0] * (1 - tx) + lut[y0, x1] * tx
    bot = lut[y1, x0] * (1 - tx) + lut[y1, x1] * tx
    return top * (1 - ty) + bot * ty


def sample_pdf_det(bins, weights, n):
    """field.py:468-498 with det=True."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.5 / n, 1.0 - 0.5 / n, steps=n, device=bins.device).expand(list(cdf.shape[:-1]) + [n]).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo, hi = torch.clamp(idx - 1, min=0), torch.clamp(idx, max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def cumprod_excl(alpha):
    ones = torch.ones_like(alpha[..., :1])
    return torch.cumprod(torch.cat([ones, 1. - alpha + 1e-7], -1), -1)


def wn_linear(x, lin):
    """Weight-normed linear on a parameter holder with weight_g / weight_v / bias (tiny IoR network only)."""
    w = lin.weight_v * (lin.weight_g / lin.weight_v.norm(dim=1, keepdim=True))
    return F.linear(x, w, lin.bias)
