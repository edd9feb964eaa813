"""autograd wrappers of the encoding kernels (get_embedder, integrated directional encoding) and the few O(points) torch
helpers the stage-2 / validation code shares (sRGB transfer, the split-sum LUT lookup of the validation images, the
weight-normed linear of the parameter-holder modules).  There is no CPU fallback: the encodings raise on tensors that are
not on the GPU, like every other op of the product path.  Formulas cite the reference (paths relative to its repository root).
"""
import torch
import torch.nn.functional as F


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


def _require_cuda(t, what):
    if not t.is_cuda:
        from ._lib import NuNerfLibraryError
        raise NuNerfLibraryError(f"{what} needs a CUDA(HIP) tensor: there is no CPU fallback for the product path")


def embed(x, n_freq):
    """get_embedder(n_freq, 3) (network/field.py:14-61) of points [P,3] on the HIP kernels: 6 frequencies and fewer through the
    SDF path's embedding kernel, up to 10 through the generic pair; input gradient included."""
    _require_cuda(x, "embed")
    if x.dim() != 2 or x.shape[1] != 3 or n_freq > 10:
        raise ValueError("embed: expects [P, 3] points and at most 10 frequencies")
    if x.shape[0] == 0:
        return x.new_zeros(0, 3 + 6 * n_freq)
    return _EmbedFn.apply(x, n_freq) if n_freq <= 6 else _EmbedNFn.apply(x, n_freq)


def ide(xyz, kappa_inv):
    """Integrated directional encoding (utils/ref_utils.py:84-114) of directions [P,3] with roughness [P,1] on the HIP kernels
    (nu_ide / nu_ide_bwd)."""
    _require_cuda(xyz, "ide")
    if xyz.shape[0] == 0:
        return xyz.new_zeros(0, 72)
    return _IdeFn.apply(xyz, kappa_inv.expand(xyz.shape[0], 1))


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
