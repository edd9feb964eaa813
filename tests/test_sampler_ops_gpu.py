"""Op-level GPU parity of the two sampler entries the chain-level test only sees together: `nu_upsample` (one NeuS
up-sampling round, renderer_zerothick.py:525-554 + field.py:468-498) and `nu_merge_sorted` (cat_z_vals, :556-570), each
called alone through the C ABI on the SAME z / sdf the oracle gets -- so the inverse-CDF sensitivity of the chain test
(differences of the previous round feeding the next) does not enter.  fp32; tolerances at each check."""
import numpy as np
import pytest
import torch

from oracle import stage1_oracle as O

pytestmark = pytest.mark.gpu


def _rays_and_sdf(R, sn, seed):
    """Rays towards the unit ball, sorted jittered depths, the SDF of a 0.5-sphere plus a smooth wobble."""
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1) * 2.5
    d = torch.nn.functional.normalize(-o + 0.35 * torch.randn(R, 3, generator=g), dim=-1)
    z = torch.linspace(1.2, 3.8, sn)[None, :] + (torch.rand(R, sn, generator=g) - 0.5) * (2.6 / sn) * 0.9
    p = o[:, None, :] + d[:, None, :] * z[..., None]
    sdf = torch.linalg.norm(p, dim=-1) - 0.5 + 0.03 * torch.sin(7.0 * p[..., 0] + 3.0 * p[..., 1])
    return o.contiguous(), d.contiguous(), z.contiguous(), sdf.contiguous()


@pytest.mark.parametrize("R,sn,n_new,rnd,clip,var", [
    (48, 32, 8, 0, False, 0.3),      # the parity configs' first round, fixed inv_s = 64
    (37, 64, 16, 2, False, 0.3),     # default sampling, third round (inv_s = 256), R not a multiple of the 4 rays per workgroup
    (48, 96, 16, 1, True, 0.3),      # clip_sample_variance: exp(10 * 0.3) = 20.1 < 128 -> the variance wins
    (5, 80, 16, 1, True, 0.6),       # ... exp(6) = 403 > 128 -> the cap wins
    (1, 64, 16, 3, False, 0.3),      # a single ray
])
def test_upsample_alone_vs_oracle(gpu, R, sn, n_new, rnd, clip, var):
    from nu_nerf_amd import _lib as L
    lib = L.load()
    o, d, z, sdf = _rays_and_sdf(R, sn, seed=11 + sn + R)
    cap = 64.0 * 2 ** rnd
    variance = torch.tensor([var])
    inv_s = torch.clamp(torch.exp(variance * 10.0), max=cap)[0] if clip else torch.tensor(cap)
    ref = O.upsample(o, d, z, sdf, n_new, inv_s)
    uv = torch.linspace(0.5 / n_new, 1.0 - 0.5 / n_new, steps=n_new).to(gpu)
    og, dg, zg, sg, vg = (t.to(gpu) for t in (o, d, z, sdf, variance))
    zn = torch.full((R, n_new), float('nan'), device=gpu)
    Xn = torch.full((R * n_new, 3), float('nan'), device=gpu)
    import ctypes
    L.check(lib.nu_upsample(L.ptr(og), L.ptr(dg), L.ptr(zg), L.ptr(sg), R, sn, L.ptr(vg), ctypes.c_float(cap),
                            1 if clip else 0, L.ptr(uv), n_new, L.ptr(zn), L.ptr(Xn), L.stream()), "nu_upsample")
    torch.cuda.synchronize()
    zn_c, Xn_c = zn.cpu(), Xn.cpu()
    assert bool(torch.isfinite(zn_c).all()) and bool(torch.isfinite(Xn_c).all())           # every slot written
    assert bool((zn_c[:, 1:] >= zn_c[:, :-1]).all())                                        # inverse CDF is monotone
    assert bool((zn_c >= z[:, :1]).all()) and bool((zn_c <= z[:, -1:]).all())               # stays inside the bins
    dz = (zn_c - ref).abs() / ref.abs().clamp(min=1.0)
    # same inputs on both sides: what is left is expf / sigmoid / cumprod last bits moved through a CDF whose steps can be 1e-5
    # of the total -- nearly all samples land within 1e-5, a few may move by a fraction of their (narrow) bin
    assert float((dz < 1e-5).float().mean()) >= 0.97, float((dz < 1e-5).float().mean())
    assert float(dz.max()) < 2e-3, float(dz.max())
    # the points handed to the next SDF evaluation are exactly o + d * z_new of the z this launch wrote
    np.testing.assert_allclose(Xn_c.view(R, n_new, 3).numpy(),
                               (o[:, None, :] + d[:, None, :] * zn_c[..., None]).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("R,sn,nn,last", [(48, 32, 8, False), (37, 96, 16, True), (1, 64, 16, False), (130, 40, 8, False)])
def test_merge_sorted_alone_is_exact(gpu, R, sn, nn, last):
    """cat_z_vals without its SDF evaluation: merge two sorted rows, carry the SDF along; a pure permutation -> bit-exact
    (distinct depths, so the order is unique)."""
    from nu_nerf_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5 + R)
    grid = 0.8 + torch.argsort(torch.rand(R, sn + nn, generator=g), dim=-1).float() * (3.0 / (sn + nn))   # distinct per row
    z = torch.sort(grid[:, :sn], dim=-1).values.contiguous()
    zn = torch.sort(grid[:, sn:], dim=-1).values.contiguous()
    s, s_n = torch.randn(R, sn, generator=g), torch.randn(R, nn, generator=g)
    zr, index = torch.sort(torch.cat([z, zn], -1), dim=-1)
    sr = torch.gather(torch.cat([s, s_n], -1), -1, index)
    zg, zng, sg, sng = (t.to(gpu).contiguous() for t in (z, zn, s, s_n))
    zo = torch.full((R, sn + nn), float('nan'), device=gpu)
    so = torch.full((R, sn + nn), float('nan'), device=gpu)
    L.check(lib.nu_merge_sorted(L.ptr(zg), L.ptr(sg), sn, L.ptr(zng), L.ptr(None if last else sng), nn, R,
                                L.ptr(zo), L.ptr(None if last else so), L.stream()), "nu_merge_sorted")
    torch.cuda.synchronize()
    assert torch.equal(zo.cpu(), zr)
    if last:
        assert bool(torch.isnan(so).all())          # the last round carries no SDF: the buffer is not touched
    else:
        assert torch.equal(so.cpu(), sr)
