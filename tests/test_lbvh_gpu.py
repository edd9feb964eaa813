"""GPU parity of the HIP LBVH (build + closest-hit traversal) against the brute-force oracle: hit flags and face
indices BIT-EXACT (integer outputs), hit distance identical float32 bits."""
import os

import numpy as np
import pytest
import torch

from oracle.lbvh_oracle import brute_force_closest_hit, MISS_INDEX

pytestmark = pytest.mark.gpu


def _rays(n, seed, surface_radius=0.5):
    g = np.random.Generator(np.random.PCG64(seed))
    def unit(x):
        return x / np.linalg.norm(x, axis=1, keepdims=True)
    k = n // 4
    # (1) camera-like: from radius 4 towards points near the object (hits and near misses)
    o1 = unit(g.normal(size=(k, 3))) * 4.0
    d1 = unit(unit(g.normal(size=(k, 3))) * g.uniform(0, 0.75, (k, 1)) - o1)
    # (2) from inside the object in random directions (always hit, back faces)
    o2 = unit(g.normal(size=(k, 3))) * g.uniform(0, 0.45, (k, 1))
    d2 = unit(g.normal(size=(k, 3)))
    # (3) restarted on the surface with the reference's 1e-5 offset (renderer_zerothick.py:1682)
    s3 = unit(g.normal(size=(k, 3))) * surface_radius
    d3 = unit(g.normal(size=(k, 3)))
    o3 = s3 + 1e-5 * d3
    # (4) grazing / axis-aligned directions with zero components
    o4 = np.stack([g.uniform(-0.6, 0.6, n - 3 * k), g.uniform(-0.6, 0.6, n - 3 * k), np.full(n - 3 * k, 3.0)], 1)
    d4 = np.tile(np.array([[0.0, 0.0, -1.0]]), (n - 3 * k, 1))
    o = np.concatenate([o1, o2, o3, o4]).astype(np.float32)
    d = np.concatenate([d1, d2, d3, d4]).astype(np.float32)
    return np.concatenate([o, d], 1)


@pytest.mark.parametrize("subdiv", [0, 2, 5])
def test_icosphere_bit_exact_vs_oracle(gpu, subdiv):
    from nu_nerf_amd.lbvh import LBVH, icosphere
    V, F = icosphere(subdiv, 0.5)           # 20, 320, 20480 faces (config 3's stand-in mesh)
    bvh = LBVH(torch.from_numpy(V).to(gpu), torch.from_numpy(F).to(gpu))
    rays = _rays(4096 if subdiv == 5 else 8192, seed=subdiv)
    hit, idx, t = bvh.intersect(torch.from_numpy(rays).to(gpu), return_t=True)
    ohit, oidx, ot = brute_force_closest_hit(V, F, rays)
    assert np.array_equal(hit.cpu().numpy(), ohit)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), ot.view(np.uint32))
    assert 0.3 < ohit.mean() < 0.95            # the ray set exercises hits and misses
    assert (oidx[ohit == 0] == MISS_INDEX).all()


def test_triangle_soup_and_tiny_meshes(gpu):
    from nu_nerf_amd.lbvh import LBVH
    g = np.random.Generator(np.random.PCG64(9))
    for nf in (1, 2, 3, 7, 1000):
        c = g.uniform(-0.5, 0.5, (nf, 1, 3))
        tri = (c + g.normal(size=(nf, 3, 3)) * 0.08).astype(np.float32)       # overlapping, non-watertight
        V = tri.reshape(-1, 3)
        F = np.arange(nf * 3, dtype=np.int32).reshape(nf, 3)
        if nf == 1000:                                                          # duplicate triangles: exact t ties
            V = np.concatenate([V, V[:30]])
            F = np.concatenate([F, np.arange(nf * 3, nf * 3 + 30, dtype=np.int32).reshape(10, 3)])
        rays = _rays(4096, seed=nf)
        bvh = LBVH(torch.from_numpy(V).to(gpu), torch.from_numpy(F).to(gpu))
        hit, idx = bvh.intersect(torch.from_numpy(rays).to(gpu))
        ohit, oidx, _ = brute_force_closest_hit(V, F, rays)
        assert np.array_equal(hit.cpu().numpy(), ohit), nf
        assert np.array_equal(idx.cpu().numpy(), oidx), nf


def test_large_batch_lbvh_equals_device_brute_force(gpu):
    """Full-size property (config 3: 4096 rays x 3 bounces is small; use 200k rays): the traversal never culls a hit."""
    from nu_nerf_amd.lbvh import LBVH, icosphere
    V, F = icosphere(5, 0.5)
    bvh = LBVH(torch.from_numpy(V).to(gpu), torch.from_numpy(F).to(gpu))
    rays = torch.from_numpy(_rays(200000, seed=77)).to(gpu)
    h1, i1 = bvh.intersect(rays)
    h2, i2 = bvh.intersect_brute(rays)
    assert torch.equal(h1, h2) and torch.equal(i1, i2)


def test_scene_dintersect_matches_analytic_sphere(gpu):
    from nu_nerf_amd.lbvh import Scene, icosphere
    V, F = icosphere(5, 0.5)
    sc = Scene(torch.from_numpy(V).to(gpu), torch.from_numpy(F).to(gpu))
    rays = torch.from_numpy(_rays(4096, seed=5)[:1024]).to(gpu)       # camera-like rays
    o, d = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    inter, hitted = sc.Dintersect(o, d)
    assert int(hitted.sum()) > 100
    # the mesh approximates a radius-0.5 sphere: hit points and interpolated normals agree with it
    p = inter['point']
    assert float((p.norm(dim=1) - 0.5).abs().max()) < 2e-3
    cosang = (inter['n'] * torch.nn.functional.normalize(p, dim=1)).sum(1)
    assert float(cosang.min()) > 0.999
    assert float(inter['t'].min()) > 0 and bool(((inter['u'] >= -1e-5) & (inter['v'] >= -1e-5) & (inter['u'] + inter['v'] <= 1 + 1e-5)).all())


def test_mask_renderer_matches_brute_force_and_the_analytic_silhouette(gpu, tmp_path):
    """N4: utils/render_mask_synthetic.py:64-75 on the LBVH.  Every pixel's hit flag equals the exhaustive tracer's; the
    silhouette of an icosphere seen from distance 4 is the disc of angular radius asin(r / 4) (up to the facets)."""
    from nu_nerf_amd.mask_render import render_masks, write_masks, pixel_directions
    from nu_nerf_amd.lbvh import LBVH, icosphere
    from nu_nerf_amd.synthetic import make_cameras
    v, f = icosphere(3, 0.5)
    V, Fc = torch.from_numpy(v).to(gpu), torch.from_numpy(f).to(gpu)
    poses = torch.from_numpy(make_cameras(3, radius=4.0, seed=5)).to(gpu)
    h = w = 96
    focal = 0.5 * w / np.tan(0.5 * 0.6911)
    K = torch.tensor([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1]], dtype=torch.float32)
    bvh = LBVH(V, Fc)
    masks = render_masks(V, Fc, K, poses, h, w, bvh=bvh)
    assert masks.shape == (3, h, w) and masks.dtype == torch.uint8 and set(np.unique(masks.cpu().numpy())) <= {0, 255}
    dirs = pixel_directions(K, h, w, gpu)
    for n in range(3):
        rd = dirs @ poses[n, :3, :3].T
        ray = torch.cat([poses[n, :3, 3].expand_as(rd), rd], 1)
        hit_b, _ = bvh.intersect_brute(ray)
        assert torch.equal(masks[n].reshape(-1) > 0, hit_b > 0)                          # bit-exact vs the O(N F) sweep
        # analytic disc: the camera looks at the origin, so the pixel's angle from the optical axis decides
        cosang = (torch.nn.functional.normalize(rd, dim=-1) @ (-torch.nn.functional.normalize(poses[n, :3, 3], dim=0)))
        ang = torch.acos(cosang.clamp(-1, 1))
        lim = np.arcsin(0.5 / 4.0)
        inside, outside = ang < lim * 0.97, ang > lim * 1.01                              # facets sit slightly inside the sphere
        m = masks[n].reshape(-1) > 0
        assert bool(m[inside].all()) and not bool(m[outside].any()) and 0.01 < float(m.float().mean()) < 0.2
    paths = write_masks(masks, str(tmp_path), ['r_0.png', 'r_1.png', 'r_2.png'])
    assert all(os.path.getsize(p) > 0 for p in paths)
