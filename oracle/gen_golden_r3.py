"""Round-3 fixture generated from the REFERENCE itself (build container only; same shims as oracle/gen_golden.py).

  ray_store.npz   the ray batches the reference builds from an already-loaded image set (network/renderer_zerothick.py:199-254):
                  `_construct_nerf_ray_batch` (NeRF-synthetic convention: one K, rays_o / rays_d in world space, masks) and
                  `_construct_ray_batch` (real captures: per-image K^-1 pixel directions + image indices) on a seeded synthetic
                  `imgs_info` (3 images of 6 x 5 pixels, seeded intrinsics / poses).  Inputs and outputs only.

Usage:  python oracle/gen_golden_r3.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import install_shims, OUT   # noqa: E402


def synthetic_imgs_info(seed=4711, imn=3, h=6, w=5):
    g = np.random.Generator(np.random.PCG64(seed))
    imgs = g.uniform(0, 1, (imn, 3, h, w)).astype(np.float32)
    masks = (g.uniform(0, 1, (imn, 1, h, w)) > 0.4).astype(np.float32)
    Ks = np.zeros((imn, 3, 3), np.float32)
    for i in range(imn):
        f = 7.0 + g.uniform(0, 1)
        Ks[i] = [[f, 0, 0.5 * w + 0.1 * i], [0, f * 1.02, 0.5 * h - 0.05 * i], [0, 0, 1]]
    poses = np.zeros((imn, 3, 4), np.float32)
    for i in range(imn):
        q, _ = np.linalg.qr(g.standard_normal((3, 3)))
        poses[i, :, :3] = q
        poses[i, :, 3] = g.standard_normal(3) * 2.0
    return {'imgs': imgs, 'masks': masks, 'Ks': Ks, 'poses': poses}


def main():
    install_shims()
    torch.manual_seed(0)
    from network.renderer_zerothick import NeROShapeRenderer  # reference
    info = synthetic_imgs_info()
    t = {k: torch.from_numpy(v) for k, v in info.items()}
    net = NeROShapeRenderer({'name': 'golden', 'network': 'shape', 'database_name': 'nerf/spherepot', 'is_nerf': True}, training=False)
    out = {'in_' + k: v for k, v in info.items()}
    nb, poses, rn, h, w = net._construct_nerf_ray_batch(t)
    assert (rn, h, w) == (90, 6, 5)
    for k, v in nb.items():
        out['nerf_' + k] = v.numpy()
    out['nerf_poses'] = poses.numpy()
    ev, _, _, _, _ = net._construct_nerf_ray_batch({k: v for k, v in t.items() if k != 'masks'}, is_train=False)
    assert sorted(ev.keys()) == ['idxs', 'rays_d', 'rays_o', 'rgbs']
    rb, poses2, rn2, h2, w2 = net._construct_ray_batch(t)
    assert (rn2, h2, w2) == (90, 6, 5)
    for k, v in rb.items():
        out['real_' + k] = v.numpy()
    out['real_poses'] = poses2.numpy()
    np.savez_compressed(os.path.join(OUT, 'ray_store.npz'), **out)
    print('wrote ray_store.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
