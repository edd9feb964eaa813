"""Synthetic Spherepot-shaped ray source (no dataset exists offline).

100 pinhole cameras on the upper hemisphere of radius 4 looking at the origin, 800x800,
camera_angle_x = 0.6911 (NeRF-synthetic convention, reference dataset/database.py:606-648); rays
are built the way `_construct_nerf_ray_batch` does (renderer_zerothick.py:222-254):
d = R . [(i-cx)/f, -(j-cy)/f, -1],  o = camera centre.  Pixels are drawn by a seeded permutation,
target colours ~ U[0,1).  Everything comes from numpy PCG64 so any machine regenerates the same rays.
"""
import math

import numpy as np


def look_at_pose(cam_pos):
    """Camera-to-world 3x4 (OpenGL/Blender axes: camera looks down -z, +y up)."""
    fwd = -cam_pos / np.linalg.norm(cam_pos)        # viewing direction
    zc = -fwd                                       # camera +z points backwards
    up = np.array([0.0, 0.0, 1.0])
    xc = np.cross(up, zc)
    if np.linalg.norm(xc) < 1e-6:
        xc = np.array([1.0, 0.0, 0.0])
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    pose = np.stack([xc, yc, zc, cam_pos], 1)
    return pose.astype(np.float32)


def make_cameras(n_cam=100, radius=4.0, seed=6033):
    rng = np.random.Generator(np.random.PCG64(seed))
    az = rng.uniform(0.0, 2 * math.pi, n_cam)
    el = rng.uniform(math.radians(10.0), math.radians(80.0), n_cam)
    pos = radius * np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
    return np.stack([look_at_pose(p) for p in pos], 0)  # [n_cam,3,4]


def make_rays(n_rays, seed=6033, n_cam=100, hw=800, camera_angle_x=0.6911, radius=4.0):
    """Returns dict of float32 arrays: rays_o [n,3], rays_d [n,3] (unnormalised, as the reference
    stores them), rgbs [n,3], idxs [n] (camera index)."""
    poses = make_cameras(n_cam, radius, seed)
    focal = 0.5 * hw / math.tan(0.5 * camera_angle_x)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    total = n_cam * hw * hw
    flat = rng.choice(total, size=n_rays, replace=False) if n_rays <= total else rng.integers(0, total, n_rays)
    cam = flat // (hw * hw)
    pix = flat % (hw * hw)
    j = (pix // hw).astype(np.float32)  # row
    i = (pix % hw).astype(np.float32)   # column
    dirs = np.stack([(i - 0.5 * hw) / focal, -(j - 0.5 * hw) / focal, -np.ones_like(i)], -1).astype(np.float32)
    R = poses[cam, :, :3]
    rays_d = np.einsum('nij,nj->ni', R, dirs).astype(np.float32)
    rays_o = poses[cam, :, 3].astype(np.float32)
    rgbs = np.random.Generator(np.random.PCG64(seed + 2)).uniform(0, 1, (n_rays, 3)).astype(np.float32)
    return {'rays_o': rays_o, 'rays_d': rays_d, 'rgbs': rgbs, 'idxs': cam.astype(np.int64)}


def make_jitter(n_rays, n_bg, seed):
    """The two uniform draws `sample_ray` consumes per step (renderer_zerothick.py:585,591)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(0, 1, (n_rays, 1)).astype(np.float32), rng.uniform(0, 1, (n_rays, n_bg)).astype(np.float32)


def make_object_rays(n_rays, seed=6033, cam_radius=4.0, aim_radius=0.7):
    """Rays from the camera shell aimed at random points within `aim_radius` of the origin: a controllable mix of rays that
    hit / graze / miss a small object (used by the stage-2 and mesh-tracing tests, where most image rays would miss)."""
    g = np.random.Generator(np.random.PCG64(seed))
    o = g.standard_normal((n_rays, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * cam_radius
    o[:, 2] = np.abs(o[:, 2])
    t = g.standard_normal((n_rays, 3))
    t = t / np.linalg.norm(t, axis=1, keepdims=True) * (aim_radius * g.uniform(0, 1, (n_rays, 1)) ** (1 / 3))
    d = t - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    rgbs = g.uniform(0, 1, (n_rays, 3))
    return {'rays_o': o.astype(np.float32), 'rays_d': d.astype(np.float32), 'rgbs': rgbs.astype(np.float32)}


def make_image_rays(cam_index, hw=800, seed=6033, n_cam=100, camera_angle_x=0.6911, radius=4.0, downsample=1.0):
    """All rays of ONE synthetic camera, row-major over the (down-sampled) image -- the validation counterpart of
    make_rays (renderer_zerothick.py:222-254 with is_train=False; imgs_info_downsample scales the intrinsics).
    Returns (dict rays_o/rays_d/rgbs [h*w,3], h, w); target colours are U[0,1) like the training pool's."""
    poses = make_cameras(n_cam, radius, seed)
    pose = poses[int(cam_index) % n_cam]
    h = w = int(downsample * hw)
    focal = 0.5 * hw / math.tan(0.5 * camera_angle_x) * (w / hw)
    j, i = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    i, j = i.reshape(-1), j.reshape(-1)
    dirs = np.stack([(i - 0.5 * w) / focal, -(j - 0.5 * h) / focal, -np.ones_like(i)], -1).astype(np.float32)
    rays_d = (dirs @ pose[:, :3].T).astype(np.float32)
    rays_o = np.broadcast_to(pose[:, 3], rays_d.shape).astype(np.float32).copy()
    rgbs = np.random.Generator(np.random.PCG64(seed + 3 + int(cam_index))).uniform(0, 1, (h * w, 3)).astype(np.float32)
    return {'rays_o': rays_o, 'rays_d': rays_d, 'rgbs': rgbs}, h, w
