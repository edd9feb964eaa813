"""Object-mask renderer on the HIP LBVH: drop-in for `utils/render_mask_synthetic.py:64-75`.

The reference builds, per training image, the pinhole rays of every pixel (`dirs = [(i - cx)/fx, -(j - cy)/fy, -1]` rotated by
the pose's 3x3, origin = the pose's translation; :52-66), traces them against the stage-1 mesh through OptiX
(`Scene.Dintersect`, :71) and writes `converged * 255` as an image (:72-74).  Here the tracing is `nu_lbvh_trace` (closest hit
with the semantics of cuda/triangle.cu:48-99); only the hit flag is needed, so no re-intersection runs.  Image files are
written without OpenCV (absent offline): binary PGM, or PNG/JPEG when Pillow is importable.
"""
import os

import numpy as np
import torch

from .lbvh import LBVH


def pixel_directions(K, h, w, device):
    """Camera-space directions of every pixel, row-major [h*w, 3] (render_mask_synthetic.py:52-58: meshgrid over (w, h)
    transposed = x along columns, y along rows)."""
    K = torch.as_tensor(K, dtype=torch.float32, device=device)
    j, i = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=device),
                          torch.arange(w, dtype=torch.float32, device=device), indexing='ij')
    return torch.stack([(i - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -torch.ones_like(i)], -1).reshape(-1, 3)


@torch.no_grad()
def render_masks(vertices, faces, Ks, poses, h, w, bvh=None):
    """uint8 masks [n_img, h, w] (255 = the pixel's ray hits the mesh).  poses [n,3,4] camera-to-world (columns = axes,
    last column = centre), Ks [n,3,3] or one [3,3] shared by all images (the reference uses Ks[0] for every image, :57)."""
    dev = vertices.device
    bvh = bvh or LBVH(vertices, faces)
    poses = torch.as_tensor(poses, dtype=torch.float32, device=dev)
    Ks = torch.as_tensor(Ks, dtype=torch.float32, device=dev)
    K0 = Ks if Ks.dim() == 2 else Ks[0]
    dirs = pixel_directions(K0, h, w, dev)
    out = torch.empty(poses.shape[0], h, w, dtype=torch.uint8, device=dev)
    for n in range(poses.shape[0]):
        rays_d = dirs @ poses[n, :3, :3].T                       # sum(dirs[..., None, :] * R, -1)   (:66)
        rays_o = poses[n, :3, 3].expand_as(rays_d)
        hit, _ = bvh.intersect(torch.cat([rays_o, rays_d], 1))
        out[n] = (hit > 0).reshape(h, w).to(torch.uint8) * 255
    return out


def write_masks(masks, out_dir, names):
    """One image file per mask under out_dir (the reference writes <name>.jpg with cv2, :74)."""
    os.makedirs(out_dir, exist_ok=True)
    arr = masks.detach().cpu().numpy()
    try:
        from PIL import Image
    except ImportError:
        Image = None
    paths = []
    for m, name in zip(arr, names):
        stem = os.path.splitext(name)[0]
        if Image is not None:
            path = os.path.join(out_dir, stem + '.png')
            Image.fromarray(np.repeat(m[:, :, None], 3, 2)).save(path)
        else:
            path = os.path.join(out_dir, stem + '.pgm')
            with open(path, 'wb') as fh:
                fh.write(b'P5\n%d %d\n255\n' % (m.shape[1], m.shape[0]))
                fh.write(m.tobytes())
        paths.append(path)
    return paths
