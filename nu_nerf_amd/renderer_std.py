"""Stage-1 renderer of the NON-zero-thickness configs (`zero_thickness: False`; every configs/shape/real/*.yaml):
drop-in for `network/renderer.py:NeROShapeRenderer` (registry network/renderer.py:2400; dispatch run_training.py:16-20).

Deltas w.r.t. the zero-thickness renderer (network/renderer.py vs network/renderer_zerothick.py):
  * `loss_normal` = sum_j w_j * max(n_j . d, 0)                        (renderer.py:693-705) -- composited by the HIP
    composite kernel as a 4th colour channel, gradients to weights and normals included;
  * colour_spec / colour_bkgr only on "candidate" rays whose sample 64 lies inside the unit sphere, colour_spec queried
    with the sphere-direction encoding of that point when shader_config.sphere_direction           (renderer.py:710-725);
  * `loss_mask` = L1(masks, acc) for NeRF-synthetic data                                            (renderer.py:477-478);
  * `train_ray_num` 1024, real-capture ray construction from poses (`_process_ray_batch`, renderer.py:347-361).
"""
import torch
import torch.nn.functional as F

from .renderer import NeROShapeRenderer as _ZeroThickRenderer


class NeROShapeRenderer(_ZeroThickRenderer):
    default_cfg = {**_ZeroThickRenderer.default_cfg, 'train_ray_num': 1024, 'downsample_ratio': 1.0, 'get_mask': False}

    def _spec_query_points(self, rays_o, rays_d, z_vals):
        S = z_vals.shape[1]
        if S <= 65:
            raise ValueError("the standard renderer takes sample 64 as the surface candidate (renderer.py:710): needs > 65 samples")
        with torch.no_grad():
            dist = z_vals[:, 65] - z_vals[:, 64]
            mid = z_vals[:, 64] + dist * 0.5
            pts = (rays_o + rays_d * mid[:, None]).contiguous()
            cand = torch.norm(pts, dim=-1) <= 1.0
        return pts, cand

    def _extra_outputs(self, outputs, nrm_sum):
        outputs['loss_normal'] = nrm_sum[:, None]

    def get_human_coordinate_poses(self, poses):
        """renderer.py:329-345 (only consumed by human_light, which is off in every config)."""
        pn = poses.shape[0]
        cam_cen = (-poses[:, :, :3].permute(0, 2, 1) @ poses[:, :, 3:])[..., 0]
        if not self.cfg['fixed_camera']:
            cam_cen = cam_cen.clone()
            cam_cen[..., 2] = 0
        Y = torch.zeros(pn, 3, device=poses.device)
        Y[:, 2] = -1.0
        Z = poses[:, 2, :3].clone()
        Z[:, 2] = 0
        Z = F.normalize(Z, dim=-1)
        X = torch.cross(Y, Z, dim=-1)
        R = torch.stack([X, Y, Z], 1)
        t = -R @ cam_cen[:, :, None]
        return torch.cat([R, t], -1)

    def _process_ray_batch(self, ray_batch, poses):
        """Real-capture rays from camera poses: o = -R^T t, d = normalize(R^T dirs), near/far from the unit sphere
        (renderer.py:347-361)."""
        rays_d = ray_batch['dirs']
        idxs = ray_batch['idxs'][..., 0]
        rays_o = (poses[:, :, :3].permute(0, 2, 1) @ -poses[:, :, 3:])[idxs, :, 0]
        rays_d = (poses[idxs, :, :3].permute(0, 2, 1) @ rays_d.unsqueeze(-1))[..., 0]
        rays_d = F.normalize(rays_d, dim=-1)
        near, far = self.near_far_from_sphere(rays_o, rays_d)
        return rays_o, rays_d, near, far, self.get_human_coordinate_poses(poses)[idxs]

    def train_step_rays(self, batch, step, rand=None, poses=None, fused=False):
        if 'dirs' in batch:
            rays_o, rays_d, near, far, hp = self._process_ray_batch(batch, poses)
        else:
            rays_o, rays_d, near, far, hp = self._process_nerf_ray_batch(batch)
            if not self.is_nerf:    # explicit rays of a real capture: bracket the unit sphere like the base class does
                near, far = self.near_far_from_sphere(rays_o, rays_d)
        outputs = self.render(rays_o, rays_d, near, far, hp, -1, self.get_anneal_val(step), is_train=True, step=step,
                              is_nerf=self.is_nerf, rand=rand, fused=fused)
        if not fused:
            outputs['loss_rgb'] = self.compute_rgb_loss(outputs['ray_rgb'], batch['rgbs'])
        if self.is_nerf and 'masks' in batch:
            outputs['loss_mask'] = F.l1_loss(batch['masks'], outputs['acc'], reduction='mean')
        return outputs


name2renderer = {'shape': NeROShapeRenderer}
