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
    candidate_rays = True       # colour_spec / colour_bkgr on candidate rays only (renderer.py:710-725): a data-dependent subset

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

    def train_step_rays(self, batch, step, rand=None, poses=None, fused=False):
        """renderer.py:459-481: the base class's step (explicit rays, or pixel directions + poses through the inherited
        _process_ray_batch / get_human_coordinate_poses) plus the mask loss of the NeRF-synthetic data."""
        outputs = super().train_step_rays(batch, step, rand=rand, fused=fused, poses=poses)
        if self.is_nerf and 'masks' in batch:
            outputs['loss_mask'] = F.l1_loss(batch['masks'], outputs['acc'], reduction='mean')
        return outputs


name2renderer = {'shape': NeROShapeRenderer}
