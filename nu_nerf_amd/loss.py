"""Loss registry with the reference's names and semantics (network/loss.py:215-227) and the trainer's
total (train/trainer_zero.py:153-161): total = sum over dict entries whose key starts with 'loss' of mean(v).

These operate on per-ray / per-point OUTPUT tensors of the renderer (tiny reductions); the heavy work and
all parameter gradients happen inside the renderer's HIP ops.
"""
import numpy as np
import torch


class Loss:
    def __call__(self, data_pr, data_gt, step, **kwargs):
        return {}


class NeRFRenderLoss(Loss):
    def __init__(self, cfg):
        pass

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        keys = ('loss_rgb', 'loss_rgb_fine', 'loss_global_rgb', 'loss_rgb_inner', 'loss_rgb0', 'loss_rgb1', 'loss_masks')
        return {k: data_pr[k] for k in keys if k in data_pr}


class EikonalLoss(Loss):
    default_cfg = {"eikonal_weight": 0.1, 'eikonal_weight_anneal_begin': 0, 'eikonal_weight_anneal_end': 0}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}

    def get_eikonal_weight(self, step):
        c = self.cfg
        if step < c['eikonal_weight_anneal_begin']:
            return 0.0
        if c['eikonal_weight_anneal_begin'] <= step < c['eikonal_weight_anneal_end']:
            return c['eikonal_weight'] * (step - c['eikonal_weight_anneal_begin']) / \
                (c['eikonal_weight_anneal_end'] - c['eikonal_weight_anneal_begin'])
        return c['eikonal_weight']

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        return {'loss_eikonal': data_pr['gradient_error'] * self.get_eikonal_weight(step)}


class StdRecorder(Loss):
    default_cfg = {'apply_std_loss': False, 'std_loss_weight': 0.01, 'std_loss_weight_type': 'constant'}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        out = {}
        if 'std' in data_pr:
            out['std'] = data_pr['std']
            if self.cfg['apply_std_loss']:
                out['loss_std'] = data_pr['std'] * self.cfg['std_loss_weight']
        return out


class OccLoss(Loss):
    def __init__(self, cfg):
        pass

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        if 'loss_occ' in data_pr:
            return {'loss_occ': torch.mean(data_pr['loss_occ']).reshape(1)}
        return {}


class InitSDFRegLoss(Loss):
    """Keeps the SDF positive outside radius 1.05 and negative inside 0.1 during the first 1000 steps
    (network/loss.py:115-149)."""

    def __init__(self, cfg):
        pass

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        reg_step, small_t, large_t = 1000, 0.1, 1.05
        if 'sdf_vals' not in data_pr or 'sdf_pts' not in data_pr or step >= reg_step:
            return {}
        norm = torch.norm(data_pr['sdf_pts'], dim=-1)
        sdf = data_pr['sdf_vals']
        dev = sdf.device
        small = norm < small_t
        if torch.sum(small) > 0:
            sl = torch.mean(torch.clamp(sdf[small] - (norm[small] - small_t), min=0.0))
            sl = torch.sum(sl) / (torch.sum(sl > 1e-5) + 1e-3)
        else:
            sl = torch.zeros(1, device=dev)
        large = norm > large_t
        if torch.sum(large) > 0:
            ll = torch.clamp((norm[large] - large_t) - sdf[large], min=0.0)
            ll = torch.sum(ll) / (torch.sum(ll > 1e-5) + 1e-3)
        else:
            ll = torch.zeros(1, device=dev)
        w = (np.cos((step / reg_step) * np.pi) + 1) / 2
        return {'loss_sdf_large': ll * w, 'loss_sdf_small': sl * w}


class MaskLoss(Loss):
    default_cfg = {'mask_loss_weight': 0.01}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        if 'loss_mask' in data_pr:
            return {'loss_mask': data_pr['loss_mask'].reshape(1) * self.cfg['mask_loss_weight']}
        return {}


class OuterRegLoss(Loss):
    default_cfg = {'outer_reg_loss_weight': 0.5}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        if 'color_bkgr' in data_pr and step >= 15000:
            return {'loss_outer_reg': torch.nn.functional.mse_loss(data_pr['color_bkgr'].flatten(),
                                                                   data_pr['color_spec'].flatten())
                    * self.cfg['outer_reg_loss_weight']}
        return {}


class NormalOrientationLoss(Loss):
    def __init__(self, cfg):
        pass

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        if 'loss_normal' in data_pr:
            return {'loss_normal': torch.mean(data_pr['loss_normal']).reshape(1)}
        return {}


name2loss = {
    'nerf_render': NeRFRenderLoss,
    'eikonal': EikonalLoss,
    'std': StdRecorder,
    'init_sdf_reg': InitSDFRegLoss,
    'occ': OccLoss,
    'mask': MaskLoss,
    'outer_reg': OuterRegLoss,
    'normal_ori': NormalOrientationLoss,
}

SPHEREPOT_LOSSES = ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'mask', 'outer_reg']


def total_loss(outputs, losses, step):
    """trainer_zero.py:153-161."""
    log = {}
    for ls in losses:
        log.update(ls(outputs, {}, step))
    total = 0
    for k, v in log.items():
        if k.startswith('loss'):
            total = total + torch.mean(v)
    return total, log
