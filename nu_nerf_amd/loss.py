"""Loss registry with the reference's names and semantics (network/loss.py:215-227) and the trainer's
total (train/trainer_zero.py:153-161): total = sum over dict entries whose key starts with 'loss' of mean(v).

These operate on per-ray / per-point OUTPUT tensors of the renderer (tiny reductions); the heavy work and
all parameter gradients happen inside the renderer's HIP ops.
"""
import numpy as np
import torch
import torch.nn.functional as F


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


class MaterialRegLoss(Loss):
    """network/loss.py:51-62: passes `loss_mat_reg` / `loss_diffuse_light` through when the renderer provides them (no shipped
    renderer does)."""

    def __init__(self, cfg):
        pass

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        return {k: data_pr[k] for k in ('loss_mat_reg', 'loss_diffuse_light') if k in data_pr}


class TransmissionRegLoss(Loss):
    """network/loss.py:166-177: 0.1 * mean(transmission_weight^2) over the inner points."""
    default_cfg = {'transmission_reg_loss_weight': 0.1}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        if 'transmission' in data_pr:
            t = data_pr['transmission']
            return {'loss_trans_reg': F.mse_loss(t, torch.zeros_like(t)) * self.cfg['transmission_reg_loss_weight']}
        return {}


class MetallicRegLoss(Loss):
    """network/loss.py:179-190: 0.1 * mean(metallic^2) over the inner points."""
    default_cfg = {'metallic_reg_loss_weight': 0.1}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        if 'metallic' in data_pr:
            m = data_pr['metallic']
            return {'loss_metal_reg': F.mse_loss(m, torch.zeros_like(m)) * self.cfg['metallic_reg_loss_weight']}
        return {}


class OuterRegLoss(Loss):
    default_cfg = {'outer_reg_loss_weight': 0.5}

    def __init__(self, cfg):
        self.cfg = {**self.default_cfg, **cfg}

    def __call__(self, data_pr, data_gt, step, *args, **kwargs):
        if 'color_bkgr' in data_pr and step >= 15000:
            return {'loss_outer_reg': F.mse_loss(data_pr['color_bkgr'].flatten(),
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
    'mat_reg': MaterialRegLoss,
    'transmission_reg': TransmissionRegLoss,
    'metallic_reg': MetallicRegLoss,
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


# ------------------------------------------------------------------------------------------------------------------
# Fused loss assembly on the HIP loss kernels (csrc/loss.hip)
# ------------------------------------------------------------------------------------------------------------------
class _FusedLossFn(torch.autograd.Function):
    """rgb, acc, rgb_bg, spec_raw, gerr, nrm_sum -> (sum of the fused loss terms, terms[6], ray_rgb, colour_spec, loss_rgb[R]).
    Two launches forward, one backward; the upstream gradient never visits the host."""

    @staticmethod
    def forward(ctx, rgb, acc, rgb_bg, spec_raw, gerr, nrm_sum, gt, cand, white_bg, exp_max, w_eik, w_reg, w_nrm, point_weight=None):
        import ctypes
        from . import _lib as L
        lib = L.load()
        lib.nu_loss_workspace_bytes.restype = ctypes.c_longlong
        L.require_cuda(rgb, acc, rgb_bg, spec_raw, gt)
        dev = rgb.device
        R, P = rgb.shape[0], gerr.shape[0]
        ts = [t.detach().contiguous() if t is not None else None for t in (rgb, acc, rgb_bg, spec_raw, gerr, nrm_sum, gt)]
        rgb_, acc_, bg_, spec_, gerr_, nrm_, gt_ = ts
        cand_ = None if cand is None else cand.to(torch.uint8).contiguous()
        # data parallelism: device scalar that turns the per-rank eikonal mean into this rank's share of the global one
        pw_ = None if point_weight is None else point_weight.detach().reshape(-1)[:1].to(device=dev, dtype=torch.float32).contiguous()
        nb = lib.nu_loss_workspace_bytes(R, P)
        ws = torch.empty((nb + 3) // 4, device=dev)
        ray_rgb, color_spec, loss_rgb, terms = (torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev),
                                                torch.empty(R, device=dev), torch.empty(6, device=dev))
        L.check(lib.nu_loss_fwd(L.ptr(rgb_), L.ptr(acc_), L.ptr(bg_), L.ptr(spec_), L.ptr(gerr_ if P else None), L.ptr(nrm_), L.ptr(gt_),
                                L.ptr(cand_), R, P, int(white_bg), ctypes.c_float(exp_max), ctypes.c_float(w_eik), ctypes.c_float(w_reg),
                                ctypes.c_float(w_nrm), L.ptr(ray_rgb), L.ptr(color_spec), L.ptr(loss_rgb), L.ptr(terms), L.ptr(pw_), L.ptr(ws),
                                ctypes.c_longlong(nb), L.stream()), "nu_loss_fwd")
        ctx.save_for_backward(rgb_, acc_, bg_, spec_, gt_, ray_rgb, color_spec, loss_rgb, terms)
        ctx.cand, ctx.nrm, ctx.pw = cand_, nrm_ is not None, pw_
        ctx.k = (R, P, int(white_bg), float(exp_max), float(w_eik), float(w_reg), float(w_nrm))
        ctx.mark_non_differentiable(terms, ray_rgb, color_spec, loss_rgb)
        return terms[4].clone(), terms, ray_rgb, color_spec, loss_rgb

    @staticmethod
    def backward(ctx, d_total, *_):
        import ctypes
        from . import _lib as L
        lib = L.load()
        rgb_, acc_, bg_, spec_, gt_, ray_rgb, color_spec, loss_rgb, terms = ctx.saved_tensors
        R, P, white_bg, exp_max, w_eik, w_reg, w_nrm = ctx.k
        dev = rgb_.device
        up = d_total.detach().reshape(1).to(torch.float32).contiguous()
        d_rgb, d_acc, d_bg, d_spec = (torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, 3, device=dev),
                                      torch.empty(R, 3, device=dev))
        d_gerr = torch.empty(P, device=dev)
        d_nrm = torch.empty(R, device=dev) if ctx.nrm else None
        L.check(lib.nu_loss_bwd(L.ptr(rgb_), L.ptr(acc_), L.ptr(bg_), L.ptr(spec_), L.ptr(gt_), L.ptr(ctx.cand), L.ptr(ray_rgb),
                                L.ptr(color_spec), L.ptr(loss_rgb), L.ptr(terms), L.ptr(up), L.ptr(ctx.pw), R, P, white_bg, ctypes.c_float(exp_max),
                                ctypes.c_float(w_eik), ctypes.c_float(w_reg), ctypes.c_float(w_nrm), L.ptr(d_rgb), L.ptr(d_acc),
                                L.ptr(d_bg), L.ptr(d_spec), L.ptr(d_gerr if P else None), L.ptr(d_nrm), L.stream()), "nu_loss_bwd")
        return d_rgb, d_acc, d_bg, d_spec, d_gerr, d_nrm, None, None, None, None, None, None, None, None


_FUSED_TYPES = None


def fused_stage1_loss(renderer, batch, step, losses, rand=None, reducer=None):
    """One training forward + the trainer's total (train/trainer_zero.py:153-161) with the loss assembly on the HIP loss
    kernels: NeRFRenderLoss (charbonier), EikonalLoss, OuterRegLoss and NormalOrientationLoss are fused; StdRecorder, OccLoss,
    InitSDFRegLoss and MaskLoss entries keep their (O(1)-sized) torch form and are added on.  Returns (total, log, outputs)
    with the same log keys and values as `total_loss` gives on the unfused outputs.

    reducer (parallel.GradAllReducer, world > 1): every mean over a data-dependent subset -- eikonal, transmission / metallic
    regularisers, occlusion loss, the real-capture outer regulariser's candidate rays -- becomes this rank's share of the mean over
    the union of all ranks' subsets (`parallel.dp_weight_outputs`; the eikonal weight goes to the loss kernels as a device scalar),
    so the data-parallel step runs the SAME fused assembly as the single-GPU step and its all-reduced gradient equals the
    single-process one up to the two approximations stated at dp_weight_outputs (the 2048-point occlusion cap, the init-SDF
    normalisers of the first 1000 steps)."""
    from .parallel import dp_weight_outputs
    pw = None
    dp = reducer is not None and not getattr(reducer, 'solo', reducer.world <= 1)
    real_cand = getattr(renderer, 'candidate_rays', False)
    if renderer.cfg['rgb_loss'] != 'charbonier' or not any(isinstance(ls, NeRFRenderLoss) for ls in losses) or (dp and real_cand):
        # (the candidate-ray regulariser of the real-capture renderer takes its weight on the eager outputs)
        out = renderer.train_step_rays(batch, step, rand=rand)
        if dp:
            dp_weight_outputs(out, reducer, renderer)
        total, log = total_loss(out, losses, step)
        return total, log, out
    out = renderer.train_step_rays(batch, step, rand=rand, fused=True)
    if dp:
        pw = dp_weight_outputs(out, reducer, renderer, fused_eikonal=True)['inner']
    raw = out.pop('_raw')
    w_eik = w_reg = w_nrm = 0.0
    rest = []
    for ls in losses:
        if isinstance(ls, EikonalLoss):
            w_eik = float(ls.get_eikonal_weight(step))
        elif isinstance(ls, OuterRegLoss):
            w_reg = float(ls.cfg['outer_reg_loss_weight']) if step >= 15000 else 0.0
        elif isinstance(ls, NormalOrientationLoss):
            w_nrm = 1.0 if 'loss_normal' in out else 0.0
        elif isinstance(ls, NeRFRenderLoss):
            pass
        else:
            rest.append(ls)
    gerr = raw['gerr']
    nrm = raw['nrm_sum'] if w_nrm else None
    total, terms, ray_rgb, color_spec, loss_rgb = _FusedLossFn.apply(
        raw['rgb'], raw['acc'], raw['rgb_bg'], raw['spec_raw'], gerr, nrm, batch['rgbs'], raw['cand'], raw['is_nerf'], raw['exp_max'],
        w_eik, w_reg, w_nrm, pw)
    cand = raw['cand']
    out['ray_rgb'] = ray_rgb
    out['color_spec'] = color_spec if cand is None else color_spec[cand]
    out['color_bkgr'] = raw['rgb_bg'] if cand is None else raw['rgb_bg'][cand]
    out['loss_rgb'] = loss_rgb
    log = {'loss_rgb': loss_rgb}
    log['loss_eikonal'] = terms[1].reshape(1)
    if w_reg:
        log['loss_outer_reg'] = terms[2]
    if w_nrm:
        log['loss_normal'] = terms[3].reshape(1)
    for ls in rest:                      # std (log only), occ, init_sdf_reg, mask: already reduced or O(points in the shell)
        extra = ls(out, {}, step)
        for k, v in extra.items():
            if k.startswith('loss'):
                total = total + torch.mean(v)
        log.update(extra)
    return total, log, out
