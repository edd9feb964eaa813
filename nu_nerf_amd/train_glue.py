"""Trainer-step glue either side of the hot path.

  WarmUpCosLR / name2lr_manager   train/lr_common_manager.py:4-52 (same class surface: construct_optimizer, __call__)
  FusedAdam                        torch.optim.Adam's update as ONE HIP launch per 80 tensors (csrc/mlp.hip: adam_kernel);
                                   state_dict() uses Adam's keys (step, exp_avg, exp_avg_sq), so optimizer states move
                                   between the two (checkpoints of the reference resume here and vice versa)
  save_checkpoint / load_checkpoint   the reference's file layout (train/trainer_zero.py:204-223)
  train_step                       the loop body of Trainer_zero.run (train/trainer_zero.py:131-161) + the gradient
                                   all-reduce of the data-parallel build
"""
import ctypes

import numpy as np
import torch

from . import _lib as L


class LearningRateManager:
    @staticmethod
    def set_lr_for_all(optimizer, lr):
        for g in optimizer.param_groups:
            g['lr'] = lr

    def construct_optimizer(self, optimizer, network):
        return optimizer(network.parameters(), lr=1e-3)

    def __call__(self, optimizer, step, *args, **kwargs):
        raise NotImplementedError


class WarmUpCosLR(LearningRateManager):
    default_cfg = {'end_warm': 5000, 'end_iter': 300000, 'lr': 5e-4}

    def __init__(self, cfg):
        cfg = {**self.default_cfg, **cfg}
        self.warm_up_end = cfg['end_warm']
        self.learning_rate_alpha = 0.05
        self.end_iter = cfg['end_iter']
        self.learning_rate = cfg['lr']

    def factor(self, step):
        if step < self.warm_up_end:
            return step / self.warm_up_end
        alpha = self.learning_rate_alpha
        progress = (step - self.warm_up_end) / (self.end_iter - self.warm_up_end)
        return (np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha

    def __call__(self, optimizer, step, *args, **kwargs):
        lr = self.learning_rate * self.factor(step)
        self.set_lr_for_all(optimizer, lr)
        return lr


name2lr_manager = {'warm_up_cos': WarmUpCosLR}


class AdamDesc(ctypes.Structure):
    """Mirror of NuAdamDesc (include/nu_nerf.h)."""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_longlong), ("blk_begin", ctypes.c_int), ("pad_", ctypes.c_int)]


class FusedAdam(torch.optim.Optimizer):
    """Adam (no weight decay, no amsgrad: what the reference trains with) on the HIP multi-tensor kernel.  Parameters
    without a gradient are skipped like torch.optim.Adam does; every parameter group keeps its own lr / betas / eps and
    every PARAMETER its own step count (bias correction), as torch.optim.Adam does.  There is no CPU fallback: `step()` on non-CUDA parameters raises."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._lib = None
        self._plans = {}          # group index -> launch plan for the current set of parameters with gradients
        self._steps = None        # one CPU vector holding every parameter's step count; state[p]['step'] is a 0-d view of it

    def _library(self):
        if self._lib is None:
            self._lib = L.load()
            assert self._lib.nu_adam_desc_size() == ctypes.sizeof(AdamDesc), "AdamDesc ABI mismatch"
        return self._lib

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()       # new state tensors: pointers and step counts are read again
        self._steps = None

    def _build_plan(self, gi, ps):
        """Everything about a group's launch that does not change from step to step: validation, lazily created Adam state,
        the step counts gathered into ONE CPU vector (state[p]['step'] stays a tensor, as torch.optim.Adam keeps it, but becomes
        a view of that vector -- one vector add per step instead of one tensor op per parameter), and per distinct step count a
        descriptor table with the parameter / moment pointers filled in (parameters that start receiving gradients later, e.g.
        deviation_network.variance at freeze_inv_s_step, or that were restored with their own counts form their own bucket;
        all counts advance together, so the buckets are static)."""
        if self._steps is None:
            cap = sum(len(g['params']) for g in self.param_groups)
            self._steps = torch.zeros(max(cap, 1))
            self._slot = {}
        buckets = {}
        for p in ps:
            L.require_cuda(p)
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("FusedAdam: contiguous fp32 parameters only")
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            slot = self._slot.setdefault(p, len(self._slot))
            if slot >= self._steps.numel():       # add_param_group after the first step
                grown = torch.zeros(2 * slot + 1)
                grown[:self._steps.numel()] = self._steps
                self._steps = grown
                self._plans.clear()
                for q, sl in self._slot.items():
                    if q in self.state and 'step' in self.state[q]:
                        self.state[q]['step'] = self._steps[sl]
            view = self._steps[slot]
            if st['step'] is not view and st['step'].data_ptr() != view.data_ptr():
                view.copy_(st['step'].to(torch.float32))
            st['step'] = view
            buckets.setdefault(int(view), []).append((p, st, slot))
        plan = []
        for count, items in buckets.items():
            descs = (AdamDesc * len(items))()
            for d, (p, st, _) in zip(descs, items):
                d.p, d.m, d.v, d.n = p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()
            plan.append({'count': count, 'params': [p for p, _, _ in items], 'descs': descs,
                         'slots': torch.tensor([sl for _, _, sl in items], dtype=torch.long)})
        return {'sig': tuple(id(p) for p in ps), 'buckets': plan}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = self._library()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            plan = self._plans.get(gi)
            if plan is None or plan['sig'] != tuple(id(p) for p in ps):
                plan = self._plans[gi] = self._build_plan(gi, ps)
            b1, b2 = group['betas']
            keep = []               # contiguous copies of gradients stay alive until every launch is enqueued
            for bk in plan['buckets']:
                descs = bk['descs']
                for d, p in zip(descs, bk['params']):
                    g = p.grad
                    if not g.is_contiguous():
                        g = g.contiguous()
                        keep.append(g)
                    d.g = g.data_ptr()
                    d.p = p.data_ptr()        # (re-read every step: `module.to(...)` gives a Parameter new storage)
                bk['count'] += 1
                self._steps[bk['slots']] += 1
                L.check(lib.nu_adam_step(descs, len(descs), ctypes.c_double(group['lr']), ctypes.c_double(b1),
                                         ctypes.c_double(b2), ctypes.c_double(group['eps']), bk['count'], L.stream()), "nu_adam_step")
            del keep
        return loss


def save_checkpoint(path, network, optimizer, step, best_para=0):
    """train/trainer_zero.py:215-223"""
    torch.save({'step': step, 'best_para': best_para, 'network_state_dict': network.state_dict(),
                'optimizer_state_dict': optimizer.state_dict()}, path)


def load_checkpoint(path, network, optimizer=None, map_location=None):
    """train/trainer_zero.py:204-213 (strict=False like the reference).  Only tensors and plain containers are read."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    network.load_state_dict(ck['network_state_dict'], strict=False)
    if optimizer is not None and 'optimizer_state_dict' in ck:
        optimizer.load_state_dict(ck['optimizer_state_dict'])
    return ck.get('best_para', 0), ck.get('step', 0)


def train_step(network, optimizer, lr_manager, losses, step, batch=None, reducer=None):
    """One iteration of Trainer_zero.run's loop body.  batch=None: the module draws from its own device-resident ray
    store (forward({'step': step})); otherwise an explicit {'rays_o','rays_d','rgbs'} batch."""
    lr = lr_manager(optimizer, step)
    optimizer.zero_grad(set_to_none=True)
    if batch is not None and hasattr(network, 'engine') and type(network).__name__ == 'NeROShapeRenderer':
        # explicit stage-1 ray batch: loss assembly on the HIP loss kernels (loss.fused_stage1_loss), same total and log; with a
        # reducer the eikonal mean takes this rank's point weight inside the kernels, so N > 1 runs the same step as N = 1
        from .loss import fused_stage1_loss
        total, log_info, _ = fused_stage1_loss(network, batch, step, losses, reducer=reducer)
        total.backward()
        if reducer is not None:
            reducer.all_reduce()
        optimizer.step()
        return total.detach(), log_info, lr
    outputs = network({'step': step}) if batch is None else network.train_step_rays(batch, step)
    if reducer is not None and not getattr(reducer, 'solo', reducer.world <= 1) and type(network).__name__ == 'NeROShapeRenderer':
        # means over data-dependent subsets (eikonal, material regularisers, occlusion loss, candidate rays) become this rank's share
        # of the mean over the union of all ranks' subsets: the all-reduced gradient is the single-process gradient on the global
        # batch (parallel.dp_weight_outputs, which also states the two approximations that remain)
        from .parallel import dp_weight_outputs
        dp_weight_outputs(outputs, reducer, network)
    log_info = {}
    for loss in losses:
        log_info.update(loss(outputs, {'step': step}, step))
    total = 0
    for k, v in log_info.items():
        if k.startswith('loss'):
            total = total + torch.mean(v)
    total.backward()
    if reducer is not None:
        reducer.all_reduce()
    optimizer.step()
    return total.detach(), log_info, lr
