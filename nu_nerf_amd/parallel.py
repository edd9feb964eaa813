"""Data-parallel over rays: one process per GPU, full model replica, ONE flat fp32 all-reduce of the parameter
gradients per step (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The reference has no multi-GPU path (`multi_gpus: True` raises NotImplementedError, train/trainer_zero.py:74-75);
SURVEY.md section 8(e) specifies this one.  Rays are independent units, so the data path needs no collective: each rank
renders its own slice of the ray batch; only the ~2.55 M parameter gradients (10.2 MB) are summed and divided by the
world size.  Parameters that never receive a gradient in stage 1 (color_network.iors.*, infinity_far_bkgr.*) are left
out of the bucket instead of relying on unused-parameter detection.

Approximation inherited from per-rank means (documented in DESIGN.md): per-point means (eikonal) are averaged per
rank before the all-reduce, so ranks with different inner-point counts weigh points slightly differently from a
single-process run over the union batch.
"""
import torch
import torch.distributed as dist


def stage1_trainable_names(module):
    """Names of the parameters that take part in the gradient bucket."""
    skip = ('color_network.iors.', 'infinity_far_bkgr.')
    return [n for n, _ in module.named_parameters() if not n.startswith(skip)]


class GradAllReducer:
    def __init__(self, module, world_size, group=None):
        self.world = world_size
        self.group = group
        named = dict(module.named_parameters())
        self.params = [named[n] for n in stage1_trainable_names(module)]
        self.numel = sum(p.numel() for p in self.params)
        self._flat = None

    def all_reduce(self):
        """Sum the gradients over ranks and divide by the world size (in place on every .grad)."""
        if self.world <= 1:
            return
        dev = self.params[0].device
        if self._flat is None or self._flat.device != dev:
            self._flat = torch.zeros(self.numel, device=dev)
        flat = self._flat
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            else:
                flat[off:off + n].zero_()
            off += n
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n


def shard_rays(batch, rank, world):
    """Contiguous slice of a ray batch for this rank (rays are independent: no data-path collective)."""
    n = next(iter(batch.values())).shape[0]
    per = (n + world - 1) // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
