"""Data-parallel over rays: one process per GPU, full model replica, ONE flat fp32 all-reduce of the parameter
gradients per step (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The reference has no multi-GPU path (`multi_gpus: True` raises NotImplementedError, train/trainer_zero.py:74-75);
SURVEY.md section 8(e) specifies this one.  Rays are independent units, so the data path needs no collective: each rank
renders its own slice of the ray batch; only the ~2.55 M parameter gradients (10.2 MB) are summed and divided by the
world size.  Parameters that never receive a gradient in stage 1 (color_network.iors.*, infinity_far_bkgr.*) are left
out of the bucket instead of relying on unused-parameter detection.

Per-ray loss terms average exactly (equal ray counts per rank).  Per-point means (eikonal over the inner points) are
taken per rank; `GradAllReducer.point_weight` gives the count ratio that makes them exact as well (one scalar
all-reduce, no host sync) -- bench.py applies it to `gradient_error`.
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
        self.in_place_calls = 0      # how often the zero-copy path ran (reported by bench.py)
        self.gathered_calls = 0

    def _shared_flat(self):
        """The renderer's backward hands out every gradient as a slice of ONE flat buffer (engine.grad_views order) and
        autograd keeps those slices (it detaches, it does not copy): when the .grad tensors tile one storage range, the
        collective runs on that range in place -- no gather, no scatter.  A parameter WITHOUT a gradient this step
        (deviation_network.variance while step < freeze_inv_s_step: every rank alike, the step decides) may leave a hole of
        at most its own size inside the range: the hole is part of the same zero-initialised buffer, is reduced along with the
        rest and stays unobserved -- .grad stays None, so Adam creates no state for it (as in the 1-GPU run)."""
        live = [p.grad for p in self.params if p.grad is not None]
        if not live:
            return None
        missing = sum(p.numel() for p in self.params if p.grad is None)
        g0 = live[0]
        store = g0.untyped_storage()
        sp = store.data_ptr()
        lo, hi, tot = None, None, 0
        for g in live:
            if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != sp:
                return None
            o, n = g.storage_offset(), g.numel()
            lo = o if lo is None else min(lo, o)
            hi = o + n if hi is None else max(hi, o + n)
            tot += n
        if tot + missing != self.numel or not (tot <= hi - lo <= tot + missing):     # overlaps, or gaps nobody owns
            return None
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(store, lo, (hi - lo,), (1,))

    def all_reduce(self):
        """Sum the gradients over ranks and divide by the world size (in place on every .grad): one collective.
        Parameters whose .grad is None are left alone (no zero gradient is materialised for them): which parameters have a
        gradient is decided by the step index and the loss set, identically on every rank."""
        if self.world <= 1:
            return
        flat = self._shared_flat()
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            self.in_place_calls += 1
            return
        self.gathered_calls += 1
        # general case (gradients from several sources): one gather, one collective, one multi-tensor scatter
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        views, off = [], 0
        for g in grads:
            views.append(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(grads, views)

    def point_weight(self, n_local, device):
        """Factor that turns a per-rank mean over n_local points into this rank's share of the mean over the union batch:
        n_local * world / sum_r n_r (a device scalar; no host sync).  Multiply a per-point loss input by it (e.g.
        outputs['gradient_error']) and the all-reduced gradient equals the single-process gradient over all ranks' rays."""
        if torch.is_tensor(n_local):       # a device-resident count (engine ctx['P_in_dev']): no host -> device copy at all
            n = n_local.reshape(-1)[:1].to(device=device, dtype=torch.float32)
        else:
            n = torch.tensor([float(n_local)], device=device)
        if self.world <= 1:
            return torch.ones(1, device=device)
        tot = n.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        return n * self.world / torch.clamp(tot, min=1.0)


def shard_rays(batch, rank, world):
    """Contiguous slice of a ray batch for this rank (rays are independent: no data-path collective)."""
    n = next(iter(batch.values())).shape[0]
    if n % world:
        raise ValueError(f"shard_rays: {n} rays do not divide over {world} ranks (per-ray loss means would not average exactly)")
    per = n // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
