"""Data-parallel over rays: one process per GPU, full model replica, ONE flat fp32 all-reduce of the parameter
gradients per step (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The reference has no multi-GPU path (`multi_gpus: True` raises NotImplementedError, train/trainer_zero.py:74-75);
SURVEY.md section 8(e) specifies this one.  Rays are independent units, so the data path needs no collective: each rank
renders its own slice of the ray batch; only the ~2.55 M parameter gradients (10.2 MB) are summed and divided by the
world size.  Parameters that never receive a gradient in stage 1 (color_network.iors.*, infinity_far_bkgr.*) are left
out of the bucket instead of relying on unused-parameter detection.

Per-ray loss terms average exactly (equal ray counts per rank).  Means over data-dependent subsets -- the inner points (eikonal,
transmission / metallic regularisers), the occlusion-loss points, the candidate rays of the real-capture outer regulariser -- are
taken per rank; `dp_weight_outputs` multiplies each by its count ratio n_local * world / sum_r n_r (`GradAllReducer.count_weights`:
one small all-reduce, no host sync for device-resident counts), which makes the all-reduced gradient the gradient of the mean over
the union of all ranks' subsets.  What stays approximate under sharding is stated at `dp_weight_outputs`.
"""
import torch
import torch.distributed as dist


_DEAD = ('.iors.', '.infinity_far_bkgr.')       # never receive a gradient: every shading network's `iors`, the far background


def stage1_trainable_names(module):
    """Names of the parameters that take part in the gradient bucket: everything except the parameters no loss reaches
    (color_network.iors.*, infinity_far_bkgr.*).  Works for the stage-2 modules too (their stage-1 network sits under
    `stage1_network.`; aliases of one Parameter are listed once)."""
    return [n for n, _ in module.named_parameters() if not any(d in '.' + n for d in _DEAD)]


class GradAllReducer:
    def __init__(self, module, world_size, group=None, always_collective=False):
        self.world = world_size
        self.group = group
        # always_collective: issue the collectives on a ONE-rank group too (every weight is exactly 1, the mean divides by 1: the step's
        # bits do not change) -- the RCCL rehearsal a one-GPU box allows (tests/test_rccl_single_rank_gpu.py)
        self.solo = world_size <= 1 and not always_collective
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
        if missing > 64:
            # more than a frozen scalar or two is missing: which parameters have a gradient may then be rank-dependent (stage 2),
            # and ranks must not disagree on the path they take -- the gathered path settles the set with a collective of its own
            return None
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(store, lo, (hi - lo,), (1,))

    def all_reduce(self):
        """Sum the gradients over ranks and divide by the world size (in place on every .grad): one collective.
        Parameters whose .grad is None are left alone (no zero gradient is materialised for them): which parameters have a
        gradient is decided by the step index and the loss set, identically on every rank."""
        if self.solo:
            return
        flat = self._shared_flat()
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            self.in_place_calls += 1
            return
        self.gathered_calls += 1
        # general case (gradients from several sources -- the stage-2 modules: two engines, frozen and data-dependent subsets): one
        # gather, one collective, one multi-tensor scatter.  Which parameters have a gradient may differ between ranks there (a rank
        # whose rays all miss the object trains no inner network that step): the union over ranks decides, a rank that lacks one
        # contributes zeros -- so every rank reduces the same buffer and Adam sees the same set of parameters everywhere.
        dev = self.params[0].device
        present = torch.tensor([1.0 if p.grad is not None else 0.0 for p in self.params], device=dev)
        dist.all_reduce(present, op=dist.ReduceOp.MAX, group=self.group)
        for p, have in zip(self.params, present.tolist()):
            if have and p.grad is None:
                p.grad = torch.zeros_like(p)
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

    def count_weights(self, counts, device):
        """[k] device tensor of n_local_i * world / sum_r n_{r,i} for k subset sizes (python numbers or device-resident counts) in
        ONE small all-reduce; a subset that is empty on every rank gets weight 1."""
        # (python numbers become device scalars through a fill launch, not a host -> device copy: a pageable copy makes the host wait
        # for the stream, and between the forward and the loss that wait is a gap of the whole enqueue lead -- about 1 ms per step)
        n = torch.cat([c.reshape(-1)[:1].to(device=device, dtype=torch.float32) if torch.is_tensor(c)
                       else torch.full((1,), float(c), dtype=torch.float32, device=device) for c in counts])
        if self.solo:
            return torch.ones_like(n)
        tot = n.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        return torch.where(tot > 0, n * self.world / torch.clamp(tot, min=1.0), torch.ones_like(n))

    def point_weight(self, n_local, device):
        """Factor that turns a per-rank mean over n_local points into this rank's share of the mean over the union batch:
        n_local * world / sum_r n_r (a device scalar; no host sync).  Multiply a per-point loss input by it (e.g.
        outputs['gradient_error']) and the all-reduced gradient equals the single-process gradient over all ranks' rays."""
        if torch.is_tensor(n_local):       # a device-resident count (engine ctx['P_in_dev']): no host -> device copy at all
            n = n_local.reshape(-1)[:1].to(device=device, dtype=torch.float32)
        else:
            n = torch.full((1,), float(n_local), dtype=torch.float32, device=device)
        if self.solo:
            return torch.ones(1, device=device)
        tot = n.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        return n * self.world / torch.clamp(tot, min=1.0)


def dp_weight_outputs(out, reducer, renderer, fused_eikonal=False):
    """Make the subset means of a rank's renderer outputs shares of the means over the union batch (in place on `out`; call between
    the forward and the loss registry).  Exact for: the eikonal mean, TransmissionRegLoss / MetallicRegLoss (network/loss.py:166-192;
    all three over the inner points), OccLoss (the points of compute_occ_loss, renderer_zerothick.py:695-723) and the real-capture
    OuterRegLoss over candidate rays (network/renderer.py:710-725) -- each gets the count ratio of ITS subset; squared terms take its
    square root on their inputs.  Returns the weights {'inner', 'occ', 'cand'} (device scalars).

    Approximations that remain under sharding (stated, not hidden):
      * the occlusion loss caps its points at occ_loss_max_pn = 2048 by a random subsample (renderer_zerothick.py:708-714): each rank
        caps ITS candidates, so N ranks may use up to N x 2048 points where the single-process step on the union batch would use
        2048 -- the same estimator on a larger sample; exact whenever no rank reaches the cap;
      * InitSDFRegLoss (network/loss.py:131-142, first 1000 steps only) normalises each sum by the number of violating points: a
        ratio of two sums, taken per rank and then averaged (exact only when the ranks' violating-point counts are equal).
    fused_eikonal: the eikonal weight is applied inside the fused HIP loss kernels (loss.fused_stage1_loss), not here."""
    ctx = renderer.engine().last_ctx
    dev = out['ray_rgb'].device if 'ray_rgb' in out else ctx['P_in_dev'].device
    cand = getattr(renderer, '_last_cand', None)   # bool [R] mask of the candidate rays, or None when every ray takes part
    n_cand = cand.sum() if cand is not None else 0
    w = reducer.count_weights([ctx['P_in_dev'], getattr(renderer, '_n_occ', 0), n_cand], dev)
    w_in, w_occ, w_cand = w[0:1], w[1:2], w[2:3]
    if not fused_eikonal and 'gradient_error' in out:
        out['gradient_error'] = out['gradient_error'] * w_in
    for k in ('transmission', 'metallic'):
        if k in out and torch.is_tensor(out[k]) and out[k].requires_grad:
            out[k] = out[k] * torch.sqrt(w_in)
    if 'loss_occ' in out and getattr(renderer, '_n_occ', 0) > 0:
        out['loss_occ'] = out['loss_occ'] * w_occ
    if cand is not None and 'color_spec' in out:
        r = torch.sqrt(w_cand)
        out['color_spec'], out['color_bkgr'] = out['color_spec'] * r, out['color_bkgr'] * r
    return {'inner': w_in, 'occ': w_occ, 'cand': w_cand}


def shard_rays(batch, rank, world):
    """Contiguous slice of a ray batch for this rank (rays are independent: no data-path collective)."""
    n = next(iter(batch.values())).shape[0]
    if n % world:
        raise ValueError(f"shard_rays: {n} rays do not divide over {world} ranks (per-ray loss means would not average exactly)")
    per = n // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
