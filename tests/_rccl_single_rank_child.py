"""Child process of tests/test_rccl_single_rank_gpu.py (not a test module): ONE rank on cuda:0 with the "nccl" backend (= RCCL),
started with RANK / WORLD_SIZE / MASTER_* in the environment like a rank of `torch.distributed.run`.  Runs one fused stage-1 train
step without a reducer and one with a GradAllReducer that issues its collectives on the one-rank group (always_collective), and checks
that the bits agree: on one rank every count ratio is exactly 1 and the mean divides by 1, so any difference would be the transport."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from helpers import golden                                        # noqa: E402
from test_stage1_gpu import CFG, make_net                         # noqa: E402
from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, fused_stage1_loss      # noqa: E402
from nu_nerf_amd.parallel import GradAllReducer                   # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=dev)                # what bench.py does for N > 1
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    g = golden("train_step20000_r48.npz")
    step = 20000                                                  # occlusion + outer-regulariser losses on, inv_s trainable
    batch = {k: torch.from_numpy(g[k]).to(dev) for k in ('rays_o', 'rays_d', 'rgbs')}
    rand = (torch.from_numpy(g['u1']).to(dev), torch.from_numpy(g['u2']).to(dev))
    losses = [name2loss[n](CFG) for n in SPHEREPOT_LOSSES + ['transmission_reg', 'metallic_reg']]

    def run(with_reducer):
        net = make_net(dev)
        red = GradAllReducer(net, 1, always_collective=True) if with_reducer else None
        total, _, _ = fused_stage1_loss(net, batch, step, losses, rand=rand, reducer=red)
        total.backward()
        if red is not None:
            red.all_reduce()
        torch.cuda.synchronize()
        return float(total.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, red

    loss0, grads0, _ = run(False)
    loss0b, grads0b, _ = run(False)
    repeatable = loss0 == loss0b and all(torch.equal(grads0[n], grads0b[n]) for n in grads0)     # the step itself, run twice
    loss1, grads1, red = run(True)
    assert red.in_place_calls == 1 and red.gathered_calls == 0, (red.in_place_calls, red.gathered_calls)   # the zero-copy flat range
    assert set(grads0) == set(grads1) and len(grads0) > 100
    worst = 0.0
    for n in grads0:
        worst = max(worst, float((grads0[n] - grads1[n]).double().norm() / (grads0[n].double().norm() + 1e-30)))
    if repeatable:
        assert loss0 == loss1 and worst == 0.0, (loss0, loss1, worst)
    else:                   # (not expected in the exact-fp32 mode: the comparison then is as good as the step's own repeatability)
        assert abs(loss0 - loss1) <= 2e-6 * abs(loss0) and worst <= 2e-6, (loss0, loss1, worst)
    # the timing reduction of bench.py (MAX over ranks) and its barrier
    t = torch.tensor([12.5], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 12.5
    dist.barrier()
    n_flat = int(red._shared_flat().numel())
    dist.destroy_process_group()
    print(f"RCCL_SINGLE_RANK_OK repeatable={repeatable} worst={worst:.1e} params_with_grad={len(grads0)} flat_elems={n_flat} loss={loss1:.6f}", flush=True)


if __name__ == "__main__":
    main()
