"""Trainer-step glue: WarmUpCos schedule, FusedAdam state compatibility with torch.optim.Adam, the
reference's checkpoint layout.  CPU part; the HIP kernel itself is checked in test_train_glue_gpu.py."""
import copy
import math

import numpy as np
import pytest
import torch

from nu_nerf_amd.train_glue import FusedAdam, WarmUpCosLR, load_checkpoint, name2lr_manager, save_checkpoint


def test_warm_up_cos_schedule_matches_reference_formula():
    # train/lr_common_manager.py:22-46 with its defaults (end_warm 5000, end_iter 300000, lr 5e-4, alpha 0.05)
    mgr = name2lr_manager['warm_up_cos']({})
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    assert mgr(opt, 0) == 0.0
    assert mgr(opt, 2500) == pytest.approx(2.5e-4)
    assert mgr(opt, 5000) == pytest.approx(5e-4)
    mid = 5e-4 * ((math.cos(math.pi * 0.5) + 1) * 0.5 * 0.95 + 0.05)
    assert mgr(opt, 152500) == pytest.approx(mid)
    assert mgr(opt, 300000) == pytest.approx(5e-4 * 0.05)
    assert opt.param_groups[0]['lr'] == pytest.approx(5e-4 * 0.05)
    assert WarmUpCosLR({'lr': 1e-3, 'end_warm': 10}).factor(5) == 0.5
    built = mgr.construct_optimizer(torch.optim.Adam, torch.nn.Linear(2, 2))
    assert isinstance(built, torch.optim.Adam) and built.param_groups[0]['lr'] == 1e-3


def test_fused_adam_has_no_cpu_fallback_and_shares_adams_state_layout(tmp_path):
    from nu_nerf_amd._lib import NuNerfLibraryError
    net = torch.nn.Linear(3, 2)
    ref = torch.optim.Adam(net.parameters(), lr=1e-3)
    net(torch.ones(1, 3)).sum().backward()
    ref.step()
    fused = FusedAdam(net.parameters(), lr=1e-3)
    fused.load_state_dict(copy.deepcopy(ref.state_dict()))                      # a reference checkpoint's optimizer state resumes here
    st = fused.state[next(iter(net.parameters()))]
    assert set(st) == {'step', 'exp_avg', 'exp_avg_sq'} and int(st['step']) == 1
    back = torch.optim.Adam(net.parameters(), lr=1e-3)
    back.load_state_dict(copy.deepcopy(fused.state_dict()))                     # ... and ours resumes in torch's Adam
    assert int(back.state[next(iter(net.parameters()))]['step']) == 1
    with pytest.raises(NuNerfLibraryError):
        fused.step()                                             # CPU parameters: refuse, do not fall back
    # checkpoint file layout of train/trainer_zero.py:215-223, read back without unpickling code
    fn = str(tmp_path / "model.pth")
    save_checkpoint(fn, net, fused, step=7, best_para=1.5)
    raw = torch.load(fn, weights_only=True)
    assert set(raw) == {'step', 'best_para', 'network_state_dict', 'optimizer_state_dict'}
    net2, opt2 = torch.nn.Linear(3, 2), torch.optim.Adam(torch.nn.Linear(3, 2).parameters())
    opt2 = torch.optim.Adam(net2.parameters())
    best, step = load_checkpoint(fn, net2, opt2)
    assert (best, step) == (1.5, 7)
    assert torch.equal(net2.weight, net.weight)
