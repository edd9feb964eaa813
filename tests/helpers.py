"""Shared test helpers: rebuild the seed-defined weights / inputs the golden vectors were made with."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

GOLDEN_CFG = {'n_samples': 32, 'n_importance': 32, 'n_bg_samples': 16}


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def parity_params(device="cpu", requires_grad=False, **init_kw):
    """The weights oracle/gen_golden.py loaded into the reference (same seeds)."""
    from nu_nerf_amd.params import init_stage1_params, randomize_for_parity
    p = randomize_for_parity(init_stage1_params(6033, **init_kw), seed=1)
    out = {}
    for k, v in p.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(device)
        if requires_grad and not k.endswith("FG_LUT"):
            t.requires_grad_(True)
        out[k] = t
    return out


def oracle_cfg(**over):
    from oracle.stage1_oracle import DEFAULT_CFG
    cfg = dict(DEFAULT_CFG)
    cfg.update(GOLDEN_CFG)
    cfg.update(over)
    return cfg


def rel_err(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


STD_CFG = {'n_samples': 64, 'n_importance': 32, 'n_bg_samples': 16, 'sphere_direction': True, 'refrac_freq': 3,
           'eikonal_weight': 0.05, 'outer_reg_loss_weight': 0.1, 'normal_ori': True, 'is_nerf': False}
