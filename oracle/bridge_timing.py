"""Bridge measurement of BASELINE.md section 3.2 (build container only): the TRUE reference (imported under the shims of
oracle/gen_golden.py) and this repo's CPU oracle, timed on identical inputs -- full training iterations (sample_ray ->
render_core -> the seven configured losses -> backward -> Adam) on the container's 8 cores.  The ratio is what links the
`cpu_baseline` of bench.py (the oracle, timed on the GPU box's host cores) to the reference itself, which cannot travel.

Usage:  python oracle/bridge_timing.py            (prints a markdown table row per configuration)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import install_shims, to_t   # noqa: E402


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from network.renderer_zerothick import NeROShapeRenderer  # reference
    from network.loss import name2loss
    from nu_nerf_amd.params import init_stage1_params
    from nu_nerf_amd.synthetic import make_rays
    from oracle import stage1_oracle as O

    for R, ns, ni, nbg in ((256, 32, 32, 32), (256, 64, 64, 32)):
        cfg = {'name': 'bridge', 'network': 'shape', 'database_name': 'nerf/spherepot', 'apply_occ_loss': True, 'occ_loss_step': 15000,
               'is_nerf': True, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1,
               'loss': ['nerf_render', 'eikonal', 'std', 'init_sdf_reg', 'occ', 'mask', 'outer_reg'],
               'n_samples': ns, 'n_importance': ni, 'n_bg_samples': nbg}
        params = init_stage1_params(6033)
        rays = make_rays(R, seed=6033)
        o, d, rgbs = (torch.from_numpy(rays[k]) for k in ('rays_o', 'rays_d', 'rgbs'))
        dn = torch.nn.functional.normalize(d, dim=-1)
        step0 = 20000

        # ---- the reference ----
        net = NeROShapeRenderer(cfg, training=False)
        net.load_state_dict(to_t(params), strict=True)
        losses = [name2loss[n](cfg) for n in cfg['loss']]
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        near, far = torch.full((R, 1), 0.8), torch.full((R, 1), 4.5)
        t_ref = []
        for it in range(4):
            t0 = time.time()
            opt.zero_grad()
            z = net.sample_ray(o, dn, near, far, 1.0)
            out = net.render_core(o, dn, z, torch.zeros(R, 3, 4), cos_anneal_ratio=net.get_anneal_val(step0 + it), step=step0 + it,
                                  is_train=True, is_nerf=True)
            out['loss_rgb'] = net.compute_rgb_loss(out['ray_rgb'], rgbs)
            log = {}
            for ls in losses:
                log.update(ls(out, {}, step0 + it))
            total = sum(torch.mean(v) for k, v in log.items() if k.startswith('loss'))
            total.backward()
            opt.step()
            if it:
                t_ref.append(time.time() - t0)
        inner = float(out['gradient_error'].numel()) / (R * z.shape[1])

        # ---- the oracle ----
        ocfg = dict(O.DEFAULT_CFG)
        ocfg.update(n_samples=ns, n_importance=ni, n_bg_samples=nbg)
        P = {}
        for k, v in params.items():
            t = torch.from_numpy(np.ascontiguousarray(v))
            if not k.endswith('FG_LUT') and not k.startswith('infinity') and '.iors.' not in k:
                t.requires_grad_(True)
            P[k] = t
        oopt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-3)
        t_or = []
        for it in range(4):
            t0 = time.time()
            oopt.zero_grad()
            total, _, _ = O.train_step(P, ocfg, o, d, rgbs, step0 + it)
            total.backward()
            oopt.step()
            if it:
                t_or.append(time.time() - t0)
        a, b = 1e3 * float(np.mean(t_ref)), 1e3 * float(np.mean(t_or))
        print(f"| R={R}, n_samples={ns}, n_importance={ni}, n_bg={nbg}, step 20000+ | {a:.0f} ms/iter ({R / a * 1e3:.0f} rays/s) | "
              f"{b:.0f} ms/iter ({R / b * 1e3:.0f} rays/s) | {a / b:.2f} | {inner:.2f} |", flush=True)


if __name__ == "__main__":
    main()
