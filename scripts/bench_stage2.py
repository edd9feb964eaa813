"""Report aid: stage-2 training step (BASELINE.json configs[2]) on one GPU -- 4096 rays, icosphere(r=0.5) with 20480
faces standing in for the stage-1 mesh, 3 bounces, segment samples 256/128/256, fp32.  Prints ms/step and rays/s.
Not the headline bench (bench.py); numbers are quoted in DESIGN.md section 9."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--object-rays', action='store_true', help='aim every ray at the object (worst case: all 3 bounces)')
    args = ap.parse_args()
    from nu_nerf_amd.stage2 import Stage2Renderer
    from nu_nerf_amd.params import init_stage1_params, init_stage2_params
    from nu_nerf_amd.lbvh import icosphere
    from nu_nerf_amd.synthetic import make_rays, make_object_rays
    from nu_nerf_amd.loss import name2loss, total_loss
    dev = torch.device('cuda:0')
    s1 = init_stage1_params(6033)
    p2 = init_stage2_params(6033, 7044, {'sphere_direction': False})
    for k, v in s1.items():
        p2['stage1_network.' + k] = v
        p2['color_network.stage1_network.' + k] = v
    cfg = {'name': 's2', 'network': 'stage2', 'is_nerf': True, 'shader_config': {'sphere_direction': False, 'human_light': False},
           'eikonal_weight': 0.02, 'freeze_inv_s_step': 5000,
           'stage1_cfg': {'is_nerf': True, 'apply_occ_loss': True, 'occ_loss_step': 15000, 'freeze_inv_s_step': 15000},
           'stage1_mesh_arrays': icosphere(5, 0.5)}
    net = Stage2Renderer(cfg, training=False)
    net.load_param_dict(p2)
    net = net.to(dev)
    losses = [name2loss[n](cfg) for n in ('eikonal', 'std', 'nerf_render')]
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-3, fused=True)
    R = args.rays
    n = args.steps + args.warmup
    pool = (make_object_rays if args.object_rays else make_rays)(R * n, seed=6033)
    pool = {k: torch.from_numpy(v).to(dev) for k, v in pool.items() if k in ('rays_o', 'rays_d', 'rgbs')}
    hit = 0.0

    def step(i):
        nonlocal hit
        b = {k: v[i * R:(i + 1) * R] for k, v in pool.items()}
        opt.zero_grad(set_to_none=True)
        out = net.train_step_rays(b, 6000 + i)
        total, _ = total_loss(out, losses, 6000 + i)
        total.backward()
        opt.step()
        hit += float(len(out['_paths']) > 1 and out['_paths'][1].shape[0]) / R
        return total
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    hit = 0.0
    t0 = time.perf_counter()
    for i in range(args.warmup, n):
        last = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"workload": "stage-2 train step, %d rays, icosphere 20480 faces, 3 bounces, 256/128/256 samples, fp32" % R,
                      "rays": "object-aimed" if args.object_rays else "Spherepot-shaped cameras", "ms_per_step": 1e3 * dt,
                      "rays_per_s": R / dt, "frac_rays_entering_object": hit / args.steps, "final_loss": float(last.detach()),
                      "max_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == '__main__':
    main()
