"""bench.py -- headline benchmark: stage-1 training throughput (train rays/s, ms/iter) on a Spherepot-shaped
synthetic workload (BASELINE.json configs[1]: 4096 rays/batch, 64 coarse + 64 importance + 32 background
samples, fp32), one process per GPU, ray batches sharded data-parallel with an RCCL all-reduce of the
parameter gradients.

    python bench.py --gpus N --steps K --warmup W            (N>1: under torch.distributed.run, or bare -- it then starts
                                                              the N ranks itself and refuses to run on fewer GPUs)

A step = ray fetch (device-resident pool) -> sampler -> render_core forward -> loss -> backward -> [all-reduce]
-> Adam.  Rank 0 prints ONE JSON line.  Extra objects:
  extra_workloads : (default N = 1 run only) short legs of the other BASELINE.json configs, timed AFTER the headline's timed
                 region in the same process, each {workload, dtype, ms_per_step, rays_per_s, roofline{...}}: config 3 (stage 2,
                 4096 rays, and the non-zero-thickness model at its real batch of 1024), config 4 (real-capture path, 8192
                 rays, bf16 storage), the reference's default stage-1 batch of 512 rays, and the headline workload in the opt-in
                 `bf16x6` mode (fp32-equivalent products on the bf16 pipe).  The headline fields never change.
  roofline     : the fp32-MFMA GEMM kernels (gemm_nt_kernel*, the dominant kernel): algorithmic FLOPs of every
                 launch / summed launch durations (HIP events on the launch stream, on one step in the middle of
                 the timed region by default: the ~330 event pairs cost that step a few ms) vs the 157.3 TFLOP/s fp32-MFMA peak
                 of MI355X_MICROARCH.md.
  cpu_baseline : the CPU oracle (oracle/stage1_oracle.py, a port of the reference's PyTorch path) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
Rehearsal switches for a one-GPU box (never a multi-GPU number; the line says so in config.rehearsal / collective_backend):
  NU_BENCH_ONE_RANK_GROUP=1 under `torch.distributed.run --nproc-per-node 1`: the N > 1 step's collectives on a one-rank RCCL group;
  NU_BENCH_DEVICE=0 NU_BENCH_BACKEND=gloo with --nproc-per-node 2: two ranks time-sharing one card.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def build_cfg(rays, real_capture=False, mlp_dtype='fp32'):
    cfg = {'name': 'bench', 'network': 'shape', 'database_name': 'synthetic/0', 'is_nerf': True, 'apply_occ_loss': True,
           'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'eikonal_weight': 0.1, 'train_ray_num': rays,
           'n_samples': 64, 'n_importance': 64, 'n_bg_samples': 32, 'up_sample_steps': 4, 'mlp_dtype': mlp_dtype}
    if real_capture:   # configs/shape/real/*.yaml: no white background, near/far from the unit sphere, 144-d outer_light
        cfg.update(is_nerf=False, shader_config={'sphere_direction': True, 'human_light': False, 'light_exp_max': 5.0})
    return cfg


def warmup_cos_lr(step, end_warm=5000, end_iter=300000, lr=5e-4, alpha=0.05):
    if step < end_warm:
        return lr * step / end_warm
    prog = (step - end_warm) / (end_iter - end_warm)
    return lr * ((np.cos(np.pi * prog) + 1.0) * 0.5 * (1 - alpha) + alpha)


def cpu_baseline(rays_cpu, step0, budget_s=20.0):
    """Oracle (port of the reference PyTorch path) on host cores: full iterations incl. backward + Adam."""
    from oracle import stage1_oracle as O
    from nu_nerf_amd.params import init_stage1_params
    R = 128
    cfg = dict(O.DEFAULT_CFG)
    params = {}
    for k, v in init_stage1_params(6033).items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        if not k.endswith('FG_LUT') and not k.startswith('infinity') and '.iors.' not in k:
            t.requires_grad_(True)
        params[k] = t
    opt = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=1e-3)
    o, d, rgb = (torch.from_numpy(rays_cpu[k][:R]) for k in ('rays_o', 'rays_d', 'rgbs'))
    times = []
    t_start = time.time()
    it = 0
    while True:
        t0 = time.time()
        opt.zero_grad()
        total, _, _ = O.train_step(params, cfg, o, d, rgb, step0 + it)
        total.backward()
        opt.step()
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        it += 1
        if (time.time() - t_start > budget_s and len(times) >= 5) or len(times) >= 8:
            break
    # the MEDIAN of at least five iterations, with the spread: on a 128-thread host a single iteration moves by +-30 % (36 rays/s in
    # round 3 against 50 in rounds 1-2 came from the mean of five)
    ms = 1e3 * float(np.median(times))
    return {"value": R / (ms / 1e3), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "statistic": "median", "iterations": len(times),
            "spread_rays_per_s": [round(R / max(times), 1), round(R / min(times), 1)],
            "sample": f"{len(times)} full iterations (fwd+bwd+Adam) of {R} rays x 160 samples after 1 warm-up, median "
                      f"{ms:.0f} ms/iter (min {1e3 * min(times):.0f}, max {1e3 * max(times):.0f}), torch CPU fp32, "
                      f"os.cpu_count()={os.cpu_count()}"}


def spawn_ranks_or_die(args):
    """Fail-closed multi-GPU launch.  `--gpus N` must mean N ranks on N GPUs:
      * launched by torch.distributed.run (WORLD_SIZE set): WORLD_SIZE must equal --gpus, otherwise exit 2;
      * launched bare with --gpus N > 1: this process has made no GPU call yet, so it starts
        `python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>` as a CHILD process (no exec) and exits
        with its code; if the box shows fewer than N devices it exits 2 with a message instead of measuring one GPU.
    NU_BENCH_DEVICE / NU_BENCH_BACKEND=gloo (all ranks on one card) exist to rehearse the multi-process path on a one-GPU box."""
    world_env = os.environ.get('WORLD_SIZE')
    if world_env is not None:
        if int(world_env) != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to report a {args.gpus}-GPU number "
                  f"from {world_env} rank(s)", file=sys.stderr, flush=True)
            raise SystemExit(2)
        return
    if args.gpus == 1:
        return
    n_dev = torch.cuda.device_count()            # counting devices does not initialise the GPU runtime on this image
    rehearsal = 'NU_BENCH_DEVICE' in os.environ
    if n_dev < args.gpus and not rehearsal:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this box shows {n_dev}: refusing to run",
              file=sys.stderr, flush=True)
        raise SystemExit(2)
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


class SumRuleError(RuntimeError):
    """The per-launch GEMM times of a bracketed step add up to more than that step's wall time: the launches overlapped
    (two streams) or the event pairing broke -- the roofline of such a step is not evidence."""


def check_sum_rule(what, gemm_s, step_ms):
    if gemm_s * 1e3 > step_ms * 1.001:
        raise SumRuleError(f"{what}: NT + TN launch time {gemm_s * 1e3:.2f} ms exceeds the bracketed step's wall time {step_ms:.2f} ms")
    return {"gemm_ms_in_bracketed_step": gemm_s * 1e3, "bracketed_step_ms": step_ms, "ok": True}


def stage2_leg(args):
    """BASELINE.json configs[2]: stage-2 training step (refraction through the stage-1 mesh, learned IoR, 3 bounces) on ONE
    GPU: 4096 rays, icosphere(r=0.5) with 20480 faces standing in for the stage-1 mesh, segment samples 256/128/256.  A parity
    case of the build, not its headline: printed in the same JSON format on request (--workload stage2) and attached to the
    default run's line as an extra workload."""
    from nu_nerf_amd.stage2 import Stage2Renderer
    from nu_nerf_amd.params import init_stage1_params, init_stage2_params
    from nu_nerf_amd.lbvh import icosphere
    from nu_nerf_amd.synthetic import make_rays, make_object_rays
    from nu_nerf_amd.loss import name2loss, total_loss
    from nu_nerf_amd.train_glue import FusedAdam
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
    if args.mlp_dtype != 'fp32':
        if args.mlp_dtype != 'bf16x6':
            raise SystemExit("--workload stage2 runs the fp32 and bf16x6 MLP modes (bf16 storage is a stage-1 mode)")
        cfg['mlp_dtype'] = args.mlp_dtype
        cfg['stage1_cfg'] = dict(cfg['stage1_cfg'], mlp_dtype=args.mlp_dtype)
    if args.thick:
        # the non-zero-thickness model (network/renderer.py:907-2398; every configs/stage2/real/*.yaml): shell refraction with
        # curvature radius + thickness network, segment samples 64/128/64, SpecInner inner shading
        from nu_nerf_amd.stage2_thick import Stage2Renderer as ThickRenderer
        cfg.update({'get_mask': False, 'is_nerf': False})
        cfg['stage1_cfg'] = dict(cfg['stage1_cfg'], get_mask=False, is_nerf=False)
        net = ThickRenderer(cfg, training=False)
        from nu_nerf_amd.params import init_stage2_thick_own_params
        net.load_param_dict(init_stage2_thick_own_params(7044, net.color_network_inner.cfg))     # fixed seeds: the same step every run
        net.load_param_dict({'stage1_network.' + k: v for k, v in s1.items()})
    else:
        net = Stage2Renderer(cfg, training=False)
        net.load_param_dict(p2)
    net = net.to(dev)
    losses = [name2loss[n](cfg) for n in ('eikonal', 'std', 'nerf_render')]
    # lr 1e-5: the full optimizer step runs, but the light paths (which rays refract, how many samples fall inside the object) stay
    # what the initial weights give for the whole run -- with untrained networks and lr 1e-3 the IoR network drifts within a few
    # steps towards invalidating rays (masked out of the loss), and the work per step with it (scripts/stage2_thick_trajectory.py)
    S2_LR = 1e-5
    opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=S2_LR)
    R, n = args.rays, args.steps + args.warmup
    pool = (make_object_rays if args.object_rays else make_rays)(R * n, seed=6033)
    pool = {k: torch.from_numpy(v).to(dev) for k, v in pool.items() if k in ('rays_o', 'rays_d', 'rgbs')}
    entered = []

    def step(i):
        b = {k: v[i * R:(i + 1) * R] for k, v in pool.items()}
        opt.zero_grad(set_to_none=True)
        out = net.train_step_rays(b, 6000 + i)
        total, _ = total_loss(out, losses, 6000 + i)
        total.backward()
        opt.step()
        entered.append(float(len(out['_paths']) > 1 and out['_paths'][1].shape[0]) / R)
        return total
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    entered.clear()
    n1, n2 = net.nets()
    engines = (n1.eng, n2.eng)
    mid = args.warmup + args.steps // 2
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if not args.no_kernel_timing:           # the event pools are made OUTSIDE the timed region (2 x 2400 events + a device sync)
        for e in engines:
            e.begin_kernel_timing(reserve=2400)
            e.ktime_on = False
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.warmup, n):
        bracket = i == mid and not args.no_kernel_timing
        if bracket:
            # one bracketed step: per-launch HIP events on the GEMMs of both engines.  That step runs on ONE stream (the
            # two-stream modes of the renderer and of the engines are switched off for it): events of launches that overlap on
            # two streams would each include the other stream's work and their sum would exceed the step (check_sum_rule)
            saved = (net._TWO_STREAM_RAYS, [e._TWO_STREAM_SAMPLES for e in engines])
            net._TWO_STREAM_RAYS = 0
            for e in engines:
                e._TWO_STREAM_SAMPLES = 0
                e.ktime_on = True
        last = step(i)
        if bracket:
            net._TWO_STREAM_RAYS = saved[0]
            for e, v in zip(engines, saved[1]):
                e._TWO_STREAM_SAMPLES = v
                e.ktime_on = False
        step_ev[i - args.warmup + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    gc.enable()
    if not args.no_kernel_timing:
        kts = [e.end_kernel_timing() for e in engines]
    step_ms = np.array([step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)])
    res = {
        "metric": "train rays/sec", "value": R / dt, "unit": "rays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt, "median_ms_per_step": float(np.median(step_ms)), "p10_ms_per_step": float(np.percentile(step_ms, 10)),
        "p90_ms_per_step": float(np.percentile(step_ms, 90)), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.mlp_dtype == 'fp32' else "bf16x6 (exact 3-way split, fp32-equivalent)", "data": "synthetic",
        "config": {"workload": ("stage-2 train step, NON-zero-thickness model, %d rays, icosphere 20480 faces (HIP LBVH), 3 shell crossings, "
                                "segment samples 64/128/64, fp32" % R) if args.thick else
                               ("stage-2 train step, %d rays, icosphere 20480 faces (HIP LBVH), 3 bounces, segment samples "
                                "256/128/256, fp32 (BASELINE.json configs[2])" % R),
                   "rays": "object-aimed" if args.object_rays else "Spherepot-shaped cameras", "adam_lr": S2_LR,
                   "adam_lr_note": "fixed small lr: the light paths (and with them the work per step) stay those of the initial "
                                   "weights for the whole run -- a fixed-path measurement, not a training trajectory",
                   "frac_rays_entering_object": float(np.mean(entered)), "final_loss": float(last.detach()),
                   "max_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}}
    if not args.no_kernel_timing and args.mlp_dtype == 'fp32':
        fl = sum(k['flops'] for k in kts); sec = sum(k['seconds'] for k in kts); ln = sum(k['launches'] for k in kts)
        tf = fl / max(sec, 1e-12) / 1e12
        tn_sec = sum(k['tn_seconds'] for k in kts)
        res["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS,
                           "traffic": None, "kernel": "gemm_nt2_kernel<*> (all NT launches of the bracketed step, both engines, one stream)",
                           "launches": ln, "avg_launch_us": 1e6 * sec / max(ln, 1),
                           "algorithmic_flops_per_launch": fl / max(ln, 1),
                           "event_pool_exhausted": any(k['event_capacity_reached'] or k['python_event_pool_exhausted'] for k in kts),
                           "sum_rule": check_sum_rule("stage 2", sec + tn_sec, float(step_ms[mid - args.warmup])),
                           "wgrad": {"achieved": sum(k['tn_flops'] for k in kts) / max(tn_sec, 1e-12) / 1e12,
                                     "launches": sum(k['tn_launches'] for k in kts)}}
    # mesh tracing of this step's shape: the first-bounce rays of one batch against the scene's LBVH (latency-bound kernel;
    # algorithmic bytes = 24 B in + 8 B out per ray)
    b = {k: v[:R] for k, v in pool.items()}
    ray = torch.cat([b['rays_o'], torch.nn.functional.normalize(b['rays_d'], dim=-1)], 1).contiguous()
    net.scene.bvh.intersect(ray)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hit, _ = net.scene.bvh.intersect(ray)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    res["lbvh_trace"] = {"rays": R, "faces": int(net.scene.bvh.n_faces), "us_per_launch": us, "rays_per_s": R / us * 1e6,
                         "algorithmic_GBps": R * 32 / us / 1e3, "hit_fraction": float(hit.mean()),
                         "note": "closest hit with four lanes per ray over 4-wide records (lbvh_trace_quad_kernel); a launch lasts as long as its longest chain of dependent record fetches (profiles/r03/lbvh_bench.txt: 2.7-5.0 G rays/s at 2^20 rays)"}
    return res


def find_traffic_profile(R, world, real_capture, mlp_dtype, object_rays, h16, p_in, p_out, nt_launches=None):
    """HBM traffic per launch of the dominant kernel from a committed rocprofv3 PMC pass (FETCH_SIZE / WRITE_SIZE in separate
    --pmc runs, gfx950 corrections applied by scripts/summarize_pmc.py) -- but only when that pass was collected on THIS
    workload: same configuration and mean inner / outer point counts within 5 % of this run's; otherwise (None, None, None):
    a constant from another run is not a measurement of this one."""
    if world != 1 or object_rays:
        return None, None, None
    prof = os.path.join(ROOT, 'profiles')
    cands = []
    for rnd in sorted(os.listdir(prof), reverse=True) if os.path.isdir(prof) else []:
        for name in sorted(os.listdir(os.path.join(prof, rnd))) if os.path.isdir(os.path.join(prof, rnd)) else []:
            if name.startswith('traffic_pmc') and name.endswith('.json'):
                cands.append(os.path.join(prof, rnd, name))
    for f in cands:
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = d.get('workload_record')
        if not isinstance(w, dict):                 # files without the workload record (rounds 1-2) cannot be matched
            continue
        if (w.get('rays') != R or bool(w.get('real_capture')) != bool(real_capture) or w.get('mlp_dtype') != mlp_dtype
                or bool(w.get('bf16_storage', False)) != bool(h16)):
            continue
        pi, po = w.get('mean_inner_points', 0), w.get('mean_outer_points', 0)
        if pi <= 0 or po <= 0 or abs(pi - p_in) > 0.05 * pi or abs(po - p_out) > 0.05 * po:
            continue
        # bytes PER LAUNCH only carry over to a run with the same launch structure (round 4 merged launches: a round-3 pass
        # averaged over 120 NT launches per step says nothing about a step of 104)
        # this run counts a batch of problems launched back to back as ONE launch (its event pair); the pass counts kernels.  A pass
        # that recorded its own event-launch count is matched on that and its per-kernel bytes are rescaled to bytes per event launch
        lps, eps = w.get('nt_launches_per_step'), w.get('nt_event_launches_per_step')
        ref = eps if eps else lps
        if nt_launches is not None and (ref is None or abs(ref - nt_launches) > 0.03 * nt_launches):
            continue
        nt_b, tn_b = d.get('gemm_nt_kernel', {}).get('hbm_bytes_per_launch'), d.get('gemm_tn_kernel', {}).get('hbm_bytes_per_launch')
        if eps and lps and nt_b:
            nt_b *= lps / eps
        teps, tlps = w.get('tn_event_launches_per_step'), w.get('tn_launches_per_step')
        if teps and tlps and tn_b:
            tn_b *= tlps / teps
        return nt_b, tn_b, os.path.relpath(f, ROOT)
    return None, None, None


def stage1_leg(args, dist_ctx):
    """One timed run of the stage-1 training step with the configuration in `args` (rays, code path, MLP arithmetic).
    dist_ctx = (rank, world, dev, rccl_ranks, coll_backend).  Returns the result dict on rank 0, None elsewhere."""
    import torch.distributed as dist
    from nu_nerf_amd.renderer import NeROShapeRenderer
    from nu_nerf_amd.params import init_stage1_params
    from nu_nerf_amd.synthetic import make_rays, make_object_rays
    from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss, fused_stage1_loss
    from nu_nerf_amd.parallel import GradAllReducer
    from nu_nerf_amd.train_glue import FusedAdam
    rank, world, dev, rccl_ranks, coll_backend = dist_ctx

    R = args.rays
    cfg = build_cfg(R, args.real_capture, args.mlp_dtype)
    torch.manual_seed(6033)
    net = NeROShapeRenderer(cfg, training=False)
    net.load_param_dict(init_stage1_params(6033, sphere_direction=args.real_capture))       # identical replica on every rank
    net = net.to(dev)
    losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
    # Adam: the HIP multi-tensor kernel (nu_nerf_amd/train_glue.py; checked against torch.optim.Adam in tests/test_train_glue_gpu.py)
    opt = FusedAdam(net.parameters(), lr=1e-3) if os.environ.get('NU_BENCH_ADAM', 'hip') == 'hip' else torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    # NU_BENCH_ONE_RANK_GROUP=1 (a one-rank launch under torch.distributed.run): the rehearsal a one-GPU box allows of the N > 1 step on the
    # real transport -- the reducer issues its collectives on the one-rank RCCL group (weights exactly 1, mean over 1: same bits)
    one_rank_group = world == 1 and coll_backend is not None
    reducer = GradAllReducer(net, world, always_collective=one_rank_group) if (world > 1 or one_rank_group) else None

    # device-resident ray pool; every rank draws a disjoint slice of the same seeded permutation
    n_iter = args.steps + args.warmup
    # --object-rays: every ray aimed at the unit sphere's interior (inner-point share ~0.5, the survey's 8.1 TFLOP/iter case:
    # SURVEY 8(d)) instead of full camera frusta (share ~0.19: most image rays miss the sphere)
    # (always 16 batches per rank, whatever --steps is: the rays -- and with them the inner / outer point counts of the workload --
    # must not depend on how long a run is, or a short profiling pass would measure another workload than the timed line)
    pool = (make_object_rays if args.object_rays else make_rays)(R * world * 16, seed=6033)
    pool_dev = {k: torch.from_numpy(v).to(dev) for k, v in pool.items() if k != 'idxs'}

    def batch_for(it):
        b = (it % 16) * world + rank
        return {k: v[b * R:(b + 1) * R] for k, v in pool_dev.items()}

    eng = net.engine()
    stats = {'P_in': 0, 'P_out': 0}

    def one_step(it):
        step = args.start_step + it
        lr = warmup_cos_lr(step)
        for g in opt.param_groups:
            g['lr'] = lr
        opt.zero_grad(set_to_none=True)
        if not args.unfused_loss:
            # loss assembly on the HIP loss kernels; with a reducer the subset means take this rank's count ratios (n_local * world /
            # sum n, device scalars; the eikonal one INSIDE those kernels): the N > 1 step is the N = 1 step + two all-reduces
            total, _, _ = fused_stage1_loss(net, batch_for(it), step, losses, reducer=reducer)
        else:
            out = net.train_step_rays(batch_for(it), step)
            if reducer is not None:     # subset means over the union of all ranks' subsets (parallel.dp_weight_outputs)
                from nu_nerf_amd.parallel import dp_weight_outputs
                dp_weight_outputs(out, reducer, net)
            total, _ = total_loss(out, losses, step)
        total.backward()
        if reducer is not None:
            reducer.all_reduce()
        opt.step()
        stats['P_in'] += eng.last_ctx['P_in']
        stats['P_out'] += eng.last_ctx['P_out']
        return total

    for it in range(args.warmup):
        one_step(it)
    torch.cuda.synchronize()
    if coll_backend is not None:
        dist.barrier()
    torch.cuda.synchronize()
    stats['P_in'] = stats['P_out'] = 0
    bracketed = []
    if not args.no_kernel_timing:
        bracketed = [args.steps // 2] if args.time_every <= 0 else [i for i in range(args.steps) if i % args.time_every == 0]
        eng.begin_kernel_timing(reserve=2 * 360 * len(bracketed), py_reserve=2 * 40 * len(bracketed))
    gc.collect()
    gc.disable()          # no collector pause inside the timed region (a full collection stalls the launch thread)
    t0 = time.perf_counter()
    last = None
    timed_steps = 0
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_ev[0].record()
    seg_count = [torch.cuda.memory_stats(dev).get('segment.all.allocated', 0)]
    two_stream_default = eng._TWO_STREAM_SAMPLES
    for it in range(args.warmup, args.warmup + args.steps):
        eng.ktime_on = (not args.no_kernel_timing) and (it - args.warmup) in bracketed
        # a bracketed step runs on ONE stream: per-launch events of launches that overlap on two streams (the small-batch mode,
        # engine._TWO_STREAM_SAMPLES) would each include the other stream's work (check_sum_rule)
        eng._TWO_STREAM_SAMPLES = 0 if eng.ktime_on else two_stream_default
        timed_steps += int(eng.ktime_on)
        last = one_step(it)
        step_ev[it - args.warmup + 1].record()
        seg_count.append(torch.cuda.memory_stats(dev).get('segment.all.allocated', 0))
    eng._TWO_STREAM_SAMPLES = two_stream_default
    torch.cuda.synchronize()
    if coll_backend is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if coll_backend is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ktime = eng.end_kernel_timing() if not args.no_kernel_timing else None
    if rank != 0:
        return None

    ms = 1e3 * elapsed / args.steps
    value = R * world * args.steps / elapsed
    step_ms = np.array([step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)])   # rank 0's stream
    if args.real_capture or args.mlp_dtype != 'fp32':
        workload = ("stage-1 train step, %s code path, %d rays/GPU x (64 + 64 + 32) samples, %s MLP GEMMs (fp32 accumulate), "
                    "synthetic cameras (BASELINE.json configs[3] when --real-capture --rays 8192 --mlp-dtype bf16)"
                    % ("real-capture (is_nerf False, sphere_direction True)" if args.real_capture else "Spherepot", R,
                       args.mlp_dtype))
    else:
        workload = ("Spherepot-shaped stage-1 train step, %d rays/GPU x (64 coarse + 64 importance + 32 bg) "
                    "samples, fp32, synthetic cameras (BASELINE.json configs[1])" % R)
    p_in, p_out = stats['P_in'] / args.steps, stats['P_out'] / args.steps
    res = {
        "metric": "train rays/sec", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "median_ms_per_step": float(np.median(step_ms)),
        "p10_ms_per_step": float(np.percentile(step_ms, 10)), "p90_ms_per_step": float(np.percentile(step_ms, 90)),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x6": "bf16x6 (exact 3-way split, fp32-equivalent)"}[args.mlp_dtype],
        "data": "synthetic",
        "config": {"workload": workload,
                   "rays_per_gpu": R, "global_rays": R * world, "samples_per_ray": 160, "start_step": args.start_step,
                   "parallelism": "dp%d" % world, "rccl_ranks": rccl_ranks, "collective_backend": coll_backend,
                   "rehearsal": "one rank, collectives issued on the one-rank group (NU_BENCH_ONE_RANK_GROUP=1): not a multi-GPU number" if one_rank_group else None,
                   "rays": "object-aimed (make_object_rays)" if args.object_rays else "Spherepot-shaped camera frusta",
                   "loss_assembly": "eager torch registry (--unfused-loss)" if args.unfused_loss else
                   "HIP loss kernels (fused_stage1_loss%s)" % ("; eikonal point weight as a device scalar" if reducer is not None else ""),
                   "grad_all_reduce": None if reducer is None else
                   ("in place on the flat gradient buffer" if reducer.gathered_calls == 0 else
                    "gathered (%d of %d steps)" % (reducer.gathered_calls, reducer.gathered_calls + reducer.in_place_calls)),
                   "mean_inner_points": p_in, "mean_outer_points": p_out,
                   "final_loss": float(last.detach()),
                   "step_ms": [round(float(v), 2) for v in step_ms],
                   "new_device_segments_per_step": [seg_count[i + 1] - seg_count[i] for i in range(args.steps)]},
    }
    if ktime is not None:
        tf = ktime['flops'] / max(ktime['seconds'], 1e-12) / 1e12
        traffic, traffic_tn, traffic_src = find_traffic_profile(R, world, args.real_capture, args.mlp_dtype, args.object_rays,
                                                                eng.h16, p_in, p_out, nt_launches=ktime['launches'] / max(timed_steps, 1))
        br_ms = float(np.sum(step_ms[bracketed]))
        common = {"algorithmic_bytes_per_launch": ktime['bytes'] / max(ktime['launches'], 1),
                  "algorithmic_flops_per_launch": ktime['flops'] / max(ktime['launches'], 1),
                  "launches": ktime['launches'], "avg_launch_us": 1e6 * ktime['seconds'] / max(ktime['launches'], 1),
                  "event_timed_steps": timed_steps,
                  "event_pool_exhausted": bool(ktime['event_capacity_reached'] or ktime['python_event_pool_exhausted']),
                  "gemm_time_share": ktime['seconds'] / (elapsed * timed_steps / args.steps),
                  "traffic_source": traffic_src if traffic is not None else
                  "null: no committed PMC pass matches this run's configuration, launch structure and mean point counts within 5 %",
                  "wgrad": {"achieved": ktime['tn_flops'] / max(ktime['tn_seconds'], 1e-12) / 1e12,
                            "launches": ktime['tn_launches'],
                            "algorithmic_bytes_per_launch": ktime['tn_bytes'] / max(ktime['tn_launches'], 1),
                            "avg_launch_us": 1e6 * ktime['tn_seconds'] / max(ktime['tn_launches'], 1),
                            "traffic": traffic_tn,
                            "time_share": ktime['tn_seconds'] / (elapsed * timed_steps / args.steps)}}
        try:
            common["sum_rule"] = check_sum_rule("stage 1", ktime['seconds'] + ktime['tn_seconds'], br_ms)
        except SumRuleError as e:
            res["roofline"] = None
            res["roofline_error"] = str(e)
            return res
        if args.mlp_dtype == 'bf16x6':
            # six bf16 MFMAs per 16-deep k-step: price the achieved rate against the bf16 pipe doing 6x the arithmetic
            res["roofline"] = {"bound": "mfma", "achieved": 6 * tf, "peak": 2516.6, "unit": "TFLOP/s (bf16 MFMA issued)",
                               "frac": 6 * tf / 2516.6, "traffic": traffic, "fp32_equivalent_tflops": tf,
                               "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, gfx950 corrections applied)",
                               "kernel": "gemm_nt6_kernel<*> (weights pre-split by the pack launch) / gemm_nt_kernel<*, split> below 320 tiles "
                                         "(6 x v_mfma_f32_32x32x16_bf16 per k-step)", **common}
        elif args.mlp_dtype != 'fp32':
            # bf16 build: 16x the fp32 MFMA rate, so the GEMMs are bound by streaming their operands; the algorithmic bytes
            # count each matrix at the width it is stored in (bf16 weight tables and hidden activations, fp32 elsewhere)
            gbs = ktime['bytes'] / max(ktime['seconds'], 1e-12) / 1e9
            stored = eng.h16
            res["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                               "traffic": traffic, "mfma_tflops": tf,
                               "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, gfx950 corrections applied)",
                               "kernel": ("gemm_nt16_kernel<*> (v_mfma_f32_32x32x16_bf16, bf16 weights and hidden activations in HBM)"
                                          if stored else "gemm_nt_kernel<*, bf16> (v_mfma_f32_32x32x16_bf16, fp32 operands in HBM)"),
                               **common}
        else:
            res["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, gfx950 corrections applied)",
                               "kernel": "gemm_nt2_kernel<*> (fp32 v_mfma_f32_32x32x2_f32, software-pipelined, 2 LDS stages)", **common}
    res["_pool"] = pool
    return res


EXTRA_LEGS = (
    # (tag, workload function, overrides of the command-line namespace)
    ("config3_stage2", 'stage2', dict(rays=4096, steps=10, warmup=4, thick=False)),
    ("stage2_thick_1024", 'stage2', dict(rays=1024, steps=20, warmup=5, thick=True)),
    ("config4_bf16", 'stage1', dict(rays=8192, steps=10, warmup=4, real_capture=True, mlp_dtype='bf16')),
    ("stage1_512rays", 'stage1', dict(rays=512, steps=30, warmup=6)),
    # the headline workload in the opt-in split-product mode (dtype says what it is): reported BESIDE the exact-fp32 headline, never instead
    ("headline_workload_bf16x6", 'stage1', dict(rays=4096, steps=12, warmup=4, mlp_dtype='bf16x6')),
)


def run_extra_legs(args, dist_ctx):
    """Short legs of the non-headline configs, each >= 10 steps after >= 4 warm-ups, run after the headline's timed region."""
    out = []
    for tag, kind, over in EXTRA_LEGS:
        a = argparse.Namespace(**vars(args))
        a.object_rays, a.real_capture, a.mlp_dtype, a.thick, a.unfused_loss, a.time_every, a.no_kernel_timing = False, False, 'fp32', False, False, 0, False
        a.start_step = 20000
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            r = stage2_leg(a) if kind == 'stage2' else stage1_leg(a, dist_ctx)
            r.pop('_pool', None)
            leg = {"tag": tag, "workload": r['config']['workload'], "dtype": r['dtype'], "steps": a.steps, "warmup": a.warmup,
                   "ms_per_step": r['ms_per_step'], "median_ms_per_step": r['median_ms_per_step'], "rays_per_s": r['value'],
                   "roofline": r.get('roofline'), "leg_wall_s": None}
            for k in ('frac_rays_entering_object', 'adam_lr', 'mean_inner_points', 'mean_outer_points', 'final_loss'):
                if k in r['config']:
                    leg[k] = r['config'][k]
            if 'roofline_error' in r:
                leg['roofline_error'] = r['roofline_error']
            if 'lbvh_trace' in r:
                leg['lbvh_trace'] = r['lbvh_trace']
        except Exception as e:          # a failing leg must not cost the headline its line: it is reported as failed
            leg = {"tag": tag, "error": "%s: %s" % (type(e).__name__, e)}
        leg["leg_wall_s"] = time.perf_counter() - t0
        out.append(leg)
        gc.collect()
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--workload', default='stage1', choices=['stage1', 'stage2'],
                    help="'stage1' = the headline (BASELINE configs[1]); 'stage2' = configs[2] on one GPU (a parity case, see DESIGN 9)")
    ap.add_argument('--thick', action='store_true', help="with --workload stage2: the non-zero-thickness model (nu_nerf_amd/stage2_thick.py)")
    ap.add_argument('--object-rays', action='store_true',
                    help='aim every ray at the object: stage 1 -> inner-point share ~0.5 (the shading stack dominates); stage 2 -> all three bounces')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--rays', type=int, default=4096, help='rays per GPU per step')
    ap.add_argument('--start-step', type=int, default=20000, help='training-step index of the first iteration')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true',
                    help="headline line only: skip the extra_workloads legs (configs 3 / 4 and the 512-ray batch) of the default N = 1 run")
    ap.add_argument('--mlp-dtype', default='fp32', choices=['fp32', 'bf16', 'bf16x6'],
                    help="'bf16' = BASELINE config 4's MLP arithmetic (not the headline: the reference computes in fp32)")
    ap.add_argument('--real-capture', action='store_true',
                    help='real-capture code path (is_nerf False, sphere_direction True): BASELINE config 4 with --rays 8192 --mlp-dtype bf16')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--unfused-loss', action='store_true', help='assemble the loss with the eager torch registry (development A/B)')
    ap.add_argument('--time-every', type=int, default=0,
                    help='bracket the GEMM launches with HIP events on every Nth step of the timed region; 0 (default): on ONE '
                         'step in the middle of it (the ~330 event pairs cost that step 2-4 ms)')
    args = ap.parse_args()
    if args.workload == 'stage2':
        if int(os.environ.get('WORLD_SIZE', 1)) != 1 or args.gpus != 1:
            raise SystemExit("--workload stage2 runs on one GPU")
        print(json.dumps(stage2_leg(args)), flush=True)
        return

    spawn_ranks_or_die(args)        # --gpus N without a launcher: start the N ranks ourselves (never returns in the parent)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    import torch.distributed as dist
    # NU_BENCH_DEVICE / NU_BENCH_BACKEND exist only to rehearse the multi-process path on a one-GPU box (gloo, all ranks on
    # cuda:0); the driver's runs use one GPU per rank and RCCL ("nccl").
    dev_index = int(os.environ.get('NU_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    grouped = world > 1 or (os.environ.get('NU_BENCH_ONE_RANK_GROUP') == '1' and 'WORLD_SIZE' in os.environ)
    if grouped:
        backend = os.environ.get('NU_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    rccl_ranks = dist.get_world_size() if grouped else 1
    if rccl_ranks != args.gpus:
        raise SystemExit(f"bench.py: process group has {rccl_ranks} rank(s), --gpus {args.gpus}")
    coll_backend = dist.get_backend() if grouped else None
    dist_ctx = (rank, world, dev, rccl_ranks, coll_backend)

    res = stage1_leg(args, dist_ctx)
    if rank == 0:
        pool = res.pop('_pool')
        headline = (args.rays == 4096 and args.mlp_dtype == 'fp32' and not args.real_capture and not args.object_rays
                    and not args.unfused_loss)
        if world == 1 and not grouped and headline and not args.no_extra:
            res["extra_workloads"] = run_extra_legs(args, dist_ctx)
        if world == 1 and not args.no_cpu_baseline and not args.real_capture and args.mlp_dtype == 'fp32':
            res["cpu_baseline"] = cpu_baseline(pool, args.start_step)
        print(json.dumps(res), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
