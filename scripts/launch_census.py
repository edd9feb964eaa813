"""Which ops the device kernels of one stage-2 step belong to: one step under torch.profiler, kernels attributed to the CPU op
(aten op or autograd Function) whose time range contains their launch -- library launches through ctypes count towards the
Function that made them.  (Python stacks are requested but this torch build returns none for these events; the by-site table
then has one row.)  Usage: python scripts/launch_census.py [thick|zero] [rays]"""
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from nu_nerf_amd.params import init_stage1_params, init_stage2_params, init_stage2_thick_own_params  # noqa: E402
from nu_nerf_amd.lbvh import icosphere  # noqa: E402
from nu_nerf_amd.synthetic import make_rays  # noqa: E402
from nu_nerf_amd.loss import name2loss, total_loss  # noqa: E402
from nu_nerf_amd.train_glue import FusedAdam  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'thick'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device('cuda:0')
s1 = init_stage1_params(6033)
cfg = {'name': 's2', 'network': 'stage2', 'is_nerf': True, 'shader_config': {'sphere_direction': False, 'human_light': False},
       'eikonal_weight': 0.02, 'freeze_inv_s_step': 5000, 'get_mask': False,
       'stage1_cfg': {'is_nerf': True, 'apply_occ_loss': True, 'occ_loss_step': 15000, 'freeze_inv_s_step': 15000, 'get_mask': False},
       'stage1_mesh_arrays': icosphere(5, 0.5)}
if which == 'thick':
    from nu_nerf_amd.stage2_thick import Stage2Renderer
    net = Stage2Renderer(cfg, training=False)
    net.load_param_dict(init_stage2_thick_own_params(7044, net.color_network_inner.cfg))
    net.load_param_dict({'stage1_network.' + k: v for k, v in s1.items()})
else:
    from nu_nerf_amd.stage2 import Stage2Renderer
    net = Stage2Renderer(cfg, training=False)
    p2 = init_stage2_params(6033, 7044, {'sphere_direction': False})
    for k, v in s1.items():
        p2['stage1_network.' + k] = v
        p2['color_network.stage1_network.' + k] = v
    net.load_param_dict(p2)
net = net.to(dev)
losses = [name2loss[n](cfg) for n in ('eikonal', 'std', 'nerf_render')]
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-3)
pool = {k: torch.from_numpy(v).to(dev) for k, v in make_rays(R * 4, seed=6033).items() if k in ('rays_o', 'rays_d', 'rgbs')}


def step(i):
    b = {k: v[i * R:(i + 1) * R] for k, v in pool.items()}
    opt.zero_grad(set_to_none=True)
    out = net.train_step_rays(b, 6000 + i)
    total, _ = total_loss(out, losses, 6000 + i)
    total.backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
import time  # noqa: E402
rows = []
for i in range(8):
    b = {k: v[(i % 4) * R:((i % 4) + 1) * R] for k, v in pool.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = net.train_step_rays(b, 6000 + i)
    t_f = time.perf_counter()
    torch.cuda.synchronize()
    t_fs = time.perf_counter()
    total, _ = total_loss(out, losses, 6000 + i)
    total.backward()
    opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((1e3 * (t_f - t0), 1e3 * (t_fs - t_f), 1e3 * (t1 - t_fs), 1e3 * (t2 - t1), 1e3 * (t2 - t0)))
a = np.array(rows[2:])
print("ms: forward on the host (incl. its device->host reads) | GPU still busy after it | loss + backward + Adam enqueue | GPU still busy "
      "after it | step (with the extra sync after the forward)")
print(np.round(a.mean(0), 2))
if len(sys.argv) > 3 and sys.argv[3] == 'cprofile':        # host time by Python function over 6 steps
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(6):
        step(i % 4)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(45)
    st.sort_stats('cumulative').print_stats(60)
    raise SystemExit(0)
if len(sys.argv) > 3 and sys.argv[3] == 'sites':           # which Python lines ask for zero-filled buffers / index lists (one step)
    sites = collections.Counter()

    def wrap(owner, name, tag):
        orig = getattr(owner, name)

        def f(*a, **k):
            fr = sys._getframe(1)
            while fr is not None and 'nu_nerf_amd' not in fr.f_code.co_filename:
                fr = fr.f_back
            if fr is not None:
                sites['%-12s %s:%d %s' % (tag, os.path.basename(fr.f_code.co_filename), fr.f_lineno, fr.f_code.co_name)] += 1
            return orig(*a, **k)
        setattr(owner, name, f)
    for nm in ('zeros', 'zeros_like', 'ones', 'ones_like', 'full', 'nonzero', 'arange', 'cat', 'where', 'index_select', 'gather'):
        wrap(torch, nm, nm)
    for nm in ('new_zeros', 'zero_', 'nonzero', 'item', 'tolist', 'index_select', 'index_copy', 'index_add_', 'index_add', 'fill_',
               'new_ones', 'new_full', 'clone', 'contiguous', '__getitem__', '__setitem__', 'gather', 'float', 'to'):
        wrap(torch.Tensor, nm, 'T.' + nm)
    step(3)
    torch.cuda.synchronize()
    for k, v in sorted(sites.items(), key=lambda kv: (kv[0].split()[0], -kv[1])):
        print('%4d  %s' % (v, k))
    raise SystemExit(0)
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
by_site, by_op = collections.Counter(), collections.Counter()
total = 0
for e in prof.events():
    nk = len(e.kernels)
    if nk == 0:
        continue
    total += nk
    site = 'autograd engine / no python frame'
    for fr in e.stack:
        m = re.search(r'(nu_nerf_amd/\w+\.py|bench\.py|launch_census\.py)\((\d+)\): (\w+)', fr)
        if m:
            site = '%s:%s %s' % m.groups()
            break
    by_site[site] += nk
    by_op[e.name] += nk
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
print("torch-launched device kernels in the step:", total)
print("---- by call site")
for k, v in by_site.most_common(45):
    print("%5d  %s" % (v, k))
print("---- by op")
for k, v in by_op.most_common(25):
    print("%5d  %s" % (v, k))
