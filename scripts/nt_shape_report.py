"""Development aid: per-shape time / TFLOP/s of the NT and TN GEMM launches inside one training step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import collections
import numpy as np
import torch
import bench
from nu_nerf_amd.renderer import NeROShapeRenderer
from nu_nerf_amd.params import init_stage1_params
from nu_nerf_amd.synthetic import make_rays
from nu_nerf_amd.loss import name2loss, SPHEREPOT_LOSSES, total_loss

dev = torch.device('cuda:0')
R = 4096
cfg = bench.build_cfg(R)
net = NeROShapeRenderer(cfg, training=False)
net.load_param_dict(init_stage1_params(6033))
net = net.to(dev)
losses = [name2loss[n](cfg) for n in SPHEREPOT_LOSSES]
pool = {k: torch.from_numpy(v).to(dev) for k, v in make_rays(R * 4, seed=6033).items() if k != 'idxs'}
eng = net.engine()
rec = []
orig_nt, orig_wg = eng.nt, eng._wgrad


def nt(A, lda, B, ldb, M, N, K, C, ldc, epi, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_nt(A, lda, B, ldb, M, N, K, C, ldc, epi, **kw); e1.record()
    g = kw.get('groups', 1)
    rec.append(('NT', M, N, K, epi, g, 2.0 * M * (kw.get('ntrue') or N) * (kw.get('ktrue') or K) * g, e0, e1))


def wg(A0, lda0, B0, ldb0, P, N1, N2, dW, ldw, db, A1, *rest):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_wg(A0, lda0, B0, ldb0, P, N1, N2, dW, ldw, db, A1, *rest); e1.record()
    groups = rest[3]
    rec.append(('TN', P, N1, N2, 2 if A1 else 1, groups, 2.0 * P * N1 * N2 * groups * (2 if A1 else 1), e0, e1))


def step(i):
    b = {k: v[i * R:(i + 1) * R] for k, v in pool.items()}
    net.zero_grad(set_to_none=True)
    out = net.train_step_rays(b, 20000 + i)
    total, _ = total_loss(out, losses, 20000 + i)
    total.backward()


step(0); step(1)
eng.nt, eng._wgrad = nt, wg
step(2)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, a, b_, c, d, g, fl, e0, e1 in rec:
    key = (kind, a, b_, c, d, g)
    t = e0.elapsed_time(e1)
    x = agg.setdefault(key, [0, 0.0, 0.0])
    x[0] += 1; x[1] += t; x[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"total GEMM ms {tot:.2f}  P_in {eng.last_ctx['P_in']} P_out {eng.last_ctx['P_out']}")
for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]} M/P={key[1]:7d} N={key[2]:4d} K/N2={key[3]:4d} epi/pairs={key[4]} g={key[5]}  x{n:2d}  {t:7.3f} ms  {fl/t/1e9:6.1f} TFLOP/s")
