"""Development aid: stage-2 golden step -- exit directions / far-ray nodes of segment 2 against the fixture's path2."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from helpers import golden
from test_stage2_gpu import build

gpu = torch.device('cuda:0')
g = golden("stage2_step6000_r24.npz")
net, cfg = build(gpu)
batch = {k: torch.from_numpy(g[k]).to(gpu) for k in ('rays_o', 'rays_d', 'rgbs')}
out = net.train_step_rays(batch, int(g['step']))


def dirn(P):
    d = P[:, -1] - P[:, 0]
    return d / np.linalg.norm(d, axis=1, keepdims=True)


for b in range(3):
    P, G = out['_paths'][b].detach().cpu().numpy(), g['path%d' % b]
    print('segment', b, P.shape, G.shape)
    if P.shape == G.shape:
        print('  cos chord dir', (dirn(P) * dirn(G)).sum(1))
        print('  start diff', np.abs(P[:, 0] - G[:, 0]).max(1))
        print('  end diff', np.abs(P[:, -1] - G[:, -1]).max(1))
    D = out['_directions'][b].detach().cpu().numpy()
    print('  cos dir vs golden chord', (D * dirn(G)).sum(1) if D.shape[0] == G.shape[0] else D.shape)
