"""Development aid: checksums of one bf16-storage training step (outputs and every gradient), to compare kernel variants
(NU_NT16_V=1|2, NU_TN_SCALAR=1) across processes -- the variants run the same arithmetic in the same order, so the checksums
must agree bit for bit."""
import hashlib
import sys

import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from test_bf16_gpu import _step  # noqa: E402

out, total, grads, net = _step(torch.device('cuda:0'), 'bf16', R=int(sys.argv[1]) if len(sys.argv) > 1 else 512)
h = hashlib.sha256(out['ray_rgb'].detach().cpu().numpy().tobytes()).hexdigest()[:12]
print('rgb', h, 'total', repr(total))
for pre in ('sdf_network.', 'outer_nerf.', 'color_network.'):
    hh = hashlib.sha256()
    for n in sorted(grads):
        if n.startswith(pre):
            hh.update(grads[n].cpu().numpy().tobytes())
    print(pre, hh.hexdigest()[:12])
