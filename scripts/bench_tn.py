"""Development aid: weight-gradient GEMM variants (NU_TN_V) at the shapes of the training step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from bench_gemm import tn
v = int(os.environ.get("NU_TN_V", "0"))
wpc = v if v else 2
print("NU_TN_V", v)
for (P, N1, N2, pairs) in ((131072, 256, 256, 1), (131072, 256, 256, 2), (524288, 256, 256, 1), (131072, 1024, 288, 1),
                           (131072, 256, 64, 1), (393216, 256, 96, 1), (524288, 128, 288, 1)):
    tiles = ((N1 + 127) // 128) * ((N2 + 127) // 128)
    for mult in (1, 2):
        S = max(1, min((P + 255) // 256, (256 * wpc * mult) // tiles))
        tn(P, N1, N2, pairs, S)
