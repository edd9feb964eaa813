import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from bench_gemm import nt
print("grid", os.environ.get("NU_NT_GRID"))
for M in (65536, 131072, 262144, 524288, 1048576):
    nt(M, 256, 256, 7)
nt(262144, 256, 256, 2)
