"""One shape of scripts/bench_nt6.py in one mode, for counter passes: python scripts/bench_nt6_one.py M N K epi mode reps"""
import sys, os
sys.argv, args = sys.argv[:1], sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import importlib.util
src = open(os.path.join(ROOT, 'scripts', 'bench_nt6.py')).read().split("\nfor (M, N, K, epi) in")[0]
ns = {'__file__': os.path.join(ROOT, 'scripts', 'bench_nt6.py'), '__name__': 'bench_nt6_defs'}
exec(compile(src, 'bench_nt6.py', 'exec'), ns)
M, N, K, epi = (int(x) for x in args[:4])
print(ns['run'](M, N, K, epi, args[4], int(args[5]) if len(args) > 5 else 10))
