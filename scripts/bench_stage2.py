"""Report aid: stage-2 training step (BASELINE.json configs[2]) on one GPU -- same as `python bench.py --workload stage2`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

if __name__ == '__main__':
    if '--workload' not in sys.argv:
        sys.argv += ['--workload', 'stage2']
    if '--steps' not in sys.argv:
        sys.argv += ['--steps', '5', '--warmup', '2']
    bench.main()
