"""bench.py's multi-GPU launch is fail-closed: `--gpus N` either runs N ranks or exits non-zero -- it never prints a line
for fewer GPUs than asked (round-1 VERDICT: a bare `--gpus 8` ran on one GPU and printed "n_gpus": 1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NU_BENCH_DEVICE')}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_world_size_mismatch_is_refused():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode == 2 and 'refusing' in r.stderr and '"metric"' not in r.stdout


def test_bare_multi_gpu_request_without_enough_devices_is_refused():
    # this container shows no GPU; a 1-GPU box shows one: either way `--gpus 2` must not fall back to one device
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 GPUs: the bare launch would really run")
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], {})
    assert r.returncode == 2 and 'needs 2 visible GPUs' in r.stderr and '"metric"' not in r.stdout
