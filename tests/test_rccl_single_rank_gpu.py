"""The collectives of the data-parallel step on the REAL transport, as far as a one-GPU box allows: one rank, backend "nccl"
(RCCL), in a child process (its process group and its HIP context stay out of the test process; two processes use the card,
well inside the box's limit).  No scaling is measured here or anywhere else in this repo: N > 1 on hardware is the driver's run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_group_runs_the_reducers_collectives_bit_exactly(gpu):
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env.pop('NU_MLP_DTYPE', None)         # the bit-identity claim is about the exact-fp32 two-stream default
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_rccl_single_rank_child.py')
    r = subprocess.run([sys.executable, child], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'RCCL_SINGLE_RANK_OK' in r.stdout, r.stdout[-2000:]
