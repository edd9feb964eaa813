cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_nets_gpu.py -m gpu -q -k "presplit" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-200 | head
