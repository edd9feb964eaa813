cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_stage1_gpu.py -m gpu -q -k "bit_reproducible" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-200 | head -5; done
