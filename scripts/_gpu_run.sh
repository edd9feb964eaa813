cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | grep -E "FAILED|Error|assert|passed|failed|mismatch" | head
