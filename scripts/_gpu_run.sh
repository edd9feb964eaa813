cd $GRAFT_REPO_ROOT
echo "== determinism bf16x6 (one stream now)"; timeout -k 10 120 python scripts/determinism_probe3.py bf16x6 512 7 2>&1 | cut -c1-120 | grep "^rep" | tail -5
for i in 1 2 3; do NU_MLP_DTYPE=bf16x6 timeout -k 10 300 python -m pytest tests/test_core_parity_gpu.py -m gpu -q -k "fused_loss_kernels_equal" 2>&1 | grep -E "^E   *Assert|passed|failed" | cut -c1-160 | head -3; done
NU_MLP_DTYPE=bf16x6 timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
