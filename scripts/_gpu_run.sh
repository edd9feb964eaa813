cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_fp32.txt 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/suite_fp32.txt | cut -c1-200
NU_MLP_DTYPE=bf16x6 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_x6.txt 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/suite_x6.txt | cut -c1-200
true
