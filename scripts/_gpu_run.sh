cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_fp32.txt 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/suite_fp32.txt | cut -c1-200
NU_MLP_DTYPE=bf16x6 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_x6.txt 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/suite_x6.txt | cut -c1-200
NU_MLP_DTYPE=bf16 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_bf16.txt 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/suite_bf16.txt | cut -c1-200 | head -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
true
