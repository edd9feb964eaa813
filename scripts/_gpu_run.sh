cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -1 > gpurun_out/r4_final_tests2.txt
NU_MLP_DTYPE=bf16x6 python -m pytest tests -q -m gpu 2>&1 | tail -1 >> gpurun_out/r4_final_tests2.txt
cat gpurun_out/r4_final_tests2.txt
