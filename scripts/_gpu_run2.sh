cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sampler_ops_gpu.py -q 2>&1 | tail -15 > gpurun_out/r4_sampler_ops_tests.txt
cat gpurun_out/r4_sampler_ops_tests.txt
STEPS=20 WARMUP=5 bash scripts/collect_profiles.sh r04_final20
