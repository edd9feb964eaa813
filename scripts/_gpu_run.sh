cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r04f > gpurun_out/r4f_collect.txt 2>&1
STEPS=20 WARMUP=5 bash scripts/collect_profiles.sh r04f_s20 >> gpurun_out/r4f_collect.txt 2>&1
STEPS=10 WARMUP=4 bash scripts/collect_profiles.sh r04f_c4 --real-capture --rays 8192 --mlp-dtype bf16 >> gpurun_out/r4f_collect.txt 2>&1
STEPS=12 WARMUP=4 bash scripts/collect_profiles.sh r04f_x6 --mlp-dtype bf16x6 >> gpurun_out/r4f_collect.txt 2>&1
STEPS=30 WARMUP=6 bash scripts/collect_profiles.sh r04f_512 --rays 512 >> gpurun_out/r4f_collect.txt 2>&1
tail -40 gpurun_out/r4f_collect.txt
