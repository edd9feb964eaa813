cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r04_final
STEPS=10 WARMUP=4 bash scripts/collect_profiles.sh r04_final_c4 --real-capture --rays 8192 --mlp-dtype bf16
STEPS=30 WARMUP=6 bash scripts/collect_profiles.sh r04_final_512 --rays 512
STEPS=12 WARMUP=4 bash scripts/collect_profiles.sh r04_final_x6 --mlp-dtype bf16x6
