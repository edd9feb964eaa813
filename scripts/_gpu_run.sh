cd $GRAFT_REPO_ROOT
STEPS=10 WARMUP=4 bash scripts/collect_profiles.sh r04_c4b --real-capture --rays 8192 --mlp-dtype bf16 2>&1 | tail -8
STEPS=12 WARMUP=4 bash scripts/collect_profiles.sh r04_x6 --mlp-dtype bf16x6 2>&1 | tail -8
