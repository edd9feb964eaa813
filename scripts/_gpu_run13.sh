cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r04_last
STEPS=20 WARMUP=5 bash scripts/collect_profiles.sh r04_last20
STEPS=30 WARMUP=6 bash scripts/collect_profiles.sh r04_last_512 --rays 512
