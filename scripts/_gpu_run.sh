cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scripts/bench_nt_a_residency.py 2>&1 | tail -8
