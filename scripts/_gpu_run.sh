cd $GRAFT_REPO_ROOT
timeout -k 10 400 python scripts/determinism_valu_victim.py 2>&1 | grep -v "first differing\|dy of that" | tail -8 | cut -c1-190
