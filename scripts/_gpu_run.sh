cd $GRAFT_REPO_ROOT
python scripts/nt_store_locality.py > gpurun_out/r4_nt_store_locality.txt 2>&1
cat gpurun_out/r4_nt_store_locality.txt
