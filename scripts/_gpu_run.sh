cd $GRAFT_REPO_ROOT
python scripts/bench_nt_row_blocks.py > gpurun_out/r4_nt_row_blocks.txt 2>&1
cat gpurun_out/r4_nt_row_blocks.txt
