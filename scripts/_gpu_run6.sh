cd $GRAFT_REPO_ROOT
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 timeout -k 10 240 python tests/_rccl_single_rank_child.py > gpurun_out/r4_rccl_single_rank.txt 2>&1
echo "child rc=$?" >> gpurun_out/r4_rccl_single_rank.txt
tail -5 gpurun_out/r4_rccl_single_rank.txt
timeout -k 10 400 python -m pytest tests/test_rccl_single_rank_gpu.py tests/test_core_parity_gpu.py tests/test_train_glue_gpu.py -q 2>&1 | tail -3
