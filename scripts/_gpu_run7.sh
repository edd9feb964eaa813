cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > gpurun_out/r4_bench_torchrun_n1.json 2> gpurun_out/r4_bench_torchrun_n1.err
echo rc=$?
tail -c 600 gpurun_out/r4_bench_torchrun_n1.json
