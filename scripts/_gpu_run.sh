cd $GRAFT_REPO_ROOT
NU_BENCH_DEVICE=0 NU_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4_rehearsal_n2.json 2> gpurun_out/r4_rehearsal_n2.err
tail -1 gpurun_out/r4_rehearsal_n2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], round(d['ms_per_step'],2), round(d['value']), d['scaling'], d['config'].get('parallelism'), d['config'].get('grad_all_reduce'))"
