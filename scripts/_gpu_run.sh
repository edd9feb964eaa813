set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
NU_BENCH_DEVICE=0 NU_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4_rehearsal_n2.json 2> gpurun_out/r4_rehearsal_n2.err || (tail -20 gpurun_out/r4_rehearsal_n2.err; exit 1)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_rehearsal_n2.json').read().strip().split('\n')[-1])
print('N=2 rehearsal', d['n_gpus'], d['ms_per_step'], d['value'], d['config'].get('grad_all_reduce'), d['config'].get('loss_assembly'))
PY
