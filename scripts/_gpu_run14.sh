cd $GRAFT_REPO_ROOT
NU_BENCH_DEVICE=0 NU_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4_rehearsal_n2_last.json 2> gpurun_out/r4_rehearsal_n2_last.err
echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_rehearsal_n2_last.json').read().strip().splitlines()[-1])
c=d['config']
print('n_gpus',d['n_gpus'],'ms/step',round(d['ms_per_step'],2),'rays/s',round(d['value']),'backend',c['collective_backend'],'all_reduce',c['grad_all_reduce'],'global_rays',c['global_rays'],'loss',c['final_loss'])
PY
