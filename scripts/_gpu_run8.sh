cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for g in 0 1; do
NU_BENCH_ONE_RANK_GROUP=$g timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$g bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r4_one_rank_group_$g.json 2> gpurun_out/r4_one_rank_group_$g.err || { echo FAILED $g; tail -20 gpurun_out/r4_one_rank_group_$g.err; exit 1; }
python - $g <<'PY'
import json,sys
g=sys.argv[1]
d=json.loads(open(f'gpurun_out/r4_one_rank_group_{g}.json').read().strip().splitlines()[-1])
c=d['config']
print('group',g,'ms/step',round(d['ms_per_step'],3),'rays/s',round(d['value']),'backend',c['collective_backend'],'all_reduce',c['grad_all_reduce'],'rehearsal',bool(c.get('rehearsal')),'loss',c['final_loss'], 'frac', round(d['roofline']['frac'],4))
PY
done
done
