cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver_style.json 2> gpurun_out/r4_bench_driver_style.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_driver_style.json').read().strip().splitlines()[-1])
r=d['roofline']
print('headline', round(d['ms_per_step'],2), round(d['value']), 'frac', round(r['frac'],4), 'traffic', r['traffic'], 'src', r.get('traffic_source'), 'wgrad', round(r['wgrad']['achieved'],1))
for e in d.get('extra_workloads', []):
    rr=e['roofline']
    print('   ', e['tag'], round(e['ms_per_step'],2), round(e['rays_per_s']), 'frac', round(rr['frac'],3), 'traffic', rr.get('traffic'))
PY
