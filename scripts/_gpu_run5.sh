cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -1 > gpurun_out/r4_last_tests.txt
cat gpurun_out/r4_last_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver_style_last.json 2> gpurun_out/r4_bench_driver_style_last.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_driver_style_last.json').read().strip().splitlines()[-1])
r=d['roofline']
print('driver-style', round(d['ms_per_step'],2), round(d['value']), 'frac', round(r['frac'],4), 'traffic', r['traffic'], 'launches', r['launches'], 'wgrad', round(r['wgrad']['achieved'],1), 'wgrad traffic', r['wgrad'].get('traffic'))
for e in d.get('extra_workloads', []):
    rr=e['roofline']
    print('   ', e['tag'], round(e['ms_per_step'],2), round(e['rays_per_s']), 'frac', round(rr['frac'],3), 'traffic', rr.get('traffic'))
print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('spread_rays_per_s'))
PY
