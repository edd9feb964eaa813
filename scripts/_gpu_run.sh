cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r4_bench_default_final2.json 2> gpurun_out/r4_bench_default_final2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_default_final2.json').read().strip().splitlines()[-1])
r=d['roofline']
print('headline', round(d['ms_per_step'],2), round(d['value']), 'frac', round(r['frac'],4), 'traffic', r['traffic'], 'launches', r['launches'], 'wgrad', round(r['wgrad']['achieved'],1), r['wgrad']['traffic'])
for e in d.get('extra_workloads', []):
    rr=e['roofline']
    print('   ', e['tag'], round(e['ms_per_step'],2), round(e['rays_per_s']), 'frac', round(rr['frac'],3), 'traffic', rr.get('traffic'))
print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('spread_rays_per_s'))
PY
