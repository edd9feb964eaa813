cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -1 > gpurun_out/r4_final_tests5.txt
NU_MLP_DTYPE=bf16x6 python -m pytest tests -q -m gpu 2>&1 | tail -1 >> gpurun_out/r4_final_tests5.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r4_final_tests5.txt
cat gpurun_out/r4_final_tests5.txt
python bench.py > gpurun_out/r4_bench_default_final5.json 2> gpurun_out/r4_bench_default_final5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_default_final5.json').read().strip().splitlines()[-1])
r=d['roofline']
print('headline', round(d['ms_per_step'],2), round(d['value']), 'frac', round(r['frac'],4), 'traffic', r['traffic'], 'launches', r['launches'], 'wgrad', round(r['wgrad']['achieved'],1))
for e in d.get('extra_workloads', []):
    rr=e['roofline']
    print('   ', e['tag'], round(e['ms_per_step'],2), round(e['rays_per_s']), 'frac', round(rr['frac'],3), 'traffic', rr.get('traffic'))
print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('spread_rays_per_s'))
PY
