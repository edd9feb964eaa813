cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -2 > gpurun_out/r4_final_tests.txt
NU_MLP_DTYPE=bf16x6 python -m pytest tests -q -m gpu 2>&1 | tail -2 >> gpurun_out/r4_final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/r4_final_tests.txt
cat gpurun_out/r4_final_tests.txt
python bench.py > gpurun_out/r4_bench_default_final.json 2> gpurun_out/r4_bench_default_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_default_final.json').read().strip().splitlines()[-1])
r=d['roofline']
print('headline', round(d['ms_per_step'],2), round(d['value']), 'frac', round(r['frac'],4), 'traffic', r['traffic'], 'launches', r['launches'], 'wgrad', round(r['wgrad']['achieved'],1), r['wgrad']['traffic'])
for e in d.get('extra_workloads', []):
    rr=e['roofline']
    print('   ', e['tag'], round(e['ms_per_step'],2), round(e['rays_per_s']), 'frac', round(rr['frac'],3), 'traffic', rr.get('traffic'))
print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('spread_rays_per_s'))
PY
bash scripts/collect_profiles.sh r04g > gpurun_out/r4g_collect.txt 2>&1
STEPS=20 WARMUP=5 bash scripts/collect_profiles.sh r04g_s20 >> gpurun_out/r4g_collect.txt 2>&1
STEPS=10 WARMUP=4 bash scripts/collect_profiles.sh r04g_c4 --real-capture --rays 8192 --mlp-dtype bf16 >> gpurun_out/r4g_collect.txt 2>&1
STEPS=12 WARMUP=4 bash scripts/collect_profiles.sh r04g_x6 --mlp-dtype bf16x6 >> gpurun_out/r4g_collect.txt 2>&1
STEPS=30 WARMUP=6 bash scripts/collect_profiles.sh r04g_512 --rays 512 >> gpurun_out/r4g_collect.txt 2>&1
grep "total kernel" gpurun_out/r4g_collect.txt
