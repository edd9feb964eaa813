set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_stage2_gpu.py tests/test_stage1_gpu.py -x -q 2>&1 | tail -6
python bench.py > gpurun_out/r4_bench_default_a.json 2> gpurun_out/r4_bench_default_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_default_a.json').read().strip().split('\n')[-1])
print('headline', d['ms_per_step'], d['value'], d['roofline'])
for e in d.get('extra_workloads',[]): print(e['workload'], e['ms_per_step'], e.get('rays_per_s'), e['roofline'].get('frac'))
print(d.get('cpu_baseline'))
PY
