cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/bench_gemm.py > gpurun_out/bench_gemm_r04.txt 2>&1
tail -30 gpurun_out/bench_gemm_r04.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r4_bench_check.json 2> gpurun_out/r4_bench_check.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_check.json').read().strip().splitlines()[-1])
for e in d.get('extra_workloads', []):
    rr=e['roofline']
    print('   ', e['tag'], round(e['ms_per_step'],2), 'frac', round(rr['frac'],3), 'traffic', rr.get('traffic'), rr.get('traffic_source'))
PY
