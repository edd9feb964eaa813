set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_stage2_gpu.py tests/test_stage2_thick_gpu.py tests/test_stage2_ops_gpu.py tests/test_eval_gpu.py -x -q 2>&1 | tail -6
for i in 1 2; do
NU_S2_STACKS=0 NU_TN_BATCH=0 NU_NT_BATCH=0 python bench.py --workload stage2 --thick --rays 1024 --steps 30 --warmup 8 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_thick_off_$i.json 2>/dev/null
python bench.py --workload stage2 --thick --rays 1024 --steps 30 --warmup 8 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_thick_on_$i.json 2>/dev/null
done
NU_S2_STACKS=0 NU_TN_BATCH=0 NU_NT_BATCH=0 python bench.py --workload stage2 --rays 4096 --steps 12 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_s2_off.json 2>/dev/null
python bench.py --workload stage2 --rays 4096 --steps 12 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_s2_on.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_thick_o*.json')+glob.glob('gpurun_out/r4_s2_o*.json')):
    d=json.loads(open(f).read().strip().split('\n')[-1]); print(f, round(d['ms_per_step'],3), round(d['value']))
PY
