cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -m gpu -q -k "bf16_storage" 2>&1 | tail -2
for i in 1 2; do
for x in 0 1; do
NU_NT16_ROWWALK=$x python bench.py --real-capture --rays 8192 --mlp-dtype bf16 --steps 10 --warmup 4 --no-extra --no-cpu-baseline > gpurun_out/c4w_${x}_$i.json 2> gpurun_out/c4w_${x}_$i.err
python - <<PY
import json
d=json.loads(open('gpurun_out/c4w_${x}_$i.json').read().strip().splitlines()[-1])
r=d['roofline']
print('roww',$x,round(d['ms_per_step'],2),'NT GB/s',round(r['achieved'],1),r['launches'],round(r['avg_launch_us'],1),'frac',round(r['frac'],3),'wgrad',round(r['wgrad']['achieved'],1))
PY
done
done
