cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_nets_gpu.py tests/test_core_parity_gpu.py tests/test_stage1_gpu.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
for x in 0 1; do
NU_TN_CUT=$x python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/cut_${x}.json 2> gpurun_out/cut_${x}.err
python - <<PY
import json
d=json.loads(open('gpurun_out/cut_${x}.json').read().strip().splitlines()[-1])
w=d['roofline']['wgrad']
print('cut',$x,round(d['ms_per_step'],2),'wgrad TF',round(w['achieved'],1),'launches',w['launches'],'avg us',round(w['avg_launch_us'],1),'share',round(w['time_share'],3), 'loss', d['config'].get('final_loss'))
PY
done
done
