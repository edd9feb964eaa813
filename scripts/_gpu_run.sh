cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py tests/test_nets_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for x in 0 1; do
NU_TN_XCD=$x python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/xcd_${x}_$i.json 2> gpurun_out/xcd_${x}_$i.err
python - <<EOF
import json
d=json.loads(open('gpurun_out/xcd_${x}_$i.json').read().strip().splitlines()[-1])
print('xcd',$x,d['ms_per_step'],d['roofline']['wgrad']['achieved'],d['roofline']['wgrad']['avg_launch_us'],d['roofline']['wgrad']['launches'])
EOF
done
done
