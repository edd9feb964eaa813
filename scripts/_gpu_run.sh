cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py tests/test_nets_gpu.py tests/test_bf16_gpu.py tests/test_stage1_gpu.py -x -q -m gpu 2>&1 | grep -E "FAILED|Error|passed|failed" | head
cat > /tmp/_p.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], round(d['ms_per_step'],2), 'ms  NT', round(r['achieved'],1), r['unit'], 'frac', round(r['frac'],3), ' wgrad', round(r['wgrad']['achieved'],1))
PY
run() { python bench.py --steps 20 --warmup 6 --no-extra --no-cpu-baseline $@ 2>/dev/null | python /tmp/_p.py "$*"; }
run; run --real-capture --rays 8192 --mlp-dtype bf16; run; run --real-capture --rays 8192 --mlp-dtype bf16
