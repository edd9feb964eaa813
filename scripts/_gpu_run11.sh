cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_occ_idx_early_ab.txt; : > $O
one() { # tag env flags
  local tag=$1; local e=$2; shift 2
  env $e python bench.py --no-cpu-baseline --no-extra --no-kernel-timing "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', '$e', round(d['ms_per_step'],3),'ms/step', 'loss', d['config'].get('final_loss'))" >> $O
}
for rep in 1 2 3; do
  one headline NU_OCC_IDX_EARLY=0 --steps 20 --warmup 5
  one headline NU_OCC_IDX_EARLY=1 --steps 20 --warmup 5
  one rays512  NU_OCC_IDX_EARLY=0 --rays 512 --steps 40 --warmup 10
  one rays512  NU_OCC_IDX_EARLY=1 --rays 512 --steps 40 --warmup 10
done
one config4 NU_OCC_IDX_EARLY=0 --real-capture --rays 8192 --mlp-dtype bf16 --steps 10 --warmup 4
one config4 NU_OCC_IDX_EARLY=1 --real-capture --rays 8192 --mlp-dtype bf16 --steps 10 --warmup 4
one config4 NU_OCC_IDX_EARLY=0 --real-capture --rays 8192 --mlp-dtype bf16 --steps 10 --warmup 4
one config4 NU_OCC_IDX_EARLY=1 --real-capture --rays 8192 --mlp-dtype bf16 --steps 10 --warmup 4
cat $O
python -m pytest tests/test_stage1_gpu.py tests/test_core_parity_gpu.py tests/test_bf16_gpu.py -q 2>&1 | tail -2
