cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_s2_uv_ab.txt; : > $O
one() { # tag flags
  local tag=$1; shift
  python bench.py --no-cpu-baseline --no-extra --no-kernel-timing "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3),'ms/step')" >> $O
}
for rep in 1 2 3; do
  cp scripts/_stage2_old.py.txt nu_nerf_amd/stage2.py
  one "old thick1024" --workload stage2 --thick --rays 1024 --steps 30 --warmup 8
  one "old config3  " --workload stage2 --rays 4096 --steps 12 --warmup 4
  cp scripts/_stage2_new.py.txt nu_nerf_amd/stage2.py
  one "new thick1024" --workload stage2 --thick --rays 1024 --steps 30 --warmup 8
  one "new config3  " --workload stage2 --rays 4096 --steps 12 --warmup 4
done
cat $O
python -m pytest tests/test_stage2_gpu.py tests/test_stage2_thick_gpu.py tests/test_stage2_ops_gpu.py tests/test_rccl_single_rank_gpu.py tests/test_core_parity_gpu.py -q 2>&1 | tail -2
