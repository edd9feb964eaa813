cd $GRAFT_REPO_ROOT
for lib in libnunerf.so libnunerf_nops1.so libnunerf_nops2.so libnunerf_nops3.so libnunerf.so; do
  echo "== $lib"
  NU_NERF_LIB=$GRAFT_REPO_ROOT/nu_nerf_amd/$lib python scripts/bench_gemm.py 2>&1 | grep -E "^NT" | tail -13
done > gpurun_out/r4_nt_nops.txt 2>&1
cat gpurun_out/r4_nt_nops.txt
