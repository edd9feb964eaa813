cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for x in 0 1; do
NU_NT6=$x rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/px6_$x -- python3 $R/bench.py --mlp-dtype bf16x6 --steps 12 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/px6_$x.json 2> $R/gpurun_out/px6_$x.err
done
cd $R
for x in 0 1; do python3 scripts/kstats.py gpurun_out/px6_$x 16 40 > gpurun_out/kstats_x6_nt6_$x.txt; rm -rf gpurun_out/px6_$x; done
grep "gemm_nt" gpurun_out/kstats_x6_nt6_0.txt | head -12; echo; grep "gemm_nt" gpurun_out/kstats_x6_nt6_1.txt | head -12; tail -1 gpurun_out/kstats_x6_nt6_0.txt; tail -1 gpurun_out/kstats_x6_nt6_1.txt
