set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pc4 -- python3 $R/bench.py --real-capture --rays 8192 --mlp-dtype bf16 --steps 10 --warmup 4 --no-extra --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pc4.json 2> $R/gpurun_out/pc4.err
cd $R
python3 scripts/kstats.py gpurun_out/pc4 14 45 > gpurun_out/kstats_c4_tm64.txt
rm -rf gpurun_out/pc4
grep nt16b gpurun_out/kstats_c4_tm64.txt
