cd $GRAFT_REPO_ROOT
STEPS=10 WARMUP=4 bash scripts/collect_profiles.sh r04_c4 --real-capture --rays 8192 --mlp-dtype bf16 2>&1 | tail -6
STEPS=30 WARMUP=6 bash scripts/collect_profiles.sh r04_512 --rays 512 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pthick -- python3 $R/bench.py --workload stage2 --thick --rays 1024 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pthick.json 2> $R/gpurun_out/pthick.err
cd $R
python3 scripts/kstats.py gpurun_out/pthick 25 60 > gpurun_out/kstats_thick_b.txt
rm -rf gpurun_out/pthick
python scripts/launch_census.py thick 1024 > gpurun_out/census_thick_1024_r4b.txt 2>&1 || true
grep -n "torch-launched\|^\[" gpurun_out/census_thick_1024_r4b.txt | head
