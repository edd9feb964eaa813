set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_core_parity_gpu.py -x -q -k "two_ray_shards" 2>&1 | tail -15
python scripts/launch_census.py thick 1024 > gpurun_out/census_thick_1024_r4a.txt 2>&1 || true
tail -40 gpurun_out/census_thick_1024_r4a.txt
