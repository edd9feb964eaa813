#!/bin/bash
# Collect the round's rocprofv3 evidence for the headline workload on the GPU box (run from the repo root):
#   kernel stats (--kernel-trace --stats), HBM traffic (FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes, as
#   MI355X_MICROARCH.md prescribes) and the matrix-pipe counters, each summarised into gpurun_out/<tag>/.
# usage: [STEPS=50 WARMUP=10] scripts/collect_profiles.sh <tag> [extra bench.py flags]
# (default: bench.py's own default step counts, so that the point counts recorded with the traffic match a default run's)
set -u
TAG=${1:-prof}; shift || true
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-50}; WARMUP=${WARMUP:-10}
B="$R/bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $B --no-kernel-timing > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $B --no-kernel-timing > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/mfma -- python3 $B --no-kernel-timing > $OUT/bench_mfma.json 2> $OUT/mfma.err
cd $R
python3 scripts/kstats.py $OUT/stats $((STEPS + WARMUP)) 40 > $OUT/kstats.txt 2>&1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 scripts/summarize_pmc.py $(ls $OUT/fetch/*/*counter_collection.csv | head -1) $(ls $OUT/write/*/*counter_collection.csv | head -1) $OUT/traffic_pmc.json $OUT/bench_fetch.json $OUT/bench_stats.json > $OUT/traffic.txt 2>&1
python3 scripts/summarize_mfma_pmc.py $(ls $OUT/mfma/*/*counter_collection.csv | head -1) $(ls $OUT/mfma/*/*kernel_trace.csv | head -1) $OUT/mfma_pmc.json > $OUT/mfma.txt 2>&1
rm -rf $OUT/fetch $OUT/write $OUT/mfma $OUT/stats
tail -3 $OUT/kstats.txt; tail -5 $OUT/traffic.txt; tail -5 $OUT/mfma.txt
