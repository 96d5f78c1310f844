#!/bin/bash
# Round profile: kernel-trace stats of the default bench + PMC traffic passes + FETCH_SIZE calibration.
# usage (on the GPU box): tools/profile_round.sh r01
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py > $OUT/bench_stats.log 2>&1
echo "stats rc=$?"; tail -1 $OUT/bench_stats.log | cut -c1-400
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-secondary > $OUT/pmc_$n.log 2>&1
  echo "pmc $c rc=$?"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -- python $R/tools/calibrate_fetch.py > $OUT/cal_$c.log 2>&1
  echo "cal $c rc=$?"
done
python $R/tools/summarize_profile.py $OUT $TAG
# the raw traces are hundreds of MB (gpurun copies back at most 64 MiB): keep the summaries and the logs only
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum_TCC_MISS_sum $OUT/cal_FETCH_SIZE $OUT/cal_WRITE_SIZE
