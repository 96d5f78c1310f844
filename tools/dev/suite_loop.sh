#!/bin/bash
# The whole GPU suite N times, one fresh process each (consecutive fresh-process passes); stops at
# the first failure.  usage: suite_loop.sh <runs> <logfile>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
n=$1; log=$2
cd $R
: > "$log"
for i in $(seq 1 "$n"); do
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/suite_run.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/suite_run.log)" | tee -a "$log"
  if [ $rc -ne 0 ]; then tail -40 gpurun_out/suite_run.log >> "$log"; break; fi
done
