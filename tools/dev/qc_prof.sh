#!/bin/bash
# kernel-trace stats of the captured QC step (C4): where do the 2.9 ms of a replay go?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_qc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/tools/qc_bench.py --model ${1:-EdgeGCN_K_Sum} --no-cpu-baseline --steps 100 --warmup 20 > $OUT/run.log 2>&1
echo "rc=$?"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f over 120 steps = %.3f ms/step" % (tot / 1e6, tot / 1e6 / 120))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-90s %6s calls %9.1f us avg %6.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
