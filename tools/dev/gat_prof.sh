#!/bin/bash
# kernel-trace stats of tools/gat_bench.py (Citeseer ODE-GAT, launch-bound)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_gat
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/tools/gat_bench.py "$@" > $OUT/run.log 2>&1
echo "rc=$?"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" > $R/gpurun_out/gat_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats -- python tools/gat_bench.py ; total kernel time %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:36]:
    print("%-100s %7s calls %9.2f us avg %6.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat $R/gpurun_out/gat_kernel_stats.txt; grep -i "ms" $OUT/run.log | tail -10
rm -rf $OUT/stats
