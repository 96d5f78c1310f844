#!/bin/bash
# kernel-trace stats of the Pubmed dense-paper dopri5 step (config C2)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_pubmed
mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys, torch, json
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import config_bench as cb
print(json.dumps(next(cb.c2_pubmed(torch.device("cuda:0"), cpu=False))))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $OUT/run.py > $OUT/run.log 2>&1
echo "rc=$?"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" > $R/gpurun_out/pubmed_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats -- config C2 (Pubmed, GCN-dense-paper ODEGCN3, dopri5): 5 warm-up + 20 timed steps; total kernel time %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print("%-100s %7s calls %9.2f us avg %6.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat $R/gpurun_out/pubmed_kernel_stats.txt; grep "ms_per_step" $OUT/run.log
rm -rf $OUT/stats
