#!/bin/bash
# kernel-trace stats of `python bench.py` (the round's committed profile: profiles/rNN_bench_kernel_stats.txt)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "profiling bench.py ..."
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py > $OUT/bench.log 2> $OUT/bench.err
echo "rc=$?"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" "$OUT/bench.log" > $R/gpurun_out/bench_kernel_stats.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)[:52]
print("# kernel                                               calls       avg_us     total_ms     pct")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-52s %7s %12.1f %12.2f %6.1f%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
line = [l for l in open(sys.argv[2]) if l.startswith("{")]
print("# bench line of the same (profiled) run:")
print(line[-1][:700] if line else "# (no bench line)")
PY
cat $R/gpurun_out/bench_kernel_stats.txt | cut -c1-200
rm -rf $OUT/stats
