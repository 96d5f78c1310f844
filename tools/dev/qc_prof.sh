#!/bin/bash
# kernel-trace stats of the QC step (C4): tools/dev/qc_prof.sh <model> [qc_bench flags...]; prints the top kernels and
# writes gpurun_out/profiles_qc/<model>_kernel_stats.txt (raw traces are deleted: gpurun copies back at most 64 MiB)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
M=${1:-EdgeGCN_K_Sum}; shift
OUT=$R/gpurun_out/prof_qc
mkdir -p $OUT $R/gpurun_out/profiles_qc
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/tools/qc_bench.py --model $M --no-cpu-baseline --steps 100 --warmup 20 "$@" > $OUT/run.log 2>&1
echo "rc=$?"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" "$M" "$*" > $R/gpurun_out/profiles_qc/${M}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats -- python tools/qc_bench.py --model %s --steps 100 --warmup 20 %s" % (sys.argv[2], sys.argv[3]))
print("# total kernel time %.1f ms over 120 steps = %.3f ms/step" % (tot / 1e6, tot / 1e6 / 120))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%-86s %6s calls %9.1f us avg %6.1f%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "^{" $OUT/run.log | cut -c1-200 >> $R/gpurun_out/profiles_qc/${M}_kernel_stats.txt
cat $R/gpurun_out/profiles_qc/${M}_kernel_stats.txt
rm -rf $OUT/stats
