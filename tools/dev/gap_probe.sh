#!/bin/bash
# How much of the benchmark step is the GPU idle?  Kernel trace of a short bench run; union of the kernel intervals of
# the last step against its wall time, and the largest gaps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_gap
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/bench.py --steps 2 --warmup 1 --no-configs --no-secondary --no-cpu-baseline > $OUT/bench.log 2>&1
echo "rc=$?"; tail -1 $OUT/bench.log | cut -c1-160
python - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
print("kernels", len(rows))
# the last step = the last 40 % of the trace (3 steps of equal length after the setup); find it via the big SpMM launches
sp = [r for r in rows if "spmm_vec4_kernel<32>" in r[2]]
per_step = 194
last = sp[-per_step:]
t0, t1 = last[0][0], rows[-1][1]
sel = [r for r in rows if r[0] >= t0]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]; gaps = []
for s, e, n, q in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print("window %.2f ms  GPU busy (union of kernels) %.2f ms  idle %.2f ms (%.1f %%)" % (wall / 1e6, busy / 1e6, (wall - busy) / 1e6, 100 * (wall - busy) / wall))
gaps.sort(reverse=True)
print("gaps: n=%d  >5us: %d  >20us: %d  sum of gaps >5us: %.2f ms" % (len(gaps), sum(g[0] > 5000 for g in gaps), sum(g[0] > 20000 for g in gaps), sum(g[0] for g in gaps if g[0] > 5000) / 1e6))
for g, n in gaps[:12]:
    print("   %.1f us before %s" % (g / 1e3, n[:70]))
# context of the two largest gaps: the five kernels on either side
ends = sorted(sel, key=lambda r: r[0])
for gi in range(2):
    g, n = gaps[gi]
    for i in range(1, len(ends)):
        if ends[i][2] == n and ends[i][0] - max(e[1] for e in ends[max(0, i - 40):i]) == g:
            print("--- gap %.1f us, %.1f ms into the step:" % (g / 1e3, (ends[i][0] - t0) / 1e6))
            for r in ends[max(0, i - 5):i + 5]:
                print("      %s%8.1f us  %s" % (">" if r is ends[i] else " ", (r[1] - r[0]) / 1e3, r[2][:90]))
            break
import collections, re
agg = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_, q_ in sel:
    if e_ > t1 - 3_000_000:                    # the trailing read-back after the last step
        continue
    key = re.sub(r"\(.*", "", n_)[-70:]
    agg[key][0] += 1; agg[key][1] += e_ - s_
print("kernels of the last step by total time:")
for k_, (c_, t_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("   %8.2f ms  %5d x  %s" % (t_ / 1e6, c_, k_))
tot = sum(e - s for s, e, n, q in sel)
print("sum of kernel durations in window %.2f ms (overlap = sum - busy = %.2f ms)" % (tot / 1e6, (tot - busy) / 1e6))
PY
rm -rf $OUT/trace
