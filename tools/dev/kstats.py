#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats output dir.  usage: kstats.py DIR [N]"""
import csv
import glob
import sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print("total %.2f ms over %d calls" % (tot / 1e6, calls))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:n]:
        print("%-90s %6d %9.1f us %8.2f ms %5.1f%%" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                     float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
