#!/usr/bin/env python3
"""A/B of libgraphode's run-time options on the contract benchmark: runs bench.py in this process's children is not
possible on the GPU boxes (no exec after HIP init), so this script is a driver that only parses: it is given the JSON
lines of several `GODE_OVERLAP=x python bench.py ...` runs on stdin and prints the comparison."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        if line.startswith("##"):
            print(line)
        continue
    d = json.loads(line)
    k = (d.get("roofline_dense") or {}).get("kernels", {})
    print("  %.4f steps/s  %.1f ms/step  spmm %.3f ms  dense %s" % (
        d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"],
        {n.split("_kernel")[0].replace("gn_gemm_", ""): v["avg_launch_ms"] for n, v in k.items()}))
