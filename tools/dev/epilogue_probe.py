#!/usr/bin/env python3
"""What do the epilogue operands cost INSIDE the gathered SpMM, and what would the same array passes cost in a streaming
kernel?  (VERDICT r02 item 1 asked to move them out.)  C5 graph, hubs-first node order as the ODE block uses it; same
process, variants interleaved, medians."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import gcn_ode, ops  # noqa: E402
from graph_odenet_amd.synth import rmat_graph  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g0 = rmat_graph(20, 10_000_000, seed=0, device=dev)
    g = gcn_ode.tuned_graph(g0, 128)[0]
    gt = g.transpose()
    n, d = g.n_rows, 128
    X, Y, K1, K2, K3, OUT, OUT2, TMP = [torch.randn(n, d, device=dev) for _ in range(8)]
    bias = torch.randn(d, device=dev)
    nd = n * d * 4 / 1e9
    cases = {
        "spmm A plain (bias+relu)                 ": (lambda: ops.spmm(g, X, bias=bias, relu=True, out=OUT), 0),
        "spmm A + 4 pre-terms                     ": (lambda: ops.spmm(g, X, bias=bias, relu=True, out=OUT, alpha=0.1,
                                                                       pre_terms=[(1.0, Y), (0.1, K1), (0.2, K2), (0.3, K3)]), 4),
        "spmm A + 1 cot term + dZ out             ": (lambda: ops.spmm(g, X, bias=bias, relu=True, out=OUT, cot_terms=[(-1.0, Y)], out2=OUT2), 2),
        "spmm A + 4 cot terms + dZ out            ": (lambda: ops.spmm(g, X, bias=bias, relu=True, out=OUT,
                                                                       cot_terms=[(-1.0, Y), (0.1, K1), (0.2, K2), (0.3, K3)], out2=OUT2), 5),
        "spmm A^T plain                           ": (lambda: ops.spmm(gt, X, out=OUT), 0),
        "lincomb 4 terms -> 1 (streaming, 5 arrays)": (lambda: ops.lincomb_(TMP, [(1.0, Y), (0.1, K1), (0.2, K2), (0.3, K3)]), 5),
        "lincomb 1 term -> 1 (copy, 2 arrays)     ": (lambda: ops.lincomb_(TMP, [(1.0, Y)]), 2),
    }
    res = {k: [] for k in cases}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(8):
        for k, (fn, _) in cases.items():
            fn()
            ev[0].record()
            for _ in range(5):
                fn()
            ev[1].record()
            torch.cuda.synchronize()
            if rep:
                res[k].append(ev[0].elapsed_time(ev[1]) / 5)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    base = med["spmm A plain (bias+relu)                 "]
    for k, (fn, extra) in cases.items():
        line = "%s %.4f ms" % (k, med[k])
        if k.startswith("spmm A +"):
            line += "   extra %.4f ms for %d arrays = %.2f TB/s" % (med[k] - base, extra, extra * nd / (med[k] - base))
        if k.startswith("lincomb"):
            line += "   %.2f TB/s" % (extra * nd / med[k])
        print(line)


if __name__ == "__main__":
    main()
