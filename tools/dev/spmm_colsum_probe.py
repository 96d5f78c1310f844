#!/usr/bin/env python3
"""What do the per-block column sums cost inside the SpMM that writes the masked cotangent (C5 size: R-MAT 2^20 nodes, d = 128)?
The same launch with and without `out2_colsum`, interleaved, HIP-event timed; and the column-sum pass over dZ it replaces."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops
from graph_odenet_amd.synth import rmat_graph

D = torch.device("cuda:0")
g = rmat_graph(20, 10_000_000, seed=0, device=D)
n, d = g.n_rows, 128
gen = torch.Generator(device=D).manual_seed(0)
X, b = torch.randn(n, d, device=D, generator=gen), torch.randn(d, device=D, generator=gen)
a0, a1 = torch.randn(n, d, device=D, generator=gen), torch.randn(n, d, device=D, generator=gen)
cot = [(-1.0, a0), (0.25, a1)]
out, out2 = torch.empty(n, d, device=D), torch.empty(n, d, device=D)
part = torch.empty(ops.spmm_y2_colsum_rows(g, d), d, device=D)
db = torch.empty(d, device=D)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


for rep in range(3):
    t_plain = timed(lambda: ops.spmm(g, X, bias=b, relu=True, out=out, cot_terms=cot, out2=out2))
    t_sums = timed(lambda: ops.spmm(g, X, bias=b, relu=True, out=out, cot_terms=cot, out2=out2, out2_colsum=part))
    t_cs_parts = timed(lambda: ops.colsum_(db, part))
    t_cs_full = timed(lambda: ops.colsum_(db, out2))
    print("SpMM + masked cotangent %7.1f us   with per-block column sums %7.1f us (+%.1f)   column sums of the %d partial rows %6.1f us   "
          "of dZ itself %6.1f us" % (t_plain, t_sums, t_sums - t_plain, part.shape[0], t_cs_parts, t_cs_full), flush=True)
