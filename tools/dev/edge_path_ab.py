#!/usr/bin/env python3
"""QC message step M = Etgt bmm(A, x[Esrc]) on a 20-molecule mini-batch: the per-target kernel (a block walks a target's
edges one after the other) against a block per edge + the per-target sum as an SpMM.  Forward only, HIP-event timed."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops, qc_layers
from graph_odenet_amd.synth import qm9_like_batch

D = torch.device("cuda:0")
h = 73
for seed in (0, 1, 2):
    x0, ef, Esrc, Etgt, batch = qm9_like_batch(20, seed=seed, device=D)
    es = qc_layers._EdgeSet(Esrc, Etgt)
    E, n = Esrc.numel(), x0.shape[0]
    deg = torch.bincount(Etgt.argmax(0), minlength=n)
    g = torch.Generator(device=D).manual_seed(seed)
    A = torch.randn(E, h, h, device=D, generator=g) / h ** 0.5
    x = torch.randn(n, h, device=D, generator=g)
    res = {}
    for name, thr in (("per-target kernel", 1 << 30), ("block per edge + SpMM", 0)):
        ops.EDGE_MSG_MIN_EDGES = thr
        for _ in range(5):
            out = ops.edge_matvec_fwd(es.Mt, es.src, A, x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize(); ev[0].record()
        for _ in range(100):
            out = ops.edge_matvec_fwd(es.Mt, es.src, A, x)
        ev[1].record(); torch.cuda.synchronize()
        res[name] = (ev[0].elapsed_time(ev[1]) * 10, out)
    a, b = res["per-target kernel"], res["block per edge + SpMM"]
    print("E %4d  n %3d  max in-degree %d   per-target kernel %6.1f us   block per edge + SpMM %6.1f us   max diff %.2e"
          % (E, n, int(deg.max()), a[0], b[0], (a[1] - b[1]).abs().max().item()), flush=True)
