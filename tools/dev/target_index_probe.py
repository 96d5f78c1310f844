#!/usr/bin/env python3
"""The per-edge target vector off the collate's dense N x E matrix: `(M != 0).to(uint8).argmax(0)` (three library launches)
against gode_dense_first_nonzero_f32 (one), on the batches of tools/config_bench.py's C4."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops
from graph_odenet_amd.synth import qm9_like_batch

D = torch.device("cuda:0")
Ms = [qm9_like_batch(20, seed=5000 + b, device=D)[3] for b in range(16)]


def timed(fn, reps=50):
    for _ in range(5):
        for M in Ms:
            fn(M)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(reps):
        for M in Ms:
            fn(M)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / (reps * len(Ms)) * 1e3


for rep in range(3):
    a = timed(lambda M: (M != 0).to(torch.uint8).argmax(0))
    b = timed(ops.dense_first_nonzero)
    print("library expression %6.1f us   one launch %6.1f us   (N x E about %d x %d)" % (a, b, Ms[0].shape[0], Ms[0].shape[1]), flush=True)
