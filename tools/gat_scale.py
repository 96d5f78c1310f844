#!/usr/bin/env python3
"""ODE-GAT training step at the benchmark scale (R-MAT 2^20 nodes / 10 M edges, d = 128, rk4): exercises the
record-balanced attention kernels end to end.  usage: gat_scale.py [rk4 steps] [heads].  Development aid; prints one line."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_odenet_amd import gat_heads, gat_models  # noqa: E402
from graph_odenet_amd.synth import rmat_graph  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = rmat_graph(20, 10_000_000, seed=0, device=dev)
n = g.n_rows
rp = g.rowptr.to(torch.int64)
tgt = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
src = g.col.to(torch.int64)
E = src.numel()
Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E, device=dev)]), torch.ones(E, device=dev), (n, E))
del g
x = torch.randn(n, 128, device=dev)
y = torch.randint(0, 16, (n,), device=dev)
idx = torch.randperm(n, device=dev)[: n // 10]
torch.manual_seed(0)
m = (gat_models if heads == 1 else gat_heads.zoo(heads)).ODEGCN3(nfeat=128, nhid=128, nclass=16, dropout=0.5, method="rk4", step_size=1.0 / steps).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4)


def step():
    m.train(); opt.zero_grad()
    out = m(x, src, tgt, Mtgt)
    loss = torch.nn.functional.nll_loss(out[idx], y[idx])
    loss.backward(); opt.step()
    return float(loss)


step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2):
    loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print("ODE-GAT heads=%d" % heads, "N=%d E=%d d=128 rk4 %d steps (%d f-evals fwd): %.1f ms/step, %.2f ms per (f-eval + adjoint stage), loss %.4f, "
      "peak memory %.1f GB" % (n, E, steps, 4 * steps, dt * 1e3, dt * 1e3 / (4 * steps), loss, torch.cuda.max_memory_allocated() / 2**30))
