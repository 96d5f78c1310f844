#!/usr/bin/env python3
"""Config C2 (BASELINE.json configs[1]): Pubmed's real topology, GCN-dense-paper variant (dense symmetric-normalised
adjacency, Glorot init, input dropout), ODEGCN3 with the reference's default dopri5 (rtol = atol = 1e-5), hidden 16;
synthetic features (the reference checkout lacks ind.pubmed.allx).  Training-step time on the GPU and, with --cpu,
of the oracle on the host cores.  Development aid; prints one line per case."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_odenet_amd import dense_paper  # noqa: E402

dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "pubmed_graph_sym.npz")))
n = int(g["n"])
idx = torch.stack([torch.from_numpy(g["rows"].astype(np.int64)), torch.from_numpy(g["cols"].astype(np.int64))])
adj = torch.sparse_coo_tensor(idx, torch.from_numpy(g["vals"]), (n, n)).to(dev).to_dense()     # dense, as the reference holds it
gen = torch.Generator().manual_seed(0)
x = (torch.rand(n, 500, generator=gen) < 0.1).float()
x = (x / x.sum(1, keepdim=True).clamp_min(1)).to(dev)
y = torch.randint(0, 3, (n,), generator=gen).to(dev)
tr = torch.arange(60, device=dev)
for method, step in ((None, None), ("rk4", 1 / 16)):
    torch.manual_seed(0)
    m = dense_paper.ODEGCN3(nfeat=500, nhid=16, nclass=3, dropout=0.5, method=method, step_size=step).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4)

    def train_step():
        m.train(); opt.zero_grad(); m.nfe = 0
        out = m(x, adj)
        nf = m.nfe; m.nfe = 0
        torch.nn.functional.nll_loss(out[tr], y[tr]).backward(); opt.step()
        return nf, m.nfe
    for _ in range(3):
        train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        nf, nb = train_step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("pubmed dense-paper nhid=16 method=%-6s %.2f ms/step (nfe_f %d, nfe_b %d)" % (method or "dopri5", dt * 1e3, nf, nb), flush=True)
