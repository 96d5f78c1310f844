#!/usr/bin/env python3
"""GAT (config C3) step timing: Citeseer's real edge list, synthetic row-normalised features (F=3703), ODEGCN3 (GAT
layers) with rk4 64 evals / dopri5; one head (the reference's layer) and 8 heads (gat_heads.py, BASELINE configs[2]);
product on the GPU.  Development aid."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_odenet_amd import gat_heads, gat_models  # noqa: E402

dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "citeseer_gat_edges.npz")))
n = int(g["n"])
src = torch.from_numpy(g["src"].astype(np.int64)).to(dev)
tgt = torch.from_numpy(g["tgt"].astype(np.int64)).to(dev)
e = src.numel()
Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e, device=dev)]), torch.ones(e, device=dev), (n, e))
gen = torch.Generator().manual_seed(0)
x = (torch.rand(n, 3703, generator=gen) < 0.01).float()
x = (x / x.sum(1, keepdim=True).clamp_min(1)).to(dev)
y = torch.randint(0, 6, (n,), generator=gen).to(dev)
idx = torch.arange(120, device=dev)
only = sys.argv[1:]          # e.g. "8:64:rk4" - restrict to some cases (profiling)
for heads, nhid in ((1, 16), (1, 64), (8, 64), (8, 128)):
    zoo = gat_models if heads == 1 else gat_heads.zoo(heads)
    for method, step in (("rk4", 1 / 16), (None, None)):
        if only and "%d:%d:%s" % (heads, nhid, method or "dopri5") not in only:
            continue
        torch.manual_seed(0)
        m = zoo.ODEGCN3(nfeat=3703, nhid=nhid, nclass=6, dropout=0.5, method=method, step_size=step).to(dev)
        from graph_odenet_amd.optim import Adam
        opt = Adam(m.parameters(), lr=0.01, weight_decay=5e-4)

        def train_step():
            m.train(); opt.zero_grad(set_to_none=False); m.nfe = 0
            out = m(x, src, tgt, Mtgt)
            nf = m.nfe; m.nfe = 0
            torch.nn.functional.nll_loss(out[idx], y[idx]).backward(); opt.step()
            return nf, m.nfe
        for _ in range(3):
            train_step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            nf, nb = train_step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("citeseer GAT heads=%d nhid=%-3d method=%-6s %.1f ms/step (nfe_f %d, nfe_b %d)" % (heads, nhid, method or "dopri5", dt * 1e3, nf, nb), flush=True)
