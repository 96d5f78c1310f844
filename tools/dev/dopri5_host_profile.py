#!/usr/bin/env python3
"""Where the host time of a Cora dopri5 training step goes (cProfile; development aid)."""
import cProfile, os, pstats, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import small_bench
from graph_odenet_amd import models
from graph_odenet_amd.optim import Adam
dev = torch.device("cuda:0")
adj, x, y, idx = small_bench.cora()
adj, x, y, idx = adj.to(dev), x.to(dev), y.to(dev), idx.to(dev)
torch.manual_seed(0)
m = models.ODEGCN3(nfeat=x.shape[1], nhid=16, nclass=7, dropout=0.5, method=None, step_size=None).to(dev)
opt = Adam(m.parameters(), lr=0.01, weight_decay=5e-4)
def step():
    m.train(); opt.zero_grad()
    out = m(x, adj)
    torch.nn.functional.nll_loss(out[idx], y[idx]).backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
