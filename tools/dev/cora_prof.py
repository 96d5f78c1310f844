import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from small_bench import cora
from graph_odenet_amd import models
dev = torch.device("cuda:0")
adj, x, y, idx = cora()
adj, x, y, idx = adj.to(dev), x.to(dev), y.to(dev), idx.to(dev)
m = models.ODEGCN3(nfeat=x.shape[1], nhid=16, nclass=7, dropout=0.5, method="rk4", step_size=1 / 16).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=0.01)
def step():
    m.train(); opt.zero_grad(); out = m(x, adj)
    torch.nn.functional.nll_loss(out[idx], y[idx]).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
