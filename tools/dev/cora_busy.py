import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import config_bench as cb
dev = torch.device("cuda:0")
# same as config_bench.c1_cora but with 200 timed steps so that rocprof totals are per-step meaningful
import numpy as np, torch.nn.functional as F
from graph_odenet_amd import models
g = dict(np.load(os.path.join(cb.GOLD, "cora_graph.npz"))); n = int(g["n"]); T = lambda a: torch.from_numpy(np.asarray(a))
adj = torch.sparse_coo_tensor(torch.stack([T(g["rows"].astype(np.int64)), T(g["cols"].astype(np.int64))]), T(g["vals"]), (n, n)).to(dev)
x = torch.zeros(n, int(g["n_feat"])); x[T(g["feat_rows"].astype(np.int64)), T(g["feat_cols"].astype(np.int64))] = T(g["feat_vals"])
x, y, idx = x.to(dev), T(g["labels"].astype(np.int64)).to(dev), T(g["idx_train"].astype(np.int64)).to(dev)
torch.manual_seed(0)
m = models.ODEGCN3(nfeat=x.shape[1], nhid=16, nclass=7, dropout=0.5, method="rk4", step_size=1 / 16).to(dev)
step = cb._trainer(m, lambda: m(x, adj), idx, y)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("wall ms/step %.3f" % ((time.perf_counter() - t0) / 200 * 1e3))
