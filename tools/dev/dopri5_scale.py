import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import models
from graph_odenet_amd.synth import rmat_graph
dev = torch.device("cuda:0")
g = rmat_graph(20, 10_000_000, seed=0, device=dev); g.transpose()
n = g.n_rows
x = torch.randn(n, 128, device=dev); y = torch.randint(0, 16, (n,), device=dev); idx = torch.randperm(n, device=dev)[: n // 10]
torch.manual_seed(42)
m = models.ODEGCN3(nfeat=128, nhid=128, nclass=16, dropout=0.5, tol=1e-3).to(dev)      # default method: dopri5
opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.train(); opt.zero_grad(); m.nfe = 0
    out = m(x, g); nf = m.nfe; m.nfe = 0
    loss = torch.nn.functional.nll_loss(out[idx], y[idx]); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print("dopri5 (tol 1e-3) at 2^20 nodes: %.1f ms/step, nfe_f %d nfe_b %d, %.2f ms per eval, loss %.4f, peak %.1f GB" % (
        (time.perf_counter() - t0) * 1e3, nf, m.nfe, (time.perf_counter() - t0) * 1e3 / (nf + m.nfe), float(loss), torch.cuda.max_memory_allocated() / 2**30))
