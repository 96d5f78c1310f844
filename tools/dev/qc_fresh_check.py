import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import qc_models
from graph_odenet_amd.synth import qm9_like_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = qc_models.EdgeGCN_K_Sum(node_features=13, edge_features=5, target_features=12, hidden_features=73, num_layers=3).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
for label, seeds in (("one batch reused", [0] * 60), ("distinct objects, same shapes", None), ("distinct batches", list(range(60)))):
    if seeds is None:
        batches = [qm9_like_batch(20, seed=0, device=dev) for _ in range(60)]
    elif len(set(seeds)) == 1:
        b = qm9_like_batch(20, seed=0, device=dev); batches = [b] * 60
    else:
        batches = [qm9_like_batch(20, seed=s, device=dev) for s in seeds]
    tg = torch.randn(20, 12, device=dev)
    def step(b):
        opt.zero_grad(set_to_none=False)
        F.mse_loss(net(*b), tg).backward(); opt.step()
    for b in batches[:10]:
        step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in batches[10:]:
        step(b)
    torch.cuda.synchronize()
    print("%-32s %.2f ms/step" % (label, (time.perf_counter() - t0) / 50 * 1e3))
