"""Where do the 2 ms between 'one batch reused' and 'a new batch every step' go?  Same graph sizes with new values every
step vs new sizes every step (development aid)."""
import os, sys, time, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import qc_models
from graph_odenet_amd.synth import qm9_like_batch
dev = torch.device("cuda:0")
for name in ("EdgeGCN_K_Sum", "MPNN_ENN_K_Set2Set"):
    torch.manual_seed(0)
    net = getattr(qc_models, name)(node_features=13, edge_features=5, target_features=12, hidden_features=73, num_layers=3).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    def run(batches, label):
        def step(b):
            x, ef, Esrc, Etgt, batch, tgt = b
            opt.zero_grad(); F.mse_loss(net(x, ef, Esrc, Etgt, batch), tgt).backward(); opt.step()
        for b in batches[:5]: step(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in batches[5:]: step(b)
        torch.cuda.synchronize(); print("%-20s %-34s %.2f ms/step" % (name, label, (time.perf_counter() - t0) / (len(batches) - 5) * 1e3), flush=True)
    mk = lambda s: qm9_like_batch(20, seed=s, device=dev) + (torch.randn(20, 12, device=dev),)
    varied = [mk(s) for s in range(100, 160)]
    run(varied, "new sizes every step")
    base = mk(7)
    same = []
    for s in range(60):
        x, ef, Esrc, Etgt, batch, tgt = base
        same.append((x + 0.01 * torch.randn_like(x), ef.clone(), Esrc.clone(), Etgt.clone(), batch.clone(), torch.randn(20, 12, device=dev)))
    run(same, "same sizes, new tensors every step")
    run([base] * 60, "one batch object reused")
