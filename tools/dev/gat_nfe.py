"""Dev check: fused GAT field vs autograd path, dopri5, nhid 64: nfe and gradient agreement."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import gat_models, odeint as OI
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "citeseer_gat_edges.npz")))
n = int(g["n"])
src = torch.from_numpy(g["src"].astype(np.int64)).to(dev); tgt = torch.from_numpy(g["tgt"].astype(np.int64)).to(dev)
e = src.numel()
Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e, device=dev)]), torch.ones(e, device=dev), (n, e))
for d in (16, 64, 128):
    torch.manual_seed(5)
    f = gat_models.ODEfunc(d).to(dev); f.set_adj(src, tgt, Mtgt)
    x0 = torch.randn(n, d, device=dev).relu()
    t = torch.tensor([0., 1.], device=dev)
    res = {}
    for fused in (True, False):
        hook = gat_models.ODEfunc.gode_fields
        if not fused:
            gat_models.ODEfunc.gode_fields = lambda self, y0: None
        try:
            f.zero_grad(); f.nfe = 0
            xi = x0.clone().requires_grad_(True)
            out = OI.odeint_adjoint(f, xi, t, 1e-5, 1e-5, None, None)[1]
            nf = f.nfe; f.nfe = 0
            out.square().sum().backward()
            res[fused] = (out.detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in f.named_parameters()}, nf, f.nfe)
        finally:
            gat_models.ODEfunc.gode_fields = hook
    a, b = res[True], res[False]
    print("d=%d nfe fused %d/%d autograd %d/%d  |out| %.3e diff %.3e  gx diff %.3e (scale %.3e)" % (
        d, a[3], a[4], b[3], b[4], b[0].abs().max(), (a[0] - b[0]).abs().max(), (a[1] - b[1]).abs().max(), b[1].abs().max()))
    for k in a[2]:
        print("   %-12s diff %.3e scale %.3e" % (k, (a[2][k] - b[2][k]).abs().max(), b[2][k].abs().max()))
