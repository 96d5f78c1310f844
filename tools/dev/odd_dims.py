"""Fused ODE fields at odd widths (generic kernel paths) against the same module driven through autograd."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import models, gat_models, odeint as OI
dev = torch.device("cuda:0")
n, E = 700, 4000
gen = torch.Generator().manual_seed(0)
r = torch.randint(0, n, (E,), generator=gen); c = torch.randint(0, n, (E,), generator=gen)
v = torch.rand(E, generator=gen); v = v / torch.zeros(n).index_add_(0, r, v)[r]
adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).to(dev)
src, tgt = r.to(dev), c.to(dev)
Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E, device=dev)]), torch.ones(E, device=dev), (n, E))
for kind in ("gcn", "gat"):
    for d in (7, 24, 96, 160):
        for method, opts in (("rk4", {"step_size": 0.25}), (None, None)):
            mod = models if kind == "gcn" else gat_models
            torch.manual_seed(d)
            f = mod.ODEfunc(d).to(dev)
            f.set_adj(*((adj,) if kind == "gcn" else (src, tgt, Mtgt)))
            x0 = torch.randn(n, d, device=dev).relu()
            res = {}
            for fused in (True, False):
                hook = mod.ODEfunc.gode_fields
                if not fused:
                    mod.ODEfunc.gode_fields = lambda self, y0: None
                try:
                    f.zero_grad(); f.nfe = 0
                    xi = x0.clone().requires_grad_(True)
                    out = OI.odeint_adjoint(f, xi, torch.tensor([0., 1.]), 1e-4, 1e-4, method, opts)[1]
                    out.square().mean().backward()
                    res[fused] = (out.detach(), xi.grad.clone(), [p.grad.clone() for p in f.parameters()], f.nfe)
                finally:
                    mod.ODEfunc.gode_fields = hook
            rel = lambda a, b: float((a - b).abs().max() / max(1e-6, float(b.abs().max())))
            errs = [rel(res[True][0], res[False][0]), rel(res[True][1], res[False][1])] + [rel(a, b) for a, b in zip(res[True][2], res[False][2])]
            print("%s d=%-3d %-6s nfe %d/%d  max rel err: state %.1e gx %.1e params %.1e" % (kind, d, method or "dopri5", res[True][3], res[False][3], errs[0], errs[1], max(errs[2:])))
