"""GAT ODE block at d=96 (3 channels per GroupNorm group, generic dense kernels): fused field and autograd path against the
fp64 oracle (rk4 + adjoint restated in oracle/solver_ref.py)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import gat_models, odeint as OI
from oracle import layers_ref as R, solver_ref as S
dev = torch.device("cuda:0")
n, E = 700, 4000
gen = torch.Generator().manual_seed(0)
r = torch.randint(0, n, (E,), generator=gen); c = torch.randint(0, n, (E,), generator=gen)
for d in (24, 96, 160):
    torch.manual_seed(d)
    f = gat_models.ODEfunc(d)
    sd = {k: v.detach().double().requires_grad_(True) for k, v in f.named_parameters()}
    Mt = torch.sparse_coo_tensor(torch.stack([c, torch.arange(E)]), torch.ones(E, dtype=torch.float64), (n, E)).coalesce()
    x0 = torch.randn(n, d, generator=gen).relu()

    class F64(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(v.detach().clone()) for v in sd.values()])
        def forward(self, t, h):
            p = list(self.ps)
            return R.gat_odefunc(t, h, r, c, Mt, p[0], p[1], p[2], p[3], p[4], p[5])
    keys = list(sd.keys())
    assert keys == ["norm1.weight", "norm1.bias", "gc1.f.weight", "gc1.f.bias", "gc1.w.weight", "gc1.w.bias"], keys
    f64 = F64()
    xr = x0.double().requires_grad_(True)
    out = S.odeint_adjoint(f64, xr, torch.tensor([0., 1.], dtype=torch.float64), 1e-4, 1e-4, "rk4", {"step_size": 0.25})[1]
    out.square().mean().backward()
    ref = (out.detach(), xr.grad, [p.grad for p in f64.ps])
    f = f.to(dev); src, tgt = r.to(dev), c.to(dev)
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E, device=dev)]), torch.ones(E, device=dev), (n, E))
    f.set_adj(src, tgt, Mtgt)
    for fused in (True, False):
        hook = gat_models.ODEfunc.gode_fields
        if not fused:
            gat_models.ODEfunc.gode_fields = lambda self, y0: None
        try:
            f.zero_grad()
            xi = x0.to(dev).requires_grad_(True)
            o = OI.odeint_adjoint(f, xi, torch.tensor([0., 1.]), 1e-4, 1e-4, "rk4", {"step_size": 0.25})[1]
            o.square().mean().backward()
        finally:
            gat_models.ODEfunc.gode_fields = hook
        rel = lambda a, b: float((a.double().cpu() - b).abs().max() / max(1e-9, float(b.abs().max())))
        print("d=%-3d %-8s vs fp64 oracle: state %.1e gx %.1e params %s" % (d, "fused" if fused else "autograd", rel(o, ref[0]), rel(xi.grad, ref[1]),
              " ".join("%.0e" % rel(p.grad, g) for p, g in zip(f.parameters(), ref[2]))))
