import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import gat_models, odeint as OI, ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
ge = dict(np.load(os.path.join(ROOT, "tests/golden/citeseer_gat_edges.npz")))
n = int(ge["n"]); T = lambda a: torch.from_numpy(np.asarray(a))
src, tgt = T(ge["src"]).long().to(dev), T(ge["tgt"]).long().to(dev)
e = src.numel()
Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e, device=dev)]), torch.ones(e, device=dev), (n, e))
d = 128
torch.manual_seed(5)
f = gat_models.ODEfunc(d).to(dev); f.set_adj(src, tgt, Mtgt)
x0 = torch.randn(n, d, device=dev); t = torch.tensor([0., 1.], device=dev)
res = {}
for fused in (True, False):
    hook = gat_models.ODEfunc.gode_fields
    if not fused: gat_models.ODEfunc.gode_fields = lambda self, y0: None
    f.zero_grad(); xi = x0.clone().requires_grad_(True)
    out = OI.odeint_adjoint(f, xi, t, 1e-5, 1e-5, "rk4", {"step_size": 0.25})[1]
    out.square().sum().backward()
    res[fused] = {k: p.grad.clone() for k, p in f.named_parameters()}
    gat_models.ODEfunc.gode_fields = hook
for k in res[True]:
    a, b = res[True][k], res[False][k]
    print(k, "err %.3e scale %.3e ratio(mean) %.4f" % ((a - b).abs().max().item(), b.abs().max().item(), (a.flatten() @ b.flatten() / (b.flatten() @ b.flatten())).item()))
# direct kernel check: generic bwd dgamma with d_out != d_in
torch.manual_seed(0)
nn_, dd, dout, G = 777, 128, 258, 32
x = torch.randn(nn_, dd, requires_grad=True); gam = (torch.rand(dd) + .5).requires_grad_(True); bet = (torch.rand(dd) - .5).requires_grad_(True)
W = (torch.randn(dd + 1, dout) / dd ** .5)
S = torch.cat([torch.full((nn_, 1), .3), F.group_norm(x, G, gam, bet, 1e-5)], 1) @ W
dS = torch.randn(nn_, dout); S.backward(dS)
dx, dgp, dbp = ops.gn_time_gemm_bwd([(1., x.detach().to(dev))], nn_, dd, G, 1e-5, gam.detach().to(dev), W.to(dev), True, dS.to(dev))
print("direct dgamma err", (dgp.sum(0).cpu() - gam.grad).abs().max().item(), gam.grad.abs().max().item(), "dbeta err", (dbp.sum(0).cpu() - bet.grad).abs().max().item())
print("---- single eval VJP")
from graph_odenet_amd.gat_ode import gat_fields
fwd, mk, plist = gat_fields(f, x0)
adj = mk()
comps = adj.new_state(x0)
a0 = torch.randn(n, d, device=dev)
out = [torch.zeros_like(c) for c in comps]
adj.eval(0.3, [[(1.0, x0)], [(1.0, a0)]] + [[(1.0, c)] for c in comps[2:]], out)
xi = x0.clone().requires_grad_(True)
tt = torch.tensor(0.3, device=dev, requires_grad=True)
fe = f(tt, xi)
vj = torch.autograd.grad(fe, (tt, xi) + tuple(f.parameters()), -a0)
pg = adj.param_grads(out)
print("f err", (out[0] - fe).abs().max().item(), "vy err", (out[1] - vj[1]).abs().max().item(), "vt", out[2].item(), vj[0].item())
for (k, p), g1, g2 in zip(f.named_parameters(), pg, vj[2:]):
    print(k, "err %.3e scale %.3e" % ((g1 - g2).abs().max().item(), g2.abs().max().item()))
print("---- finite difference on norm1.bias[3], norm1.weight[5]")
with torch.no_grad():
    for name, par, idx in (("beta", f.norm1.bias, 3), ("gamma", f.norm1.weight, 5)):
        vals = []
        for eps_ in (1e-2, -1e-2):
            par[idx] += eps_
            vals.append(float((f(torch.tensor(0.3, device=dev), x0).double() * (-a0).double()).sum()))
            par[idx] -= eps_
        fd = (vals[0] - vals[1]) / 2e-2
        k = 1 if name == "beta" else 0
        print(name, "FD %.4f fused %.4f autograd %.4f" % (fd, pg[k][idx].item(), vj[2 + k][idx].item()))
