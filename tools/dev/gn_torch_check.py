"""torch 2.10.0+rocm7.0 F.group_norm backward on 2-D input: CPU vs GPU dgamma / dbeta (cited by functional.py; HISTORY.md)."""
import torch, torch.nn.functional as F
torch.manual_seed(0)
for n, c, g in ((3327, 128, 32), (120, 64, 32), (257, 16, 16), (3327, 129, 1)):
    x = torch.randn(n, c); w = torch.rand(c) + .5; b = torch.rand(c) - .5; go = torch.randn(n, c)
    res = {}
    for dev in ("cpu", "cuda:0"):
        xi = x.to(dev).detach().clone().requires_grad_(True); wi = w.to(dev).detach().clone().requires_grad_(True); bi = b.to(dev).detach().clone().requires_grad_(True)
        out = F.group_norm(xi, g, wi, bi, 1e-5)
        out.backward(go.to(dev))
        res[dev] = (out.detach().cpu(), xi.grad.cpu(), wi.grad.cpu(), bi.grad.cpu())
    names = ("out", "dx", "dgamma", "dbeta")
    print(n, c, g, " ".join("%s err %.2e (scale %.2e)" % (k, (a - b_).abs().max().item(), a.abs().max().item()) for k, a, b_ in zip(names, res["cpu"], res["cuda:0"])))
