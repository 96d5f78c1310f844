import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import ops
dev = torch.device("cuda:0")
for nn_, dd, dout, G, unit in ((777, 128, 258, 32, False), (3327, 128, 258, 32, False), (3327, 128, 258, 32, True), (3327, 128, 128, 32, True), (3327, 96, 96, 32, True), (4100, 128, 258, 32, False), (2000, 128, 258, 32, False)):
    torch.manual_seed(0)
    x = torch.randn(nn_, dd, requires_grad=True)
    gam = (torch.ones(dd) if unit else torch.rand(dd) + .5).requires_grad_(True)
    bet = (torch.zeros(dd) if unit else torch.rand(dd) - .5).requires_grad_(True)
    W = (torch.randn(dd + 1, dout) / dd ** .5)
    S = torch.cat([torch.full((nn_, 1), .3), F.group_norm(x, G, gam, bet, 1e-5)], 1) @ W
    dS = torch.randn(nn_, dout); S.backward(dS)
    dx, dgp, dbp = ops.gn_time_gemm_bwd([(1., x.detach().to(dev))], nn_, dd, G, 1e-5, gam.detach().to(dev), W.to(dev), True, dS.to(dev))
    print(nn_, dd, dout, unit, "parts", dgp.shape[0], "dgamma err %.3e / %.3e" % ((dgp.sum(0).cpu() - gam.grad).abs().max().item(), gam.grad.abs().max().item()),
          "dbeta err %.3e" % (dbp.sum(0).cpu() - bet.grad).abs().max().item(), "nonzero part rows", int((dgp.abs().sum(1) > 0).sum()))
