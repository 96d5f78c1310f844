#!/usr/bin/env python3
"""Per-parameter gradient differences between the one-launch and multi-launch dense half (development aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import _lib, gat_models, gat_heads
from graph_odenet_amd.models import ODEBlock
lib = _lib.load()
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "citeseer_gat_edges.npz")))
n = int(g["n"]); src = torch.from_numpy(g["src"].astype(np.int64)); tgt = torch.from_numpy(g["tgt"].astype(np.int64))
E = src.numel()
Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
D = torch.device("cuda:0")
heads, d = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
x0 = (torch.randn(n, d, generator=torch.Generator().manual_seed(3)) * 0.5).to(D)
gout = torch.randn(n, d, generator=torch.Generator().manual_seed(4)).to(D)
res = {}
for fused in (0, 1):
    lib.gode_set_option(b"small_fused", fused)
    torch.manual_seed(9)
    fn = gat_models.ODEfunc(d) if heads == 1 else gat_heads.ODEfunc(d, heads)
    blk = ODEBlock(fn, method="rk4", step_size=1.0 / steps).to(D)
    xg = x0.clone().requires_grad_(True)
    out = blk(xg, src.to(D), tgt.to(D), Mtgt.to(D))
    out.backward(gout)
    res[fused] = (out.detach().clone(), xg.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()})
print("forward diff %.3e, dx diff %.3e (|dx| %.3e)" % ((res[1][0] - res[0][0]).abs().max().item(), (res[1][1] - res[0][1]).abs().max().item(), res[0][1].abs().max().item()))
for k in res[0][2]:
    a, b = res[1][2][k], res[0][2][k]
    print("%-28s |ref| %.3e  diff %.3e" % (k, b.abs().max().item(), (a - b).abs().max().item()))
