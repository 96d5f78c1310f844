#!/usr/bin/env python3
"""Randomised H-head layer cases against the fp64 oracle (development aid)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd.gat_heads import MultiHeadGraphConvolution
from oracle import layers_ref as R
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(40):
    H = [1, 2, 3, 5, 8, 16][int(torch.randint(0, 6, (1,), generator=g))]
    o = [1, 2, 3, 4, 8, 16, 32][int(torch.randint(0, 7, (1,), generator=g))]
    n = int(torch.randint(2, 400, (1,), generator=g))
    E = int(torch.randint(1, 3000, (1,), generator=g))
    nin = int(torch.randint(1, 20, (1,), generator=g))
    src = torch.randint(0, n, (E,), generator=g)
    tgt = torch.randint(0, max(1, n - int(torch.randint(0, 3, (1,), generator=g))), (E,), generator=g)
    vals = torch.rand(E, generator=g) + 0.25 if case % 2 else torch.ones(E)
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), vals, (n, E))
    torch.manual_seed(case)
    layer = MultiHeadGraphConvolution(nin, H * o, heads=H)
    with torch.no_grad():
        for hd in layer.heads:
            hd.w.bias.add_(torch.randn(1) * 3)
    x = torch.randn(n, nin, generator=g)
    gout = torch.randn(n, H * o, generator=g)
    xd = x.double().requires_grad_(True)
    heads = [[p.detach().double().requires_grad_(True) for p in (hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias)] for hd in layer.heads]
    ref = R.gat_multihead_layer(xd, src, tgt, Mtgt.double(), heads)
    ref.backward(gout.double())
    layer = layer.to(dev)
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg, src.to(dev), tgt.to(dev), Mtgt.to(dev))
    out.backward(gout.to(dev))
    def rel(a, b):
        return float((a.detach().cpu().double() - b).abs().max() / max(1.0, float(b.abs().max())))
    errs = [rel(out, ref.detach()), rel(xg.grad, xd.grad)]
    for hd, ps in zip(layer.heads, heads):
        errs += [rel(p.grad, q.grad) for p, q in zip((hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias), ps)]
    ok = max(errs) < 5e-5
    bad += (not ok)
    print("case %2d H=%2d o=%2d n=%3d E=%4d nin=%2d vals=%d: max rel err %.1e %s" % (case, H, o, n, E, nin, case % 2, max(errs), "" if ok else "  <-- FAIL"), flush=True)
print("failures:", bad)
