import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import ops
from graph_odenet_amd.gat_layers import EdgeGraph
dev = torch.device("cuda:0")
for o in (16, 32, 64, 128):
    gen = torch.Generator().manual_seed(o)
    n, E = 70000, 300000
    src = torch.randint(0, n, (E,), generator=gen); tgt = torch.randint(0, n - 100, (E,), generator=gen); tgt[:3000] = 5
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    eg = EdgeGraph(src.to(dev), tgt.to(dev), Mtgt.to(dev))
    Ps, Pt, A2 = torch.randn(n, o, generator=gen), torch.randn(n, o, generator=gen), torch.randn(n, 2, generator=gen)
    bf, bw = torch.randn(o, generator=gen), torch.randn(1, generator=gen)
    dout = torch.randn(n, o, generator=gen)
    # reference in double on CPU, edges in canonical order
    s, t = eg.src.cpu().long(), eg.tgt.cpu().long()
    Psd, Ptd, A2d = Ps.double().requires_grad_(True), Pt.double().requires_grad_(True), A2.double().requires_grad_(True)
    z = Psd[s] + Ptd[t] + bf.double()
    y = z.relu()
    a = A2d[s, 0] + A2d[t, 1] + bw.double()
    w = torch.exp(a - a.max())
    den = torch.zeros(n, dtype=torch.float64).index_add_(0, t, w) + 1e-6
    out = torch.zeros(n, o, dtype=torch.float64).index_add_(0, t, y * w[:, None]) / den[:, None]
    out.backward(dout.double())
    f = dict(dtype=torch.float32, device=dev)
    Pg, Qg, Ag = Ps.to(dev), Pt.to(dev), A2.to(dev)          # kept alive: the struct holds raw pointers
    proj = ops.gat_proj(Pg, Qg, Ag)
    ad, amax = torch.empty(E, **f), torch.empty(1, **f)
    outd, wd, dend = torch.empty(n, o, **f), torch.empty(E, **f), torch.empty(n, **f)
    ops.gat_logits(proj, bw.to(dev), eg.src, eg.tgt, ad, amax)
    ops.gat_agg_fwd(eg, proj, o, bf.to(dev), ad, amax, 1e-6, outd, wd, dend)
    dz, da = torch.empty(E, o, **f), torch.empty(E, **f)
    dPs, dPt, dA2 = torch.empty(n, o, **f), torch.empty(n, o, **f), torch.empty(n, 2, **f)
    ops.gat_vjp(eg, proj, o, bf.to(dev), ad, amax, wd, dend, outd, dz, da, dPs, dPt, dA2, dout=dout.to(dev))
    e = lambda x, r: float((x.cpu().double() - r).abs().max())
    print("o=%d out %.2e dPs %.2e dPt %.2e dAs %.2e dAt %.2e" % (o, e(outd, out.detach()), e(dPs, Psd.grad), e(dPt, Ptd.grad),
                                                                e(dA2[:, 0], A2d.grad[:, 0]), e(dA2[:, 1], A2d.grad[:, 1])))
