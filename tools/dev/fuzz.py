"""Randomised cross-check of the graph kernels against dense / index_add references (development aid)."""
import os, sys, itertools, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import ops, graph as G
from graph_odenet_amd.gat_layers import EdgeGraph
dev = torch.device("cuda:0")
bad = 0


def err(a, b):
    b = b.to(a.device)
    return float((a.double() - b.double()).abs().max() / max(1.0, float(b.double().abs().max())))


gen = torch.Generator().manual_seed(0)
# ---- SpMM ----
for n, d, mode, split in itertools.product((1, 63, 1000, 70000), (1, 3, 4, 7, 16, 32, 64, 73, 128, 256), ("vals", "pattern"), (None, 8)):
    m = max(1, n // 2 + 3)
    nnz = min(n * 6 + 5, 300000)
    r = torch.randint(0, n, (nnz,), generator=gen); c = torch.randint(0, m, (nnz,), generator=gen)
    if n > 10:
        r[: nnz // 5] = 3                      # a hub row
        r[r == 7] = 8                          # an empty row
    v = torch.randn(nnz, generator=gen) if mode == "vals" else None
    g = G.from_coo(r.to(dev), c.to(dev), None if v is None else v.to(dev), n, m, split=split)
    X = torch.randn(m, d, generator=gen)
    A = torch.sparse_coo_tensor(torch.stack([r, c]), v if v is not None else torch.ones(nnz), (n, m)).coalesce()
    if v is None:
        A = torch.sparse_coo_tensor(A.indices(), A.values(), (n, m))      # duplicates summed, as the product does
    ref = torch.sparse.mm(A.double(), X.double()).float()
    bias = torch.randn(d, generator=gen)
    out = ops.spmm(g, X.to(dev), bias=bias.to(dev), relu=True)
    e = err(out, torch.relu(ref + bias))
    if e > 2e-5:
        bad += 1; print("SPMM  n=%d d=%d %s split=%s err %.2e" % (n, d, mode, split, e))
    outT = ops.spmm(g.transpose(), torch.randn(n, d, generator=gen).to(dev))
print("spmm done")
# ---- GAT small path ----
for n, E, o in itertools.product((5, 300, 3000), (0, 40, 5000), (1, 7, 16, 33, 128, 200, 512)):
    if E == 0 and n != 5:
        continue
    src = torch.randint(0, n, (E,), generator=gen); tgt = torch.randint(0, max(1, n - 2), (E,), generator=gen)
    if E > 100:
        tgt[: E // 4] = 1
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    eg = EdgeGraph(src.to(dev), tgt.to(dev), Mtgt.to(dev))
    Ps, Pt, A2 = torch.randn(n, o, generator=gen), torch.randn(n, o, generator=gen), torch.randn(n, 2, generator=gen)
    bf, bw, dout = torch.randn(o, generator=gen), torch.randn(1, generator=gen), torch.randn(n, o, generator=gen)
    s, t = eg.src.cpu().long(), eg.tgt.cpu().long()
    Psd, Ptd, A2d = Ps.double().requires_grad_(True), Pt.double().requires_grad_(True), A2.double().requires_grad_(True)
    a = A2d[s, 0] + A2d[t, 1] + bw.double()
    w = torch.exp(a - (a.max() if E else 0.0))
    den = torch.zeros(n, dtype=torch.float64).index_add_(0, t, w) + 1e-6
    outr = torch.zeros(n, o, dtype=torch.float64).index_add_(0, t, (Psd[s] + Ptd[t] + bf.double()).relu() * w[:, None]) / den[:, None]
    outr.backward(dout.double())
    f = dict(dtype=torch.float32, device=dev)
    Pg, Qg, Ag = Ps.to(dev), Pt.to(dev), A2.to(dev)
    proj = ops.gat_proj(Pg, Qg, Ag)
    ad, amax = torch.empty(E, **f), torch.empty(1, **f)
    outd, wd, dend = torch.empty(n, o, **f), torch.empty(E, **f), torch.empty(n, **f)
    bfd, bwd = bf.to(dev), bw.to(dev)
    ops.gat_logits(proj, bwd, eg.src, eg.tgt, ad, amax)
    ops.gat_agg_fwd(eg, proj, o, bfd, ad, amax, 1e-6, outd, wd, dend)
    dz, da = torch.empty(E, o, **f), torch.empty(E, **f)
    dPs, dPt, dA2 = torch.empty(n, o, **f), torch.empty(n, o, **f), torch.empty(n, 2, **f)
    dd = dout.to(dev)
    if True:
        ops.gat_vjp(eg, proj, o, bfd, ad, amax, wd, dend, outd, dz, da, dPs, dPt, dA2, dout=dd)
        es = [err(outd, outr.detach()), err(dPs, Psd.grad), err(dPt, Ptd.grad), err(dA2, A2d.grad)]
    if max(es) > 5e-5:
        bad += 1; print("GAT   n=%d E=%d o=%d errs %s" % (n, E, o, ["%.1e" % x for x in es]))
print("gat done")
# ---- QC edge matvec + segment attention ----
for n, E, h in itertools.product((3, 200), (1, 50, 5000), (1, 5, 73, 100)):
    src = torch.randint(0, n, (E,), generator=gen); tgt = torch.randint(0, n, (E,), generator=gen)
    A = torch.randn(E, h, h, generator=gen) / max(1, h) ** 0.5; X = torch.randn(n, h, generator=gen)
    Mt = G.incidence_from_index(tgt.to(dev).to(torch.int32), n)
    out = ops.edge_matvec_fwd(Mt, src.to(dev).to(torch.int32), A.to(dev), X.to(dev))
    ref = torch.zeros(n, h, dtype=torch.float64).index_add_(0, tgt, torch.bmm(A.double(), X.double()[src].unsqueeze(-1)).squeeze(-1))
    e = err(out, ref.float())
    if e > 2e-5:
        bad += 1; print("QC    n=%d E=%d h=%d err %.2e" % (n, E, h, e))
print("qc done;", "FAILURES: %d" % bad if bad else "all within tolerance")
