"""Generates tests/golden/*.npz by importing the REFERENCE's own classes from
/root/reference (build container only; the reference never travels to the GPU box).

Run:  python tests/golden/make_golden.py
What is captured (inputs AND expected outputs, fp32):
  gcn_layer.npz      GraphConvolution / FixedGraphConvolution fwd + grads   (GCN/layers.py)
  gcn_odefunc_*.npz  ODEfunc fwd + VJP at d in {16, 64, 128}                (GCN/models.py:161-179)
  gcn_odefunc2.npz   ODEfunc2 fwd + VJP                                     (GCN/models.py:551-575)
  gcn3_cora.npz      GCN3 / RGCN3 eval logits on real Cora with fixed weights (GCN/models.py, GCN/utils.py)
  cora_graph.npz / citeseer_graph.npz   loader outputs (GCN/utils.py:134-229, GAT/utils.py:187-209)
  gat_layer.npz      GAT GraphConvolution fwd + grads                        (GAT/layers.py)
  gat_odefunc.npz    GAT ODEfunc fwd + VJP                                   (GAT/models.py:161-179)
  qc_layers.npz      MPNN_enn_edge (T=1,3) and EdgeGraphConvolution fwd + grads (QC/mpnn.py, QC/layers.py)
  scatter_kat.npz    the scatter_add docstring known-answer vector           (QC/torch_scatter.py:207-218)
  pubmed_graph_sym.npz  Pubmed topology, D^-1/2 (A+I) D^-1/2                 (GCN-dense-paper/utils.py:70-110)
  gcn_depth_models.npz  depth-sweep model family (GCNK*, RESK*) eval outputs + one gradient (GCN/models.py:255-522)
  gcn_variants.npz   GCN-mlp-sum layer / ODEfunc / models and GCN-dense-paper models (their layers.py, models.py)
  train_traj_qc.npz  five training steps of the reference's EdgeGCN_K_Sum / MPNN_ENN_K_Set2Set on a synthetic batch
  train_traj_cora.npz  ten training steps of the reference's GCN3 / RGCN3norm (GCN) and GCN3 (GAT) on Cora: loss trajectory
  gat_zoo.npz        non-ODE GAT model zoo eval outputs + one gradient (GAT/models.py)
  set2set.npz        the reference's Set2Set readout alone: q_star + gradients (QC/set2set.py:6-75)
  qc_models.npz      QC model zoo outputs + small gradients on a synthetic batch (QC/layer_models.py:27-232)

`torchdiffeq` is absent from the image; an EMPTY stand-in module object is registered so
that `models.py` imports.  No solver is ever called through it (parity at the solver
boundary stays unpinned, see oracle/solver_ref.py).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def ref_import(subdir, *names):
    """Import reference modules by bare name from REF/subdir (as its scripts do)."""
    for n in ("layers", "models", "utils", "mpnn"):
        sys.modules.pop(n, None)
    sys.path.insert(0, os.path.join(REF, subdir))
    try:
        return [importlib.import_module(n) for n in names]
    finally:
        sys.path.pop(0)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def rand_graph(n, nnz, seed, normalize=True, dup=True):
    g = torch.Generator().manual_seed(seed)
    r = torch.randint(0, n, (nnz,), generator=g)
    c = torch.randint(0, n, (nnz,), generator=g)
    if dup:   # uncoalesced: repeat some entries (torch.spmm sums duplicates)
        r = torch.cat([r, r[:nnz // 10]])
        c = torch.cat([c, c[:nnz // 10]])
    v = torch.rand(r.numel(), generator=g) + 0.1
    if normalize:
        deg = torch.zeros(n).index_add_(0, r, v)
        v = v / deg[r]
    return r, c, v


def main():
    sys.dont_write_bytecode = True
    stub = types.ModuleType("torchdiffeq")
    stub.odeint_adjoint = None
    stub.odeint = None
    sys.modules["torchdiffeq"] = stub
    # scipy >= 1.8 moved this private module; GCN/utils.py:8 imports an unused symbol from it
    import scipy.sparse.linalg as spla
    alias = types.ModuleType("scipy.sparse.linalg.eigen.arpack")
    alias.eigsh = spla.eigsh
    sys.modules.setdefault("scipy.sparse.linalg.eigen", types.ModuleType("scipy.sparse.linalg.eigen"))
    sys.modules["scipy.sparse.linalg.eigen.arpack"] = alias

    torch.manual_seed(0)

    # ---------------- GCN layer --------------------------------------------------
    layers, models = ref_import("GCN", "layers", "models")
    n, fi, fo = 300, 40, 16
    r, c, v = rand_graph(n, 1500, 1)
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    x = torch.randn(n, fi, requires_grad=True)
    gc = layers.GraphConvolution(fi, fo)
    out = gc(x, adj)
    gout = torch.randn_like(out)
    out.backward(gout)
    save("gcn_layer.npz", rows=r, cols=c, vals=v, n=n, x=x, weight=gc.weight, bias=gc.bias, out=out, gout=gout,
         gx=x.grad, gw=gc.weight.grad, gb=gc.bias.grad)

    # ---------------- ODEfunc fwd + VJP ------------------------------------------
    for d in (16, 64, 128):
        torch.manual_seed(10 + d)
        n = 257
        r, c, v = rand_graph(n, 1800, 2 + d)
        adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
        f = models.ODEfunc(d)
        with torch.no_grad():   # non-trivial affine so that gamma/beta paths are exercised
            f.norm1.weight.uniform_(0.5, 1.5)
            f.norm1.bias.uniform_(-0.5, 0.5)
        f.set_adj(adj)
        x = torch.randn(n, d, requires_grad=True)
        t = torch.tensor(0.37)
        out = f(t, x)
        gout = torch.randn_like(out)
        out.backward(gout)
        save("gcn_odefunc_%d.npz" % d, rows=r, cols=c, vals=v, n=n, t=t, x=x, gn_w=f.norm1.weight, gn_b=f.norm1.bias,
             W=f.gc1.weight, b=f.gc1.bias, out=out, gout=gout, gx=x.grad, g_gn_w=f.norm1.weight.grad,
             g_gn_b=f.norm1.bias.grad, gW=f.gc1.weight.grad, gb=f.gc1.bias.grad)

    # ---------------- ODEfunc2 ----------------------------------------------------
    torch.manual_seed(5)
    d, n = 64, 200
    r, c, v = rand_graph(n, 1200, 7)
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    f2 = models.ODEfunc2(d, 0.5)
    f2.set_adj(adj)
    x = torch.randn(n, d, requires_grad=True)
    t = torch.tensor(0.61)
    out = f2(t, x)
    gout = torch.randn_like(out)
    out.backward(gout)
    sd = {k.replace(".", "__"): p for k, p in f2.state_dict().items()}
    gd = {"g__" + k.replace(".", "__"): p.grad for k, p in f2.named_parameters()}
    save("gcn_odefunc2.npz", rows=r, cols=c, vals=v, n=n, t=t, x=x, out=out, gout=gout, gx=x.grad, **sd, **gd)

    # ---------------- real Cora / Citeseer through the reference loader -----------
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        (gutils,) = ref_import("GCN", "utils")
        for ds in ("cora", "citeseer"):
            adj, feats, labels, itr, iva, ite = gutils.load_data_new(ds)
            idx = adj._indices()
            fnz = feats.nonzero()
            save("%s_graph.npz" % ds, rows=idx[0].to(torch.int32), cols=idx[1].to(torch.int32),
                 vals=adj._values(), n=adj.shape[0], feat_rows=fnz[:, 0].to(torch.int32),
                 feat_cols=fnz[:, 1].to(torch.int32), feat_vals=feats[fnz[:, 0], fnz[:, 1]],
                 n_feat=feats.shape[1], labels=labels.to(torch.int16), idx_train=itr.to(torch.int32),
                 idx_val=iva.to(torch.int32), idx_test=ite.to(torch.int32))
            if ds == "cora":
                torch.manual_seed(42)
                res = {}
                for name in ("GCN3", "RGCN3"):
                    m = getattr(models, name)(nfeat=feats.shape[1], nhid=16, nclass=int(labels.max()) + 1, dropout=0.5)
                    m.eval()
                    with torch.no_grad():
                        res[name + "__out"] = m(feats, adj)
                    for k, p in m.state_dict().items():
                        res[name + "__" + k.replace(".", "__")] = p
                save("gcn3_cora.npz", **res)
        # GAT edge list of citeseer (GAT/utils.py:187-209)
        (gatutils,) = ref_import("GAT", "utils")
        src, tgt, Mtgt, feats, labels, itr, iva, ite = gatutils.load_data_new("citeseer")
        mi = Mtgt._indices()
        save("citeseer_gat_edges.npz", src=src.to(torch.int32), tgt=tgt.to(torch.int32),
             m_rows=mi[0].to(torch.int32), m_cols=mi[1].to(torch.int32), m_vals=Mtgt._values(), n=Mtgt.shape[0])
    finally:
        os.chdir(cwd)

    # ---------------- GAT layer + ODEfunc ------------------------------------------
    glayers, gmodels = ref_import("GAT", "layers", "models")
    torch.manual_seed(3)
    n, e, fi, fo = 120, 700, 24, 16
    g = torch.Generator().manual_seed(11)
    src = torch.randint(0, n, (e,), generator=g)
    tgt = torch.randint(0, n - 10, (e,), generator=g)     # last 10 nodes never a target -> 0/eps rows
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e)]), torch.ones(e), (n, e))
    layer = glayers.GraphConvolution(fi, fo)
    x = torch.randn(n, fi, requires_grad=True)
    out = layer(x, src, tgt, Mtgt)
    gout = torch.randn_like(out)
    out.backward(gout)
    save("gat_layer.npz", src=src, tgt=tgt, n=n, x=x, f_w=layer.f.weight, f_b=layer.f.bias, w_w=layer.w.weight,
         w_b=layer.w.bias, out=out, gout=gout, gx=x.grad, g_f_w=layer.f.weight.grad, g_f_b=layer.f.bias.grad,
         g_w_w=layer.w.weight.grad, g_w_b=layer.w.bias.grad)
    d = 64
    f = gmodels.ODEfunc(d)
    f.set_adj(src, tgt, Mtgt)
    x = torch.randn(n, d, requires_grad=True)
    t = torch.tensor(0.25)
    out = f(t, x)
    gout = torch.randn_like(out)
    out.backward(gout)
    save("gat_odefunc.npz", src=src, tgt=tgt, n=n, t=t, x=x, gn_w=f.norm1.weight, gn_b=f.norm1.bias,
         f_w=f.gc1.f.weight, f_b=f.gc1.f.bias, w_w=f.gc1.w.weight, w_b=f.gc1.w.bias, out=out, gout=gout, gx=x.grad,
         g_f_w=f.gc1.f.weight.grad, g_f_b=f.gc1.f.bias.grad, g_w_w=f.gc1.w.weight.grad, g_w_b=f.gc1.w.bias.grad,
         g_gn_w=f.norm1.weight.grad, g_gn_b=f.norm1.bias.grad)

    # ---------------- QC layers ------------------------------------------------------
    qlayers, qmpnn = ref_import("QC", "layers", "mpnn")
    torch.manual_seed(4)
    n, e, h = 14, 24, 73
    g = torch.Generator().manual_seed(13)
    Esrc = torch.randint(0, n, (e,), generator=g)
    etgt = torch.randint(0, n, (e,), generator=g)
    Etgt = torch.zeros(n, e)
    Etgt[etgt, torch.arange(e)] = 1.0
    edge_data = (torch.randn(e, h, h, generator=g) * 0.1).requires_grad_(True)
    x = torch.randn(n, h, generator=g).requires_grad_(True)
    res = dict(Esrc=Esrc, etgt=etgt, n=n, x=x, edge_data=edge_data)
    for T in (1, 3):
        m = qmpnn.MPNN_enn_edge(5, h)
        m.set_T(T)
        x.grad = None
        edge_data.grad = None
        out = m(x, Esrc, Etgt, edge_data)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        res.update({"T%d__out" % T: out, "T%d__gout" % T: gout, "T%d__gx" % T: x.grad.clone(),
                    "T%d__gedge" % T: edge_data.grad.clone()})
        for k, p in m.update_net.state_dict().items():
            res["T%d__gru__%s" % (T, k)] = p
        for k, p in m.update_net.named_parameters():
            res["T%d__ggru__%s" % (T, k)] = p.grad
    egc = qlayers.EdgeGraphConvolution(h, h)
    x.grad = None
    edge_data.grad = None
    out = egc(x, Esrc, Etgt, edge_data)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    res.update(egc__weight=egc.weight, egc__bias=egc.bias, egc__out=out, egc__gout=gout, egc__gx=x.grad,
               egc__gedge=edge_data.grad, egc__gw=egc.weight.grad, egc__gb=egc.bias.grad)
    save("qc_layers.npz", **res)

    # ---------------- scatter_add docstring KAT (QC/torch_scatter.py:207-218) ---------
    save("scatter_kat.npz", src=np.array([[2, 0, 1, 4, 3], [0, 2, 1, 3, 4]], dtype=np.float32),
         index=np.array([[4, 5, 4, 2, 3], [0, 0, 2, 2, 1]], dtype=np.int64),
         out=np.array([[0, 0, 4, 3, 3, 0], [2, 4, 4, 0, 0, 0]], dtype=np.float32))


def pubmed_topology():
    """C2: Pubmed's real topology through the dense-paper normalisation
    (GCN-dense-paper/utils.py:70-110: normalize_adj(adj + I)).  Features are absent from the reference
    checkout (`.MISSING_LARGE_BLOBS`), so only the graph is captured."""
    import pickle as pkl
    import networkx as nx
    import scipy.sparse as sp
    sys.path.insert(0, os.path.join(REF, "GCN-dense-paper"))
    for n in ("utils",):
        sys.modules.pop(n, None)
    utils = importlib.import_module("utils")
    sys.path.pop(0)
    with open(os.path.join(REF, "data", "ind.pubmed.graph"), "rb") as f:
        graph = pkl.load(f, encoding="latin1")
    adj = nx.adjacency_matrix(nx.from_dict_of_lists(graph))
    adj = utils.normalize_adj(adj + sp.eye(adj.shape[0])).tocoo().astype(np.float32)
    save("pubmed_graph_sym.npz", rows=adj.row.astype(np.int32), cols=adj.col.astype(np.int32), vals=adj.data,
         n=adj.shape[0])


if __name__ == "__main__":
    if os.environ.get("GOLDEN_ONLY") == "pubmed":
        sys.dont_write_bytecode = True
        import scipy.sparse.linalg as spla
        alias = types.ModuleType("scipy.sparse.linalg.eigen.arpack")
        alias.eigsh = spla.eigsh
        sys.modules.setdefault("scipy.sparse.linalg.eigen", types.ModuleType("scipy.sparse.linalg.eigen"))
        sys.modules["scipy.sparse.linalg.eigen.arpack"] = alias
        pubmed_topology()
    elif os.environ.get("GOLDEN_ONLY") not in ("qc_models", "set2set", "depth", "gat_zoo", "gat_heads", "variants", "gat_edges", "train_traj"):
        main()
        pubmed_topology()
        qc_models_golden()
        set2set_golden()
        depth_models_golden()
        gat_zoo_golden()
        variants_golden()


def qc_models_golden():
    """QC model zoo (QC/layer_models.py) on a small synthetic batch: outputs and a few gradients of the
    reference classes themselves (hidden 16 to keep the fixture small)."""
    sys.path.insert(0, os.path.join(ROOT_REPO))
    from graph_odenet_amd.synth import qm9_like_batch
    for n in ("layers", "models", "mpnn", "set2set", "layer_models", "torch_scatter", "torch_geometric_utils"):
        sys.modules.pop(n, None)
    sys.path.insert(0, os.path.join(REF, "QC"))
    try:
        lm = importlib.import_module("layer_models")
    finally:
        sys.path.pop(0)
    x, ef, Esrc, Etgt, batch = qm9_like_batch(4, seed=3)
    res = dict(x=x, ef=ef, Esrc=Esrc, etgt=Etgt.argmax(0), batch=batch, n=x.shape[0])
    torch.manual_seed(21)
    for name in ("MPNN_ENN_K_Sum", "MPNN_ENN_K_Set2Set", "EdgeGCN_K_Sum", "EdgeGCN_K_Set2Set", "EdgeRES1_K_Set2Set"):
        m = getattr(lm, name)(node_features=13, edge_features=5, target_features=12, hidden_features=16, num_layers=3,
                              s2s_processing_steps=3, dropout=0.0)
        m.eval()
        out = m(x, ef, Esrc, Etgt, batch)
        gout = torch.randn_like(out)
        out.backward(gout)
        res[name + "__out"] = out
        res[name + "__gout"] = gout
        for k, p in m.state_dict().items():
            res[name + "__sd__" + k.replace(".", "__")] = p
        for k, p in m.named_parameters():
            if p.grad is not None and p.numel() <= 600:          # small ones only: biases, norms, output layers
                res[name + "__g__" + k.replace(".", "__")] = p.grad
    # the fixed-depth classes of QC/models.py (train_egcn_multitask.py); EdgeGCN3_* read self.type without setting it
    sys.modules.pop("models", None)
    sys.path.insert(0, os.path.join(REF, "QC"))
    try:
        qm = importlib.import_module("models")
    finally:
        sys.path.pop(0)
    for name, kw in (("MPNN_ENN_Sum", {}), ("MPNN_ENN_Set2Set", dict(processing_steps=3)), ("EdgeGCN3_Sum", {}),
                     ("EdgeGCN3_Set2Set", dict(processing_steps=3))):
        m = getattr(qm, name)(13, 5, 16, 12, **kw)
        if not hasattr(m, "type"):
            m.type = "regression"
        m.eval()
        out = m(x, ef, Esrc, Etgt, batch)
        gout = torch.randn_like(out)
        out.backward(gout)
        res[name + "__out"] = out
        res[name + "__gout"] = gout
        for k, p in m.state_dict().items():
            res[name + "__sd__" + k.replace(".", "__")] = p
        for k, p in m.named_parameters():
            if p.grad is not None and p.numel() <= 600:
                res[name + "__g__" + k.replace(".", "__")] = p.grad
    save("qc_models.npz", **res)


def depth_models_golden():
    """Depth-sweep model family of GCN/train_layers.py (GCN/models.py:255-522): eval-mode log-probabilities and the
    gradient of the first layer's bias for every class at two depths, on a 60-node random graph, hidden 8."""
    stub = types.ModuleType("torchdiffeq"); stub.odeint_adjoint = None; stub.odeint = None
    sys.modules["torchdiffeq"] = stub
    (models,) = ref_import("GCN", "models")
    n, nfeat, nhid, ncls = 60, 12, 8, 4
    r, c, v = rand_graph(n, 300, seed=11)
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(n, nfeat, generator=gen)
    gout = torch.randn(n, ncls, generator=gen)
    res = dict(rows=r, cols=c, vals=v, n=n, x=x, gout=gout)
    torch.manual_seed(13)
    for name, depths in (("GCNK", (2, 4)), ("GCNKnorm", (2, 5)), ("RESK1", (3, 5)), ("RESK2", (4, 7)),
                         ("RESK1norm", (3, 6)), ("RESK2norm", (4, 7)), ("RESK", (5, 6)), ("RESKnorm", (5, 6))):
        for nl in depths:
            kw = dict(residue_layers=3) if name in ("RESK", "RESKnorm") else {}
            mdl = getattr(models, name)(nfeat=nfeat, nhid=nhid, nclass=ncls, dropout=0.5, nlayers=nl, **kw)
            mdl.eval()
            out = mdl(x, adj)
            out.backward(gout)
            key = "%s_%d" % (name, nl)
            res[key + "__out"] = out
            res[key + "__gbias0"] = mdl.gcs[0].bias.grad
            for k, p in mdl.state_dict().items():
                res[key + "__sd__" + k.replace(".", "__")] = p
    save("gcn_depth_models.npz", **res)


def gat_zoo_golden():
    """Non-ODE members of the GAT model zoo (GAT/models.py) on a 40-node / 160-edge random multigraph, hidden 8:
    eval-mode outputs and the gradient of the first layer's message bias."""
    stub = types.ModuleType("torchdiffeq"); stub.odeint_adjoint = None; stub.odeint = None
    sys.modules["torchdiffeq"] = stub
    (gmodels,) = ref_import("GAT", "models")
    n, E, nfeat, nhid, ncls = 40, 160, 10, 8, 3
    gen = torch.Generator().manual_seed(21)
    src = torch.randint(0, n, (E,), generator=gen)
    tgt = torch.randint(0, n - 4, (E,), generator=gen)          # the last 4 nodes receive nothing
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    x = torch.randn(n, nfeat, generator=gen)
    gout = torch.randn(n, ncls, generator=gen)
    res = dict(src=src, tgt=tgt, n=n, x=x, gout=gout)
    torch.manual_seed(22)
    for name, kw in (("GCN", {}), ("RGCN2", {}), ("GCN3", {}), ("GCN3norm", {}), ("RGCN3", {}), ("RGCN3norm", {}),
                     ("RGCN3fullnorm", {}), ("GCNK", dict(nlayers=4)), ("RESK2", dict(nlayers=5)),
                     ("RESK1norm", dict(nlayers=4))):
        mdl = getattr(gmodels, name)(nfeat=nfeat, nhid=nhid, nclass=ncls, dropout=0.5, **kw)
        mdl.eval()
        out = mdl(x, src, tgt, Mtgt)
        out.backward(gout)
        first = mdl.gcs[0] if hasattr(mdl, "gcs") else mdl.gc1
        res[name + "__out"] = out
        res[name + "__gbias0"] = first.f.bias.grad
        for k, p in mdl.state_dict().items():
            res[name + "__sd__" + k.replace(".", "__")] = p
    save("gat_zoo.npz", **res)


def gat_heads_golden():
    """H = 4 instances of the reference's GAT layer (GAT/layers.py GraphConvolution) on one 50-node / 220-edge random
    multigraph, outputs concatenated: forward, and the gradients of a random cotangent w.r.t. the input and every
    head's parameters.  The last nodes receive no edge; head 2 gets a large logit bias so that the heads' maxima differ
    by far more than eps resolves."""
    (glayers,) = ref_import("GAT", "layers")
    n, E, nin, H, o = 50, 220, 12, 4, 5
    gen = torch.Generator().manual_seed(31)
    src = torch.randint(0, n, (E,), generator=gen)
    tgt = torch.randint(0, n - 3, (E,), generator=gen)
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    x = torch.randn(n, nin, generator=gen).requires_grad_(True)
    gout = torch.randn(n, H * o, generator=gen)
    torch.manual_seed(32)
    layers = [glayers.GraphConvolution(nin, o) for _ in range(H)]
    with torch.no_grad():
        layers[2].w.bias.add_(7.5)
        layers[1].w.weight.mul_(4.0)
    out = torch.cat([l(x, src, tgt, Mtgt) for l in layers], 1)
    out.backward(gout)
    res = dict(src=src, tgt=tgt, n=n, x=x, gout=gout, out=out, gx=x.grad)
    for h, l in enumerate(layers):
        for k, p in l.named_parameters():
            res["h%d__%s" % (h, k.replace(".", "__"))] = p
            res["h%d__grad__%s" % (h, k.replace(".", "__"))] = p.grad
    save("gat_heads.npz", **res)


def variants_golden():
    """GCN-mlp-sum (MLP graph layer + one model) and GCN-dense-paper (one model; eval mode, so its input dropout is the
    identity) on a 50-node random graph: outputs, input / parameter gradients, state dicts."""
    stub = types.ModuleType("torchdiffeq"); stub.odeint_adjoint = None; stub.odeint = None
    sys.modules["torchdiffeq"] = stub
    n, nfeat, nhid, ncls = 50, 9, 8, 3
    r, c, v = rand_graph(n, 260, seed=31)
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    gen = torch.Generator().manual_seed(32)
    x = torch.randn(n, nfeat, generator=gen)
    res = dict(rows=r, cols=c, vals=v, n=n, x=x)
    torch.manual_seed(33)
    mlayers, mmodels = ref_import("GCN-mlp-sum", "layers", "models")
    lay = mlayers.GraphConvolution(nfeat, nhid)
    xi = x.clone().requires_grad_(True)
    out = lay(xi, adj)
    gout = torch.randn(out.shape, generator=gen)
    out.backward(gout)
    res.update(mlp_layer__out=out, mlp_layer__gout=gout, mlp_layer__gx=xi.grad)
    for k, p in lay.state_dict().items():
        res["mlp_layer__sd__" + k.replace(".", "__")] = p
    for k, p in lay.named_parameters():
        res["mlp_layer__g__" + k.replace(".", "__")] = p.grad
    f = mmodels.ODEfunc(nhid)
    f.set_adj(adj)
    h = torch.randn(n, nhid, generator=gen)
    res.update(mlp_odefunc__x=h, mlp_odefunc__out=f(torch.tensor(0.4), h))
    for k, p in f.state_dict().items():
        res["mlp_odefunc__sd__" + k.replace(".", "__")] = p
    gm = torch.randn(n, ncls, generator=gen)
    res["gmodel"] = gm
    for tag, mods, name, kw in (("mlp", mmodels, "RGCN3norm", {}), ("mlp", mmodels, "RESK2", dict(nlayers=5)),
                                ("dense", ref_import("GCN-dense-paper", "layers", "models")[1], "RGCN3fullnorm", {}),
                                ("dense", ref_import("GCN-dense-paper", "layers", "models")[1], "GCNK", dict(nlayers=3))):
        mdl = getattr(mods, name)(nfeat=nfeat, nhid=nhid, nclass=ncls, dropout=0.5, **kw)
        mdl.eval()
        o = mdl(x, adj)
        o.backward(gm)
        key = "%s_%s" % (tag, name)
        res[key + "__out"] = o
        for k, p in mdl.state_dict().items():
            res[key + "__sd__" + k.replace(".", "__")] = p
        k0, p0 = next(iter(mdl.named_parameters()))
        res[key + "__g0"] = p0.grad
    save("gcn_variants.npz", **res)


def gat_edges_golden():
    """Edge lists the GAT loader derives from the Planetoid graphs (GAT/utils.py:187-209), for both datasets present."""
    import scipy.sparse.linalg as spla
    alias = types.ModuleType("scipy.sparse.linalg.eigen.arpack")
    alias.eigsh = spla.eigsh
    sys.modules.setdefault("scipy.sparse.linalg.eigen", types.ModuleType("scipy.sparse.linalg.eigen"))
    sys.modules["scipy.sparse.linalg.eigen.arpack"] = alias
    cwd = os.getcwd()
    os.chdir(REF)                                  # the loaders open "data/ind.*" relative to the checkout root
    try:
        (gatutils,) = ref_import("GAT", "utils")
        for name in ("cora", "citeseer"):
            src, tgt, Mtgt, feats, labels, itr, iva, ite = gatutils.load_data_new(name)
            mi = Mtgt._indices()
            save("%s_gat_edges.npz" % name, src=src.to(torch.int32), tgt=tgt.to(torch.int32),
                 m_rows=mi[0].to(torch.int32), m_cols=mi[1].to(torch.int32), m_vals=Mtgt._values(), n=Mtgt.shape[0])
    finally:
        os.chdir(cwd)


def train_traj_golden():
    """Ten Adam steps (lr .01, wd 5e-4, dropout 0: no RNG involved) of the reference's own GCN3 / RGCN3norm (GCN) and
    GCN3 (GAT) on Cora, from a saved initial state: training-loss trajectory and final eval logits of 64 test nodes."""
    stub = types.ModuleType("torchdiffeq"); stub.odeint_adjoint = None; stub.odeint = None
    sys.modules["torchdiffeq"] = stub
    sys.path.insert(0, ROOT_REPO)
    from graph_odenet_amd.data import load_captured, load_captured_gat
    import torch.nn.functional as F
    res = {}
    for variant, names in (("GCN", ("GCN3", "RGCN3norm")), ("GAT", ("GCN3",))):
        (mods,) = ref_import(variant, "models")
        data = load_captured("cora") if variant == "GCN" else load_captured_gat("cora")
        *graph, x, y, itr, iva, ite = data
        for name in names:
            torch.manual_seed(5)
            m = getattr(mods, name)(nfeat=x.shape[1], nhid=16, nclass=7, dropout=0.0)
            key = "%s_%s" % (variant, name)
            for k, p in m.state_dict().items():
                res[key + "__sd__" + k.replace(".", "__")] = p.clone()
            opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4)
            losses = []
            for _ in range(10):
                m.train(); opt.zero_grad()
                loss = F.nll_loss(m(x, *graph)[itr], y[itr])
                loss.backward(); opt.step()
                losses.append(float(loss))
            m.eval()
            res[key + "__losses"] = torch.tensor(losses, dtype=torch.float64)
            res[key + "__logits"] = m(x, *graph)[ite[:64]]
    save("train_traj_cora.npz", **res)
    # QC: five Adam steps (lr 1e-3, QC/train_egcn.py:122) of two reference models on the 4-molecule synthetic batch
    from graph_odenet_amd.synth import qm9_like_batch
    for n in ("layers", "models", "mpnn", "set2set", "layer_models", "torch_scatter", "torch_geometric_utils"):
        sys.modules.pop(n, None)
    sys.path.insert(0, os.path.join(REF, "QC"))
    try:
        lm = importlib.import_module("layer_models")
    finally:
        sys.path.pop(0)
    x, ef, Esrc, Etgt, batch = qm9_like_batch(4, seed=3)
    torch.manual_seed(8)
    tgt = torch.randn(4, 12)
    res = dict(x=x, ef=ef, Esrc=Esrc, etgt=Etgt.argmax(0), batch=batch, n=x.shape[0], target=tgt)
    for name in ("EdgeGCN_K_Sum", "MPNN_ENN_K_Set2Set"):
        m = getattr(lm, name)(node_features=13, edge_features=5, target_features=12, hidden_features=16, num_layers=3,
                              s2s_processing_steps=3, dropout=0.0)
        for k, p in m.state_dict().items():
            res[name + "__sd__" + k.replace(".", "__")] = p.clone()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        losses = []
        m.train()
        for _ in range(5):
            opt.zero_grad()
            loss = F.mse_loss(m(x, ef, Esrc, Etgt, batch), tgt)
            loss.backward(); opt.step()
            losses.append(float(loss))
        res[name + "__losses"] = torch.tensor(losses, dtype=torch.float64)
        res[name + "__out"] = m(x, ef, Esrc, Etgt, batch)
    save("train_traj_qc.npz", **res)


def set2set_golden():
    """The reference's Set2Set module (QC/set2set.py:6-75) alone: h=24, 4 processing steps, 6 graphs of
    uneven size; inputs, lstm parameters, q_star and the gradients of x and of the lstm parameters."""
    for n in ("set2set", "torch_scatter", "torch_geometric_utils"):
        sys.modules.pop(n, None)
    sys.path.insert(0, os.path.join(REF, "QC"))
    try:
        s2s = importlib.import_module("set2set")
    finally:
        sys.path.pop(0)
    torch.manual_seed(33)
    sizes = [5, 1, 17, 9, 2, 30]
    batch = torch.cat([torch.full((n,), b, dtype=torch.int64) for b, n in enumerate(sizes)])
    x = torch.randn(batch.numel(), 24, requires_grad=True)
    m = s2s.Set2Set(24, 4, 1)
    out = m(x, batch)
    gout = torch.randn_like(out)
    out.backward(gout)
    res = dict(x=x.detach(), batch=batch, out=out.detach(), gout=gout, gx=x.grad)
    for k, p in m.state_dict().items():
        res["sd__" + k.replace(".", "__")] = p
    for k, p in m.named_parameters():
        res["g__" + k.replace(".", "__")] = p.grad
    save("set2set.npz", **res)


ROOT_REPO = os.path.dirname(os.path.dirname(OUT))
if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "train_traj":
    sys.dont_write_bytecode = True
    ROOT_REPO = os.path.dirname(os.path.dirname(OUT))
    train_traj_golden()
if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "gat_edges":
    sys.dont_write_bytecode = True
    gat_edges_golden()
if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "variants":
    sys.dont_write_bytecode = True
    variants_golden()
if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "gat_heads":
    gat_heads_golden()

if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "gat_zoo":
    sys.dont_write_bytecode = True
    gat_zoo_golden()
if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "depth":
    sys.dont_write_bytecode = True
    depth_models_golden()
if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "set2set":
    sys.dont_write_bytecode = True
    set2set_golden()
if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY") == "qc_models":
    sys.dont_write_bytecode = True
    stub = types.ModuleType("torchdiffeq"); stub.odeint_adjoint = None; stub.odeint = None
    sys.modules["torchdiffeq"] = stub
    qc_models_golden()
