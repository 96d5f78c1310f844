#!/usr/bin/env python3
"""Training-step times of BASELINE.json's other configurations (parity-test cases of the contract, timed here so that
the numbers come from the same box as bench.py's line):

  C1  Cora, ODEGCN3 (GCN layers, hidden 16), rk4 16 steps = 64 f-evals             (configs[0])
  C2  Pubmed topology, GCN-dense-paper ODEGCN3 (hidden 16), dopri5 rtol=atol=1e-5    (configs[1]; synthetic features)
  C3  Citeseer edge list, GAT ODEGCN3 with 8 heads x 8 (hidden 64) and one head (hidden 16), rk4 64 f-evals (configs[2])
  C4  QM9-like batches of 20 molecules, EdgeGCN_K_Sum / MPNN_ENN_K_Set2Set h=73 T=3, 16 distinct batches cycled (configs[3])

One step = forward + backward + Adam, as in the reference's training scripts.  `python tools/config_bench.py` prints one
JSON object; bench.py calls all_configs() for its "secondary" block.  Inputs come from tests/golden (captured graph
topologies) and graph_odenet_amd.synth."""
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")     # graph_odenet_amd/hipgraph.py (stand-alone runs)

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _time_steps(fn, warm=3, n=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r


def _trainer(m, fwd, idx, y, lr=0.01, wd=5e-4):
    from graph_odenet_amd.optim import Adam
    opt = Adam(m.parameters(), lr=lr, weight_decay=wd)

    def step():
        m.train(); opt.zero_grad(); m.nfe = 0
        out = fwd()
        nf = m.nfe; m.nfe = 0
        F.nll_loss(out[idx], y[idx]).backward(); opt.step()
        return nf, m.nfe
    return step


def c1_cora(dev):
    from graph_odenet_amd import models
    g = dict(np.load(os.path.join(GOLD, "cora_graph.npz")))
    n = int(g["n"])
    T = lambda a: torch.from_numpy(np.asarray(a))   # noqa: E731
    adj = torch.sparse_coo_tensor(torch.stack([T(g["rows"].astype(np.int64)), T(g["cols"].astype(np.int64))]), T(g["vals"]), (n, n)).to(dev)
    x = torch.zeros(n, int(g["n_feat"]))
    x[T(g["feat_rows"].astype(np.int64)), T(g["feat_cols"].astype(np.int64))] = T(g["feat_vals"])
    x, y, idx = x.to(dev), T(g["labels"].astype(np.int64)).to(dev), T(g["idx_train"].astype(np.int64)).to(dev)
    torch.manual_seed(0)
    m = models.ODEGCN3(nfeat=x.shape[1], nhid=16, nclass=7, dropout=0.5, method="rk4", step_size=1 / 16).to(dev)
    ms, (nf, nb) = _time_steps(_trainer(m, lambda: m(x, adj), idx, y))
    return {"workload": "Cora 2708 nodes, ODEGCN3 hidden 16, rk4 64 f-evals", "ms_per_step": round(ms, 3), "nfe_f": nf, "nfe_b": nb}


def c2_pubmed(dev):
    from graph_odenet_amd import dense_paper
    g = dict(np.load(os.path.join(GOLD, "pubmed_graph_sym.npz")))
    n = int(g["n"])
    idx = torch.stack([torch.from_numpy(g["rows"].astype(np.int64)), torch.from_numpy(g["cols"].astype(np.int64))])
    adj = torch.sparse_coo_tensor(idx, torch.from_numpy(g["vals"]), (n, n)).to(dev).to_dense()    # dense, as the reference holds it
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(n, 500, generator=gen) < 0.1).float()
    x = (x / x.sum(1, keepdim=True).clamp_min(1)).to(dev)
    y, tr = torch.randint(0, 3, (n,), generator=gen).to(dev), torch.arange(60, device=dev)
    torch.manual_seed(0)
    m = dense_paper.ODEGCN3(nfeat=500, nhid=16, nclass=3, dropout=0.5).to(dev)                    # default method: dopri5
    ms, (nf, nb) = _time_steps(_trainer(m, lambda: m(x, adj), tr, y))
    return {"workload": "Pubmed 19717 nodes (real topology, synthetic features), GCN-dense-paper ODEGCN3 hidden 16, dopri5 tol 1e-5",
            "ms_per_step": round(ms, 3), "nfe_f": nf, "nfe_b": nb}


def c3_citeseer_gat(dev, heads, nhid):
    from graph_odenet_amd import gat_heads, gat_models
    g = dict(np.load(os.path.join(GOLD, "citeseer_gat_edges.npz")))
    n = int(g["n"])
    src, tgt = torch.from_numpy(g["src"].astype(np.int64)).to(dev), torch.from_numpy(g["tgt"].astype(np.int64)).to(dev)
    e = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e, device=dev)]), torch.ones(e, device=dev), (n, e))
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(n, 3703, generator=gen) < 0.01).float()
    x = (x / x.sum(1, keepdim=True).clamp_min(1)).to(dev)
    y, idx = torch.randint(0, 6, (n,), generator=gen).to(dev), torch.arange(120, device=dev)
    zoo = gat_models if heads == 1 else gat_heads.zoo(heads)
    torch.manual_seed(0)
    m = zoo.ODEGCN3(nfeat=3703, nhid=nhid, nclass=6, dropout=0.5, method="rk4", step_size=1 / 16).to(dev)
    ms, (nf, nb) = _time_steps(_trainer(m, lambda: m(x, src, tgt, Mtgt), idx, y))
    return {"workload": "Citeseer 3327 nodes / %d edges, GAT ODEGCN3 %d head(s), hidden %d, rk4 64 f-evals" % (e, heads, nhid),
            "ms_per_step": round(ms, 3), "nfe_f": nf, "nfe_b": nb}


def c4_qc(dev, model_name, captured=False):
    from graph_odenet_amd import qc_models
    from graph_odenet_amd.synth import qm9_like_batch
    torch.manual_seed(0)
    net = getattr(qc_models, model_name)(node_features=13, edge_features=5, target_features=12, hidden_features=73,
                                         num_layers=3).to(dev)
    batches = []
    for b in range(16):
        x, ef, Esrc, Etgt, batch = qm9_like_batch(20, seed=b, device=dev)
        batches.append((x, ef, Esrc, Etgt, batch, torch.randn(20, 12, generator=torch.Generator().manual_seed(b)).to(dev)))
    from graph_odenet_amd.optim import Adam
    opt = Adam(net.parameters(), lr=1e-3)
    it = [0]

    def step():
        x, ef, Esrc, Etgt, batch, tgt = batches[it[0] % len(batches)]
        it[0] += 1
        opt.zero_grad()
        F.mse_loss(net(x, ef, Esrc, Etgt, batch), tgt).backward()
        opt.step()
    ms, _ = _time_steps(step, warm=len(batches), n=2 * len(batches))     # first pass: every batch shape seen once
    res = {"workload": "%s h=73 T=3, 20 QM9-like molecules per step, 16 distinct batches cycled (each shape seen once before "
                       "timing; tools/qc_bench.py times never-repeating batches)" % model_name,
           "ms_per_step": round(ms, 3), "graphs_per_s": round(20e3 / ms, 1)}
    if not captured:
        return res
    # the same batches padded to shape buckets, one HIP-graph replay per step (qc_step.CapturedQCStep); only when this
    # process honours replayed memset nodes (graph_odenet_amd/hipgraph.py).  Stand-alone runs only: bench.py keeps
    # graph captures of arbitrary autograd away from the process that has to print the contract line.
    try:
        from graph_odenet_amd import hipgraph
        from graph_odenet_amd.qc_batch import pad_batch
        from graph_odenet_amd.qc_step import CapturedQCStep
        if hipgraph.memset_nodes_ok(dev):
            padded = [pad_batch(*b[:5])[:5] + (b[5],) for b in batches]
            opt2 = Adam(net.parameters(), lr=1e-3)
            cstep = CapturedQCStep(net, opt2, F.mse_loss)
            jt = [0]

            def step2():
                cstep(*padded[jt[0] % len(padded)])
                jt[0] += 1
            ms2, _ = _time_steps(step2, warm=4 * len(padded), n=2 * len(padded))
            res["ms_per_step_captured"] = round(ms2, 3)
            res["shape_buckets"] = len(cstep.buckets)
        else:
            res["ms_per_step_captured"] = None
    except Exception as e:
        res["ms_per_step_captured"] = "error: %s: %s" % (type(e).__name__, e)
    return res


def all_configs(dev, qc_captured=False):
    out = {}
    for key, fn in (("C1_cora_gcn_ode_rk4", lambda: c1_cora(dev)),
                    ("C2_pubmed_dense_paper_ode_dopri5", lambda: c2_pubmed(dev)),
                    ("C3_citeseer_gat_8head_ode_rk4", lambda: c3_citeseer_gat(dev, 8, 64)),
                    ("C3_citeseer_gat_1head_ode_rk4", lambda: c3_citeseer_gat(dev, 1, 16)),
                    ("C4_qc_edge_gcn_sum", lambda: c4_qc(dev, "EdgeGCN_K_Sum", qc_captured)),
                    ("C4_qc_mpnn_enn_set2set", lambda: c4_qc(dev, "MPNN_ENN_K_Set2Set", qc_captured))):
        try:
            out[key] = fn()
        except Exception as e:                     # a secondary number must never take the contract line down
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    print(json.dumps(all_configs(torch.device("cuda:0"), qc_captured=True), indent=1))
