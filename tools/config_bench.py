#!/usr/bin/env python3
"""Training-step times of BASELINE.json's other configurations (parity-test cases of the contract, timed here so that
the numbers come from the same box as bench.py's line):

  C1  Cora, ODEGCN3 (GCN layers, hidden 16), rk4 16 steps = 64 f-evals             (configs[0])
  C2  Pubmed topology, GCN-dense-paper ODEGCN3 (hidden 16), dopri5 rtol=atol=1e-5    (configs[1]; synthetic features)
  C3  Citeseer edge list, GAT ODEGCN3 with 8 heads x 8 (hidden 64) and one head (hidden 16), rk4 64 f-evals (configs[2])
  C4  QM9-like batches of 20 molecules, EdgeGCN_K_Sum / MPNN_ENN_K_Set2Set h=73 T=3, a NEW batch every step (configs[3])

One step = forward + backward + Adam, as in the reference's training scripts.  Every entry also carries the CPU leg
(`cpu_baseline_ms_per_step`, `cores`, `host_cpu_count`, `cpu_sample`): the oracle's restatement of the same model
(oracle/models_ref.py) with the same weights, timed on this box's host cores in the same run.  `python tools/config_bench.py` prints one
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


def _time_steps(fn, warm=5, n=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r


def _cpu_fields(ms, sample):
    """The fields every configuration adds for its CPU leg: the oracle (oracle/models_ref.py = the reference's model
    restated from oracle/layers_ref.py + oracle/solver_ref.py, same weights) timed on this box's host cores."""
    return {"cpu_baseline_ms_per_step": round(ms, 2), "cores": torch.get_num_threads(), "host_cpu_count": os.cpu_count(),
            "cpu_kind": "port", "cpu_sample": sample}


def _median_ms(fn, reps=3):
    fn()                                            # one warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[len(ts) // 2]


def _cpu_full_step(loss_fn, params, lr, wd):
    """One reference-style training step on the host: zero_grad, forward, loss, backward, torch.optim.Adam
    (GCN/train_res.py:63-79, QC/util.py:146-211)."""
    opt = torch.optim.Adam(list(params.values()), lr=lr, weight_decay=wd)

    def step():
        opt.zero_grad()
        loss_fn().backward()
        opt.step()
    return _median_ms(step)


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


def c1_cora(dev, cpu=True):
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
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ms, (nf, nb) = _time_steps(_trainer(m, lambda: m(x, adj), idx, y))
    res = {"workload": "Cora 2708 nodes, ODEGCN3 hidden 16, rk4 64 f-evals", "ms_per_step": round(ms, 3), "nfe_f": nf, "nfe_b": nb}
    yield dict(res)              # GPU part done (all_configs times every configuration on the GPU before any CPU leg)
    if cpu:
        from oracle import models_ref as M
        p = M.leaves(sd0)
        xc, ac, yc, ic = x.cpu(), adj.cpu(), y.cpu(), idx.cpu()
        t = _cpu_full_step(lambda: F.nll_loss(M.odegcn3(p, xc, ac, 0.5, True, "rk4", 1 / 16)[0][ic], yc[ic]), p, 0.01, 5e-4)
        res.update(_cpu_fields(t, "whole training steps of the oracle's ODEGCN3 (rk4, 64 + 64 evaluations, Adam): 1 warm-up, median of 3"))
    yield res


def c2_pubmed(dev, cpu=True):
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
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ms, (nf, nb) = _time_steps(_trainer(m, lambda: m(x, adj), tr, y))
    res = {"workload": "Pubmed 19717 nodes (real topology, synthetic features), GCN-dense-paper ODEGCN3 hidden 16, dopri5 tol 1e-5",
           "ms_per_step": round(ms, 3), "nfe_f": nf, "nfe_b": nb}
    yield dict(res)              # GPU part done (all_configs times every configuration on the GPU before any CPU leg)
    if cpu:
        # bounded sample: one evaluation of the oracle's ODE function (dense 19717 x 19717 adjacency, as the reference
        # holds it) and one evaluation with its VJP, scaled by the evaluation counts of the GPU step above; the two
        # graph-convolution layers around the block timed once with their backward
        from oracle import layers_ref as R, models_ref as M
        p = M.leaves(sd0)
        ac, xc = adj.cpu(), x.cpu()
        fp = [p["gc2.odefunc.norm1.weight"], p["gc2.odefunc.norm1.bias"], p["gc2.odefunc.gc1.weight"], p["gc2.odefunc.gc1.bias"]]
        with torch.no_grad():
            h = torch.relu(R.graph_convolution(xc, ac, p["gc1.weight"], p["gc1.bias"]))
            t_f = _median_ms(lambda: R.odefunc(torch.tensor(0.3), h, ac, *fp))

        def fb():
            hg = h.clone().requires_grad_(True)
            out = R.odefunc(torch.tensor(0.3), hg, ac, *fp)
            out.backward(torch.ones_like(out))
        t_fb = _median_ms(fb)

        def layers():
            h1 = torch.relu(R.graph_convolution(xc, ac, p["gc1.weight"], p["gc1.bias"]))
            R.graph_convolution(h1, ac, p["gc3.weight"], p["gc3.bias"]).sum().backward()
        t_l = _median_ms(layers, reps=1)
        res.update(_cpu_fields(t_l + nf * t_f + nb * t_fb,
                               "oracle ODE function on the dense adjacency: f-eval %.1f ms, f-eval + VJP %.1f ms (1 warm-up, median of 3 each), "
                               "first + last layer forward and backward %.1f ms; scaled to the GPU step's %d forward and %d adjoint evaluations"
                               % (t_f, t_fb, t_l, nf, nb)))
    yield res


def c3_citeseer_gat(dev, heads, nhid, cpu=True):
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
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ms, (nf, nb) = _time_steps(_trainer(m, lambda: m(x, src, tgt, Mtgt), idx, y))
    res = {"workload": "Citeseer 3327 nodes / %d edges, GAT ODEGCN3 %d head(s), hidden %d, rk4 64 f-evals" % (e, heads, nhid),
           "ms_per_step": round(ms, 3), "nfe_f": nf, "nfe_b": nb}
    yield dict(res)              # GPU part done (all_configs times every configuration on the GPU before any CPU leg)
    if cpu:
        from oracle import models_ref as M
        p = M.leaves(sd0)
        xc, sc, tc, mc, yc, ic = x.cpu(), src.cpu(), tgt.cpu(), Mtgt.cpu(), y.cpu(), idx.cpu()
        nh = heads if heads > 1 else None
        t = _cpu_full_step(lambda: F.nll_loss(M.gat_odegcn3(p, xc, sc, tc, mc, nh, 0.5, True, "rk4", 1 / 16)[0][ic], yc[ic]),
                           p, 0.01, 5e-4)
        res.update(_cpu_fields(t, "whole training steps of the oracle's GAT ODEGCN3 (%d reference layer(s) per graph layer, rk4, 64 + 64 "
                                  "evaluations, Adam): 1 warm-up, median of 3" % heads))
    yield res


def c4_qc(dev, model_name, cpu=True, n_timed=100):
    """C4 as the reference trains it (QC/util.py:146-211): a NEW batch every step - every step meets node / edge counts
    it may never have seen, and the per-batch graph conversion is part of the step.  Three modes of the same loop body
    (graph_odenet_amd/qc_train.py): eager (dense Etgt, as the reference's collate hands it over), prepared (index
    vectors from the loader) and captured (one HIP-graph replay per shape bucket).  `ms_per_step` is the DEFAULT mode of
    qc_train.TrainStep ("prepared": index vectors taken off the dense matrix on the device, no host synchronisation) on
    the dense-Etgt batches the reference's collate emits; the eager / loader-prepared / captured times are beside it."""
    from graph_odenet_amd import hipgraph, qc_models
    from graph_odenet_amd.optim import Adam
    from graph_odenet_amd.qc_train import TrainStep
    from graph_odenet_amd.synth import qm9_like_batch
    n_warm = 30
    # ONE list of never-repeating batches, resident before any timing, walked once per mode (the captured mode walks the
    # warm-up part several times first: every shape bucket has to be met twice and captured)
    pool = []
    for b in range(n_warm + n_timed):
        x, ef, Esrc, Etgt, batch = qm9_like_batch(20, seed=5000 + b, device=dev)
        pool.append((x, ef, Esrc, Etgt, batch, torch.randn(20, 12, generator=torch.Generator().manual_seed(b)).to(dev)))

    def run(mode):
        torch.manual_seed(0)
        net = getattr(qc_models, model_name)(node_features=13, edge_features=5, target_features=12, hidden_features=73,
                                             num_layers=3).to(dev)
        step = TrainStep(net, Adam(net.parameters(), lr=1e-3), F.mse_loss, mode=mode)
        bs = pool
        if mode == "prepared":
            bs = [(b[0], b[1], b[2], b[3].argmax(0), b[4], b[5]) for b in pool]    # what a loader has before the dense matrix
        torch.cuda.synchronize()
        for rep in range(5 if mode == "captured" else 1):
            for b in bs[:n_warm]:
                step(*b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in bs[n_warm:]:
            step(*b)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n_timed
        extra = {}
        if step._captured is not None:
            extra = {"shape_buckets_seen": len(step._captured.buckets),
                     "shape_buckets_captured": sum(1 for b in step._captured.buckets.values() if b.graph is not None)}
        return ms, step.mode, extra, net

    res = {"workload": "%s h=73 T=3, 20 QM9-like molecules per step, a NEW batch every step (never-repeating shapes; the "
                       "per-batch graph conversion is timed), %d timed steps per mode" % (model_name, n_timed)}
    modes = ["eager", "prepared", "auto"] + (["captured"] if hipgraph.memset_nodes_ok(dev) else [])
    net = None
    run("eager")                                               # a throw-away pass: first-use costs of every kernel of the model
    for k, mode in enumerate(modes):
        try:
            ms, ran, extra, net = run(mode)
            if mode == "auto":                                 # the default of qc_train.TrainStep on dense-Etgt batches
                res["ms_per_step"], res["default_mode"] = round(ms, 3), ran
                res["graphs_per_s"] = round(20e3 / ms, 1)
            else:
                res["ms_per_step_" + ("loader_prepared" if mode == "prepared" else mode)] = round(ms, 3)
            res.update(extra)
        except Exception as e:
            res["ms_per_step" if mode == "auto" else "ms_per_step_" + mode] = "error: %s: %s" % (type(e).__name__, e)
    yield dict(res)              # GPU part done (all_configs times every configuration on the GPU before any CPU leg)
    if cpu and net is not None:
        from oracle import models_ref as M
        p = M.leaves(net.state_dict())
        x, ef, Esrc, Etgt, batch, tgt = (t.cpu() for t in pool[0])
        fn = M.QC_MODELS[model_name]
        kw = dict(training=True) if model_name == "EdgeGCN_K_Sum" else {}
        t = _cpu_full_step(lambda: F.mse_loss(fn(p, x, ef, Esrc, Etgt, batch, 20, **kw), tgt), p, 1e-3, 0.0)
        res.update(_cpu_fields(t, "whole training steps of the oracle's %s on ONE batch of 20 molecules (dense Etgt, bmm messages, "
                                  "torch Adam): 1 warm-up, median of 3" % model_name))
    yield res


def all_configs(dev, cpu=True, only=None):
    """Every configuration's GPU timing FIRST, then the CPU legs: a CPU leg is tens of seconds of 128-thread work, and the
    launch-bound configurations timed right after one came out 15-45 % slow (host clocks / thread pool still busy:
    EdgeGCN 1.72 ms against 1.17 ms in a process without CPU legs)."""
    out, gens = {}, {}
    table = (("C1_cora_gcn_ode_rk4", lambda: c1_cora(dev, cpu)),
             ("C2_pubmed_dense_paper_ode_dopri5", lambda: c2_pubmed(dev, cpu)),
             ("C3_citeseer_gat_8head_ode_rk4", lambda: c3_citeseer_gat(dev, 8, 64, cpu)),
             ("C3_citeseer_gat_1head_ode_rk4", lambda: c3_citeseer_gat(dev, 1, 16, cpu)),
             ("C4_qc_edge_gcn_sum", lambda: c4_qc(dev, "EdgeGCN_K_Sum", cpu)),
             ("C4_qc_mpnn_enn_set2set", lambda: c4_qc(dev, "MPNN_ENN_K_Set2Set", cpu)))
    for key, fn in table:
        if only and not any(key.startswith(o) for o in only):
            continue
        try:
            gens[key] = fn()
            out[key] = next(gens[key])                 # the GPU part
        except Exception as e:                         # a secondary number must never take the contract line down
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            gens.pop(key, None)
        torch.cuda.empty_cache()
    for key, g in gens.items():                        # the CPU legs (a generator without one simply ends)
        try:
            for res in g:
                out[key] = res
        except Exception as e:
            out[key]["cpu_error"] = "%s: %s" % (type(e).__name__, e)
    return out


if __name__ == "__main__":
    # stand-alone, and as bench.py's child process (a fresh process: HIP-graph captures of arbitrary autograd stay away
    # from the process that has to print the contract line): one JSON object on the LAST line of stdout
    cpu_leg = "--no-cpu" not in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]          # e.g. C4 or C2_pubmed (development)
    print(json.dumps(all_configs(torch.device("cuda:0"), cpu=cpu_leg, only=only), indent=None if "--one-line" in sys.argv else 1))
