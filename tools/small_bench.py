#!/usr/bin/env python3
"""Small-graph (Cora, config C1) training-step timing: ODEGCN3 d=16/64, rk4 (64 evals) and dopri5,
product on the GPU vs the oracle on the host cores.  Development aid; prints one line per case."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_odenet_amd import models  # noqa: E402


def cora():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "cora_graph.npz")))
    n = int(g["n"])
    T = lambda a: torch.from_numpy(np.asarray(a))   # noqa: E731
    adj = torch.sparse_coo_tensor(torch.stack([T(g["rows"].astype(np.int64)), T(g["cols"].astype(np.int64))]), T(g["vals"]), (n, n))
    x = torch.zeros(n, int(g["n_feat"]))
    x[T(g["feat_rows"].astype(np.int64)), T(g["feat_cols"].astype(np.int64))] = T(g["feat_vals"])
    return adj, x, T(g["labels"].astype(np.int64)), T(g["idx_train"].astype(np.int64))


def main():
    dev = torch.device("cuda:0")
    adj, x, y, idx = cora()
    adj_g, x_g, y_g, idx_g = adj.to(dev), x.to(dev), y.to(dev), idx.to(dev)
    only = sys.argv[1:]                       # e.g. "16:rk4" - restrict to some cases (profiling)
    for nhid in (16, 64):
        for method, step in (("rk4", 1 / 16), (None, None)):
            if only and "%d:%s" % (nhid, method or "dopri5") not in only:
                continue
            torch.manual_seed(0)
            m = models.ODEGCN3(nfeat=x.shape[1], nhid=nhid, nclass=7, dropout=0.5, method=method, step_size=step).to(dev)
            from graph_odenet_amd.optim import Adam
            opt = Adam(m.parameters(), lr=0.01, weight_decay=5e-4)

            def train_step():
                m.train(); opt.zero_grad(set_to_none=False); m.nfe = 0
                out = m(x_g, adj_g)
                nf = m.nfe; m.nfe = 0
                loss = torch.nn.functional.nll_loss(out[idx_g], y_g[idx_g])
                loss.backward(); opt.step()
                return nf, m.nfe
            for _ in range(3):
                train_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_it = 10
            for _ in range(n_it):
                nf, nb = train_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_it
            print("cora nhid=%-3d method=%-6s  %.2f ms/step  (nfe_f %d, nfe_b %d)  %.1f steps/s" % (
                nhid, method or "dopri5", dt * 1e3, nf, nb, 1 / dt), flush=True)


if __name__ == "__main__":
    main()
