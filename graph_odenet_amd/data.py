"""Planetoid citation-graph inputs for the harness.

Two sources, same return value `(adj, features, labels, idx_train, idx_val, idx_test)` as the
reference's `load_data_new` (GCN/utils.py:134-202):

  * load_planetoid(name, data_dir): own reader of the `ind.<name>.*` pickle files (the reference's
    reader no longer imports on current scipy/numpy).  Steps: stack allx/tx, undo the
    test-index permutation, pad Citeseer's isolated test nodes, row-normalise the features, build
    A + I from the adjacency lists and normalise it (`row`: D^-1(A+I) as GCN/utils.py:186,205-212;
    `sym`: D^-1/2 (A+I) D^-1/2 as GCN-dense-paper/utils.py:104-110; `sum`: A+I as GCN-sum/utils.py).
  * load_captured(name): the outputs of the reference loader captured in tests/golden/*_graph.npz
    (Cora, Citeseer) for machines without the raw files.
"""
import os
import pickle
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden")


def _coo(rows, cols, vals, n):
    idx = torch.from_numpy(np.vstack([rows, cols]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(np.asarray(vals, dtype=np.float32)), (n, n))


def load_captured(name):
    path = os.path.join(GOLDEN, "%s_graph.npz" % name)
    if not os.path.exists(path):
        raise FileNotFoundError("no captured loader output for %r (have: cora, citeseer)" % name)
    g = np.load(path)
    n = int(g["n"])
    feats = torch.zeros(n, int(g["n_feat"]))
    feats[torch.from_numpy(g["feat_rows"].astype(np.int64)), torch.from_numpy(g["feat_cols"].astype(np.int64))] = \
        torch.from_numpy(g["feat_vals"])
    t = lambda k: torch.from_numpy(g[k].astype(np.int64))   # noqa: E731
    return _coo(g["rows"], g["cols"], g["vals"], n), feats, t("labels"), t("idx_train"), t("idx_val"), t("idx_test")


def undirected_edge_list(graph):
    """Each undirected edge of a {node: [neighbours]} dict ONCE, in the order the reference's GAT loader gets them from
    networkx (GAT/utils.py:187-189: `nx.from_dict_of_lists(graph)` then `G.edges`): nodes in the dict's order, each
    with its not-yet-finished neighbours in the order they were first attached.  (Messages therefore flow in one
    direction per undirected edge - a property of the reference.)"""
    adj = {}
    for u in graph:
        adj.setdefault(u, {})
    seen = set()
    for u, nbrs in graph.items():                 # from_dict_of_lists: an edge enters when its first endpoint comes up
        for v in nbrs:
            if v not in seen:
                adj.setdefault(v, {})
                adj[u][v] = True
                adj[v][u] = True
        seen.add(u)
    done, edges = set(), []
    for u, nbrs in adj.items():                   # Graph.edges: (u, v) for v in adj[u] unless v is already finished
        for v in nbrs:
            if v not in done:
                edges.append((u, v))
        done.add(u)
    return np.asarray(edges, dtype=np.int64).reshape(-1, 2)


def _gat_tuple(edges, n, rest):
    src, tgt = torch.from_numpy(edges[:, 0].copy()), torch.from_numpy(edges[:, 1].copy())
    e = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e)]), torch.ones(e), (n, e))
    return (src, tgt, Mtgt) + tuple(rest)


def load_captured_gat(name):
    """(src, tgt, Mtgt, features, labels, idx_train, idx_val, idx_test) as GAT/utils.py:load_data_new returns them, from
    the captured outputs (tests/golden/<name>_gat_edges.npz + <name>_graph.npz)."""
    path = os.path.join(GOLDEN, "%s_gat_edges.npz" % name)
    if not os.path.exists(path):
        raise FileNotFoundError("no captured GAT edge list for %r (have: cora, citeseer)" % name)
    g = np.load(path)
    edges = np.stack([g["src"].astype(np.int64), g["tgt"].astype(np.int64)], 1)
    return _gat_tuple(edges, int(g["n"]), load_captured(name)[1:])


def load_planetoid_gat(name, data_dir):
    """The same from the raw `ind.<name>.*` files (own reader + undirected_edge_list)."""
    with open(os.path.join(data_dir, "ind.%s.graph" % name), "rb") as f:
        graph = pickle.load(f, encoding="latin1")
    rest = load_planetoid(name, data_dir)[1:]
    return _gat_tuple(undirected_edge_list(graph), rest[0].shape[0], rest)


def load_planetoid(name, data_dir, norm="row"):
    import scipy.sparse as sp

    def rd(suffix):
        with open(os.path.join(data_dir, "ind.%s.%s" % (name, suffix)), "rb") as f:
            return pickle.load(f, encoding="latin1") if sys.version_info[0] >= 3 else pickle.load(f)
    x, y, tx, ty, allx, ally, graph = (rd(s) for s in ("x", "y", "tx", "ty", "allx", "ally", "graph"))
    test_idx = np.array([int(line) for line in open(os.path.join(data_dir, "ind.%s.test.index" % name))])
    lo, hi = test_idx.min(), test_idx.max()
    n_test_span = hi - lo + 1
    if tx.shape[0] != n_test_span:                      # Citeseer: isolated test nodes have no row in tx/ty
        tx_full = sp.lil_matrix((n_test_span, tx.shape[1]))
        ty_full = np.zeros((n_test_span, ty.shape[1]))
        tx_full[np.sort(test_idx) - lo, :] = tx
        ty_full[np.sort(test_idx) - lo, :] = ty
        tx, ty = tx_full, ty_full
    feats = sp.vstack((allx, tx)).tolil()
    labels = np.vstack((ally, ty))
    order = np.sort(test_idx)
    feats[test_idx, :] = feats[order, :]
    labels[test_idx, :] = labels[order, :]
    n = feats.shape[0]
    # features: D^-1 X
    feats = sp.csr_matrix(feats, dtype=np.float64)
    rs = np.asarray(feats.sum(1)).ravel()
    inv = np.divide(1.0, rs, out=np.zeros_like(rs), where=rs != 0)
    feats = sp.diags(inv).dot(feats)
    # adjacency from the neighbour lists (undirected, duplicates collapse, self loops as listed) + I
    r = np.concatenate([np.full(len(v), k, dtype=np.int64) for k, v in graph.items()])
    c = np.concatenate([np.asarray(v, dtype=np.int64) for v in graph.values()])
    a = sp.coo_matrix((np.ones(r.size), (r, c)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.float64)
    # networkx keeps a listed self loop as weight 1; the reference adds I on top (2 on Citeseer self loops)
    a = (a + sp.eye(n)).tocsr()
    deg = np.asarray(a.sum(1)).ravel()
    if norm == "row":
        a = sp.diags(np.divide(1.0, deg, out=np.zeros_like(deg), where=deg != 0)).dot(a)
    elif norm == "sym":
        d = np.power(deg, -0.5, out=np.zeros_like(deg), where=deg != 0)
        a = sp.diags(d).dot(a).dot(sp.diags(d))
    elif norm != "sum":
        raise ValueError("norm must be row, sym or sum")
    a = a.tocoo()
    adj = _coo(a.row, a.col, a.data, n)
    features = torch.from_numpy(np.asarray(feats.todense(), dtype=np.float32))
    lab = torch.from_numpy(np.argmax(labels, axis=1).astype(np.int64))
    n_lab = y.shape[0]
    return (adj, features, lab, torch.arange(n_lab), torch.arange(n_lab, n_lab + 500),
            torch.from_numpy(order.astype(np.int64)))
