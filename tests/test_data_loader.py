"""Own Planetoid reader vs the loader outputs captured from the reference (tests/golden/*_graph.npz).
Needs the raw `ind.*` files, which exist only in the build container (skipped elsewhere)."""
import os

import numpy as np
import pytest
import torch

from graph_odenet_amd.data import load_captured, load_captured_gat, load_planetoid, load_planetoid_gat

RAW = "/root/reference/data"


@pytest.mark.parametrize("name", ["cora", "citeseer"])
def test_own_reader_matches_reference_loader(name):
    if not os.path.exists(os.path.join(RAW, "ind.%s.graph" % name)):
        pytest.skip("raw Planetoid files not present on this machine")
    adj, x, y, itr, iva, ite = load_planetoid(name, RAW)
    radj, rx, ry, ritr, riva, rite = load_captured(name)
    assert adj.shape == radj.shape
    assert torch.allclose(adj.to_dense(), radj.to_dense(), atol=1e-7)
    assert torch.allclose(x, rx, atol=1e-7)
    assert torch.equal(y, ry) and torch.equal(itr, ritr) and torch.equal(iva, riva) and torch.equal(ite, rite)


def test_captured_shapes():
    adj, x, y, itr, iva, ite = load_captured("cora")
    assert adj.shape == (2708, 2708) and adj._nnz() == 13264 and x.shape == (2708, 1433)
    assert itr.numel() == 140 and iva.numel() == 500 and ite.numel() == 1000 and int(y.max()) == 6


@pytest.mark.parametrize("name,n_edges", [("cora", 5278), ("citeseer", 4676)])
def test_own_gat_edge_list_matches_reference_loader(name, n_edges):
    """undirected_edge_list restates networkx's edge order (GAT/utils.py:187-189); compared with the edge lists captured
    from the reference's GAT loader."""
    src, tgt, Mtgt, x, y, itr, iva, ite = load_captured_gat(name)
    assert src.numel() == n_edges and Mtgt.shape == (x.shape[0], n_edges)
    if not os.path.exists(os.path.join(RAW, "ind.%s.graph" % name)):
        pytest.skip("raw Planetoid files not present on this machine")
    s2, t2, M2, x2, y2, *_ = load_planetoid_gat(name, RAW)
    assert torch.equal(s2, src) and torch.equal(t2, tgt)
    assert torch.equal(M2.coalesce().indices(), Mtgt.coalesce().indices())
    assert torch.allclose(x2, x, atol=1e-7) and torch.equal(y2, y)
