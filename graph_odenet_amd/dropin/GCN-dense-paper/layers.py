"""Bare-name shim for GCN-dense-paper/ (Glorot-initialised layers, GCN-dense-paper/layers.py:26-29)."""
from graph_odenet_amd.dense_paper import FixedGraphConvolution, GraphConvolution  # noqa: F401
