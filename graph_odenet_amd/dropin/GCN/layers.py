"""Bare-name shim: `from layers import GraphConvolution, FixedGraphConvolution` (GCN/models.py:4)."""
from graph_odenet_amd.layers import *  # noqa: F401,F403
from graph_odenet_amd.layers import FixedGraphConvolution, GraphConvolution  # noqa: F401
