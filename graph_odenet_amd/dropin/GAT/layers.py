"""Bare-name shim: `from layers import ...` (GAT/models.py:4)."""
from graph_odenet_amd.gat_layers import FixedGraphConvolution, GraphConvolution  # noqa: F401
