"""Bare-name shim for GCN-mlp-sum/ (MLP graph layer, GCN-mlp-sum/layers.py)."""
from graph_odenet_amd.mlp_sum import MLP, FixedGraphConvolution, GraphConvolution, MyLinear, NonLinear  # noqa: F401
