"""Bare-name shim for GCN-sum/ (its layers.py is GCN/layers.py; only utils.py's normalisation differs)."""
from graph_odenet_amd.layers import *  # noqa: F401,F403
from graph_odenet_amd.layers import FixedGraphConvolution, GraphConvolution  # noqa: F401
