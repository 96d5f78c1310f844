"""Bare-name shim for QC/layer_models.py:5 (`from mpnn import MPNN_enn_edge as MPNN_enn`)."""
from graph_odenet_amd.qc_layers import MPNN_enn_edge  # noqa: F401
