"""Bare-name shim for QC/layer_models.py:7 (`from layers import ..., EdgeGraphConvolution`)."""
from graph_odenet_amd.qc_layers import EdgeGraphConvolution  # noqa: F401
