"""Bare-name shim for QC/torch_scatter.py (`from torch_scatter import scatter_add`)."""
from graph_odenet_amd.qc_models import scatter_add  # noqa: F401
