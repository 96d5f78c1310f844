"""Bare-name shim for QC/models.py (`import models`, QC/train_egcn_multitask.py:29)."""
from graph_odenet_amd.qc_models import EdgeGCN3_Set2Set, EdgeGCN3_Sum, MPNN_ENN_Set2Set, MPNN_ENN_Sum  # noqa: F401
