"""Bare-name shim for QC/set2set.py (`from set2set import Set2Set`)."""
from graph_odenet_amd.qc_models import Set2Set  # noqa: F401
