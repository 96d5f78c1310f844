"""Bare-name shim: `import models` (GAT/train_res.py:14)."""
from graph_odenet_amd.gat_models import GCN3, ODEBlock, ODEfunc, ODEGCN3  # noqa: F401
