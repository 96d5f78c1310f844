"""Bare-name shim: `import models` (GAT/train_res.py:14, whose model_dict names seven classes of the zoo)."""
from graph_odenet_amd.gat_models import *  # noqa: F401,F403
from graph_odenet_amd.gat_models import ODEBlock, ODEfunc, ODEfunc2  # noqa: F401
