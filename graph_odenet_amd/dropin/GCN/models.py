"""Bare-name shim: `import models` (GCN/train_res.py:14)."""
from graph_odenet_amd.models import *  # noqa: F401,F403
from graph_odenet_amd.models import (GCN, GCN3, RGCN3, RGCN3norm, RGCN3fullnorm, ODEfunc, ODEBlock,  # noqa: F401
                                     ODEGCN3, ODEGCN3fullnorm, ODEfunc2, GCNK, ODEK1, ODEK2)
