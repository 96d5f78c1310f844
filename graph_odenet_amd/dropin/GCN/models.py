"""Bare-name shim: `import models` (GCN/train_res.py:14)."""
from graph_odenet_amd.models import (GCN, GCN3, GCN3norm, GCNK, ODEBlock, ODEGCN2, ODEGCN3,  # noqa: F401
                                     ODEGCN3fullnorm, ODEK1, ODEK2, ODEfunc, ODEfunc2, RGCN2, RGCN3, RGCN3fullnorm,
                                     RGCN3norm)
