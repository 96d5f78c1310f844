"""Bare-name shim for GCN-sum/ (`import models`, GCN-sum/train_res.py)."""
from graph_odenet_amd.models import *  # noqa: F401,F403
from graph_odenet_amd.models import ODEBlock, ODEfunc, ODEfunc2  # noqa: F401
