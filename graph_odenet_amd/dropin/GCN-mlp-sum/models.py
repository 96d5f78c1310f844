"""Bare-name shim for GCN-mlp-sum/ (`import models`)."""
from graph_odenet_amd.mlp_sum import *  # noqa: F401,F403
from graph_odenet_amd.mlp_sum import ODEBlock, ODEfunc, ODEfunc2  # noqa: F401
