"""Bare-name shim for GCN-dense-paper/ (`import models`: input dropout in every model)."""
from graph_odenet_amd.dense_paper import *  # noqa: F401,F403
from graph_odenet_amd.dense_paper import ODEBlock, ODEfunc, ODEfunc2  # noqa: F401
