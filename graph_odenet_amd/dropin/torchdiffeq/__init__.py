"""Stands in for the third-party `torchdiffeq` package at the seam GCN/models.py:5
(`from torchdiffeq import odeint_adjoint as odeint`): same public signature, own solver."""
from graph_odenet_amd.odeint import odeint, odeint_adjoint  # noqa: F401
