"""graph_odenet_amd — MI355X-native hot path of phcavelar/graph-odenet.

Message passing (GCN SpMM, GAT edge-softmax aggregation, QC edge-conditioned messages)
integrated through an ODE residual block, as hand-written HIP kernels (csrc/, C ABI in
include/graphode.h) behind the reference's layers.py / models.py / torchdiffeq API.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "graph", "ops", "layers", "models", "odeint", "solver"]
