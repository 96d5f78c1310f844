"""graph_odenet_amd — MI355X-native hot path of phcavelar/graph-odenet.

Message passing (GCN SpMM, GAT edge-softmax aggregation, QC edge-conditioned messages)
integrated through an ODE residual block, as hand-written HIP kernels (csrc/, C ABI in
include/graphode.h) behind the reference's layers.py / models.py / torchdiffeq API.
"""
from . import _lib  # noqa: F401

# Importing the package changes nothing outside it.  Replayed memset nodes are unreliable on this ROCm unless the HIP
# runtime's graph fast path is off (hipgraph.py): libgraphode's own captured solves launch no memset and are safe either
# way; a program that wants qc_step.CapturedQCStep to capture arbitrary autograd calls
# `graph_odenet_amd.hipgraph.prefer_safe_graphs()` (or exports DEBUG_CLR_GRAPH_PACKET_CAPTURE=0) before its first GPU
# call - the package's own entry points (train_res, train_layers, bench.py, tools/) do.  Until round 2 the import did
# that by itself, which silently changed the HIP graph path of every other library in the host process.

__all__ = ["_lib", "graph", "ops", "layers", "models", "odeint", "solver"]
