"""graph_odenet_amd — MI355X-native hot path of phcavelar/graph-odenet.

Message passing (GCN SpMM, GAT edge-softmax aggregation, QC edge-conditioned messages)
integrated through an ODE residual block, as hand-written HIP kernels (csrc/, C ABI in
include/graphode.h) behind the reference's layers.py / models.py / torchdiffeq API.
"""
from . import _lib  # noqa: F401
from . import hipgraph as _hipgraph

# Replayed memset nodes are unreliable on this ROCm unless the HIP runtime's graph fast path is off (hipgraph.py; no
# measurable cost on the captured solves: Cora step 3.57 vs 3.62 ms).  Only effective - and only attempted - while the
# process has not made its first HIP call; an explicit setting of the variable is left alone.
_hipgraph.prefer_safe_graphs()

__all__ = ["_lib", "graph", "ops", "layers", "models", "odeint", "solver"]
