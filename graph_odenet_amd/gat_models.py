"""GAT-family models (reference: GAT/models.py - the GCN zoo with (x, src, tgt, Mtgt) plumbing).
Same plan interpreter as graph_odenet_amd.models; only the layer type and the graph arguments differ."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .gat_layers import FixedGraphConvolution, GraphConvolution
from .models import ODEBlock, _PlanModel, _gn


class ODEfunc(nn.Module):
    """relu(gc1([t | norm1(x)])) with the edge-attention layer (reference: GAT/models.py:161-179)."""

    def __init__(self, dim):
        super(ODEfunc, self).__init__()
        self.norm1 = _gn(dim)
        self.gc1 = FixedGraphConvolution(dim + 1, dim)
        self.nfe = 0

    def set_adj(self, src, tgt, Mtgt):
        self.gc1.set_adj(src, tgt, Mtgt)

    _gode_counts_nfe = True

    def forward(self, t, x):
        self.nfe += 1
        xn = self.norm1(x)
        ttx = torch.cat([torch.ones_like(xn[:, :1]) * t, xn], 1)
        return F.relu(self.gc1(ttx))

    def gode_plan_token(self, y0):
        from .gat_layers import edge_graph
        layer = self.gc1
        if not torch.is_tensor(layer.src) or layer.src.dim() != 1:
            return None
        return ("gat", id(edge_graph(layer.src, layer.tgt, layer.Mtgt)))

    def gode_fields(self, y0):
        """Hook for graph_odenet_amd.odeint: fused forward / adjoint kernel sequences (gat_ode.py)."""
        from .gat_ode import gat_fields
        return gat_fields(self, y0)


class _GatPlan(_PlanModel):
    def forward(self, x, src, tgt, Mtgt):
        return self.run_plan(x, (src, tgt, Mtgt))


class GCN3(_GatPlan):
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("gc", "gc2"), ("relu",), ("drop",), ("gc", "gc3"))

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(GCN3, self).__init__()
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = GraphConvolution(nhid, nhid)
        self.gc3 = GraphConvolution(nhid, nclass)
        self.dropout = dropout


class ODEGCN3(_GatPlan):
    """GAT/models.py:204-226."""
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("gc", "gc2"), ("gc", "gc3"))
    ode_attr = "gc2"

    def __init__(self, nfeat, nhid, nclass, dropout, method=None, step_size=None, tol=1e-5):
        super(ODEGCN3, self).__init__()
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = ODEBlock(ODEfunc(nhid), tol=tol, method=method, step_size=step_size)
        self.gc3 = GraphConvolution(nhid, nclass)
        self.dropout = dropout
