"""GAT-family models (reference: GAT/models.py - the GCN zoo with (x, src, tgt, Mtgt) plumbing).
Every class of graph_odenet_amd.models exists here under the same name, assembled from the edge-attention layers:
the model classes are kit-generic (models._PlanModel.kit), so this module defines the two ODE functions of the GAT
variant and re-binds the zoo to them."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .gat_layers import FixedGraphConvolution, GraphConvolution
from . import models as _gcn_models
from .models import ODEBlock, _gn


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


class ODEfunc2(nn.Module):
    """Two stacked (edge-attention layer -> relu -> GroupNorm) with the time column re-attached before each layer
    (reference: GAT/models.py ODEfunc2; the un-normalised state enters the first layer, as there)."""

    def __init__(self, dim, dropout):
        super(ODEfunc2, self).__init__()
        self.norm1, self.norm2 = _gn(dim), _gn(dim)
        self.gc1 = FixedGraphConvolution(dim + 1, dim)
        self.gc2 = FixedGraphConvolution(dim + 1, dim)
        self.dropout = dropout
        self.nfe = 0

    def set_adj(self, src, tgt, Mtgt):
        self.gc1.set_adj(src, tgt, Mtgt)
        self.gc2.set_adj(src, tgt, Mtgt)

    def forward(self, t, x):
        self.nfe += 1
        tt = torch.ones_like(x[:, :1]) * t
        x = self.norm1(F.relu(self.gc1(torch.cat([tt, x], 1))))
        return self.norm2(F.relu(self.gc2(torch.cat([tt, x], 1))))


class GatKit:
    """Layer classes of the GAT variant (reference: GAT/layers.py, GAT/models.py)."""
    GraphConvolution = GraphConvolution
    ODEfunc = ODEfunc
    ODEfunc2 = ODEfunc2
    input_dropout = False


def _forward(self, x, src, tgt, Mtgt):
    return self._body(self._input(x), (src, tgt, Mtgt))


_gcn_models.rebind_zoo(globals(), __name__, GatKit, forward=_forward, what="GAT")
