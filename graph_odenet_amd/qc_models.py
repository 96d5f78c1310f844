"""QC (QM9) model zoo with the reference's class names, constructor kwargs, sub-module names and
forward(node_features, edge_features, Esrc, Etgt, batch) API (reference: QC/layer_models.py:27-232),
on graph_odenet_amd.qc_layers.  Dense pieces (edge encoder, transition MLPs, GRU / LSTM cells) stay PyTorch
modules with the reference's nesting so that checkpoints load; the message passing runs on libgraphode and the
Set2Set readout replaces the reference's per-graph Python loop (QC/set2set.py:59-75) by a segment softmax.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .functional import GroupNorm
from .qc_layers import EdgeGraphConvolution, MPNN_enn_edge


# ---- dense helpers with the reference's module nesting (QC/layers.py:10-86) ------------------------
class MyLinear(nn.Module):
    """x @ W + b with W stored (in, out) and U(-1/sqrt(out), 1/sqrt(out)) init."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        bound = 1.0 / math.sqrt(out_features)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x):
        y = torch.mm(x, self.weight)
        return y if self.bias is None else y + self.bias


class NonLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True, f=F.relu):
        super().__init__()
        self.linear = MyLinear(in_features, out_features, bias=bias)
        self.f = f

    def forward(self, x):
        return self.f(self.linear(x))


class MLP(nn.Module):
    def __init__(self, in_features, layer_sizes, out_features, bias=True):
        super().__init__()
        sizes = [in_features] + list(layer_sizes)
        self.layers = nn.Sequential(*([NonLinear(a, b, bias=bias) for a, b in zip(sizes[:-1], sizes[1:])] +
                                      [MyLinear(sizes[-1], out_features, bias=bias)]))

    def forward(self, x):
        return self.layers(x)


class TransitionMLP(nn.Module):
    """One hidden layer of width (in+out)//2."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.mlp = MLP(in_features, [(in_features + out_features) // 2], out_features, bias=bias)

    def forward(self, x):
        return self.mlp(x)


class EdgeEncoderMLP(nn.Module):
    """edge features -> one h x h matrix per edge."""

    def __init__(self, edge_features, node_features, bias=True):
        super().__init__()
        self.mlp = TransitionMLP(edge_features, node_features * node_features, bias=bias)
        self.nf = node_features

    def forward(self, x):
        return self.mlp(x).reshape(x.size(0), self.nf, self.nf)


# ---- readouts ---------------------------------------------------------------------------------------
def segment_sum(x, batch, n_graphs):
    """scatter_add(x, batch, dim=0, dim_size=n_graphs) (QC/torch_scatter.py:170-221)."""
    return torch.zeros(n_graphs, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, batch, x)


class Set2Set(nn.Module):
    """Set2Set pooling (Vinyals et al. 2015) with the reference's parameters (`lstm`), vectorised over graphs."""

    def __init__(self, in_channels, processing_steps, num_layers=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, 2 * in_channels
        self.processing_steps, self.num_layers = processing_steps, num_layers
        self.lstm = nn.LSTM(self.out_channels, self.in_channels, num_layers)

    def forward(self, x, batch):
        nb = int(batch.max().item()) + 1
        h = (x.new_zeros(self.num_layers, nb, self.in_channels), x.new_zeros(self.num_layers, nb, self.in_channels))
        q_star = x.new_zeros(nb, self.out_channels)
        for _ in range(self.processing_steps):
            q, h = self.lstm(q_star.unsqueeze(0), h)
            q = q.view(nb, self.in_channels)
            e = (x * q[batch]).sum(-1)
            # per-graph softmax: shift by the graph's maximum, normalise by the graph's sum
            emax = torch.full((nb,), -float("inf"), dtype=x.dtype, device=x.device).scatter_reduce(0, batch, e, "amax")
            w = torch.exp(e - emax[batch])
            a = w / torch.zeros(nb, dtype=x.dtype, device=x.device).index_add_(0, batch, w)[batch]
            r = segment_sum(a.unsqueeze(1) * x, batch, nb)
            q_star = torch.cat([q, r], -1)
        return q_star

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


def _output_function(kind, target_features):
    if kind == "regression" or target_features == 1:
        return lambda x: x
    return lambda x: F.log_softmax(x, dim=1)


# ---- models -------------------------------------------------------------------------------------------
class _QCBase(nn.Module):
    def _finish(self, kind, target_features):
        self.type = kind
        self.output_function = _output_function(kind, target_features)


class MPNN_ENN_K_Sum(_QCBase):
    def __init__(self, node_features=None, edge_features=None, target_features=1, hidden_features=73, num_layers=3,
                 s2s_processing_steps=12, type="regression", dropout=0.5, **kwargs):
        super().__init__()
        self.input = nn.Linear(node_features, hidden_features)
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)
        self.mpnn = MPNN_enn_edge(edge_features, hidden_features)
        self.mpnn.set_T(num_layers)
        self.output = nn.Linear(hidden_features, target_features)
        self._finish(type, target_features)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        nb = int(batch.max().item()) + 1
        x = self.mpnn(self.input(node_features), Esrc, Etgt, self.ee(edge_features))
        return self.output_function(segment_sum(self.output(x), batch, nb))


class MPNN_ENN_K_Set2Set(_QCBase):
    def __init__(self, node_features=None, edge_features=None, target_features=1, hidden_features=73, num_layers=3,
                 s2s_processing_steps=12, type="regression", dropout=0.5, **kwargs):
        super().__init__()
        self.input = nn.Linear(node_features, hidden_features)
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)
        self.mpnn = MPNN_enn_edge(edge_features, hidden_features)
        self.mpnn.set_T(num_layers)
        self.s2s = Set2Set(hidden_features, s2s_processing_steps, num_layers=1)
        self.output = nn.Linear(hidden_features, target_features)
        self._finish(type, target_features)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self.mpnn(self.input(node_features), Esrc, Etgt, self.ee(edge_features))
        x = self.s2s(x, batch)[:, :x.size(1)]
        return self.output_function(self.output(x))


class _EdgeGCNStack(_QCBase):
    def _stack(self, x, Esrc, Etgt, ef):
        for gc in self.gcmid[:-1]:
            x = F.dropout(F.relu(gc(x, Esrc, Etgt, ef)), self.dropout, training=self.training)
        return self.gcmid[-1](x, Esrc, Etgt, ef)

    def _init_stack(self, node_features, edge_features, target_features, hidden_features, num_layers, dropout):
        self.mlpin = TransitionMLP(node_features, hidden_features)
        self.gcmid = nn.ModuleList([EdgeGraphConvolution(hidden_features, hidden_features) for _ in range(num_layers)])
        self.mlpout = TransitionMLP(hidden_features, target_features)
        self.dropout = dropout
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)


class EdgeGCN_K_Sum(_EdgeGCNStack):
    def __init__(self, node_features=None, edge_features=None, target_features=1, hidden_features=73, num_layers=3,
                 s2s_processing_steps=12, type="regression", dropout=0.5, **kwargs):
        super().__init__()
        self._init_stack(node_features, edge_features, target_features, hidden_features, num_layers, dropout)
        self._finish(type, target_features)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        nb = int(batch.max().item()) + 1
        x = self._stack(self.mlpin(node_features), Esrc, Etgt, self.ee(edge_features))
        return self.output_function(segment_sum(self.mlpout(x), batch, nb))


class EdgeGCN_K_Set2Set(_EdgeGCNStack):
    def __init__(self, node_features=None, edge_features=None, target_features=1, hidden_features=73, num_layers=3,
                 s2s_processing_steps=12, type="regression", dropout=0.5, **kwargs):
        super().__init__()
        self._init_stack(node_features, edge_features, target_features, hidden_features, num_layers, dropout)
        self.s2s = Set2Set(hidden_features, s2s_processing_steps, num_layers=1)
        self._finish(type, target_features)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self._stack(self.mlpin(node_features), Esrc, Etgt, self.ee(edge_features))
        x = self.s2s(x, batch)[:, :x.size(1)]
        return self.output_function(self.mlpout(x))


class RESKnorm(nn.Module):
    """Residual stack of edge-conditioned convolutions with GroupNorm (QC/layer_models.py:203-232)."""

    def __init__(self, nfeat, nhid, nclass, nlayers=3, residue_layers=1):
        super().__init__()
        if nlayers < 2 + residue_layers:
            raise ValueError("Can't make a Residual GCN with less than {} layers using {} layers for each residual block"
                             .format(2 + residue_layers, residue_layers))
        self.n_layers = nlayers
        self.gcs = nn.ModuleList([EdgeGraphConvolution(nfeat, nhid)] +
                                 [EdgeGraphConvolution(nhid, nhid) for _ in range(nlayers - 2)] +
                                 [EdgeGraphConvolution(nhid, nclass)])
        self.norms = nn.ModuleList([GroupNorm(min(32, nhid), nhid) for _ in range(nlayers - 2)])
        self.residue_layers = residue_layers

    def forward(self, x, Esrc, Etgt, ef):
        countdown, r = 1, None
        for gc, norm in zip(self.gcs[:-1], self.norms):      # zip stops at len(norms) = nlayers - 2, as in the reference
            countdown -= 1
            if countdown == 0:
                r, countdown = x, self.residue_layers
            x = norm(F.relu(gc(x, Esrc, Etgt, ef)))
            if countdown == 1:
                x = x + r
        if countdown > 1:
            x = x + r
        return self.gcs[-1](x, Esrc, Etgt, ef)


class EdgeRES1_K_Set2Set(_QCBase):
    def __init__(self, node_features=None, edge_features=None, target_features=1, hidden_features=73, num_layers=3,
                 s2s_processing_steps=12, type="regression", dropout=0.5, **kwargs):
        super().__init__()
        self.mlpin = TransitionMLP(node_features, hidden_features)
        self.gcmid = RESKnorm(hidden_features, hidden_features, hidden_features, nlayers=num_layers, residue_layers=1)
        self.mlpout = TransitionMLP(hidden_features, target_features)
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)
        self.s2s = Set2Set(hidden_features, s2s_processing_steps, num_layers=1)
        self._finish(type, target_features)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self.gcmid(self.mlpin(node_features), Esrc, Etgt, self.ee(edge_features))
        x = self.s2s(x, batch)[:, :x.size(1)]
        return self.output_function(self.mlpout(x))
