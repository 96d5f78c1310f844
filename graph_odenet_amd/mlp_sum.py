"""The GCN-mlp-sum variant of the reference (GCN-mlp-sum/layers.py, models.py): the graph layer aggregates an MLP of
the features,  out = A @ (relu(x W1 + b1) W2 + b2)  with both widths equal to out_features and no bias after the
aggregation; the model zoo is GCN/models.py's.  The two dense products run on the rectangular fp32-MFMA kernels (functional.dense, csrc/rect.hip), the aggregation is
gode_spmm_csr_f32; the ODE functions of this variant take the solver's generic (autograd) field."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.module import Module
from torch.nn.parameter import Parameter

from . import models
from .functional import graph_aggregate
from .graph import as_graph
from .models import ODEBlock, _gn  # noqa: F401


class MyLinear(Module):
    """x W + b with W stored in x out (GCN-mlp-sum/layers.py:10-30)."""

    def __init__(self, in_features, out_features, bias=True):
        super(MyLinear, self).__init__()
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input):
        from .functional import dense
        return dense(input, self.weight, self.bias)


class NonLinear(Module):
    def __init__(self, in_features, out_features, bias=True, f=F.relu):
        super(NonLinear, self).__init__()
        self.linear = MyLinear(in_features, out_features, bias=bias)
        self.bias = bias
        self.f = f

    def forward(self, input):
        return self.f(self.linear(input))


class MLP(Module):
    """NonLinear layers of the given sizes, then a MyLinear to out_features (GCN-mlp-sum/layers.py:46-63; the
    out_features=None branch of the reference reads an undefined name and is refused here)."""

    def __init__(self, in_features, layer_sizes, out_features=None, bias=True):
        super(MLP, self).__init__()
        if out_features is None:
            raise ValueError("MLP: out_features is required")
        sizes = list(layer_sizes)
        ins = [in_features] + sizes[:-1]
        self.layers = nn.Sequential(*([NonLinear(i, o, bias=bias) for i, o in zip(ins, sizes)] +
                                      [MyLinear(sizes[-1], out_features, bias=bias)]))

    def forward(self, input):
        return self.layers(input)


class GraphConvolution(Module):
    def __init__(self, in_features, out_features, bias=True):
        super(GraphConvolution, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.mlp = MLP(in_features, [out_features], out_features, bias=bias)

    def forward(self, input, adj):
        return graph_aggregate(as_graph(adj), self.mlp(input), None, relu=False)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class FixedGraphConvolution(GraphConvolution):
    def __init__(self, in_features, out_features, bias=True):
        super(FixedGraphConvolution, self).__init__(in_features, out_features, bias)
        self.adj = torch.Tensor([[1]])

    def set_adj(self, adj):
        self.adj = adj

    def forward(self, input):
        return GraphConvolution.forward(self, input, self.adj)


class ODEfunc(nn.Module):
    """relu(gc1([t | norm1(x)])) (GCN-mlp-sum/models.py:161-179)."""

    def __init__(self, dim):
        super(ODEfunc, self).__init__()
        self.norm1 = _gn(dim)
        self.gc1 = FixedGraphConvolution(dim + 1, dim)
        self.nfe = 0

    def set_adj(self, adj):
        self.gc1.set_adj(adj)

    def forward(self, t, x):
        self.nfe += 1
        xn = self.norm1(x)
        return F.relu(self.gc1(torch.cat([torch.ones_like(xn[:, :1]) * t, xn], 1)))


class ODEfunc2(nn.Module):
    def __init__(self, dim, dropout):
        super(ODEfunc2, self).__init__()
        self.norm1, self.norm2 = _gn(dim), _gn(dim)
        self.gc1 = FixedGraphConvolution(dim + 1, dim)
        self.gc2 = FixedGraphConvolution(dim + 1, dim)
        self.dropout = dropout
        self.nfe = 0

    def set_adj(self, adj):
        self.gc1.set_adj(adj)
        self.gc2.set_adj(adj)

    def forward(self, t, x):
        self.nfe += 1
        tt = torch.ones_like(x[:, :1]) * t
        x = self.norm1(F.relu(self.gc1(torch.cat([tt, x], 1))))
        return self.norm2(F.relu(self.gc2(torch.cat([tt, x], 1))))


class MlpSumKit:
    GraphConvolution = GraphConvolution
    ODEfunc = ODEfunc
    ODEfunc2 = ODEfunc2
    input_dropout = False


models.rebind_zoo(globals(), __name__, MlpSumKit, what="GCN-mlp-sum")
