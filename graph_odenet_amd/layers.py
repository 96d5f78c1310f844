"""GCN-family graph layers with the reference's names, constructor signatures, parameter
names, initialisation and forward API (GCN/layers.py:9-83 == GCN-sum/layers.py; the
dense-adjacency variant of GCN-dense-paper/layers.py differs only in init), computing on
the HIP kernels of libgraphode.so.

    GraphConvolution(in_features, out_features, bias=True).forward(input, adj)
    FixedGraphConvolution(in_features, out_features, bias=True).set_adj(adj); .forward(input)

`adj` may be an (uncoalesced) sparse COO tensor, a dense N x N tensor or a CSRGraph; it is
normalised once per tensor object (graph.as_graph).
"""
import math

import torch
from torch.nn.modules.module import Module
from torch.nn.parameter import Parameter

from . import ops
from .graph import as_graph


_SQUARE = (16, 32, 64, 128)


def _xt_g(x, g):
    """x^T g for tall x (n x d) and g (n x c): the weight gradient of a rectangular layer.  The library GEMM of this image
    runs the reduction over 2^20 rows in ONE workgroup column (1.25 ms for 128 x 16 outputs); as a batched product over
    256 row blocks plus a sum it takes 0.12 ms (tools/dev/narrow_mm_probe.py)."""
    n = x.shape[0]
    if n < 65536 or not (x.is_contiguous() and g.is_contiguous()):
        return torch.mm(x.t(), g)
    B = 256
    m = n // B
    out = torch.bmm(x[:m * B].view(B, m, x.shape[1]).transpose(1, 2), g[:m * B].view(B, m, g.shape[1])).sum(0)
    if m * B < n:
        out += torch.mm(x[m * B:].t(), g[m * B:])
    return out


class _GraphConvFn(torch.autograd.Function):
    """output = A @ (input @ W) + bias   (GCN/layers.py:31-37)."""

    @staticmethod
    def forward(ctx, graph, x, weight, bias):
        # square layers of the widths the fused kernels cover run X W on the exact-fp32 MFMA kernel of the ODE function
        # (no GroupNorm, no time row): 0.43 ms at 2^20 x 128 x 128 against 1.9 ms for the library GEMM of this image
        ctx.square = (x.shape[1] == weight.shape[1] and weight.shape[0] in _SQUARE and x.shape[0] >= 4096
                      and weight.is_contiguous())
        if ctx.square:
            support = ops.gn_time_gemm([(1.0, x)], x.shape[0], x.shape[1], 0, 0.0, None, None, weight, False, 0.0)
        else:
            support = torch.mm(x, weight).contiguous()       # plain dense GEMM (rocBLAS); see DESIGN.md
        out = ops.spmm(graph, support, bias=bias, relu=False)
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        g = grad_out.contiguous()
        d_support = ops.spmm(ctx.graph.transpose(), g)       # A^T dY
        gx = gw = gb = None
        n, d = x.shape
        if ctx.needs_input_grad[1]:
            if ctx.square:
                gx, _, _ = ops.gn_time_gemm_bwd([(1.0, x)], n, d, 0, 0.0, None, weight, False, d_support,
                                                want_affine_grads=False)
            else:
                gx = torch.mm(d_support, weight.t())
        if ctx.needs_input_grad[2]:
            if ctx.square:
                gw = torch.empty_like(weight)
                ops.reduce_parts_(gw.view(-1), ops.wgrad([(1.0, x)], n, d, 0, 0.0, None, None, d_support, False))
            else:
                gw = _xt_g(x, d_support)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gb = torch.empty_like(weight[0])
            ops.colsum_(gb, g)
        return None, gx, gw, gb


class GraphConvolution(Module):
    """Simple GCN layer (reference: GCN/layers.py:9-42)."""

    def __init__(self, in_features, out_features, bias=True):
        super(GraphConvolution, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        # GCN/layers.py:25-29: U(-1/sqrt(out), 1/sqrt(out)) for weight and bias
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        return _GraphConvFn.apply(as_graph(adj), input.contiguous(), self.weight, self.bias)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class FixedGraphConvolution(GraphConvolution):
    """The same layer with the adjacency held as a plain attribute, so that an ODE solver can call
    f(t, x) (reference: GCN/layers.py:46-83; `adj` is not a buffer there either)."""

    def __init__(self, in_features, out_features, bias=True):
        super(FixedGraphConvolution, self).__init__(in_features, out_features, bias)
        self.adj = torch.Tensor([[1]])

    def set_adj(self, adj):
        self.adj = adj

    def forward(self, input):
        return GraphConvolution.forward(self, input, self.adj)
