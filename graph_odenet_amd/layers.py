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

# ---- the feature x weight product of a layer (GCN/layers.py:32 `support = torch.mm(input, self.weight)`) --------------
# Three own kernels, no library GEMM:
#   * square weights of the widths the fused ODE kernels cover: the MFMA kernels of csrc/gemm.hip (no GroupNorm, no
#     time row);
#   * a feature matrix that is mostly zeros (Cora's bag-of-words X is 98.7 % zero: 124 MFLOP of dense product for
#     49 216 non-zeros): X is converted ONCE per tensor object to CSR and X W runs on the aggregation
#     kernel with W as the dense operand (gode_spmm_csr_f32), X^T dS on its transpose.  Explicit: a sparse tensor as
#     the layer input, or sparse_input=True; default "auto": from the SECOND time the same tensor object arrives (a
#     freshly dropped-out copy per step, as GCN-dense-paper makes, never pays for a conversion), for inputs that need
#     no gradient, have >= SPARSE_MIN_FEATURES columns and <= SPARSE_MAX_DENSITY non-zeros (sparse_features);
#   * everything else: the rectangular fp32-MFMA kernels of csrc/rect.hip.
# Output widths that are not a multiple of 4 (Cora's 7 classes) are zero-padded to the next multiple, so that the
# aggregation that follows runs on the 16-byte SpMM kernel instead of the scalar one; the layer returns the first
# out_features columns.
SPARSE_MIN_FEATURES = 64
SPARSE_MAX_DENSITY = 0.25
SPARSE_INPUT = "auto"      # default of every call site: "auto" | True | False (see sparse_features)
_feat_seen = {}            # id(tensor) -> [weakref, version, data_ptr, sightings, CSRGraph | False | None]


def _feat_evict(key):
    _feat_seen.pop(key, None)


def invalidate_sparse_features(x=None):
    """Forget the cached CSR form of `x` (or of every tensor): call it after writing to a feature matrix through `.data`
    or an aliasing view, which does not advance the tensor's version counter."""
    if x is None:
        _feat_seen.clear()
    else:
        _feat_seen.pop(id(x), None)


def sparse_features(x, mode=None):
    """CSRGraph of a mostly-zero feature matrix, or None (dense path).

    mode (default: the module switch SPARSE_INPUT):
      True    convert at the FIRST sighting (one host synchronisation, one sort) whatever the density: the CSR route from
              step 1 on - the reproducible choice for a fixed bag-of-words matrix;
      False   never: always the dense rectangular kernel;
      "auto"  the first sighting of a tensor object runs dense, the second decides once by density (<= SPARSE_MAX_DENSITY,
              one host synchronisation) - so a dropped-out copy per step never pays for a conversion, but STEP 1 OF A RUN
              AND THE LATER STEPS USE DIFFERENT KERNELS for the same product (fp32-MFMA rows of 16 against CSR rows
              summed in column order: equal to rounding, not bit for bit).
    A sparse tensor (COO / CSR layout) passed as the layer input takes the CSR route directly, under every mode.
    Inside a HIP-graph capture nothing is decided or converted (both need the host): a tensor already converted keeps
    its route, any other runs dense.  The cache is keyed on the tensor object, its version counter and its storage
    address; writes through `.data` or an alias need invalidate_sparse_features."""
    import weakref
    if mode is None:
        mode = SPARSE_INPUT
    if mode is False or x.requires_grad or x.dim() != 2 or x.shape[1] < SPARSE_MIN_FEATURES or x.shape[0] == 0:
        return None
    key = id(x)
    rec = _feat_seen.get(key)
    capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
    if rec is None or rec[0]() is not x or rec[1] != x._version or rec[2] != x.data_ptr():
        if capturing:
            return None
        try:
            rec = _feat_seen[key] = [weakref.ref(x, lambda _r, k=key: _feat_evict(k)), x._version, x.data_ptr(), 1, None]
        except TypeError:
            return None
        if mode is not True:
            return None
    else:
        rec[3] += 1
    if rec[4] is None:                                       # decide once (one host synchronisation)
        if capturing:
            return None
        if mode is True:
            rec[4] = as_graph(x)
        else:
            dens = float(torch.count_nonzero(x)) / x.numel()
            rec[4] = as_graph(x) if dens <= SPARSE_MAX_DENSITY else False
    return rec[4] or None


def _pad4(c):
    return (c + 3) // 4 * 4


class _GraphConvFn(torch.autograd.Function):
    """output = A @ (input @ W) + bias   (GCN/layers.py:31-37)."""

    @staticmethod
    def forward(ctx, graph, x, weight, bias, sparse_mode=None):
        is_graph = not torch.is_tensor(x)
        n, f = (x.n_rows, x.n_cols) if is_graph else x.shape
        c = weight.shape[1]
        cp = _pad4(c)
        ctx.square = (not is_graph and f == c and weight.shape[0] in _SQUARE and n >= 4096 and weight.is_contiguous())
        ctx.xs = None
        w = weight.contiguous()
        if ctx.square:
            # 0.43 ms at 2^20 x 128 x 128 against 1.9 ms for the library GEMM of this image
            support = ops.gn_time_gemm([(1.0, x)], n, f, 0, 0.0, None, None, w, False, 0.0)
        else:
            ctx.xs = x if not torch.is_tensor(x) else sparse_features(x, sparse_mode)
            if ctx.xs is not None:
                support = ops.spmm(ctx.xs, w if cp == c else torch.nn.functional.pad(w, (0, cp - c)))
            else:
                support = ops.rect_gemm(x, w, pad_to=cp)
        b = bias
        if bias is not None and cp != c:
            b = torch.nn.functional.pad(bias, (0, cp - c))
        out = ops.spmm(graph, support, bias=b, relu=False)
        ctx.graph = graph
        ctx.has_bias = bias is not None
        ctx.c = c
        if torch.is_tensor(x):
            ctx.save_for_backward(x, weight)
        else:                                                 # a CSRGraph input (sparse features handed over as such)
            ctx.save_for_backward(weight)
        ctx.x_is_graph = not torch.is_tensor(x)
        return out if cp == c else out[:, :c]

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.x_is_graph:
            (weight,), x = ctx.saved_tensors, ctx.xs
        else:
            x, weight = ctx.saved_tensors
        c, cp = ctx.c, _pad4(ctx.c)
        g = grad_out.contiguous() if cp == c else torch.nn.functional.pad(grad_out, (0, cp - c))
        d_support = ops.spmm(ctx.graph.transpose(), g)       # A^T dY   (n x cp; the pad columns stay zero)
        gx = gw = gb = None
        n, d = (x.n_rows, x.n_cols) if ctx.x_is_graph else x.shape
        w = weight.contiguous()
        if ctx.needs_input_grad[1] and not ctx.x_is_graph:
            if ctx.square:
                gx, _, _ = ops.gn_time_gemm_bwd([(1.0, x)], n, d, 0, 0.0, None, w, False, d_support,
                                                want_affine_grads=False)
            else:
                gx = ops.rect_gemm_nt(d_support[:, :c], w)
        if ctx.needs_input_grad[2]:
            if ctx.square:
                gw = torch.empty_like(w)
                ops.reduce_parts_(gw.view(-1), ops.wgrad([(1.0, x)], n, d, 0, 0.0, None, None, d_support, False))
            elif ctx.xs is not None:
                gw = ops.spmm(ctx.xs.transpose(), d_support)           # X^T dS on the transposed CSR of X
                gw = gw if cp == c else gw[:, :c].contiguous()
            else:
                gw = ops.rect_wgrad(x, d_support[:, :c])
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gb = torch.empty(c, dtype=torch.float32, device=g.device)
            ops.colsum_(gb, grad_out.contiguous())
        return None, gx, gw, gb, None


class GraphConvolution(Module):
    """Simple GCN layer (reference: GCN/layers.py:9-42).  Extension: `sparse_input` ("auto" | True | False, default the
    module switch SPARSE_INPUT) chooses the route of `input @ weight` for a mostly-zero input (sparse_features); a sparse
    tensor (`features.to_sparse()` / `.to_sparse_csr()`) as `input` always takes the CSR route."""

    def __init__(self, in_features, out_features, bias=True, sparse_input=None):
        super(GraphConvolution, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.sparse_input = sparse_input
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
        if input.layout != torch.strided:                     # sparse features handed over as such: CSR(X) from step 1
            return _GraphConvFn.apply(as_graph(adj), as_graph(input), self.weight, self.bias, None)
        return _GraphConvFn.apply(as_graph(adj), input.contiguous(), self.weight, self.bias, self.sparse_input)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class FixedGraphConvolution(GraphConvolution):
    """The same layer with the adjacency held as a plain attribute, so that an ODE solver can call
    f(t, x) (reference: GCN/layers.py:46-83; `adj` is not a buffer there either)."""

    def __init__(self, in_features, out_features, bias=True, sparse_input=None):
        super(FixedGraphConvolution, self).__init__(in_features, out_features, bias, sparse_input)
        self.adj = torch.Tensor([[1]])

    def set_adj(self, adj):
        self.adj = adj

    def forward(self, input):
        return GraphConvolution.forward(self, input, self.adj)
