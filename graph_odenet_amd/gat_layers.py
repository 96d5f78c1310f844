"""GAT-family layers with the reference's names and API (GAT/layers.py:11-127):

    GraphConvolution(in_features, out_features, bias=True, act=F.relu, eps=1e-6)
        .f = nn.Linear(2*in, out)   .w = nn.Linear(2*in, 1)      (the `bias` argument is accepted and
        .forward(x, src, tgt, Mtgt)                               ignored, exactly as in the reference)
    FixedGraphConvolution(...).set_adj(src, tgt, Mtgt); .forward(x)

The reference gathers h = [x[src] | x[tgt]] (E x 2i) and applies f and w per edge.  Here f and w are
applied per NODE (one dense GEMM on N rows, `x @ [Wf_src | Wf_tgt | ww_src | ww_tgt]`), and the
per-edge work (logit, global-max shift, exp, per-target normalised sum of relu messages) runs in the
HIP kernels of csrc/edge.hip (a wave per target on citation-graph sizes, nnz-balanced records above 65 536
targets); the backward pass returns the edge cotangents to the nodes with one fused launch (small graphs) or inside
the record kernel plus source-side SpMMs (large graphs).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.module import Module

from . import ops
from .graph import CSRGraph, as_graph, incidence_from_index

_edge_cache = {}


class EdgeGraph:
    """(src, tgt, Mtgt) normalised once.  The edges are re-ordered by target (a stable sort, so the edges of one
    target keep their relative order - the order torch.spmm sums a coalesced Mtgt): edge k is then position k of the
    CSR of Mtgt, every per-edge array of the kernels is a stream, and `Mt` carries the balanced record list of the
    SpMM.  An Mtgt whose rows disagree with `tgt` (never produced by the reference's loaders) keeps the given order
    and the edge-id indirection instead (`canonical` False)."""

    def __init__(self, src, tgt, Mtgt):
        self.n = Mtgt.shape[0] if torch.is_tensor(Mtgt) else Mtgt.n_rows
        self.E = src.numel()
        mt = as_graph(Mtgt)
        self.Mt_given = mt                      # CSR of Mtgt over the edge ids as given (general-activation path)
        if mt.n_cols != self.E:
            raise ValueError("Mtgt must be N x E with E = len(src)")
        cols = mt.col.to(torch.int64)
        if cols.numel() and torch.bincount(cols, minlength=self.E).max().item() > 1:
            raise NotImplementedError("Mtgt with more than one entry per edge column is not supported")
        rp = mt.rowptr.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(self.n, device=rp.device), rp[1:] - rp[:-1])
        src64, tgt64 = src.to(torch.int64), tgt.to(torch.int64)
        self.canonical = bool(mt.nnz == self.E and (self.E == 0 or bool((tgt64[cols] == rows).all().item())))
        if self.canonical:
            self.perm = cols                                              # edge id at CSR position k
            self.src = src64[cols].to(torch.int32).contiguous()
            self.tgt = rows.to(torch.int32).contiguous()
            self.Mt = CSRGraph(mt.rowptr, torch.arange(self.E, device=rp.device), mt.val, self.n, self.E)
        else:
            self.perm = None
            self.src = src.to(torch.int32).contiguous()
            self.tgt = tgt.to(torch.int32).contiguous()
            self.Mt = mt
        self.Ms_inc = incidence_from_index(self.src, self.n)     # N x E, row = src of the (re-ordered) edge
        self.Mt_inc = incidence_from_index(self.tgt, self.n)     # N x E, row = tgt
        self._scratch = None

    def maxpath_scratch(self):
        if self._scratch is None:
            from . import _lib
            nb = _lib.load().gode_gat_maxpath_scratch_bytes(self.E)
            self._scratch = torch.empty(nb, dtype=torch.uint8, device=self.src.device)
        return self._scratch


def edge_graph(src, tgt, Mtgt):
    key = (id(src), id(tgt), id(Mtgt))
    hit = _edge_cache.get(key)
    if hit is not None and hit[0] is src and hit[1] is tgt and hit[2] is Mtgt:
        return hit[3]
    g = EdgeGraph(src, tgt, Mtgt)
    if len(_edge_cache) > 64:
        _edge_cache.clear()
    _edge_cache[key] = (src, tgt, Mtgt, g)     # strong refs: ids stay valid while cached
    return g


class _EdgeAttentionFn(torch.autograd.Function):
    """out = per-target softmax-weighted sum of relu messages, from the node-level projections
    Ps, Pt (N x o: message parts by source / by target) and A2 (N x 2: logit parts)."""

    @staticmethod
    def forward(ctx, eg, Ps, Pt, A2, bf, bw, eps):
        Ps, Pt, A2 = Ps.contiguous(), Pt.contiguous(), A2.contiguous()
        n, o = Ps.shape
        f = dict(dtype=torch.float32, device=Ps.device)
        a, amax = torch.empty(eg.E, **f), torch.empty(1, **f)
        out, w, den = torch.empty(n, o, **f), torch.empty(eg.E, **f), torch.empty(n, **f)
        proj = ops.gat_proj(Ps, Pt, A2)
        ops.gat_logits(proj, bw, eg.src, eg.tgt, a, amax)
        ops.gat_agg_fwd(eg, proj, o, bf, a, amax, eps, out, w, den)
        ctx.eg = eg
        ctx.save_for_backward(Ps, Pt, A2, bf, a, amax, w, den, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        Ps, Pt, A2, bf, a, amax, w, den, out = ctx.saved_tensors
        eg = ctx.eg
        n, o = Ps.shape
        f = dict(dtype=torch.float32, device=Ps.device)
        dz, da = torch.empty(eg.E, o, **f), torch.empty(eg.E, **f)
        dPs, dPt, dA2 = torch.empty(n, o, **f), torch.empty(n, o, **f), torch.empty(n, 2, **f)
        proj = ops.gat_proj(Ps, Pt, A2)
        ops.gat_vjp(eg, proj, o, bf, a, amax, w, den, out, dz, da, dPs, dPt, dA2, dout=dout.contiguous())
        dbf = torch.empty_like(bf)
        ops.colsum_(dbf, dPt)                                    # sum_e dz_e = sum_v (sum over the edges entering v)
        dbw = dA2[:, 1].sum().reshape(1)
        return None, dPs, dPt, dA2, dbf, dbw, None


def _gat_forward_general(layer, x, src, tgt, Mtgt):
    """The layer with an arbitrary activation callable (the `act=` constructor argument, GAT/layers.py:16): an
    activation that is not relu cannot be folded into the aggregation kernels, so the per-edge messages are formed as
    the reference forms them (GAT/layers.py:40-46: gather both endpoints, two Linear layers, act) with PyTorch ops and
    only the two per-target sums (GAT/layers.py:53,55) run on libgraphode's SpMM."""
    from .functional import graph_aggregate
    eg = edge_graph(src, tgt, Mtgt)
    if eg.E == 0:
        return torch.zeros(x.shape[0], layer.out_features, dtype=x.dtype, device=x.device) + 0.0 * x.sum()
    h = torch.cat([x.index_select(0, src), x.index_select(0, tgt)], 1)
    y = layer.act(layer.f(h))
    a = layer.w(h)
    e = torch.exp(a - a.max())                                   # global maximum, as GAT/layers.py:47
    num = graph_aggregate(eg.Mt_given, y * e)
    den = graph_aggregate(eg.Mt_given, e)
    return num / (den + layer.eps)


def _gat_forward(layer, x, src, tgt, Mtgt):
    if layer.act is not F.relu:
        return _gat_forward_general(layer, x, src, tgt, Mtgt)
    eg = edge_graph(src, tgt, Mtgt)
    i, o = layer.in_features, layer.out_features
    Wf, ww = layer.f.weight, layer.w.weight                      # (o, 2i), (1, 2i)
    # node-level projections (dense GEMMs), split by role: message parts by source / target, logit parts
    from .functional import dense
    Ps, Pt = dense(x, Wf[:, :i].t()), dense(x, Wf[:, i:].t())
    A2 = dense(x, torch.stack([ww[0, :i], ww[0, i:]], 1))
    if eg.E == 0:                                                # no edges: every node is 0 / eps = 0
        return torch.zeros(x.shape[0], o, dtype=x.dtype, device=x.device) + 0.0 * (Ps.sum() + Pt.sum() + A2.sum())
    return _EdgeAttentionFn.apply(eg, Ps, Pt, A2, layer.f.bias, layer.w.bias, layer.eps)


class GraphConvolution(Module):
    """GAT layer (reference: GAT/layers.py:11-63)."""

    def __init__(self, in_features, out_features, bias=True, act=F.relu, eps=1e-6):
        super(GraphConvolution, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.f = nn.Linear(2 * in_features, out_features)
        self.w = nn.Linear(2 * in_features, 1)
        self.eps = eps
        self.act = act
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.f.weight)
        nn.init.xavier_uniform_(self.w.weight)

    def forward(self, x, src, tgt, Mtgt):
        return _gat_forward(self, x, src, tgt, Mtgt)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class FixedGraphConvolution(GraphConvolution):
    """The same layer with (src, tgt, Mtgt) held as attributes (reference: GAT/layers.py:67-127)."""

    def __init__(self, in_features, out_features, bias=True, act=F.relu, eps=1e-6):
        super(FixedGraphConvolution, self).__init__(in_features, out_features, bias, act, eps)
        self.src = self.tgt = self.Mtgt = torch.Tensor([[1]])

    def set_adj(self, src, tgt, Mtgt):
        self.src, self.tgt, self.Mtgt = src, tgt, Mtgt

    def forward(self, x):
        return _gat_forward(self, x, self.src, self.tgt, self.Mtgt)
