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

from . import ops
from .functional import GroupNorm
from .functional import Linear as GodeLinear
from .graph import RECORDS_MIN_NNZ, CSRGraph, csr_from_assignment, incidence_from_index
from .qc_layers import EdgeGraphConvolution, MPNN_enn_edge, shared_edge_data


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
        if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32:
            from .functional import affine
            return affine(x, self.weight, self.bias)          # gode_gemm_f32 (+ bias epilogue), own autograd
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
        first, last = self.mlp.layers[0], self.mlp.layers[1]
        if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and first.f is F.relu:
            # both layers, their biases and the relu (and, backward, its mask) on gode_gemm_f32: functional.mlp2
            from .functional import mlp2
            return mlp2(x, first.linear.weight, first.linear.bias, last.weight, last.bias)
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
_seg_cache = {}


class _Segments:
    """Graph membership of a batch vector, converted once per tensor object: nodes of graph b are
    perm[segptr[b] : segptr[b+1]] (perm is None when `batch` is already sorted, as the reference's collate emits it)."""

    def __init__(self, batch):
        b64 = batch.to(torch.int64)
        n = b64.numel()
        known = getattr(batch, "_gode_n_graphs", None)
        if known is not None:
            # static batch of a captured step (qc_step.py): the number of graphs is a property of the bucket and the
            # batch vector is sorted (collate order) - no host synchronisation, so the conversion can be captured
            self.nb, uns = int(known), 0
        elif n:
            # one host synchronisation for both facts the layout depends on: number of graphs, sortedness
            unsorted = (b64[1:] < b64[:-1]).any().to(torch.int64) if n > 1 else torch.zeros((), dtype=torch.int64, device=b64.device)
            mx, uns = torch.stack([b64.max(), unsorted]).tolist()
            self.nb = int(mx) + 1
        else:
            self.nb, uns = 0, 0
        # nodes grouped by graph, ascending inside a graph: one launch for a mini-batch (graph.csr_from_assignment ->
        # csrc/convert.hip), the sort-based path for large inputs
        g = csr_from_assignment(b64, self.nb, None, records=False) if self.nb > 0 else None
        self.segptr = g.rowptr if g is not None else torch.zeros(1, dtype=torch.int32, device=batch.device)
        self.perm = g.col if (uns and g is not None) else None
        self.index = b64
        # sorted batch vector: the rows of the membership matrix are the segments themselves and g IS that matrix
        self.incidence = g if (g is not None and not uns and n < RECORDS_MIN_NNZ) else None

    def sum_matrix(self):
        """nb x N membership matrix (pattern-only CSR) for the per-graph sum."""
        if self.incidence is None:
            if self.perm is None:            # sorted batch vector: the rows are the segments themselves
                n = self.index.numel()
                self.incidence = CSRGraph(self.segptr, torch.arange(n, device=self.index.device), None, self.nb, n,
                                          records=n >= RECORDS_MIN_NNZ)
            else:
                self.incidence = incidence_from_index(self.index, self.nb)
        return self.incidence


def _segments(batch):
    key = id(batch)
    hit = _seg_cache.get(key)
    if hit is not None and hit[0] is batch and hit[2] == batch._version:
        return hit[1]
    seg = _Segments(batch)
    if len(_seg_cache) > 64:
        _seg_cache.clear()
    _seg_cache[key] = (batch, seg, batch._version)
    return seg


class _SegmentSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, x):
        ctx.seg = seg
        return ops.spmm(seg.sum_matrix(), x.contiguous())

    @staticmethod
    def backward(ctx, dout):
        return None, dout.index_select(0, ctx.seg.index)


def segment_sum(x, batch, n_graphs=None):
    """scatter_add(x, batch, dim=0, dim_size=n_graphs) (QC/torch_scatter.py:170-221) as a pattern-only SpMM over the
    graph-membership matrix: fixed summation order, no atomics."""
    seg = _segments(batch)
    if n_graphs is not None and n_graphs != seg.nb:
        raise ValueError("segment_sum: batch holds %d graphs, dim_size says %d" % (seg.nb, n_graphs))
    return _SegmentSumFn.apply(seg, x)


class _SegmentAttentionFn(torch.autograd.Function):
    """r_b = sum_i softmax_b(<x_i, q_b>)_i x_i   (the loop body of QC/set2set.py:63-74)."""

    @staticmethod
    def forward(ctx, seg, x, q):
        x, q = x.contiguous(), q.contiguous()
        a, r = ops.segment_attention_fwd(seg.segptr, seg.perm, x, q)
        ctx.seg = seg
        ctx.save_for_backward(x, q, a)
        return r

    @staticmethod
    def backward(ctx, dr):
        x, q, a = ctx.saved_tensors
        dx, dq = ops.segment_attention_bwd(ctx.seg.segptr, ctx.seg.perm, x, q, a, dr.contiguous())
        return None, dx, dq


class _LstmCellFn(torch.autograd.Function):
    """(h', c') = LSTMCell(x, (h, c)) on an nn.LSTM's own layer-0 parameters (QC/set2set.py:44-47,61): one launch forward,
    one backward (csrc/lstm.hip) - no library GEMM on a 20-row batch."""

    @staticmethod
    def forward(ctx, x, h, c, w_ih, w_hh, b_ih, b_hh):
        x, h, c = x.contiguous(), h.contiguous(), c.contiguous()
        w_ih, w_hh = w_ih.contiguous(), w_hh.contiguous()
        h_out, c_out, gates = ops.lstm_cell_fwd(x, h, c, w_ih, w_hh, b_ih, b_hh)
        ctx.has_bias = b_ih is not None
        ctx.save_for_backward(x, h, c, w_ih, w_hh, gates, c_out)
        return h_out, c_out

    @staticmethod
    def backward(ctx, dh_out, dc_out):
        x, h, c, w_ih, w_hh, gates, c_out = ctx.saved_tensors
        dh_out = dh_out.contiguous() if dh_out is not None else None
        dc_out = dc_out.contiguous() if dc_out is not None else None
        dx, dh, dc, dw_ih, dw_hh, db_ih, db_hh = ops.lstm_cell_bwd(
            x, h, c, w_ih, w_hh, gates, c_out, dh_out, dc_out, has_bias=ctx.has_bias,
            want_dx=ctx.needs_input_grad[0], want_dh=ctx.needs_input_grad[1])
        return dx, dh, (dc if ctx.needs_input_grad[2] else None), dw_ih, dw_hh, db_ih, db_hh


SET2SET_ONE_NODE = True      # False: the readout as a chain of per-step autograd nodes (kept for the A/B test)


class _Set2SetFn(torch.autograd.Function):
    """The whole readout loop of QC/set2set.py:50-75 as ONE autograd node and ONE launch per direction (csrc/set2set.hip:
    a workgroup per graph walks its row of the LSTM state and its nodes through every processing step - nothing couples two
    graphs of a batch).  As a chain of per-step nodes the readout was ~50 launches forward, ~50 backward and ~50 elementwise
    launches of autograd summing the LSTM's per-step weight gradients; here those gradients are one small product over the
    saved gate cotangents (ops.set2set_bwd)."""

    @staticmethod
    def forward(ctx, seg, x, w_ih, w_hh, b_ih, b_hh, steps):
        x = x.contiguous()
        w_ih, w_hh = w_ih.contiguous(), w_hh.contiguous()
        saved = ops.set2set_fwd(seg.segptr, seg.perm, x, w_ih, w_hh, b_ih, b_hh, steps, seg.nb)
        ctx.seg, ctx.has_bias = seg, b_ih is not None
        ctx.save_for_backward(x, w_ih, w_hh, *saved)
        return saved[0][steps]

    @staticmethod
    def backward(ctx, dq_star):
        x, w_ih, w_hh = ctx.saved_tensors[:3]
        dx, dw_ih, dw_hh, db_ih, db_hh = ops.set2set_bwd(ctx.seg.segptr, ctx.seg.perm, x, w_ih, w_hh, ctx.has_bias,
                                                         ctx.saved_tensors[3:], dq_star)
        return None, dx, dw_ih, dw_hh, db_ih, db_hh, None


class Set2Set(nn.Module):
    """Set2Set pooling (Vinyals et al. 2015) with the reference's parameters (`lstm`).  The per-graph softmax loop of
    the reference is one kernel per processing step (csrc/segment.hip); the single-layer LSTM step runs as the fused
    cell of csrc/lstm.hip on the same `lstm.*` parameters (gate order i, f, g, o as in nn.LSTM)."""

    def __init__(self, in_channels, processing_steps, num_layers=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, 2 * in_channels
        self.processing_steps, self.num_layers = processing_steps, num_layers
        self.lstm = nn.LSTM(self.out_channels, self.in_channels, num_layers)

    def _lstm_step(self, q_star, h):
        if self.num_layers != 1:
            q, h = self.lstm(q_star.unsqueeze(0), h)
            return q.view(q_star.shape[0], self.in_channels), h
        l = self.lstm
        if q_star.is_cuda and q_star.dtype == torch.float32 and \
                ops.lstm_cell_supported(q_star.shape[0], self.out_channels, self.in_channels):
            hx, cx = _LstmCellFn.apply(q_star, h[0][0], h[1][0], l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0)
        else:                                # outside the fused cell's limits: the library cell on the same parameters
            hx, cx = torch.lstm_cell(q_star, (h[0][0], h[1][0]), l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0)
        return hx, (hx.unsqueeze(0), cx.unsqueeze(0))

    def forward(self, x, batch):
        seg = _segments(batch)
        nb = seg.nb
        l = self.lstm
        if SET2SET_ONE_NODE and self.num_layers == 1 and x.is_cuda and x.dtype == torch.float32 and self.processing_steps > 0 and \
                (getattr(l, "bias_ih_l0", None) is None) == (getattr(l, "bias_hh_l0", None) is None) and \
                ops.set2set_supported(self.in_channels) and nb > 0:
            return _Set2SetFn.apply(seg, x, l.weight_ih_l0, l.weight_hh_l0, getattr(l, "bias_ih_l0", None),
                                    getattr(l, "bias_hh_l0", None), self.processing_steps)
        h = (x.new_zeros(self.num_layers, nb, self.in_channels), x.new_zeros(self.num_layers, nb, self.in_channels))
        q_star = x.new_zeros(nb, self.out_channels)
        for _ in range(self.processing_steps):
            q, h = self._lstm_step(q_star, h)
            r = _SegmentAttentionFn.apply(seg, x, q)
            q_star = torch.cat([q, r], -1)
        return q_star

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


def _output_function(kind, target_features):
    if kind == "regression" or target_features == 1:
        return lambda x: x
    return lambda x: F.log_softmax(x, dim=1)


get_output_function = _output_function       # the reference's name (QC/layer_models.py:10)


class UnimplementedModel(nn.Module):
    """Placeholder QC/train_egcn.py's model_dict maps three of its choices to (QC/layer_models.py:19-24)."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("Model not implemented yet")

    def forward(self, *args, **kwargs):
        raise NotImplementedError("Model not implemented yet")


# ---- models -------------------------------------------------------------------------------------------
class _QCBase(nn.Module):
    def _finish(self, kind, target_features):
        self.type = kind
        self.output_function = _output_function(kind, target_features)


class MPNN_ENN_K_Sum(_QCBase):
    def __init__(self, node_features=None, edge_features=None, target_features=1, hidden_features=73, num_layers=3,
                 s2s_processing_steps=12, type="regression", dropout=0.5, **kwargs):
        super().__init__()
        self.input = GodeLinear(node_features, hidden_features)
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)
        self.mpnn = MPNN_enn_edge(edge_features, hidden_features)
        self.mpnn.set_T(num_layers)
        self.output = GodeLinear(hidden_features, target_features)
        self._finish(type, target_features)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self.mpnn(self.input(node_features), Esrc, Etgt, self.ee(edge_features))
        return self.output_function(segment_sum(self.output(x), batch))


class MPNN_ENN_K_Set2Set(_QCBase):
    def __init__(self, node_features=None, edge_features=None, target_features=1, hidden_features=73, num_layers=3,
                 s2s_processing_steps=12, type="regression", dropout=0.5, **kwargs):
        super().__init__()
        self.input = GodeLinear(node_features, hidden_features)
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)
        self.mpnn = MPNN_enn_edge(edge_features, hidden_features)
        self.mpnn.set_T(num_layers)
        self.s2s = Set2Set(hidden_features, s2s_processing_steps, num_layers=1)
        self.output = GodeLinear(hidden_features, target_features)
        self._finish(type, target_features)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self.mpnn(self.input(node_features), Esrc, Etgt, self.ee(edge_features))
        x = self.s2s(x, batch)[:, :x.size(1)]
        return self.output_function(self.output(x))


class _EdgeGCNStack(_QCBase):
    def _stack(self, x, Esrc, Etgt, ef):
        with shared_edge_data(ef):       # the layers share the edge matrices: one pass forms their gradient (qc_layers.MessageChain)
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
        x = self._stack(self.mlpin(node_features), Esrc, Etgt, self.ee(edge_features))
        return self.output_function(segment_sum(self.mlpout(x), batch))


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


# ---- the fixed-depth models of QC/models.py (used by QC/train_egcn_multitask.py) ------------------------------------
class _Fixed(nn.Module):
    """`type` decides between raw outputs and log_softmax (QC/models.py:36,64; the two EdgeGCN3 classes read
    self.type without ever setting it - it is set here)."""

    def _out(self, x):
        return F.log_softmax(x, dim=1) if self.type == "classification" else x


class MPNN_ENN_Sum(_Fixed):
    """QC/models.py:10-36: input Linear, edge encoder, MPNN_enn_edge (its own default T = 8), output Linear, per-graph sum."""

    def __init__(self, node_features, edge_features, hidden_features, out_features, processing_steps=12,
                 type="regression", **kwargs):
        super().__init__()
        self.input = GodeLinear(node_features, hidden_features)
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)
        self.mpnn = MPNN_enn_edge(edge_features, hidden_features)
        self.output = GodeLinear(hidden_features, out_features)
        self.type = type

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self.mpnn(self.input(node_features), Esrc, Etgt, self.ee(edge_features))
        return self._out(segment_sum(self.output(x), batch))


class MPNN_ENN_Set2Set(_Fixed):
    """QC/models.py:38-66."""

    def __init__(self, node_features, edge_features, hidden_features, out_features, processing_steps=12,
                 type="regression", **kwargs):
        super().__init__()
        self.input = GodeLinear(node_features, hidden_features)
        self.ee = EdgeEncoderMLP(edge_features, hidden_features)
        self.mpnn = MPNN_enn_edge(edge_features, hidden_features)
        self.s2s = Set2Set(hidden_features, processing_steps, num_layers=1)
        self.output = GodeLinear(hidden_features, out_features)
        self.type = type

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self.mpnn(self.input(node_features), Esrc, Etgt, self.ee(edge_features))
        return self._out(self.output(self.s2s(x, batch)[:, :x.size(1)]))


class _EdgeGCN3(_Fixed):
    def _init3(self, node_features, edge_features, hidden_features, out_features, dropout, type):
        self.gc1 = EdgeGraphConvolution(node_features, hidden_features)
        self.gc2 = EdgeGraphConvolution(hidden_features, hidden_features)
        self.gc3 = EdgeGraphConvolution(hidden_features, out_features)
        self.dropout = dropout
        self.ee1 = EdgeEncoderMLP(edge_features, hidden_features)
        self.ee2 = EdgeEncoderMLP(edge_features, hidden_features)
        self.ee3 = EdgeEncoderMLP(edge_features, out_features)
        self.type = type

    def _convs(self, x, edge_features, Esrc, Etgt):
        for gc, ee in ((self.gc1, self.ee1), (self.gc2, self.ee2)):
            x = F.dropout(F.relu(gc(x, Esrc, Etgt, ee(edge_features))), self.dropout, training=self.training)
        return self.gc3(x, Esrc, Etgt, self.ee3(edge_features))


class EdgeGCN3_Sum(_EdgeGCN3):
    """QC/models.py:69-103: three edge-conditioned convolutions, each with its own edge encoder, per-graph sum."""

    def __init__(self, node_features, edge_features, hidden_features, out_features, dropout=0, type="regression"):
        super().__init__()
        self._init3(node_features, edge_features, hidden_features, out_features, dropout, type)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        return self._out(segment_sum(self._convs(node_features, edge_features, Esrc, Etgt), batch))


class EdgeGCN3_Set2Set(_EdgeGCN3):
    """QC/models.py:106-143."""

    def __init__(self, node_features, edge_features, hidden_features, out_features, dropout=0, processing_steps=8,
                 type="regression"):
        super().__init__()
        self._init3(node_features, edge_features, hidden_features, out_features, dropout, type)
        self.s2s = Set2Set(out_features, processing_steps, num_layers=1)

    def forward(self, node_features, edge_features, Esrc, Etgt, batch):
        x = self._convs(node_features, edge_features, Esrc, Etgt)
        return self._out(self.s2s(x, batch)[:, :x.size(1)])


def scatter_add(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
    """The one way the reference calls QC/torch_scatter.py:170 on this path: rows of a 2-D tensor summed per graph
    (dim = 0, no `out`, zero fill)."""
    if dim not in (0, -2) or out is not None or fill_value != 0 or src.dim() != 2:
        raise NotImplementedError("scatter_add: only scatter_add(x[N, d], batch[N], dim=0, dim_size=B) is on the hot path")
    return segment_sum(src, index, dim_size)
