"""QC (QM9) edge-conditioned layers with the reference's names and API:

    MPNN_enn_edge(edge_data_dim, node_data_hidden_dim=200).set_T(t); .forward(x, Esrc, Etgt, edge_data)
                                                                       (QC/mpnn.py:5-32)
    EdgeGraphConvolution(in_features, out_features, node_layers=1, edge_layers=1, bias=True)
        .forward(input, Esrc, Etgt, edge_data)                        (QC/layers.py:114-154)

`Etgt` is the reference's DENSE N x E incidence (QC/datasets/utils.py:214); it is converted once per
tensor object to CSR (values kept).  The message step M = Etgt @ bmm(edge_data, x[Esrc]) runs in
csrc/edge.hip without materialising the E x h edge messages; the GRU update of QC/mpnn.py:30 is the fused cell of
csrc/gru.hip on the nn.GRUCell's own parameters (the GRU half).
"""
import math

import torch
import torch.nn as nn
from torch.nn.modules.module import Module
from torch.nn.parameter import Parameter

from . import ops
from .graph import as_graph, csr_from_assignment, incidence_from_index

_cache = {}


class _EdgeSet:
    """(Esrc, Etgt) of one mini-batch, converted once per tensor pair.  In training every step brings a new batch, so
    the conversion sits on the step's critical path: for the reference's dense Etgt (one entry per edge column) it is
    a column-wise argmax, two stable sorts and two counts, with a single host synchronisation (the validity check)."""

    def __init__(self, Esrc, Etgt):
        self.E = Esrc.numel()
        self.src = Esrc.to(torch.int32).contiguous()
        if torch.is_tensor(Etgt) and Etgt.layout == torch.strided and Etgt.dim() == 2:
            self.n = Etgt.shape[0]
            if Etgt.shape[1] != self.E:
                raise ValueError("Etgt must be N x E with E = len(Esrc)")
            nz = Etgt != 0
            per_col = nz.sum(0)
            # the validity check is this conversion's one host synchronisation; inside a HIP-graph capture (qc_step.py:
            # batches built by qc_batch.pad_batch, valid by construction) nothing may synchronise and it is skipped
            if self.E and not torch.cuda.is_current_stream_capturing() and int(per_col.max().item()) > 1:
                raise NotImplementedError("Etgt with more than one entry per edge column is not supported")
            tgt = nz.to(torch.uint8).argmax(0)                                   # the entry's row (0 for an empty column)
            val = Etgt.gather(0, tgt.unsqueeze(0)).squeeze(0).to(torch.float32)    # its value (0 for an empty column)
            self.Mt = csr_from_assignment(tgt, self.n, val)                      # N x E CSR with values
            self.edge_row = torch.where(per_col > 0, tgt, torch.full_like(tgt, -1)).to(torch.int32)
            self.edge_val = val.contiguous()
        else:                                                                    # sparse / CSRGraph input: general path
            self.Mt = as_graph(Etgt)
            self.n = self.Mt.n_rows
            if self.Mt.n_cols != self.E:
                raise ValueError("Etgt must be N x E with E = len(Esrc)")
            rp = self.Mt.rowptr.to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(self.n, device=rp.device), rp[1:] - rp[:-1])
            cols = self.Mt.col.to(torch.int64)
            if cols.numel() and torch.bincount(cols, minlength=self.E).max().item() > 1:
                raise NotImplementedError("Etgt with more than one entry per edge column is not supported")
            self.edge_row = torch.full((self.E,), -1, dtype=torch.int32, device=rp.device)
            self.edge_row[cols] = rows.to(torch.int32)
            self.edge_val = torch.zeros(self.E, dtype=torch.float32, device=rp.device)
            self.edge_val[cols] = self.Mt.val if self.Mt.val is not None else 1.0
        self.Ms_inc = incidence_from_index(self.src, self.n)

    @classmethod
    def from_index(cls, Esrc, etgt, n_nodes, values=None):
        """The same object from the per-edge TARGET INDEX vector (what a loader has before it scatters it into the
        reference's dense N x E matrix, QC/datasets/utils.py:194-214): no dense matrix, no validity check, hence no
        host synchronisation - a training loop can hand the result to the models in place of `Etgt`."""
        self = cls.__new__(cls)
        self.E = Esrc.numel()
        self.n = int(n_nodes)
        if etgt.numel() != self.E:
            raise ValueError("from_index: one target per edge")
        self.src = Esrc.to(torch.int32).contiguous()
        tgt = etgt.to(torch.int64)
        val = torch.ones(self.E, dtype=torch.float32, device=Esrc.device) if values is None else values.to(torch.float32)
        self.Mt = csr_from_assignment(tgt, self.n, val)            # record list (one sync) only from 65 536 edges on
        self.edge_row = tgt.to(torch.int32).contiguous()
        self.edge_val = val.contiguous()
        self.Ms_inc = incidence_from_index(self.src, self.n)
        return self


def prepared_edges(Esrc, etgt, n_nodes, values=None):
    """Loader-side form of (Esrc, Etgt): pass the result wherever a model takes `Etgt` (see _EdgeSet.from_index)."""
    return _EdgeSet.from_index(Esrc, etgt, n_nodes, values)


def _edges(Esrc, Etgt):
    if isinstance(Etgt, _EdgeSet):                   # prepared by the loader (qc_layers.prepared_edges)
        if Etgt.E != Esrc.numel():
            raise ValueError("prepared edges do not belong to this Esrc")
        return Etgt
    key = (id(Esrc), id(Etgt))
    hit = _cache.get(key)
    if hit is not None and hit[0] is Esrc and hit[1] is Etgt:
        return hit[2]
    es = _EdgeSet(Esrc, Etgt)
    if len(_cache) > 64:
        _cache.clear()
    _cache[key] = (Esrc, Etgt, es)
    return es


class MessageChain:
    """Book-keeping for the T message steps of one forward pass that use the SAME edge matrices (QC/mpnn.py:27-30: `for t in
    range(T)` on one `edge_data`; the layer stack of QC/layer_models.py likewise).  Autograd would form the edge-matrix
    gradient of every step as its own E x h x h array and add them (T writes of 4 E h^2 bytes and T - 1 additions: 17 MB
    each for a QM9 mini-batch).  With a chain, the backward of a step only records its (dM, x) pair, and the backward of the
    FIRST step - the last to run, every later step's input depends on its output - forms the sum over all steps in one
    pass (gode_edge_outer_sum_f32).  A step whose backward should run after that one (it cannot in a chain; a caller who
    re-uses a chain across independent branches could make it) falls back to its own gradient, so the result is right
    either way."""

    def __init__(self):
        self.n_steps = 0
        self.pending = []          # (dM, x) of the steps whose edge-matrix gradient is still owed
        self.flushed = False


class _EdgeMessageFn(torch.autograd.Function):
    """M = Etgt @ bmm(edge_data, x[Esrc])   (QC/mpnn.py:27-29 == QC/layers.py:143-145)."""

    @staticmethod
    def forward(ctx, es, x, edge_data, chain):
        x = x.contiguous()
        edge_data = edge_data.contiguous()
        ctx.es = es
        ctx.chain, ctx.idx = chain, 0
        if chain is not None:
            ctx.idx = chain.n_steps
            chain.n_steps += 1
        ctx.save_for_backward(x, edge_data)
        return ops.edge_matvec_fwd(es.Mt, es.src, edge_data, x)

    @staticmethod
    def backward(ctx, dM):
        x, A = ctx.saved_tensors
        es, chain = ctx.es, ctx.chain
        dM = dM.contiguous()
        want_dA = ctx.needs_input_grad[2]
        deferred = chain is not None and want_dA and not chain.flushed and ops.edge_outer_sum_supported(x.shape[1])
        dA, dxe = ops.edge_matvec_bwd(es.edge_row, es.edge_val, es.src, A, x, dM, want_dA=want_dA and not deferred,
                                      want_dx=ctx.needs_input_grad[1])
        dx = ops.spmm(es.Ms_inc, dxe) if dxe is not None else None
        if deferred:
            chain.pending.append((dM, x))
            if ctx.idx == 0 or len(chain.pending) == ops.EDGE_OUTER_MAX_TERMS:
                dA = ops.edge_outer_sum(es.edge_row, es.edge_val, es.src, chain.pending, A)
                chain.pending = []
                if ctx.idx == 0:
                    chain.flushed = True
        return None, dx, dA, None


_CHAINS = {}          # id(edge_data) -> MessageChain of the forward pass that declared the tensor shared (shared_edge_data)


class shared_edge_data:
    """`with shared_edge_data(edge_data): ...` - every edge_message on this tensor inside the block (layers called with the
    reference's own signatures, QC/layers.py:143-145) joins one MessageChain."""

    def __init__(self, edge_data):
        self.key = id(edge_data)
        self.on = torch.is_grad_enabled() and edge_data.requires_grad

    def __enter__(self):
        if self.on:
            _CHAINS[self.key] = MessageChain()
        return self

    def __exit__(self, *exc):
        _CHAINS.pop(self.key, None)
        return False


def edge_message(x, Esrc, Etgt, edge_data, chain=None):
    """chain: a MessageChain shared by the steps of ONE forward pass that use this edge_data (optional; saves T - 1
    gradient arrays and additions in the backward pass); default: the one a surrounding shared_edge_data block declared."""
    if chain is None:
        chain = _CHAINS.get(id(edge_data))
    return _EdgeMessageFn.apply(_edges(Esrc, Etgt), x, edge_data, chain)


class GruChain:
    """The T applications of ONE update cell inside a message-passing loop (QC/mpnn.py:30): the backward of a step writes its
    weight-gradient partials into a shared buffer, and the backward of the first step - the last to run - sums the partial
    rows of all steps once (gode_gru_wreduce_f32).  Autograd would otherwise reduce T times and add the four parameter
    gradients T - 1 times each.  As with MessageChain, a step whose backward runs after that sum falls back to its own
    gradients."""

    def __init__(self):
        self.n_steps = 0
        self.part, self.plen, self.written, self.flushed = None, 0, 0, False


class _GruUpdateFn(torch.autograd.Function):
    """x' = GRUCell([x | m], x) on the parameters of an nn.GRUCell(2h, h) (QC/mpnn.py:12,30): one launch forward, three
    backward (csrc/gru.hip) - no concatenated input, no library GEMM whose shape changes with every batch."""

    @staticmethod
    def forward(ctx, x, m, w_ih, w_hh, b_ih, b_hh, chain):
        x, m = x.contiguous(), m.contiguous()
        out, gates = ops.gru_cell_fwd(x, m, w_ih.contiguous(), w_hh.contiguous(), b_ih, b_hh)
        ctx.has_bias = b_ih is not None
        ctx.chain, ctx.idx = chain, 0
        if chain is not None:
            ctx.idx = chain.n_steps
            chain.n_steps += 1
        ctx.save_for_backward(x, m, w_ih, w_hh, gates)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, m, w_ih, w_hh, gates = ctx.saved_tensors
        chain = ctx.chain
        n, h = x.shape
        if chain is not None and not chain.flushed and n > 0:
            plen = ops.gru_wgrad_part_len(n, h)
            if chain.part is None:
                chain.part, chain.plen = torch.empty(chain.n_steps * plen, dtype=torch.float32, device=x.device), plen
            if plen == chain.plen and chain.written < chain.n_steps:
                slot = chain.written
                chain.written += 1
                dx, dm = ops.gru_cell_bwd(x, m, w_ih.contiguous(), w_hh.contiguous(), gates, dout.contiguous(), ctx.has_bias,
                                          part_out=chain.part[slot * plen:(slot + 1) * plen])[:2]
                if ctx.idx != 0:
                    return dx, dm, None, None, None, None, None
                chain.flushed = True
                rows = chain.written * (plen // (3 * h * (3 * h + 2)))
                dw_ih, dw_hh, db_ih, db_hh = ops.gru_wreduce(chain.part, rows, w_ih, w_hh, ctx.has_bias)
                return dx, dm, dw_ih, dw_hh, db_ih, db_hh, None
        dx, dm, dw_ih, dw_hh, db_ih, db_hh = ops.gru_cell_bwd(x, m, w_ih.contiguous(), w_hh.contiguous(), gates,
                                                              dout.contiguous(), ctx.has_bias)
        return dx, dm, dw_ih, dw_hh, db_ih, db_hh, None


GRU_MAX_H = 640                  # 8 rows x 8h floats of LDS per block (csrc/gru.hip) must fit 160 KB
GRU_MAX_ROWS = 65535 * 64        # grid.y of the weight-gradient launch (64-row chunks)


def gru_update(cell, x, m, chain=None):
    """cell(torch.cat([x, m], 1), x) for cell = nn.GRUCell(2h, h), fused (chain: a GruChain shared by the applications of
    this cell in one forward pass - optional).  Outside the fused kernels' limits (h > 640,
    more than 4.19 M rows, a state that is not fp32 on the GPU) the update is the module's own call on the
    concatenated input - the reference's line (QC/mpnn.py:30) on the GPU library path - instead of an error."""
    if cell.input_size != 2 * cell.hidden_size or x.shape[1] != cell.hidden_size:
        raise ValueError("gru_update: the update cell must be GRUCell(2h, h) on n x h states")
    if cell.hidden_size > GRU_MAX_H or x.shape[0] > GRU_MAX_ROWS or not x.is_cuda or x.dtype != torch.float32 or \
            m.dtype != torch.float32:
        return cell(torch.cat([x, m], 1), x)
    return _GruUpdateFn.apply(x, m, cell.weight_ih, cell.weight_hh, getattr(cell, "bias_ih", None),
                              getattr(cell, "bias_hh", None), chain)


class MPNN_enn_edge(nn.Module):
    """QC/mpnn.py:5-32."""

    def __init__(self, edge_data_dim, node_data_hidden_dim=200):
        super(MPNN_enn_edge, self).__init__()
        self.e_d = edge_data_dim
        self.h_d = node_data_hidden_dim
        self.update_net = nn.GRUCell(self.h_d * 2, self.h_d)
        self.T = 8

    def set_T(self, t):
        self.T = t

    def forward(self, x, Esrc, Etgt, edge_data):
        chain = MessageChain() if (torch.is_grad_enabled() and edge_data.requires_grad) else None
        gchain = GruChain() if (torch.is_grad_enabled() and self.update_net.weight_ih.requires_grad) else None
        for t in range(self.T):
            node_msg = edge_message(x, Esrc, Etgt, edge_data, chain)
            x = gru_update(self.update_net, x, node_msg, gchain)
        return x


class EdgeGraphConvolution(Module):
    """QC/layers.py:114-154."""

    def __init__(self, in_features, out_features, node_layers=1, edge_layers=1, bias=True):
        super(EdgeGraphConvolution, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
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

    def forward(self, input, Esrc, Etgt, edge_data):
        from .functional import dense
        support = dense(input, self.weight)
        output = edge_message(support, Esrc, Etgt, edge_data)
        if self.bias is not None:
            return output + self.bias
        return output

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
