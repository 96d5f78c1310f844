"""ORACLE (test infrastructure, NOT product code): whole-model restatements on the CPU.

The reference's model classes that BASELINE.json's configurations train, as plain functions of a `state_dict` (the
reference's own parameter names) built from oracle.layers_ref and oracle.solver_ref.  Used by tests/ and by the
`cpu_baseline` legs of bench.py / tools/config_bench.py (the reference's CPU path timed on the host cores beside the GPU
number); the product (graph_odenet_amd/) never imports this.

Pinned: the QC models against tests/golden/qc_models.npz (outputs and bias gradients of the reference's own classes,
tests/test_oracle_golden.py); the graph-convolution stacks through layers_ref's golden vectors.  The ODE blocks go through
oracle.solver_ref, whose dependency (torchdiffeq) is absent here: parity unpinned at that seam.
"""
import torch
import torch.nn.functional as F

from . import layers_ref as R
from . import solver_ref as S


class _Field(torch.nn.Module):
    """An ODE function as torchdiffeq sees it: an nn.Module called as f(t, y) whose parameters() the adjoint
    differentiates.  `params`: tensors (leaf, requires_grad as the caller set them); `fn(t, y, *params)`."""

    def __init__(self, fn, params):
        super().__init__()
        self.fn = fn
        self.ps = torch.nn.ParameterList([p if isinstance(p, torch.nn.Parameter) else torch.nn.Parameter(p) for p in params])
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1
        return self.fn(t, y, *self.ps)


def ode_block(field, x, tol=1e-5, method=None, step_size=None):
    """GCN/models.py:189-193 (ODEBlock.forward): odeint_adjoint(f, x, [0, 1], rtol=tol, atol=tol)[1]; `method` /
    `step_size` are the build's documented extension (None = the reference's default, dopri5)."""
    opts = {"step_size": step_size} if step_size is not None else None
    return S.odeint_adjoint(field, x, torch.tensor([0.0, 1.0]), tol, tol, method, opts)[1]


def leaves(sd):
    """Detached float32 CPU copies of a state_dict that require grad (what an optimiser would hold)."""
    return {k: v.detach().to("cpu", torch.float32).clone().requires_grad_(True) for k, v in sd.items()}


def odegcn3(p, x, adj, dropout=0.0, training=False, method=None, step_size=None, tol=1e-5, input_dropout=False):
    """GCN/models.py:204-218 (ODEGCN3.forward); input_dropout: GCN-dense-paper/models.py:221-227 (dropout on the input
    features first, dense adjacency).  Returns (log-probabilities, field) - field.nfe counts the evaluations."""
    if input_dropout:
        x = F.dropout(x, dropout, training=training)
    h = F.relu(R.graph_convolution(x, adj, p["gc1.weight"], p["gc1.bias"]))
    h = F.dropout(h, dropout, training=training)
    f = _Field(lambda t, y, gw, gb, w, b: R.odefunc(t, y, adj, gw, gb, w, b),
               [p["gc2.odefunc.norm1.weight"], p["gc2.odefunc.norm1.bias"], p["gc2.odefunc.gc1.weight"],
                p["gc2.odefunc.gc1.bias"]])
    h = ode_block(f, h, tol, method, step_size)
    return F.log_softmax(R.graph_convolution(h, adj, p["gc3.weight"], p["gc3.bias"]), dim=1), f


def _heads(p, prefix, n_heads):
    if n_heads is None:                      # the reference's one-head layer: parameters f.*, w.* directly
        return [[p[prefix + k] for k in ("f.weight", "f.bias", "w.weight", "w.bias")]]
    return [[p["%sheads.%d.%s" % (prefix, h, k)] for k in ("f.weight", "f.bias", "w.weight", "w.bias")] for h in range(n_heads)]


def gat_odegcn3(p, x, src, tgt, Mtgt, n_heads=None, dropout=0.0, training=False, method=None, step_size=None, tol=1e-5):
    """GAT/models.py:204-218 (ODEGCN3.forward over (src, tgt, Mtgt)); n_heads: H reference layers side by side in every
    layer (BASELINE configs[2]), None: the reference's single layer."""
    h = F.relu(R.gat_multihead_layer(x, src, tgt, Mtgt, _heads(p, "gc1.", n_heads)))
    h = F.dropout(h, dropout, training=training)
    flat = [t for hd in _heads(p, "gc2.odefunc.gc1.", n_heads) for t in hd]
    nh = len(flat) // 4

    def fn(t, y, gw, gb, *hp):
        return R.gat_multihead_odefunc(t, y, src, tgt, Mtgt, gw, gb, [list(hp[4 * i:4 * i + 4]) for i in range(nh)])
    f = _Field(fn, [p["gc2.odefunc.norm1.weight"], p["gc2.odefunc.norm1.bias"]] + flat)
    h = ode_block(f, h, tol, method, step_size)
    out = R.gat_multihead_layer(h, src, tgt, Mtgt, _heads(p, "gc3.", n_heads if "gc3.heads.0.f.weight" in p else None))
    return F.log_softmax(out, dim=1), f


def _mlp2(p, prefix, x):
    """QC/layers.py:46-73 (MLP with one hidden NonLinear layer = TransitionMLP): relu(x W1 + b1) W2 + b2, MyLinear weights
    are in x out (QC/layers.py:10-30)."""
    h = F.relu(torch.mm(x, p[prefix + "layers.0.linear.weight"]) + p[prefix + "layers.0.linear.bias"])
    return torch.mm(h, p[prefix + "layers.1.weight"]) + p[prefix + "layers.1.bias"]


def edge_encoder(p, edge_features, hidden):
    """QC/layers.py:75-86 (EdgeEncoderMLP.forward): the E x h x h edge matrices."""
    return _mlp2(p, "ee.mlp.mlp.", edge_features).reshape(edge_features.shape[0], hidden, hidden)


def mpnn_enn_k_set2set(p, x, edge_features, Esrc, Etgt, batch, n_graphs, T=3, processing_steps=12):
    """QC/layer_models.py:55-81 (MPNN_ENN_K_Set2Set.forward, type 'regression')."""
    hidden = p["input.weight"].shape[0]
    A = edge_encoder(p, edge_features, hidden)
    h = F.linear(x, p["input.weight"], p["input.bias"])
    gru = lambda inp, hx: torch.gru_cell(inp, hx, p["mpnn.update_net.weight_ih"], p["mpnn.update_net.weight_hh"],   # noqa: E731
                                         p["mpnn.update_net.bias_ih"], p["mpnn.update_net.bias_hh"])
    h = R.mpnn_enn_edge(h, Esrc, Etgt, A, gru, T)
    q = R.set2set(h, batch, n_graphs, p["s2s.lstm.weight_ih_l0"], p["s2s.lstm.weight_hh_l0"], p["s2s.lstm.bias_ih_l0"],
                  p["s2s.lstm.bias_hh_l0"], processing_steps)[:, :hidden]
    return F.linear(q, p["output.weight"], p["output.bias"])


def edge_gcn_k_sum(p, x, edge_features, Esrc, Etgt, batch, n_graphs, num_layers=3, dropout=0.5, training=False):
    """QC/layer_models.py:84-122 (EdgeGCN_K_Sum.forward, type 'regression')."""
    hidden = p["gcmid.0.weight"].shape[1]
    A = edge_encoder(p, edge_features, hidden)
    h = _mlp2(p, "mlpin.mlp.", x)
    for k in range(num_layers - 1):
        h = F.relu(R.edge_graph_convolution(h, Esrc, Etgt, A, p["gcmid.%d.weight" % k], p["gcmid.%d.bias" % k]))
        h = F.dropout(h, dropout, training=training)
    k = num_layers - 1
    h = R.edge_graph_convolution(h, Esrc, Etgt, A, p["gcmid.%d.weight" % k], p["gcmid.%d.bias" % k])
    h = _mlp2(p, "mlpout.mlp.", h)
    return torch.zeros(n_graphs, h.shape[1], dtype=h.dtype).index_add_(0, batch, h)      # scatter_add over the graphs


QC_MODELS = {"MPNN_ENN_K_Set2Set": mpnn_enn_k_set2set, "EdgeGCN_K_Sum": edge_gcn_k_sum}
