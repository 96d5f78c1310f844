"""GCN-family model zoo with the reference's class names, constructor kwargs
(nfeat=, nhid=, nclass=, dropout=[, nlayers=]), sub-module / state_dict key names and
forward(x, adj) API (reference: GCN/models.py), built on graph_odenet_amd.layers and our solver.

Design: the reference writes every variant out by hand (22 near-identical classes).  Here a model
is a short *plan* - a tuple of steps interpreted by `_PlanModel.forward` - so a variant is its
constructor (which fixes the attribute names the state_dict needs) plus one line of plan:

    ("gc", name)     x = self.<name>(x, *graph)          graph convolution / ODE block
    ("relu",)        x = relu(x)
    ("drop",)        x = dropout(x, p=self.dropout, training=self.training)
    ("norm", name)   x = self.<name>(x)                  GroupNorm
    ("save",)        r = x                               residual tap
    ("add",)         x = x + r
    ("head", n)      x = x[:, :n]                        (ODEGCN2-style truncation)

Extensions (default to the reference's behaviour): ODEBlock(odefunc, tol=1e-5, method=None,
step_size=None) - method=None is adaptive dopri5 with rtol=atol=tol exactly as
GCN/models.py:192 calls torchdiffeq; method='rk4' + step_size=1/16 is the fixed 64-eval
integration the benchmark metric is quoted on.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .functional import GroupNorm, gn_time_linear, graph_aggregate
from .gcn_ode import (GcnOdeAdjointField, GcnOdeField, GcnOdePartAdjointField, GcnOdePartField, GcnOdeSpec, _Shared,
                      odefunc_apply, tuned_graph)
from .graph import as_graph
from .layers import FixedGraphConvolution, GraphConvolution
from .odeint import odeint_adjoint as odeint


def _gn(dim):
    """nn.GroupNorm(min(32, dim), dim) of the reference; the subclass routes 2-D GPU inputs to libgraphode."""
    return GroupNorm(min(32, dim), dim)


class _PlanModel(nn.Module):
    """Interprets `self.plan`; the trailing log_softmax is common to every reference model.  `kit` names the layer
    classes a model is assembled from (set after the GCN classes below; gat_models.py re-binds it to the GAT ones),
    and the graph travels as `*graph`: (adj,) here, (src, tgt, Mtgt) for the GAT variant."""
    plan = ()
    kit = None
    ode_attr = None          # name of the attribute holding the ODEBlock (for the nfe property)

    def run_plan(self, x, graph_args):
        saved = None
        for step in self.plan:
            op = step[0]
            if op == "gc":
                x = getattr(self, step[1])(x, *graph_args)
            elif op == "relu":
                x = F.relu(x)
            elif op == "drop":
                x = F.dropout(x, self.dropout, training=self.training)
            elif op == "norm":
                x = getattr(self, step[1])(x)
            elif op == "save":
                saved = x
            elif op == "add":
                x = x + saved
            elif op == "head":
                x = x[:, :step[1]]
            else:
                raise ValueError("unknown plan step %r" % (step,))
        return F.log_softmax(x, dim=1)

    def forward(self, x, adj):
        return self._body(self._input(x), (adj,))

    def _input(self, x):
        # GCN-dense-paper/models.py applies dropout to the input features of every model (e.g. :17, :222)
        return F.dropout(x, self.dropout, training=self.training) if self.kit.input_dropout else x

    def _body(self, x, graph):
        """Model body over the graph argument tuple; `forward` only fixes the public signature of the variant."""
        return self.run_plan(x, graph)

    # `model.nfe = 0` is executed on every model by the harness (GCN/train_res.py:64); ODE models
    # forward it to their block, the others just keep the number.
    @property
    def nfe(self):
        return getattr(self, self.ode_attr).nfe if self.ode_attr else self.__dict__.get("_nfe", 0)

    @nfe.setter
    def nfe(self, value):
        if self.ode_attr:
            getattr(self, self.ode_attr).nfe = value
        else:
            self.__dict__["_nfe"] = value


# ---- the ODE function and block ----------------------------------------------------------------
class ODEfunc(nn.Module):
    """f(t, x) = relu(gc1([t | norm1(x)])), counting calls in `nfe` (reference: GCN/models.py:161-179)."""

    _gode_counts_nfe = True
    layer_cls = FixedGraphConvolution       # variants (dense_paper.py) substitute their own initialisation
    node_order = "auto"                     # gcn_ode.tuned_graph: "auto" | "given" | "degree" (set by ODEBlock)

    def __init__(self, dim):
        super(ODEfunc, self).__init__()
        self.norm1 = _gn(dim)
        self.gc1 = self.layer_cls(dim + 1, dim)
        self.nfe = 0
        self._shared = None

    def set_adj(self, adj):
        self.gc1.set_adj(adj)

    def _spec(self):
        return GcnOdeSpec(as_graph(self.gc1.adj), self.gc1.weight, self.gc1.bias, self.norm1.weight,
                          self.norm1.bias, self.norm1.num_groups, self.norm1.eps)

    def forward(self, t, x):
        self.nfe += 1
        return odefunc_apply(self.gc1.adj, float(t), x, self.gc1.weight, self.gc1.bias, self.norm1.weight,
                             self.norm1.bias, self.norm1.num_groups, self.norm1.eps)

    def gode_plan_token(self, y0):
        """Identity of the graph behind this ODE function (odeint keys its captured solves on it)."""
        if self.gc1.bias is None or y0.dim() != 2 or self.gc1.adj is None:
            return None
        if getattr(self.gc1.adj, "is_partitioned", False):
            return None                         # a collective per f-eval: no captured solve
        return ("gcn", id(as_graph(self.gc1.adj)))

    def gode_fields(self, y0):
        """Hook for graph_odenet_amd.odeint: fused forward / adjoint fields (None -> autograd path)."""
        if self.gc1.bias is None or y0.dim() != 2:
            return None
        plist = [p for p in self.parameters() if p.requires_grad]
        if len(plist) != 4:
            return None           # frozen parameters: take the generic autograd path
        spec = self._spec()
        sh = self._shared
        if sh is None or sh.n != y0.shape[0] or sh.d != y0.shape[1] or sh.S.device != y0.device:
            sh = self._shared = _Shared(spec.graph, spec.d, y0.device)
        names = {id(self.norm1.weight): "gamma", id(self.norm1.bias): "beta",
                 id(self.gc1.weight): "W", id(self.gc1.bias): "b"}
        order = [names[id(p)] for p in plist]
        if getattr(spec.graph, "is_partitioned", False):
            return GcnOdePartField(spec, sh), (lambda: GcnOdePartAdjointField(spec, sh, order)), tuple(plist)
        # large graphs whose hot rows crowd a few address residues: integrate on the hubs-first renumbering
        # (gcn_ode.tuned_graph - a deterministic function of the graph and `node_order`); the solver permutes the
        # state rows on entry and exit, everything in between is row-local or the SpMM itself
        spec.graph, rows, inverse = tuned_graph(spec.graph, spec.d, self.node_order)

        def mark(field):
            field.row_order, field.row_inverse = rows, inverse
            return field
        return mark(GcnOdeField(spec, sh)), (lambda: mark(GcnOdeAdjointField(spec, sh, order))), tuple(plist)


class ODEfunc2(nn.Module):
    """Two stacked (FixedGC -> relu -> GroupNorm), time column re-attached before each convolution
    (reference: GCN/models.py:551-575).  norm1 is folded into the prologue of the second dense product;
    norm2 has no consumer inside f and runs as the stand-alone GroupNorm kernel."""

    layer_cls = FixedGraphConvolution

    def __init__(self, dim, dropout):
        super(ODEfunc2, self).__init__()
        self.norm1, self.norm2 = _gn(dim), _gn(dim)
        self.gc1 = self.layer_cls(dim + 1, dim)
        self.gc2 = self.layer_cls(dim + 1, dim)
        self.dropout = dropout
        self.nfe = 0

    def set_adj(self, adj):
        self.gc1.set_adj(adj)
        self.gc2.set_adj(adj)

    def forward(self, t, x):
        self.nfe += 1
        t = float(t)
        h = graph_aggregate(as_graph(self.gc1.adj), gn_time_linear(x, t, self.gc1.weight), self.gc1.bias, relu=True)
        s = gn_time_linear(h, t, self.gc2.weight, self.norm1.weight, self.norm1.bias,
                           self.norm1.num_groups, self.norm1.eps)
        h = graph_aggregate(as_graph(self.gc2.adj), s, self.gc2.bias, relu=True)
        return self.norm2(h)


class ODEBlock(nn.Module):
    """y(1) of y' = odefunc(t, y), y(0) = x (reference: GCN/models.py:181-201)."""

    def __init__(self, odefunc, tol=1e-5, method=None, step_size=None, node_order=None):
        super(ODEBlock, self).__init__()
        self.odefunc = odefunc
        if node_order is not None:            # extension: "auto" (default rule) | "given" | "degree" (gcn_ode.tuned_graph)
            from .gcn_ode import NODE_ORDERS
            if node_order not in NODE_ORDERS:
                raise ValueError("ODEBlock: node_order must be one of %s" % (NODE_ORDERS,))
            odefunc.node_order = node_order
        self.integration_time = torch.tensor([0, 1]).float()
        self.tol = tol
        self.method = method
        self.step_size = step_size

    def forward(self, x, *graph):
        self.integration_time = self.integration_time.type_as(x)
        self.odefunc.set_adj(*graph)
        options = None if self.step_size is None else {"step_size": self.step_size}
        # the reference keeps `out[1]` of the stack over integration_time (GCN/models.py:200); asking for that state alone
        # saves three passes over it here and in the backward pass (odeint._OdeintAdjoint, last_only)
        if self.integration_time.numel() == 2:
            return odeint(self.odefunc, x, self.integration_time, rtol=self.tol, atol=self.tol,
                          method=self.method, options=options, _last_only=True)
        return odeint(self.odefunc, x, self.integration_time, rtol=self.tol, atol=self.tol,
                      method=self.method, options=options)[1]

    @property
    def nfe(self):
        return self.odefunc.nfe

    @nfe.setter
    def nfe(self, value):
        self.odefunc.nfe = value


# ---- fixed-depth variants (reference: GCN/models.py:8-253) ----------------------------------------
class GCN(_PlanModel):
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("gc", "gc2"))

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(GCN, self).__init__()
        self.gc1 = self.kit.GraphConvolution(nfeat, nhid)
        self.gc2 = self.kit.GraphConvolution(nhid, nclass)
        self.dropout = dropout


class RGCN2(_PlanModel):
    def __init__(self, nfeat, nhid, nclass, dropout):
        super(RGCN2, self).__init__()
        if nhid < nclass:
            raise ValueError("nhid must be equal or larger than nclass")
        self.gc1 = self.kit.GraphConvolution(nfeat, nhid)
        self.gc2 = self.kit.GraphConvolution(nhid, nhid)
        self.nclass = nclass
        self.dropout = dropout
        self.plan = (("gc", "gc1"), ("relu",), ("drop",), ("save",), ("gc", "gc2"), ("add",), ("head", nclass))


class ODEGCN2(_PlanModel):
    """The reference's forward reads self.nclass which its __init__ never sets; it is set here."""
    ode_attr = "gc2"

    def __init__(self, nfeat, nhid, nclass, dropout, method=None, step_size=None, tol=1e-5):
        super(ODEGCN2, self).__init__()
        if nhid < nclass:
            raise ValueError("nhid must be equal or larger than nclass")
        self.gc1 = self.kit.GraphConvolution(nfeat, nhid)
        self.gc2 = ODEBlock(self.kit.ODEfunc(nhid), tol=tol, method=method, step_size=step_size)
        self.nclass = nclass
        self.dropout = dropout
        self.plan = (("gc", "gc1"), ("relu",), ("gc", "gc2"), ("head", nclass))


class _Three(_PlanModel):
    """gc1 (nfeat->nhid), gc2 (nhid->nhid or ODE block), gc3 (nhid->nclass) + optional GroupNorms."""

    def __init__(self, nfeat, nhid, nclass, dropout, norms=(), ode=None):
        super(_Three, self).__init__()
        self.gc1 = self.kit.GraphConvolution(nfeat, nhid)
        if "norm1" in norms:
            self.norm1 = _gn(nhid)
        self.gc2 = self.kit.GraphConvolution(nhid, nhid) if ode is None else ODEBlock(self.kit.ODEfunc(nhid), **ode)
        if "norm2" in norms:
            self.norm2 = _gn(nhid)
        self.gc3 = self.kit.GraphConvolution(nhid, nclass)
        self.dropout = dropout


class GCN3(_Three):
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("gc", "gc2"), ("relu",), ("drop",), ("gc", "gc3"))

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(GCN3, self).__init__(nfeat, nhid, nclass, dropout)


class GCN3norm(_Three):
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("gc", "gc2"), ("relu",), ("norm", "norm2"), ("gc", "gc3"))

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(GCN3norm, self).__init__(nfeat, nhid, nclass, dropout, norms=("norm2",))


class RGCN3(_Three):
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("save",), ("gc", "gc2"), ("relu",), ("drop",), ("add",), ("gc", "gc3"))

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(RGCN3, self).__init__(nfeat, nhid, nclass, dropout)


class RGCN3norm(_Three):
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("save",), ("gc", "gc2"), ("relu",), ("norm", "norm2"), ("add",),
            ("gc", "gc3"))

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(RGCN3norm, self).__init__(nfeat, nhid, nclass, dropout, norms=("norm2",))


class RGCN3fullnorm(_Three):
    plan = (("gc", "gc1"), ("relu",), ("norm", "norm1"), ("save",), ("gc", "gc2"), ("relu",), ("norm", "norm2"),
            ("add",), ("gc", "gc3"))

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(RGCN3fullnorm, self).__init__(nfeat, nhid, nclass, dropout, norms=("norm1", "norm2"))


class ODEGCN3(_Three):
    """The north-star model: gc1 -> relu -> dropout -> ODEBlock -> gc3 (reference: GCN/models.py:204-226)."""
    plan = (("gc", "gc1"), ("relu",), ("drop",), ("gc", "gc2"), ("gc", "gc3"))
    ode_attr = "gc2"

    def __init__(self, nfeat, nhid, nclass, dropout, method=None, step_size=None, tol=1e-5):
        super(ODEGCN3, self).__init__(nfeat, nhid, nclass, dropout, ode=dict(tol=tol, method=method, step_size=step_size))


class ODEGCN3fullnorm(_Three):
    plan = (("gc", "gc1"), ("relu",), ("norm", "norm1"), ("gc", "gc2"), ("gc", "gc3"))
    ode_attr = "gc2"

    def __init__(self, nfeat, nhid, nclass, dropout, method=None, step_size=None, tol=1e-5):
        super(ODEGCN3fullnorm, self).__init__(nfeat, nhid, nclass, dropout, norms=("norm1",),
                                              ode=dict(tol=tol, method=method, step_size=step_size))


# ---- depth-parametrised variants (reference: GCN/models.py:255-600) -------------------------------
class _Deep(_PlanModel):
    """`self.gcs` ModuleList: first layer, a list of middle blocks, last layer."""

    def _build(self, nfeat, nhid, nclass, dropout, middle):
        self.gcs = nn.ModuleList([self.kit.GraphConvolution(nfeat, nhid)] + list(middle) + [self.kit.GraphConvolution(nhid, nclass)])
        self.dropout = dropout

    @property
    def nfe(self):
        return sum(b.nfe for b in self.gcs if isinstance(b, ODEBlock))

    @nfe.setter
    def nfe(self, value):
        for b in self.gcs:
            if isinstance(b, ODEBlock):
                b.nfe = value


class GCNK(_Deep):
    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=2):
        super(GCNK, self).__init__()
        if nlayers < 2:
            raise ValueError("Can't make a GCN with less than 2 layers")
        self.n_layers = nlayers
        self._build(nfeat, nhid, nclass, dropout, [self.kit.GraphConvolution(nhid, nhid) for _ in range(nlayers - 2)])

    def _body(self, x, graph):
        for gc in self.gcs[:-1]:
            x = F.dropout(F.relu(gc(x, *graph)), self.dropout, training=self.training)
        return F.log_softmax(self.gcs[-1](x, *graph), dim=1)


class _ResDeep(_Deep):
    """Residual stacks of the depth sweep (reference: GCN/models.py:280-522).  One rule covers the family: the
    state is saved every `residue_layers` middle layers and added back after the last layer of the block (also
    after a trailing, incomplete block); `with_norm` replaces the middle dropout by a GroupNorm;
    `residue_layers` = 0 means no skip connections at all (GCNKnorm)."""
    with_norm = False
    min_layers_msg = "Can't make a Residual GCN with less than {} layers using {} layers for each residual block"

    def _setup(self, nfeat, nhid, nclass, dropout, nlayers, residue_layers):
        need = 2 + residue_layers
        if nlayers < need:
            raise ValueError(self.min_layers_msg.format(need, residue_layers))
        self.n_layers = nlayers
        self._build(nfeat, nhid, nclass, dropout, [self.kit.GraphConvolution(nhid, nhid) for _ in range(nlayers - 2)])
        if self.with_norm:
            self.norms = nn.ModuleList([_gn(nhid) for _ in range(nlayers - 2)])
        self.residue_layers = residue_layers

    def _body(self, x, graph):
        x = F.dropout(F.relu(self.gcs[0](x, *graph)), self.dropout, training=self.training)
        span = self.residue_layers
        left, saved = 0, None                     # layers left in the current residual block
        for k, gc in enumerate(self.gcs[1:-1]):
            if span and left == 0:
                saved, left = x, span
            x = F.relu(gc(x, *graph))
            x = self.norms[k](x) if self.with_norm else F.dropout(x, self.dropout, training=self.training)
            if span:
                left -= 1
                if left == 0:
                    x = x + saved
        if span and left > 0:
            x = x + saved
        return F.log_softmax(self.gcs[-1](x, *graph), dim=1)


class GCNKnorm(_ResDeep):
    with_norm = True

    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=2):
        super(GCNKnorm, self).__init__()
        if nlayers < 2:
            raise ValueError("Can't make a GCN with less than 2 layers")
        self._setup(nfeat, nhid, nclass, dropout, nlayers, 0)


class RESK(_ResDeep):
    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=3, residue_layers=1):
        super(RESK, self).__init__()
        self._setup(nfeat, nhid, nclass, dropout, nlayers, residue_layers)


class RESKnorm(_ResDeep):
    with_norm = True

    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=3, residue_layers=1):
        super(RESKnorm, self).__init__()
        self._setup(nfeat, nhid, nclass, dropout, nlayers, residue_layers)


class RESK1(_ResDeep):
    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=3):
        super(RESK1, self).__init__()
        self._setup(nfeat, nhid, nclass, dropout, nlayers, 1)


class RESK2(_ResDeep):
    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=4):
        super(RESK2, self).__init__()
        self._setup(nfeat, nhid, nclass, dropout, nlayers, 2)


class RESK1norm(_ResDeep):
    with_norm = True

    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=3):
        super(RESK1norm, self).__init__()
        self._setup(nfeat, nhid, nclass, dropout, nlayers, 1)


class RESK2norm(_ResDeep):
    with_norm = True

    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=4):
        super(RESK2norm, self).__init__()
        self._setup(nfeat, nhid, nclass, dropout, nlayers, 2)


class _OdeDeep(_Deep):
    def _body(self, x, graph):
        x = F.dropout(F.relu(self.gcs[0](x, *graph)), self.dropout, training=self.training)
        for block in self.gcs[1:-1]:
            x = block(x, *graph)
        return F.log_softmax(self.gcs[-1](x, *graph), dim=1)


class ODEK1(_OdeDeep):
    """First layer, (nlayers-2) ODE blocks, last layer."""

    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=3, method=None, step_size=None):
        super(ODEK1, self).__init__()
        if nlayers < 3:
            raise ValueError("Can't make a Residual GCN with less than 3 layers")
        self.n_layers = nlayers
        self._build(nfeat, nhid, nclass, dropout,
                    [ODEBlock(self.kit.ODEfunc(nhid), method=method, step_size=step_size) for _ in range(nlayers - 2)])


class ODEK2(_OdeDeep):
    """Blocks of ODEfunc2 (+ one ODEfunc block when nlayers is odd).  Quirk Q1 of SURVEY.md is kept on
    purpose: the reference passes `dropout` as the ODEBlock tolerance (GCN/models.py:587)."""

    def __init__(self, nfeat, nhid, nclass, dropout, nlayers=4, method=None, step_size=None):
        super(ODEK2, self).__init__()
        if nlayers < 4:
            raise ValueError("Can't make a Residual GCN with less than 4 layers using 2 layers for each residual block")
        self.n_layers = nlayers
        middle = [ODEBlock(self.kit.ODEfunc2(nhid, dropout), dropout, method=method, step_size=step_size)
                  for _ in range((nlayers - 2) // 2)]
        if nlayers % 2 == 1:
            middle.append(ODEBlock(self.kit.ODEfunc(nhid), method=method, step_size=step_size))
        self._build(nfeat, nhid, nclass, dropout, middle)


class GcnKit:
    """Layer classes of the GCN variant (reference: GCN/layers.py, GCN/models.py; GCN-sum is the same code)."""
    GraphConvolution = GraphConvolution
    ODEfunc = ODEfunc
    ODEfunc2 = ODEfunc2
    input_dropout = False


_PlanModel.kit = GcnKit

ZOO = ("GCN", "RGCN2", "ODEGCN2", "GCN3", "GCN3norm", "RGCN3", "RGCN3norm", "RGCN3fullnorm", "ODEGCN3", "ODEGCN3fullnorm",
       "GCNK", "GCNKnorm", "RESK1", "RESK2", "RESK", "RESK1norm", "RESK2norm", "RESKnorm", "ODEK1", "ODEK2")


def rebind_zoo(namespace, module_name, kit, forward=None, what=""):
    """Defines every class of ZOO in `namespace` as a subclass assembled from `kit` (optionally with another public
    forward signature): how the GAT, dense-paper and MLP-sum variants get their `models` module."""
    for name in ZOO:
        attrs = {"kit": kit, "__module__": module_name,
                 "__doc__": "%s of the %s variant (same body as graph_odenet_amd.models.%s)." % (name, what, name)}
        if forward is not None:
            attrs["forward"] = forward
        namespace[name] = type(name, (globals()[name],), attrs)
