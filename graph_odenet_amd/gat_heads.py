"""H-head edge attention (BASELINE.json configs[2]: "Citeseer GAT 8-head ODEBlock rk4"; SURVEY.md §8(d) C3: "8
independent reference-style heads batched").  The reference's layer (GAT/layers.py:11-63) has one head; an H-head
layer here is H of those layers on the same graph with their outputs concatenated,

    out = [ head_0(x) | head_1(x) | ... | head_{H-1}(x) ]          head_h : in_features -> out_features / H,

each head with its own f / w Linear layers, its own global logit maximum (:47) and its own eps-regularised
per-target normalisation (:53) - the parameters are literally H reference layers (`heads.<h>.f.weight`, ...).

Execution: the H heads run as ONE head on the H-fold graph.  Virtual node v*H + h carries head h of node v, so the
N x (H*o) projection matrices of all heads - produced by ONE dense product with the heads' weights side by side,
square (d+1) x d inside an ODE function where H*o = d - ARE the (N*H) x o matrices of the virtual nodes, and the
(N*H) x o result is the N x (H*o) concatenation: no copy in either direction.  Edge s -> t becomes the H edges
s*H+h -> t*H+h.  Every aggregation / VJP / scatter kernel of the one-head path runs unchanged on that graph; the
per-head maximum is handled by gode_gat_logits_heads_f32 (logits shifted by their head's maximum, the aggregation then
runs with amax = 0) and gode_gat_maxpath_heads_f32 (gradient path through each head's maximum).  The per-head message
biases are folded into the target-side projection (z_e = Ps[src] + (Pt[tgt] + bf_h)); the logit biases are added by the
logits kernel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.module import Module

from . import _lib, ops
from . import models as _gcn_models
from .gat_layers import EdgeGraph, GraphConvolution, edge_graph
from .gat_ode import GatOdeAdjointField, GatOdeField
from .models import _gn


def heads_graph(eg, heads):
    """The H-fold EdgeGraph of `eg` (cached on it)."""
    cache = eg.__dict__.setdefault("_heads", {})
    hit = cache.get(heads)
    if hit is not None:
        return hit
    if not eg.canonical:
        raise NotImplementedError("multi-head attention needs an Mtgt whose rows agree with tgt")
    dev = eg.src.device
    h = torch.arange(heads, device=dev)
    src_v = (eg.src.to(torch.int64)[:, None] * heads + h).reshape(-1)
    tgt_v = (eg.tgt.to(torch.int64)[:, None] * heads + h).reshape(-1)
    ev = eg.E * heads
    val = eg.Mt.val.repeat_interleave(heads) if eg.Mt.val is not None else torch.ones(ev, device=dev)
    M = torch.sparse_coo_tensor(torch.stack([tgt_v, torch.arange(ev, device=dev)]), val, (eg.n * heads, ev))
    egv = EdgeGraph(src_v, tgt_v, M)
    if not egv.canonical:
        raise RuntimeError("H-fold graph lost its target order")
    egv.base, egv.heads = eg, heads
    cache[heads] = egv
    return egv


class _EdgeAttentionHeadsFn(torch.autograd.Function):
    """Concatenated outputs of H heads from Ps, Pt (N x H*o, biases folded into Pt) and A2 (N x 2H: per head the
    logit part by source and by target, bias folded into the latter)."""

    @staticmethod
    def forward(ctx, egv, heads, Ps, Pt, A2, eps):
        Ps, Pt, A2 = Ps.contiguous(), Pt.contiguous(), A2.contiguous()
        n, d = Ps.shape
        o, nv = d // heads, n * heads
        f = dict(dtype=torch.float32, device=Ps.device)
        a, zero = torch.empty(egv.E, **f), torch.zeros(1, **f)
        out, w, den = torch.empty(n, d, **f), torch.empty(egv.E, **f), torch.empty(nv, **f)
        proj = ops.gat_proj(Ps.view(nv, o), Pt.view(nv, o), A2.view(nv, 2))
        ops.gat_logits_heads(proj, egv.src, egv.tgt, heads, a)
        ops.gat_agg_fwd(egv, proj, o, torch.zeros(o, **f), a, zero, eps, out.view(nv, o), w, den)
        ctx.egv, ctx.heads = egv, heads
        ctx.save_for_backward(Ps, Pt, A2, a, w, den, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        Ps, Pt, A2, a, w, den, out = ctx.saved_tensors
        egv, heads = ctx.egv, ctx.heads
        n, d = Ps.shape
        o, nv = d // heads, n * heads
        f = dict(dtype=torch.float32, device=Ps.device)
        dz, da = torch.empty(egv.E, o, **f), torch.empty(egv.E, **f)
        dPs, dPt, dA2 = torch.empty(n, d, **f), torch.empty(n, d, **f), torch.empty(n, 2 * heads, **f)
        proj = ops.gat_proj(Ps.view(nv, o), Pt.view(nv, o), A2.view(nv, 2))
        ops.gat_vjp(egv, proj, o, torch.zeros(o, **f), a, torch.zeros(1, **f), w, den, out.view(nv, o), dz, da,
                    dPs.view(nv, o), dPt.view(nv, o), dA2.view(nv, 2), dout=dout.contiguous().view(nv, o), heads=heads)
        return None, None, dPs, dPt, dA2, None


class MultiHeadGraphConvolution(Module):
    """`heads` reference layers (GAT/layers.py:11-63) on one graph, outputs concatenated."""

    def __init__(self, in_features, out_features, heads=8, bias=True, act=F.relu, eps=1e-6):
        super(MultiHeadGraphConvolution, self).__init__()
        if heads < 1 or out_features % heads:
            raise ValueError("MultiHeadGraphConvolution: %d heads do not divide %d output features" % (heads, out_features))
        if heads > 64:
            raise ValueError("MultiHeadGraphConvolution: at most 64 heads")
        self.in_features, self.out_features, self.n_heads = in_features, out_features, heads
        self.eps, self.act = eps, act
        self.heads = nn.ModuleList([GraphConvolution(in_features, out_features // heads, bias, act, eps)
                                    for _ in range(heads)])

    def packed(self):
        """(Wsrc, Wtgt: i x H*o;  Wlog: i x 2H;  bf: H*o;  ba: 2H = [0, bw_0, 0, bw_1, ...]) as differentiable
        functions of the heads' parameters."""
        # a handful of launches whatever the head count (one stack per parameter kind, one permuted copy per block): the
        # per-head cat / stack / zeros_like form cost ~30 launches per call and ~150 per training step at 8 heads
        i, H, o = self.in_features, self.n_heads, self.out_features // self.n_heads
        Wf = torch.stack([hd.f.weight for hd in self.heads])                    # H x o x 2i
        Wsrc = Wf[:, :, :i].permute(2, 0, 1).reshape(i, H * o)
        Wtgt = Wf[:, :, i:].permute(2, 0, 1).reshape(i, H * o)
        ww = torch.stack([hd.w.weight[0] for hd in self.heads])                 # H x 2i
        Wlog = ww.view(H, 2, i).permute(2, 0, 1).reshape(i, 2 * H)              # column 2h: source part, 2h+1: target part
        bf = torch.cat([hd.f.bias for hd in self.heads])
        bw = torch.cat([hd.w.bias for hd in self.heads])
        ba = torch.stack([torch.zeros_like(bw), bw], 1).reshape(2 * H)
        return Wsrc, Wtgt, Wlog, bf, ba

    def forward(self, x, src, tgt, Mtgt):
        return _heads_forward(self, x, src, tgt, Mtgt)

    def __repr__(self):
        return "%s (%d -> %d, %d heads)" % (self.__class__.__name__, self.in_features, self.out_features, self.n_heads)


def _heads_forward(layer, x, src, tgt, Mtgt):
    if layer.act is not F.relu:          # any other activation: every head through the general (unfused) layer path
        return torch.cat([hd(x, src, tgt, Mtgt) for hd in layer.heads], 1)
    eg = edge_graph(src, tgt, Mtgt)
    Wsrc, Wtgt, Wlog, bf, ba = layer.packed()
    from .functional import dense
    Ps, Pt, A2 = dense(x, Wsrc), dense(x, Wtgt, bf), dense(x, Wlog, ba)
    if eg.E == 0:
        return torch.zeros(x.shape[0], layer.out_features, dtype=x.dtype, device=x.device) + 0.0 * (Ps.sum() + Pt.sum() + A2.sum())
    return _EdgeAttentionHeadsFn.apply(heads_graph(eg, layer.n_heads), layer.n_heads, Ps, Pt, A2, layer.eps)


class FixedMultiHeadGraphConvolution(MultiHeadGraphConvolution):
    """The same layer with (src, tgt, Mtgt) held as attributes (the role of GAT/layers.py:67-127)."""

    def __init__(self, in_features, out_features, heads=8, bias=True, act=F.relu, eps=1e-6):
        super(FixedMultiHeadGraphConvolution, self).__init__(in_features, out_features, heads, bias, act, eps)
        self.src = self.tgt = self.Mtgt = torch.Tensor([[1]])

    def set_adj(self, src, tgt, Mtgt):
        self.src, self.tgt, self.Mtgt = src, tgt, Mtgt

    def forward(self, x):
        return _heads_forward(self, x, self.src, self.tgt, self.Mtgt)


# ---- fused ODE function -------------------------------------------------------------------------------------------
PAD_LOGITS_MIN_ROWS = 4096


class GatHeadsSpec:
    """Packed description of ODEfunc(dim, heads): theta = [Wsrc | Wtgt | Wlog | bf | bw | gamma | beta] with
    Wsrc, Wtgt (d+1) x d (head h in columns h*o..), Wlog (d+1) x 2H, bf d, bw H."""

    def __init__(self, egv, layer, norm):
        self.eg, self.layer, self.norm = egv, layer, norm           # eg: the H-fold graph (what the kernels see)
        self.heads = layer.n_heads
        self.d, self.i = layer.out_features, layer.in_features
        if self.i != self.d + 1:
            raise ValueError("GatHeadsSpec: the ODE layer maps d+1 -> d features")
        self.o = self.d // self.heads
        self.groups, self.eps_gn, self.eps = int(norm.num_groups), float(norm.eps), float(layer.eps)
        dev = norm.weight.device
        f = dict(dtype=torch.float32, device=dev)
        i, d, H = self.i, self.d, self.heads
        self.Wsrc, self.Wtgt, self.Wlog = torch.empty(i, d, **f), torch.empty(i, d, **f), torch.empty(i, 2 * H, **f)
        self.bf, self.bw = torch.empty(d, **f), torch.empty(H, **f)
        # Large graphs: the 2H logit columns ride the square MFMA kernels as a zero-padded (d+1) x d block (0.4-0.5 ms
        # per product at 2^20 x 128 against 2.2 / 5.5 / 3.8 ms for the generic kernels on a (d+1) x 16 block); small
        # graphs are launch-bound and keep the compact product.
        self.pad_logits = (egv.base.n >= PAD_LOGITS_MIN_ROWS and d in (16, 32, 64, 128) and 4 < 2 * H <= d)
        self.Wlog_pad = torch.zeros(i, d, **f) if self.pad_logits else None
        self.Wpacked = None
        self.n = egv.base.n
        self.refresh()
        self.gamma, self.beta = norm.weight.detach(), norm.bias.detach()
        self.n = egv.base.n
        self.off, p = {}, 0
        for name, ln in (("Wsrc", i * d), ("Wtgt", i * d), ("Wlog", i * 2 * H), ("bf", d), ("bw", H), ("gamma", d), ("beta", d)):
            self.off[name] = (p, p + ln)
            p += ln
        self.n_theta = p

    def refresh(self):
        with torch.no_grad():
            Wsrc, Wtgt, Wlog, bf, ba = self.layer.packed()
            self.Wsrc.copy_(Wsrc); self.Wtgt.copy_(Wtgt); self.Wlog.copy_(Wlog); self.bf.copy_(bf); self.bw.copy_(ba[1::2])
            if self.Wlog_pad is not None:
                self.Wlog_pad[:, :2 * self.heads].copy_(Wlog)
            if not self.pad_logits:
                from .gat_ode import _repack
                _repack(self, self.heads)

    def views(self, theta):
        v = {k: theta[a:b] for k, (a, b) in self.off.items()}
        v["Wsrc"], v["Wtgt"] = v["Wsrc"].view(self.i, self.d), v["Wtgt"].view(self.i, self.d)
        v["Wlog"] = v["Wlog"].view(self.i, 2 * self.heads)
        return v


class _HeadsWork:
    def __init__(self, spec, device):
        n, d, o, H, E = spec.n, spec.d, spec.o, spec.heads, spec.eg.E
        nv = n * H
        lib = _lib.load()
        f = dict(dtype=torch.float32, device=device)
        self.X = torch.empty(n, d, **f)
        self.Ps, self.Pt, self.A2 = torch.empty(n, d, **f), torch.empty(n, d, **f), torch.empty(n, 2 * H, **f)
        self.dPs, self.dPt, self.dA2 = torch.empty(n, d, **f), torch.empty(n, d, **f), torch.empty(n, 2 * H, **f)
        self.a, self.zero, self.bf0 = torch.empty(max(E, 1), **f)[:E], torch.zeros(1, **f), torch.zeros(o, **f)
        self.wgt, self.den = torch.zeros(max(E, 1), **f)[:E], torch.empty(nv, **f)
        self.dz, self.da = torch.zeros(max(E, 1), o, **f)[:E], torch.zeros(max(E, 1), **f)[:E]
        self.proj = ops.gat_proj(self.Ps.view(nv, o), self.Pt.view(nv, o), self.A2.view(nv, 2))
        self.np_b = lib.gode_gemm_bwd_parts(n)
        self.gp, self.bp = torch.empty(3 * self.np_b, d, **f), torch.empty(3 * self.np_b, d, **f)
        npw = lib.gode_wgrad_parts(n)
        self.wp = [torch.empty(npw, spec.i * d, **f), torch.empty(npw, spec.i * d, **f), torch.empty(npw, spec.i * 2 * H, **f)]
        self.ba_grad = torch.empty(2 * H, **f)
        # extras of the C-level dopri5 step (csrc/gat_driver.hip, heads > 1)
        u8 = dict(dtype=torch.uint8, device=device)
        self.zeros = torch.zeros(max(o, 1), **f)
        self.pair = torch.empty(2 * H, **f)
        self.heads_scratch = torch.empty(max(lib.gode_gat_heads_scratch_bytes(E, H), 16), **u8)
        self.colsum_scratch = torch.empty(max(lib.gode_colsum_scratch_bytes(n, d), 16), **u8)
        self.colsum_scratch2 = torch.empty(max(lib.gode_colsum_scratch_bytes(n, 2 * H), 16), **u8)
        self.err_scratch = torch.empty(lib.gode_rk_errnorm_scratch_bytes(), **u8)
        self.small_part = None
        self.step_parts = None
        if not spec.pad_logits and lib.gode_gat_small_supported(n, d, spec.groups, H):
            self.small_part = ops.gat_small_part(n, d, H, device)
        if spec.pad_logits:
            self.A2pad, self.dA2pad = torch.empty(n, d, **f), torch.zeros(n, d, **f)     # columns >= 2H of dA2pad stay 0
            self.wp[2] = torch.empty(npw, spec.i * d, **f)
            self.gWlog_pad = torch.empty(spec.i, d, **f)


class GatHeadsField(GatOdeField):
    """f(t, x) = relu(heads([t | GroupNorm(x)])) as a kernel sequence (the one-head sequence of gat_ode.py on the
    H-fold graph).  Adaptive steps run as one C call (csrc/gat_driver.hip with heads = H) on launch-bound graphs; above
    PAD_LOGITS_MIN_ROWS nodes the logit columns ride the padded square kernels and the solver takes the per-stage
    path."""

    def __init__(self, spec, work):
        self.s, self.w = spec, work
        self.heads = spec.heads
        self.token = ("gat-heads", id(spec.eg))

    @property
    def dopri5_step_native(self):
        return None if self.s.pad_logits else self._dopri5_step

    def _dopri5_step(self, y, kk, y1, t, h, rtol, atol):
        return GatOdeField.dopri5_step_native(self, y, kk, y1, t, h, rtol, atol)

    def _structs(self, adjoint):
        s, w, eg = self.s, self.w, self.s.eg
        fs = _lib.GatOdeFunc()
        fs.mt = ops._edge_csr(eg, s.o + 4)
        for name, gph in (("ms_inc", eg.Ms_inc), ("mt_inc", eg.Mt_inc)):
            gs = _lib.Graph()
            gs.rowptr, gs.col, gs.val = gph.rowptr.data_ptr(), gph.col.data_ptr(), None
            gs.items, gs.n_items = (gph.items.data_ptr() if gph.items is not None else None), gph.n_items
            gs.long_rows = gph.long_rows.data_ptr() if gph.long_rows is not None else None
            gs.n_long = gph.n_long
            part = gph.partial(s.o) if adjoint else None
            gs.partial = part.data_ptr() if part is not None else None
            gs.n_rows, gs.nnz = gph.n_rows, gph.nnz
            setattr(fs, name, gs)
        p = lambda t: (t.data_ptr() or None) if t is not None else None      # noqa: E731
        fs.src, fs.tgt, fs.n_edges = p(eg.src), p(eg.tgt), eg.E
        fs.n, fs.d, fs.groups, fs.eps_gn, fs.eps, fs.heads = s.n, s.d, s.groups, s.eps_gn, s.eps, s.heads
        fs.Wsrc, fs.Wtgt, fs.Wlog = s.Wsrc.data_ptr(), s.Wtgt.data_ptr(), s.Wlog.data_ptr()
        fs.bf, fs.bw, fs.gamma, fs.beta = s.bf.data_ptr(), s.bw.data_ptr(), s.gamma.data_ptr(), s.beta.data_ptr()
        fs.Wpacked = s.Wpacked.data_ptr() if s.Wpacked is not None else None
        ws = _lib.GatWorkspace()
        for k in ("X", "Ps", "Pt", "A2", "a", "wgt", "den", "zeros", "heads_scratch"):
            setattr(ws, k, p(getattr(w, k)))
        ws.amax, ws.logits_scratch = p(w.zero), p(w.heads_scratch)          # unused by the heads sequence, must be set
        if adjoint:
            for k in ("dz", "da", "dPs", "dPt", "dA2", "pair", "gp", "bp", "colsum_scratch", "colsum_scratch2"):
                setattr(ws, k, p(getattr(w, k)))
            for j in range(3):
                ws.wp[j] = w.wp[j].data_ptr()
            ws.maxpath_scratch = p(w.heads_scratch)
            ws.small_part = p(w.small_part)
        return fs, ws

    def _project(self, t, y_terms):
        s, w = self.s, self.w
        x_out = w.X if len(y_terms) > 1 else None
        terms = [(1.0, w.X)] if x_out is not None else y_terms
        if self.small():
            ops.gat_project_small(y_terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wsrc, s.Wtgt, s.Wlog, s.heads, s.bf, t,
                                  w.Ps, w.Pt, w.A2, x_out=x_out, packed=s.Wpacked)
            return terms
        ops.gn_time_gemm_pair(y_terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wsrc, s.Wtgt, True, t, w.Ps, w.Pt,
                              x_out=x_out)
        if s.pad_logits:
            ops.gn_time_gemm(terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wlog_pad, True, t, out=w.A2pad)
            w.A2.copy_(w.A2pad[:, :2 * s.heads])
        else:
            ops.gn_time_gemm(terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wlog, True, t, out=w.A2)
        w.Pt.add_(s.bf)                                  # per-head message biases, folded into the target-side part
        return terms

    def _forward(self, t, y_terms, out):
        s, w, eg = self.s, self.w, self.s.eg
        terms = self._project(t, y_terms)
        if self.raw_logits():
            # launch-bound graphs: no launch that shifts the logits - the aggregation reduces its head's partial maxima
            ops.gat_logits_heads_raw(w.proj, eg.src, eg.tgt, s.heads, w.a, w.heads_scratch, bw=s.bw)
            ops.gat_agg_heads_fwd(eg, w.proj, s.o, w.bf0, w.a, w.heads_scratch, s.heads, s.eps, out.view(s.n * s.heads, s.o),
                                  w.wgt, w.den)
            return terms
        ops.gat_logits_heads(w.proj, eg.src, eg.tgt, s.heads, w.a, bw=s.bw)
        ops.gat_agg_fwd(eg, w.proj, s.o, w.bf0, w.a, w.zero, s.eps, out.view(s.n * s.heads, s.o), w.wgt, w.den)
        return terms

    def raw_logits(self):
        return self.s.n * self.s.heads <= 65536 and self.s.eg.E > 8192 and self.small()


class GatHeadsAdjointField(GatHeadsField):
    """Components [y, a, a_t, theta], theta laid out as GatHeadsSpec.off."""

    def __init__(self, spec, work, order):
        GatHeadsField.__init__(self, spec, work)
        self.order = order
        self.n_components = 4
        self.ratio_groups = [[0], [1], [2], [3]]

    new_state = GatOdeAdjointField.new_state
    # fixed-grid steps: the small components advance once per RK step (gat_ode.GatOdeAdjointField)
    deferred_components = GatOdeAdjointField.deferred_components
    DEFER_SMALL = True
    begin_rk4_step = GatOdeAdjointField.begin_rk4_step
    finish_rk4_step = GatOdeAdjointField.finish_rk4_step
    _stage_part = GatOdeAdjointField._stage_part

    def _dopri5_step(self, y, kk, y1, t, h, rtol, atol):
        return GatOdeAdjointField.dopri5_step_native(self, y, kk, y1, t, h, rtol, atol)

    def param_grads(self, comps):
        s = self.s
        i, o, H = s.i, s.o, s.heads
        v = s.views(comps[3].clone())                    # one copy; everything below is a view of it or one permuted copy
        m = {"gamma": v["gamma"], "beta": v["beta"]}
        gWf = torch.cat([v["Wsrc"].view(i, H, o).permute(1, 2, 0), v["Wtgt"].view(i, H, o).permute(1, 2, 0)], 2)    # H x o x 2i
        gww = v["Wlog"].view(i, H, 2).permute(1, 2, 0).reshape(H, 1, 2 * i)                                         # H x 1 x 2i
        gbf, gbw = v["bf"].view(H, o), v["bw"].view(H, 1)
        for h in range(H):
            m["Wf%d" % h], m["bf%d" % h], m["ww%d" % h], m["bw%d" % h] = gWf[h], gbf[h], gww[h], gbw[h]
        return [m[k] for k in self.order]

    def eval(self, t, terms, out):
        s, w = self.s, self.w
        eg, n, d, H = s.eg, s.n, s.d, s.heads
        nv, o = n * H, s.o
        xt = self._forward(t, terms[0], out[0])
        g = s.views(out[3])
        ops.gat_vjp(eg, w.proj, o, w.bf0, w.a, w.zero, w.wgt, w.den, out[0].view(nv, o), w.dz, w.da, w.dPs.view(nv, o),
                    w.dPt.view(nv, o), w.dA2.view(nv, 2), cot_terms=terms[1], cot_scale=-1.0, heads=H,
                    raw_scratch=w.heads_scratch if self.raw_logits() else None, defer_maxpath=self.raw_logits())
        if self.small():
            # (on the raw-logit route the per-head max-path sums are taken off dA2 inside this launch)
            part, deferred = self._stage_part(t)
            ops.gat_dense_vjp_small(xt, n, d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wsrc, s.Wtgt, s.Wlog, H, w.dPs, w.dPt, w.dA2,
                                    out[1], part,
                                    maxfix=(w.heads_scratch, eg.src, eg.tgt) if self.raw_logits() and eg.E > 0 else None,
                                    packed=s.Wpacked)
            if not deferred:
                ops.gat_small_finish(part, n, d, H, t, out[3], out[2])
            return
        from .gat_ode import MERGED_FINISH_MAX_ROWS
        merged = n <= MERGED_FINISH_MAX_ROWS and s.groups > 0 and not s.pad_logits      # one reduction launch per stage
        if merged:
            n_a = ops.colsum_parts(w.dPt, w.colsum_scratch)
            n_b = ops.colsum_parts(w.dA2, w.colsum_scratch2)
        else:
            ops.colsum_(g["bf"], w.dPt)                  # biases sit on the target side: column sums of its gradient
            ops.colsum_(w.ba_grad, w.dA2)
            g["bw"].copy_(w.ba_grad[1::2])
        nb = w.np_b
        affine = s.groups > 0
        Wl, dAl = s.Wlog, w.dA2
        if s.pad_logits:
            w.dA2pad[:, :2 * H].copy_(w.dA2)
            Wl, dAl = s.Wlog_pad, w.dA2pad
        for j, (Wj, dPj) in enumerate(((s.Wsrc, w.dPs), (s.Wtgt, w.dPt), (Wl, dAl))):
            ops.gn_time_gemm_bwd(xt, n, d, s.groups, s.eps_gn, s.gamma, Wj, True, dPj, out=out[1],
                                 pre_terms=[(1.0, out[1])] if j else None,
                                 parts=(w.gp[j * nb:(j + 1) * nb], w.bp[j * nb:(j + 1) * nb]) if affine else None)
        if affine and not merged:
            ops.reduce_parts2_(g["gamma"], w.gp, g["beta"], w.bp)
        elif not affine:
            g["gamma"].zero_(); g["beta"].zero_()
        for j, dPj in enumerate((w.dPs, w.dPt, dAl)):
            ops.wgrad(xt, n, d, s.groups, s.eps_gn, s.gamma, s.beta, dPj, True, part=w.wp[j])
        if merged:
            i, npw = s.i, w.wp[0].shape[0]
            ops.reduce_segments_([
                (g["Wsrc"], w.wp[0], npw, i * d, 0, 1, i * d, s.Wsrc[0], d),          # row 0 of each block = its time row
                (g["Wtgt"], w.wp[1], npw, i * d, 0, 1, i * d, s.Wtgt[0], d),
                (g["Wlog"], w.wp[2], npw, i * 2 * H, 0, 1, i * 2 * H, s.Wlog[0], 2 * H),
                (g["bf"], w.colsum_scratch, n_a, d, 0, 1, d, None, 0),
                (g["bw"], w.colsum_scratch2, n_b, 2 * H, 1, 2, H, None, 0),           # the odd columns of the n x 2H sums
                (g["gamma"], w.gp, 3 * nb, d, 0, 1, d, None, 0),
                (g["beta"], w.bp, 3 * nb, d, 0, 1, d, None, 0)], t, out[2])
            return
        ops.reduce_parts2_(g["Wsrc"].view(-1), w.wp[0], g["Wtgt"].view(-1), w.wp[1])
        if s.pad_logits:
            ops.reduce_parts_(w.gWlog_pad.view(-1), w.wp[2])
            g["Wlog"].copy_(w.gWlog_pad[:, :2 * H])
        else:
            ops.reduce_parts_(g["Wlog"].view(-1), w.wp[2])
        ops.time_row_fixup3_([g["Wsrc"][0], g["Wtgt"][0], g["Wlog"][0]], [s.Wsrc[0], s.Wtgt[0], s.Wlog[0]], t, out[2])


class ODEfunc(nn.Module):
    """relu(gc1([t | norm1(x)])) with an H-head layer: GAT/models.py:161-179 with `heads` reference layers side by
    side (dim must be a multiple of heads)."""

    _gode_counts_nfe = True

    def __init__(self, dim, heads=8):
        super(ODEfunc, self).__init__()
        self.norm1 = _gn(dim)
        self.gc1 = FixedMultiHeadGraphConvolution(dim + 1, dim, heads)
        self.nfe = 0

    def set_adj(self, src, tgt, Mtgt):
        self.gc1.set_adj(src, tgt, Mtgt)

    def forward(self, t, x):
        self.nfe += 1
        xn = self.norm1(x)
        return F.relu(self.gc1(torch.cat([torch.ones_like(xn[:, :1]) * t, xn], 1)))

    def _egv(self):
        layer = self.gc1
        if not torch.is_tensor(layer.src) or layer.src.dim() != 1:
            return None
        eg = edge_graph(layer.src, layer.tgt, layer.Mtgt)
        return heads_graph(eg, layer.n_heads) if eg.canonical and eg.E > 0 else None

    def gode_plan_token(self, y0):
        egv = self._egv()
        return None if egv is None else ("gat-heads", id(egv))

    def gode_fields(self, y0):
        layer, norm = self.gc1, self.norm1
        egv = self._egv()
        if egv is None or layer.act is not F.relu or y0.dim() != 2 or egv.base.n != y0.shape[0]:
            return None
        plist = [p for p in self.parameters() if p.requires_grad]
        names = {id(norm.weight): "gamma", id(norm.bias): "beta"}
        for h, hd in enumerate(layer.heads):
            names.update({id(hd.f.weight): "Wf%d" % h, id(hd.f.bias): "bf%d" % h, id(hd.w.weight): "ww%d" % h,
                          id(hd.w.bias): "bw%d" % h})
        if len(plist) != len(names) or any(id(p) not in names for p in plist):
            return None
        key = ("heads", layer.out_features, y0.device)
        spec = GatHeadsSpec(egv, layer, norm)
        work = egv.__dict__.setdefault("_ode_work", {}).get(key)
        if work is None:
            work = egv.__dict__["_ode_work"][key] = _HeadsWork(spec, y0.device)
        order = [names[id(p)] for p in plist]
        return GatHeadsField(spec, work), (lambda: GatHeadsAdjointField(spec, work, order)), tuple(plist)


class ODEfunc2(nn.Module):
    """Two stacked (H-head layer -> relu -> GroupNorm) with the time column re-attached before each layer: the GAT
    variant's ODEfunc2 (GAT/models.py:551-575) with `heads` reference layers side by side in both positions."""

    def __init__(self, dim, dropout, heads=8):
        super(ODEfunc2, self).__init__()
        self.norm1, self.norm2 = _gn(dim), _gn(dim)
        self.gc1 = FixedMultiHeadGraphConvolution(dim + 1, dim, heads)
        self.gc2 = FixedMultiHeadGraphConvolution(dim + 1, dim, heads)
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


# ---- model zoo ----------------------------------------------------------------------------------------------------
_zoos = {}
MIN_HEAD_WIDTH = 4


def zoo(heads=8):
    """The 23 model classes of the GAT variant with H-head layers wherever `heads` divides the layer's output width
    into heads of at least MIN_HEAD_WIDTH features (the hidden layers and the ODE block; a class-count output layer
    keeps one head, as in the usual GAT output layer).  Returns a namespace object with the classes as attributes."""
    hit = _zoos.get(heads)
    if hit is not None:
        return hit

    def layer(in_features, out_features, *args, **kw):
        if out_features % heads == 0 and out_features // heads >= MIN_HEAD_WIDTH:
            return MultiHeadGraphConvolution(in_features, out_features, heads, *args, **kw)
        return GraphConvolution(in_features, out_features, *args, **kw)

    def odefunc(dim):
        return ODEfunc(dim, heads)

    def odefunc2(dim, dropout):
        return ODEfunc2(dim, dropout, heads)

    kit = type("GatHeadsKit%d" % heads, (), {"GraphConvolution": staticmethod(layer), "ODEfunc": staticmethod(odefunc),
                                             "ODEfunc2": staticmethod(odefunc2), "input_dropout": False})

    def _forward(self, x, src, tgt, Mtgt):
        return self._body(self._input(x), (src, tgt, Mtgt))

    ns = {}
    _gcn_models.rebind_zoo(ns, __name__, kit, forward=_forward, what="%d-head GAT" % heads)
    out = _zoos[heads] = type("GatHeadsZoo%d" % heads, (), ns)
    return out
