"""Fused GAT ODE function  f(t, x) = relu(EdgeAttention([t | GroupNorm(x)]))  (reference: GAT/models.py:172-179 ->
GAT/layers.py:95-122) and its vector-Jacobian product as plain kernel sequences over the C ABI - no autograd
graph, no per-edge E x 2i tensor:

  forward : gode_gn_time_gemm_f32 (GroupNorm + time column + node-level projections P = [t|xn] Wcat, 2o+2 columns)
            gode_edge_softmax_logits_f32 (a_e, global max), gode_edge_softmax_agg_f32_fwd (per-target normalised sum)
  adjoint : + gode_edge_softmax_agg_f32_bwd, the global-max path, 4 incidence SpMMs scattering dz / da into dP,
            gode_gn_time_gemm_bwd_f32 (dx, dgamma, dbeta), gode_wgrad_f32 (dWcat), column sums for the biases.

The weights of the two Linear layers (f: o x 2i, w: 1 x 2i with i = d+1) are packed once per solve into
Wcat (i x (2o+2)) = [Wf_src^T | Wf_tgt^T | ww_src^T | ww_tgt^T]; the adjoint integrates the gradient in that
layout and converts it back to the parameters' layout once, at the end of the solve.
"""
import torch

from . import ops
from .gat_layers import edge_graph
from .solver import Field


class GatOdeSpec:
    def __init__(self, eg, layer, norm):
        self.eg, self.layer, self.norm = eg, layer, norm
        self.d = layer.out_features
        self.i = layer.in_features                    # d + 1 (time column first)
        if self.i != self.d + 1:
            raise ValueError("GatOdeSpec: the ODE layer maps d+1 -> d features")
        self.groups, self.eps_gn = int(norm.num_groups), float(norm.eps)
        self.eps = float(layer.eps)
        self.Wcat = torch.empty(self.i, 2 * self.d + 2, dtype=torch.float32, device=layer.f.weight.device)
        self.refresh()
        self.bf, self.bw = layer.f.bias.detach(), layer.w.bias.detach()
        self.gamma, self.beta = norm.weight.detach(), norm.bias.detach()
        self.n = eg.n


    def refresh(self):
        """Re-pack the two Linear weights into Wcat in place (at the start of every solve: the parameters move
        between solves, the buffer - and a HIP graph captured over it - does not)."""
        Wf, ww = self.layer.f.weight.detach(), self.layer.w.weight.detach()
        i = self.i
        torch.cat([Wf[:, :i].t(), Wf[:, i:].t(), ww[:, :i].t(), ww[:, i:].t()], 1, out=self.Wcat)


class _Work:
    def __init__(self, spec, device):
        n, o = spec.n, spec.d
        self.P = torch.empty(n, 2 * o + 2, dtype=torch.float32, device=device)
        self.dP = torch.empty(n, 2 * o + 2, dtype=torch.float32, device=device)
        self.g = torch.empty(n, o, dtype=torch.float32, device=device)


class GatOdeField(Field):
    n_components = 1
    fused = True

    def __init__(self, spec, work):
        self.s, self.w = spec, work
        self.token = ("gat", id(spec.eg))

    def prepare(self):
        self.s.refresh()

    def _forward(self, t, y_terms, out):
        s, w = self.s, self.w
        ops.gn_time_gemm(y_terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wcat, True, t, out=w.P)
        a, amax = ops.edge_softmax_logits(w.P, s.d, s.bw, s.eg.src, s.eg.tgt)
        _, wgt, den = ops.edge_softmax_agg_fwd(s.eg.Mt, s.eg.src, s.eg.tgt, w.P, s.d, s.bf, a, amax, s.eps, out=out)
        return a, amax, wgt, den      # the outer relu of ODEfunc is the identity on a weighted mean of relu's

    def eval(self, t, terms, out):
        self._forward(t, terms[0], out[0])


class GatOdeAdjointField(GatOdeField):
    """Components: [y, a, a_t, Wcat, bf, bw, gamma, beta]."""

    def __init__(self, spec, work, order):
        super().__init__(spec, work)
        self.order = order
        self.n_components = 8
        self.ratio_groups = [[0], [1], [2], [3, 4, 5, 6, 7]]

    def new_state(self, y_end):
        s = self.s
        z = torch.zeros_like
        return [y_end.clone(), z(y_end), torch.zeros(1, dtype=torch.float32, device=y_end.device),
                z(s.Wcat), z(s.bf), z(s.bw), z(s.gamma), z(s.beta)]

    def param_grads(self, comps):
        s = self.s
        i, o = s.i, s.d
        gW = comps[3]
        gWf = torch.cat([gW[:, :o].t(), gW[:, o:2 * o].t()], 1).contiguous()                     # o x 2i
        gww = torch.cat([gW[:, 2 * o], gW[:, 2 * o + 1]]).view(1, 2 * i).contiguous()              # 1 x 2i
        m = {"gamma": comps[6], "beta": comps[7], "Wf": gWf, "bf": comps[4], "ww": gww, "bw": comps[5]}
        return [m[k] for k in self.order]

    def eval(self, t, terms, out):
        s, w = self.s, self.w
        eg, n, o = s.eg, s.n, s.d
        y_terms = terms[0]
        a, amax, wgt, den = self._forward(t, y_terms, out[0])
        # cotangent -a, masked by the outer relu (o > 0)
        ops.lincomb_(w.g, [(-c, x) for (c, x) in terms[1]])
        w.g.mul_(out[0] > 0)
        dz, da = ops.edge_softmax_agg_bwd(eg.Mt, eg.src, eg.tgt, w.P, o, s.bf, wgt, den, out[0], w.g)
        if eg.E > 0:                                    # path through the global max (GAT/layers.py:47)
            da.index_add_(0, torch.argmax(a).view(1), -da.sum().view(1))      # no host sync (graph-capturable)
        dP = w.dP
        ops.spmm(eg.Ms_inc, dz, out=dP[:, :o])
        ops.spmm(eg.Mt_inc, dz, out=dP[:, o:2 * o])
        da2 = da.view(-1, 1)
        ops.spmm(eg.Ms_inc, da2, out=dP[:, 2 * o:2 * o + 1])
        ops.spmm(eg.Mt_inc, da2, out=dP[:, 2 * o + 1:2 * o + 2])
        ops.colsum_(out[4], dz)
        ops.colsum_(out[5], da2)
        _, dgp, dbp = ops.gn_time_gemm_bwd(y_terms, n, o, s.groups, s.eps_gn, s.gamma, s.Wcat, True, dP, out=out[1])
        part = ops.wgrad(y_terms, n, o, s.groups, s.eps_gn, s.gamma, s.beta, dP, True)
        ops.reduce_parts_(out[3].view(-1), part)
        out[2].copy_((out[3][0] * s.Wcat[0]).sum().reshape(1))
        out[3][0].mul_(t)
        if dgp is not None:
            ops.reduce_parts_(out[6], dgp)
            ops.reduce_parts_(out[7], dbp)
        else:
            out[6].zero_(); out[7].zero_()


def gat_fields(odefunc, y0):
    """Hook body for gat_models.ODEfunc.gode_fields."""
    layer, norm = odefunc.gc1, odefunc.norm1
    import torch.nn.functional as F
    if layer.act is not F.relu or y0.dim() != 2 or not torch.is_tensor(layer.src) or layer.src.dim() != 1:
        return None
    plist = [p for p in odefunc.parameters() if p.requires_grad]
    names = {id(norm.weight): "gamma", id(norm.bias): "beta", id(layer.f.weight): "Wf", id(layer.f.bias): "bf",
             id(layer.w.weight): "ww", id(layer.w.bias): "bw"}
    if len(plist) != 6 or any(id(p) not in names for p in plist):
        return None
    eg = edge_graph(layer.src, layer.tgt, layer.Mtgt)
    if eg.n != y0.shape[0]:
        return None
    spec = GatOdeSpec(eg, layer, norm)
    work = _Work(spec, y0.device)
    order = [names[id(p)] for p in plist]
    return GatOdeField(spec, work), (lambda: GatOdeAdjointField(spec, work, order)), tuple(plist)
