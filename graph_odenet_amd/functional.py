"""Differentiable building blocks over the C ABI (torch.autograd.Function wrappers).

    gn_time_linear(x, t, W, gamma, beta, groups, eps)   S = [t | GroupNorm(x)] @ W     (GCN/models.py:175-177 + layers.py:70)
    graph_aggregate(graph, S, bias, relu)                relu?(A @ S + bias)             (GCN/layers.py:71-75, models.py:178)

    dense(x, W, bias)                                    x @ W (+ bias), tall and skinny (torch.mm call sites of the graph layers)
    affine / linear / Linear / mlp2                      x @ W + b of any shape on the tiled GEMM (QC/layers.py MyLinear, nn.Linear, TransitionMLP)

Used by modules that are not covered by a fused ODE field (ODEfunc2, stand-alone calls).
"""
import torch

from . import ops


class _GnTimeLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, W, gamma, beta, groups, eps, has_time):
        x = x.contiguous()
        n, d = x.shape
        S = ops.gn_time_gemm([(1.0, x)], n, d, groups, eps, gamma, beta, W, has_time, t)
        ctx.meta = (t, groups, eps, has_time)
        ctx.has_affine = gamma is not None
        ctx.save_for_backward(x, W, gamma, beta)
        return S

    @staticmethod
    def backward(ctx, dS):
        x, W, gamma, beta = ctx.saved_tensors
        t, groups, eps, has_time = ctx.meta
        n, d = x.shape
        dS = dS.contiguous()
        dx, dgp, dbp = ops.gn_time_gemm_bwd([(1.0, x)], n, d, groups, eps, gamma, W, has_time, dS,
                                            want_affine_grads=ctx.has_affine)
        part = ops.wgrad([(1.0, x)], n, d, groups, eps, gamma, beta, dS, has_time)
        gW = torch.empty_like(W)
        ops.reduce_parts_(gW.view(-1), part)
        if has_time:
            gW[0].mul_(t)
        gg = gb = None
        if ctx.has_affine and dgp is not None:
            gg = torch.empty_like(gamma)
            gb = torch.empty_like(beta)
            ops.reduce_parts2_(gg, dgp, gb, dbp)
        return dx, None, gW, gg, gb, None, None, None


def gn_time_linear(x, t, W, gamma=None, beta=None, groups=0, eps=1e-5, has_time=True):
    return _GnTimeLinearFn.apply(x, float(t), W, gamma, beta, int(groups), float(eps), bool(has_time))


class _DenseFn(torch.autograd.Function):
    """y = x @ W for any (K x M) weight on the fp32-MFMA kernels of csrc/rect.hip, with its autograd (dy W^T, x^T dy):
    the `torch.mm(input, self.weight)` / nn.Linear call sites of the reference's layers outside the fused paths
    (GCN/layers.py:32, GCN-mlp-sum/layers.py, GAT/layers.py:43-45 split per node, QC/layers.py:143)."""

    @staticmethod
    def forward(ctx, x, W):
        from .layers import sparse_features, _pad4
        x, W = x.contiguous(), W.contiguous()
        ctx.save_for_backward(x, W)
        # a mostly-zero input (Citeseer's 3327 x 3703 bag of words under the GAT input layer's three projections) runs
        # as CSR(X) @ W on the aggregation kernel from its second sighting (layers.sparse_features)
        ctx.xs = sparse_features(x)
        if ctx.xs is None:
            return ops.rect_gemm(x, W)
        m, mp = W.shape[1], _pad4(W.shape[1])
        y = ops.spmm(ctx.xs, W if mp == m else torch.nn.functional.pad(W, (0, mp - m)))
        return y if mp == m else y[:, :m].contiguous()

    @staticmethod
    def backward(ctx, dy):
        from .layers import _pad4
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        gx = ops.rect_gemm_nt(dy, W) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            if ctx.xs is None:
                gw = ops.rect_wgrad(x, dy)
            else:
                m, mp = W.shape[1], _pad4(W.shape[1])
                gw = ops.spmm(ctx.xs.transpose(), dy if mp == m else torch.nn.functional.pad(dy, (0, mp - m)))
                gw = gw if mp == m else gw[:, :m].contiguous()
        return gx, gw


def dense(x, W, bias=None):
    """x @ W (+ bias): W is (in, out).  For an nn.Linear pass `lin.weight.t()` (the transposed view is made contiguous
    once per call: weights are small next to the activations)."""
    y = _DenseFn.apply(x, W)
    return y if bias is None else y + bias


class _AffineFn(torch.autograd.Function):
    """y = x @ op(W) + b on csrc/mlp.hip's tiled fp32-MFMA GEMM (any shape): W is (in, out) with w_out_in False
    (QC/layers.py MyLinear: torch.mm(input, self.weight) + self.bias) or (out, in) as nn.Linear stores it."""

    @staticmethod
    def forward(ctx, x, W, b, w_out_in):
        x, W = x.contiguous(), W.contiguous()
        ctx.w_out_in = w_out_in
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return ops.gemm(x, W, trans_b=w_out_in, bias=b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.gemm(dy, W, trans_b=not ctx.w_out_in)
        if ctx.needs_input_grad[1]:
            gw = ops.gemm(dy, x, trans_a=True) if ctx.w_out_in else ops.gemm(x, dy, trans_a=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
            ops.colsum_(gb, dy)
        return gx, gw, gb, None


def affine(x, W, bias=None):
    """x @ W + bias, W stored (in, out)."""
    return _AffineFn.apply(x, W, bias, False)


def linear(x, weight, bias=None):
    """torch.nn.functional.linear on own kernels: weight stored (out, in)."""
    return _AffineFn.apply(x, weight, bias, True)


class Linear(torch.nn.Linear):
    """nn.Linear (same parameters and state_dict keys) whose 2-D GPU path runs on gode_gemm_f32."""

    def forward(self, x):
        if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32:
            return linear(x, self.weight, self.bias)
        return super().forward(x)


class _Mlp2Fn(torch.autograd.Function):
    """y = relu(x W1 + b1) W2 + b2 - the TransitionMLP of QC/layers.py:67-77, in particular the edge encoder
    (5 -> 2667 -> 5329 on the ~760 edge rows of a QM9 batch: QC/layers.py:79-86) - as two launches forward and five
    backward of gode_gemm_f32 (large second layers: gode_pgemm_bf16x3 from operands cut once per step) with bias / relu /
    relu-mask fused into the epilogues:
        H = relu(x W1 + b1);  y = H W2 + b2
        dW2 = H^T dy;  db2 = colsum(dy);  dH = (dy W2^T) * [H > 0];  dW1 = x^T dH;  db1 = colsum(dH);  dx = dH W1^T"""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        x, W1, W2 = x.contiguous(), W1.contiguous(), W2.contiguous()
        H = ops.gemm(x, W1, bias=b1, relu=True)
        # large second layer (the edge encoder: 760 x 2667 x 5329): the three big products of the step run on the bf16
        # matrix cores from exact cuts (csrc/pgemm.hip); H and W2 are cut once here and serve the backward pass too
        ctx.cuts = None
        if ops.pgemm_pays(H.shape[0], W2.shape[1], W2.shape[0]):
            Hc, W2c = ops.cut3(H), ops.cut3(W2)
            y = ops.pgemm(Hc, W2c, bias=b2)
            ctx.cuts = (Hc, W2c)
        else:
            y = ops.gemm(H, W2, bias=b2)
        ctx.save_for_backward(x, H, W1, W2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, H, W1, W2 = ctx.saved_tensors
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        f = dict(dtype=torch.float32, device=dy.device)
        big = ctx.cuts is not None
        dyc = ops.cut3(dy) if big and (need[3] or need[0] or need[1] or need[2]) else None
        gW2 = None
        if need[3]:
            gW2 = ops.pgemm(ctx.cuts[0], dyc, trans_a=True) if big else ops.gemm(H, dy, trans_a=True)
        gb2 = ops.colsum_(torch.empty(dy.shape[1], **f), dy) if (ctx.has_bias[1] and need[4]) else None
        gx = gW1 = gb1 = None
        if need[0] or need[1] or need[2]:
            dH = ops.pgemm(dyc, ctx.cuts[1], trans_b=True, mask=H) if big else ops.gemm(dy, W2, trans_b=True, mask=H)
            if need[1]:
                gW1 = ops.gemm(x, dH, trans_a=True)
            if ctx.has_bias[0] and need[2]:
                gb1 = ops.colsum_(torch.empty(dH.shape[1], **f), dH)
            if need[0]:
                gx = ops.gemm(dH, W1, trans_b=True)
        ctx.cuts = None
        return gx, gW1, gb1, gW2, gb2


def mlp2(x, W1, b1, W2, b2):
    return _Mlp2Fn.apply(x, W1, b1, W2, b2)


class _GraphAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, S, bias, relu):
        S = S.contiguous()
        out = ops.spmm(graph, S, bias=bias, relu=relu)
        ctx.graph, ctx.relu, ctx.has_bias = graph, relu, bias is not None
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = g.contiguous()
        dZ = g * (out > 0).to(g.dtype) if ctx.relu else g
        dS = ops.spmm(ctx.graph.transpose(), dZ)
        gb = None
        if ctx.has_bias:
            gb = torch.empty(out.shape[1], dtype=torch.float32, device=out.device)
            ops.colsum_(gb, dZ)
        return None, dS, gb, None


def graph_aggregate(graph, S, bias=None, relu=False):
    return _GraphAggregateFn.apply(graph, S, bias, bool(relu))


class _GroupNorm2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        x = x.contiguous()
        ctx.meta = (groups, eps)
        ctx.save_for_backward(x, gamma)
        return ops.group_norm_fwd(x, groups, eps, gamma, beta)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        groups, eps = ctx.meta
        dx, dgp, dbp = ops.group_norm_bwd(x, groups, eps, gamma, dy.contiguous(), want_affine_grads=gamma is not None)
        gg = gb = None
        if gamma is not None:
            gg = torch.empty_like(gamma)
            gb = torch.empty_like(gamma)
            ops.reduce_parts2_(gg, dgp, gb, dbp)
        return dx, gg, gb, None, None


class GroupNorm(torch.nn.GroupNorm):
    """nn.GroupNorm whose 2-D (nodes x channels) GPU path runs on libgraphode.

    Same parameters / state_dict keys as nn.GroupNorm.  Besides keeping the normalisation on the same
    arithmetic as the fused ODE kernels, this avoids torch-ROCm's own GroupNorm backward on 2-D inputs, whose
    dgamma / dbeta are wrong for more than a few hundred rows (torch 2.10.0+rocm7.0; tools/dev/gn_torch_check.py:
    CPU vs GPU differ by 100 % at 3327 x 128)."""

    def forward(self, x):
        if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32:
            return _GroupNorm2dFn.apply(x, self.weight, self.bias, self.num_groups, self.eps)
        return super().forward(x)


def group_norm(num_groups, num_channels):
    return GroupNorm(num_groups, num_channels)
