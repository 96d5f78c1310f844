"""Thin, validating Python wrappers over the C ABI (include/graphode.h).

Every function here launches HIP kernels on torch's current stream and returns
immediately.  Inputs must be CUDA fp32 contiguous tensors: there is deliberately no CPU
or eager fallback (a missing library or a CPU tensor raises).
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lincomb, ptr, stream_ptr


def _need(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("graph_odenet_amd: %s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("graph_odenet_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("graph_odenet_amd: %s must be contiguous" % name)


def _need_terms(terms, name):
    n = None
    for c, t in terms:
        _need(t, name)
        if n is None:
            n = t.numel()
        elif t.numel() != n:
            raise ValueError("graph_odenet_amd: %s terms differ in size" % name)
    return n


def spmm(graph, X, bias=None, relu=False, out=None, cot_terms=None, out2=None, pre_terms=None, alpha=1.0, out2_colsum=None):
    """Y = (sum pre_terms) + alpha * relu?(A @ X + bias); optionally out2 = (sum cot_terms) * (A@X+bias > 0).
    out2_colsum ([spmm_y2_colsum_rows(graph, d), d], with cot_terms): the launch also leaves per-block column sums of out2
    there - their sum over the rows is out2.sum(0)."""
    lib = _lib.load()
    _need(X, "X")
    _need(bias, "bias")
    if getattr(graph, "is_partitioned", False):
        # one graph over several GPUs (partition.py): all-gather the operand rows, then the local row block
        if X.dim() != 2 or X.shape[0] != graph.n_cols:
            raise ValueError("spmm: X has shape %s, this rank owns %d rows" % (tuple(X.shape), graph.n_cols))
        X, graph = graph.gather(X), graph.local
    if X.dim() != 2 or X.shape[0] != graph.n_cols:
        raise ValueError("spmm: X has shape %s, graph is %d x %d" % (tuple(X.shape), graph.n_rows, graph.n_cols))
    d = X.shape[1]
    if bias is not None and bias.numel() != d:
        raise ValueError("spmm: bias has %d elements, expected %d" % (bias.numel(), d))
    ldy = d
    if out is None:
        out = torch.empty(graph.n_rows, d, dtype=torch.float32, device=X.device)
    else:
        if tuple(out.shape) != (graph.n_rows, d):
            raise ValueError("spmm: out has wrong shape")
        if out.is_contiguous():
            _need(out, "out")
        else:                                  # a column block of a wider row-major matrix (leading dimension ldy)
            if not out.is_cuda or out.dtype != torch.float32 or out.stride(1) != 1 or out.stride(0) < d:
                raise ValueError("spmm: out must be contiguous or a unit-stride column block of a row-major matrix")
            ldy = out.stride(0)
    ep = None
    if bias is not None or relu or cot_terms is not None or pre_terms is not None or alpha != 1.0:
        ep = _lib.SpmmEpilogue()
        ep.bias = bias.data_ptr() if bias is not None else None
        ep.relu = 1 if relu else 0
        ep.alpha = float(alpha)
        if pre_terms is not None:
            if _need_terms(pre_terms, "pre") != graph.n_rows * d:
                raise ValueError("spmm: pre terms have wrong size")
            ep.pre = lincomb(pre_terms)
        if cot_terms is not None:
            if _need_terms(cot_terms, "cot") != graph.n_rows * d:
                raise ValueError("spmm: cotangent terms have wrong size")
            ep.cot = lincomb(cot_terms)
            if out2 is None:
                out2 = torch.empty(graph.n_rows, d, dtype=torch.float32, device=X.device)
            _need(out2, "out2")
            ep.Y2 = out2.data_ptr()
            if out2_colsum is not None:
                _need(out2_colsum, "out2_colsum")
                if out2_colsum.numel() != spmm_y2_colsum_rows(graph, d) * d or out2_colsum.numel() == 0:
                    raise ValueError("spmm: out2_colsum must be spmm_y2_colsum_rows(graph, d) x d (and the shape must support it)")
                ep.Y2_colsum = out2_colsum.data_ptr()
    partial = graph.partial(d)
    rc = lib.gode_spmm_csr_f32(ptr(graph.rowptr), ptr(graph.col), ptr(graph.val),
                               ptr(graph.items), graph.n_items,
                               ptr(graph.long_rows), graph.n_long, ptr(partial),
                               ptr(X), d, ptr(out), ldy, graph.n_rows, d,
                               ctypes.byref(ep) if ep is not None else None, stream_ptr())
    check(rc, "gode_spmm_csr_f32")
    return (out, out2) if cot_terms is not None else out


def spmm_y2_colsum_rows(graph, d):
    """Rows of the out2_colsum array of spmm() for this graph and width (0: this shape runs on kernels without it)."""
    return int(_lib.load().gode_spmm_y2_colsum_rows(graph.n_items if graph.items is not None else graph.n_rows, graph.n_long, int(d)))


def lincomb_(out, terms):
    """out = sum_j coef_j * tensor_j (in place on `out`)."""
    lib = _lib.load()
    _need(out, "out")
    if _need_terms(terms, "lincomb") != out.numel():
        raise ValueError("lincomb: size mismatch")
    lc = lincomb(terms)
    check(lib.gode_lincomb_f32(ptr(out), ctypes.byref(lc), out.numel(), stream_ptr()), "gode_lincomb_f32")
    return out


def lincomb_multi_(outs, terms_list):
    """outs[c] = sum_j coef_cj * tensor_cj for up to four components of different lengths, in ONE launch."""
    lib = _lib.load()
    k = len(outs)
    if k != len(terms_list) or not 1 <= k <= 4:
        raise ValueError("lincomb_multi: one term list per output, at most four")
    lcs = (_lib.LinComb * k)()
    ptrs = (ctypes.c_void_p * k)()
    ns = (ctypes.c_int64 * k)()
    for c, (out, terms) in enumerate(zip(outs, terms_list)):
        _need(out, "out")
        if _need_terms(terms, "lincomb") != out.numel():
            raise ValueError("lincomb_multi: size mismatch in component %d" % c)
        lcs[c] = lincomb(terms)
        ptrs[c] = out.data_ptr()
        ns[c] = out.numel()
    check(lib.gode_lincomb_multi_f32(ptrs, lcs, ns, k, stream_ptr()), "gode_lincomb_multi_f32")
    return outs


_red_scratch = {}
_retired = []      # outgrown scratch buffers stay allocated: captured HIP graphs may still hold their addresses


def _scratch(device, nbytes):
    """Reduction scratch of the CURRENT stream of `device`: launches on one stream are ordered, so they can share a
    buffer; two streams (or two threads on their own streams) each get their own."""
    key = (device, _lib.current_stream_handle(device.index if device.index is not None else torch.cuda.current_device()))
    buf = _red_scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _red_scratch[key] = buf
    return buf


def rk_error_sumsq(y0, y1, err_terms, rtol, atol, out=None):
    """sum_i (err_i / (atol + rtol*max(|y0_i|,|y1_i|)))^2 as a 1-element fp64 device tensor."""
    lib = _lib.load()
    _need(y0, "y0"); _need(y1, "y1")
    n = y0.numel()
    if _need_terms(err_terms, "err") != n or y1.numel() != n:
        raise ValueError("rk_error_sumsq: size mismatch")
    if out is None:
        out = torch.empty(1, dtype=torch.float64, device=y0.device)
    sc = _scratch(y0.device, lib.gode_rk_errnorm_scratch_bytes())
    lc = lincomb(err_terms)
    check(lib.gode_rk_errnorm_f32(ptr(out), ptr(y0), ptr(y1), ctypes.byref(lc), float(rtol), float(atol), n,
                                  ptr(sc), stream_ptr()), "gode_rk_errnorm_f32")
    return out


def rk_scaled_sumsq(terms, y, rtol, atol, out=None):
    """sum_i ((sum terms)_i / (atol + rtol*|y_i|))^2 as a 1-element fp64 device tensor."""
    lib = _lib.load()
    _need(y, "y")
    n = y.numel()
    if _need_terms(terms, "terms") != n:
        raise ValueError("rk_scaled_sumsq: size mismatch")
    if out is None:
        out = torch.empty(1, dtype=torch.float64, device=y.device)
    sc = _scratch(y.device, lib.gode_rk_errnorm_scratch_bytes())
    lc = lincomb(terms)
    check(lib.gode_rk_scaled_sumsq_f32(ptr(out), ctypes.byref(lc), ptr(y), float(rtol), float(atol), n,
                                       ptr(sc), stream_ptr()), "gode_rk_scaled_sumsq_f32")
    return out


def gn_time_gemm(x_terms, n_rows, d_in, groups, eps, gamma, beta, W, has_time, t, out=None, x_out=None):
    """S = [t | GroupNorm(sum x_terms)] @ W   (W is (d_in+has_time) x d_out); x_out (optional) receives sum x_terms."""
    lib = _lib.load()
    _need(W, "W"); _need(gamma, "gamma"); _need(beta, "beta")
    if _need_terms(x_terms, "x") != n_rows * d_in:
        raise ValueError("gn_time_gemm: x terms have wrong size")
    if W.dim() != 2 or W.shape[0] != d_in + (1 if has_time else 0):
        raise ValueError("gn_time_gemm: W has shape %s" % (tuple(W.shape),))
    d_out = W.shape[1]
    if out is None:
        out = torch.empty(n_rows, d_out, dtype=torch.float32, device=W.device)
    _need(out, "out")
    lc = lincomb(x_terms)
    if x_out is not None:
        _need(x_out, "x_out")
        if x_out.numel() != n_rows * d_in:
            raise ValueError("gn_time_gemm: x_out has wrong size")
    check(lib.gode_gn_time_gemm_xout_f32(ctypes.byref(lc), n_rows, d_in, groups, float(eps), ptr(gamma), ptr(beta),
                                         ptr(W), d_out, 1 if has_time else 0, float(t), ptr(out), ptr(x_out),
                                         stream_ptr()), "gode_gn_time_gemm_xout_f32")
    return out


def gn_time_gemm_pair(x_terms, n_rows, d, groups, eps, gamma, beta, Wa, Wb, has_time, t, out_a, out_b, x_out=None):
    """out_a = [t | GN(x)] @ Wa, out_b = [t | GN(x)] @ Wb for two square weight matrices, in one launch."""
    lib = _lib.load()
    for tns, nm in ((Wa, "Wa"), (Wb, "Wb"), (gamma, "gamma"), (beta, "beta"), (out_a, "out_a"), (out_b, "out_b"), (x_out, "x_out")):
        _need(tns, nm)
    if _need_terms(x_terms, "x") != n_rows * d:
        raise ValueError("gn_time_gemm_pair: x terms have wrong size")
    k = d + (1 if has_time else 0)
    if tuple(Wa.shape) != (k, d) or tuple(Wb.shape) != (k, d) or out_a.numel() != n_rows * d or out_b.numel() != n_rows * d:
        raise ValueError("gn_time_gemm_pair: weights must be (%d, %d) and outputs n x d" % (k, d))
    lc = lincomb(x_terms)
    check(lib.gode_gn_time_gemm_pair_f32(ctypes.byref(lc), n_rows, d, groups, float(eps), ptr(gamma), ptr(beta), ptr(Wa), ptr(Wb),
                                         1 if has_time else 0, float(t), ptr(out_a), ptr(out_b), ptr(x_out), stream_ptr()),
          "gode_gn_time_gemm_pair_f32")


def gat_small_supported(n_rows, d, groups, heads=1):
    """The one-launch dense kernels of csrc/gat_small.hip cover this shape (and the option small_fused is on)."""
    lib = _lib.load()
    return bool(lib.gode_get_option(b"small_fused")) and bool(lib.gode_gat_small_supported(n_rows, d, groups, heads))


def gat_small_part(n_rows, d, heads, device):
    lib = _lib.load()
    return torch.empty(lib.gode_gat_small_parts(n_rows, d), lib.gode_gat_small_part_len(d, heads), dtype=torch.float32, device=device)


def gat_small_pack(Wsrc, Wtgt, Wlog, heads, out=None):
    """The LDS images of the weights for gat_project_small / gat_dense_vjp_small (once per solve; csrc/gat_small.hip)."""
    lib = _lib.load()
    _need(Wsrc, "Wsrc"); _need(Wtgt, "Wtgt"); _need(Wlog, "Wlog"); _need(out, "packed")
    d = Wsrc.shape[1]
    n = lib.gode_gat_small_pack_len(d, heads)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=Wsrc.device)
    elif out.numel() != n:
        raise ValueError("gat_small_pack: packed buffer must hold %d floats" % n)
    check(lib.gode_gat_small_pack_f32(ptr(Wsrc), ptr(Wtgt), ptr(Wlog), d, heads, ptr(out), stream_ptr()), "gode_gat_small_pack_f32")
    return out


def gat_project_small(x_terms, n_rows, d, groups, eps, gamma, beta, Wsrc, Wtgt, Wlog, heads, pt_bias, t, Ps, Pt, A2, x_out=None,
                      packed=None):
    """Ps, Pt (+ pt_bias), A2 = [t | GN(x)] @ [Wsrc | Wtgt | Wlog] in one launch (x = sum of x_terms, written to x_out)."""
    lib = _lib.load()
    for tns, nm in ((gamma, "gamma"), (beta, "beta"), (Wsrc, "Wsrc"), (Wtgt, "Wtgt"), (Wlog, "Wlog"), (pt_bias, "pt_bias"),
                    (Ps, "Ps"), (Pt, "Pt"), (A2, "A2"), (x_out, "x_out")):
        _need(tns, nm)
    if _need_terms(x_terms, "x") != n_rows * d:
        raise ValueError("gat_project_small: x terms have wrong size")
    if tuple(Wsrc.shape) != (d + 1, d) or tuple(Wtgt.shape) != (d + 1, d) or tuple(Wlog.shape) != (d + 1, 2 * heads):
        raise ValueError("gat_project_small: weights must be (d+1) x d, (d+1) x d, (d+1) x 2H")
    if Ps.numel() != n_rows * d or Pt.numel() != n_rows * d or A2.numel() != n_rows * 2 * heads or \
            (x_out is not None and x_out.numel() != n_rows * d) or (pt_bias is not None and pt_bias.numel() != d):
        raise ValueError("gat_project_small: output buffers have wrong size")
    lc = lincomb(x_terms)
    check(lib.gode_gat_project_small_f32(ctypes.byref(lc), n_rows, d, groups, float(eps), ptr(gamma), ptr(beta), ptr(Wsrc), ptr(Wtgt),
                                         ptr(Wlog), heads, ptr(pt_bias), float(t), ptr(Ps), ptr(Pt), ptr(A2), ptr(x_out),
                                         ptr(packed), stream_ptr()), "gode_gat_project_small_f32")


def gat_dense_vjp_small(x_terms, n_rows, d, groups, eps, gamma, beta, Wsrc, Wtgt, Wlog, heads, dPs, dPt, dA2, ka, part,
                        out_scale=1.0, pre_terms=None, maxfix=None, packed=None):
    """k_a and the block partials of every parameter gradient of a GAT adjoint stage in one launch (csrc/gat_small.hip).
    maxfix = (scratch, esrc, etgt): close the per-head max-path step that gat_vjp(..., defer_maxpath=True) left open."""
    lib = _lib.load()
    for tns, nm in ((gamma, "gamma"), (beta, "beta"), (Wsrc, "Wsrc"), (Wtgt, "Wtgt"), (Wlog, "Wlog"), (dPs, "dPs"), (dPt, "dPt"),
                    (dA2, "dA2"), (ka, "ka"), (part, "part")):
        _need(tns, nm)
    if _need_terms(x_terms, "x") != n_rows * d:
        raise ValueError("gat_dense_vjp_small: x terms have wrong size")
    if tuple(Wsrc.shape) != (d + 1, d) or tuple(Wtgt.shape) != (d + 1, d) or tuple(Wlog.shape) != (d + 1, 2 * heads):
        raise ValueError("gat_dense_vjp_small: weights must be (d+1) x d, (d+1) x d, (d+1) x 2H")
    if dPs.numel() != n_rows * d or dPt.numel() != n_rows * d or dA2.numel() != n_rows * 2 * heads or ka.numel() != n_rows * d or \
            part.numel() != lib.gode_gat_small_parts(n_rows, d) * lib.gode_gat_small_part_len(d, heads):
        raise ValueError("gat_dense_vjp_small: buffers have wrong size")
    lc = lincomb(x_terms)
    pre = None
    if pre_terms:
        if _need_terms(pre_terms, "pre") != n_rows * d:
            raise ValueError("gat_dense_vjp_small: pre terms have wrong size")
        pre = lincomb(pre_terms)
    check(lib.gode_gat_dense_vjp_small_f32(ctypes.byref(lc), n_rows, d, groups, float(eps), ptr(gamma), ptr(beta), ptr(Wsrc), ptr(Wtgt),
                                           ptr(Wlog), heads, ptr(dPs), ptr(dPt), ptr(dA2), float(out_scale),
                                           ctypes.byref(pre) if pre is not None else None, ptr(ka), ptr(part),
                                           ptr(maxfix[0]) if maxfix else None, ptr(maxfix[1]) if maxfix else None,
                                           ptr(maxfix[2]) if maxfix else None, maxfix[1].numel() if maxfix else 0, ptr(packed),
                                           stream_ptr()),
          "gode_gat_dense_vjp_small_f32")


def gat_small_finish(part, n_rows, d, heads, t, ktheta, kat):
    lib = _lib.load()
    _need(part, "part"); _need(ktheta, "ktheta"); _need(kat, "kat")
    if part.numel() != lib.gode_gat_small_parts(n_rows, d) * lib.gode_gat_small_part_len(d, heads) or \
            ktheta.numel() != lib.gode_gat_ode_theta_len_heads(d, heads) or kat.numel() != 1:
        raise ValueError("gat_small_finish: buffers have wrong size")
    check(lib.gode_gat_small_finish_f32(ptr(part), n_rows, d, heads, float(t), ptr(ktheta), ptr(kat), stream_ptr()),
          "gode_gat_small_finish_f32")


def gat_small_finish_step(parts, n_rows, d, heads, ts, ws, theta, at):
    """theta += sum_s ws[s] * k_theta(stage s), at likewise, from the partial buffers of the len(ts) <= 4 stages of one
    fixed-grid RK step laid out one after the other in `parts` (csrc/gat_small.hip: one closing launch per STEP)."""
    lib = _lib.load()
    _need(parts, "parts"); _need(theta, "theta"); _need(at, "a_t")
    k = len(ts)
    if parts.numel() < k * lib.gode_gat_small_parts(n_rows, d) * lib.gode_gat_small_part_len(d, heads) or \
            theta.numel() != lib.gode_gat_ode_theta_len_heads(d, heads) or at.numel() != 1:
        raise ValueError("gat_small_finish_step: buffers have wrong size")
    tsa, wsa = (ctypes.c_float * 4)(*([float(v) for v in ts] + [0.0] * (4 - k))), (ctypes.c_float * 4)(*([float(v) for v in ws] + [0.0] * (4 - k)))
    check(lib.gode_gat_small_finish_step_f32(ptr(parts), n_rows, d, heads, k, tsa, wsa, ptr(theta), ptr(at), stream_ptr()),
          "gode_gat_small_finish_step_f32")


def gn_time_gemm_bwd(x_terms, n_rows, d_in, groups, eps, gamma, W, has_time, dS, out_scale=1.0, out=None,
                     want_affine_grads=True, pre_terms=None, parts=None):
    """Returns (dx, dgamma_part, dbeta_part); parts are [n_part, d_in] block partials (or None).
    With pre_terms the first output is (sum pre_terms) + out_scale * dx."""
    lib = _lib.load()
    _need(W, "W"); _need(gamma, "gamma"); _need(dS, "dS")
    if _need_terms(x_terms, "x") != n_rows * d_in:
        raise ValueError("gn_time_gemm_bwd: x terms have wrong size")
    d_out = W.shape[1]
    if out is None:
        out = torch.empty(n_rows, d_in, dtype=torch.float32, device=W.device)
    _need(out, "out")
    dg = db = None
    if want_affine_grads and groups > 0:
        n_part = lib.gode_gemm_bwd_parts(n_rows)
        if parts is not None:
            dg, db = parts
            _need(dg, "dgamma parts"); _need(db, "dbeta parts")
            if dg.numel() != n_part * d_in or db.numel() != n_part * d_in:
                raise ValueError("gn_time_gemm_bwd: partial buffers have wrong size")
        else:
            dg = torch.empty(n_part, d_in, dtype=torch.float32, device=W.device)
            db = torch.empty(n_part, d_in, dtype=torch.float32, device=W.device)
    lc = lincomb(x_terms)
    pre = None
    if pre_terms is not None:
        if _need_terms(pre_terms, "pre") != n_rows * d_in:
            raise ValueError("gn_time_gemm_bwd: pre terms have wrong size")
        pre = lincomb(pre_terms)
    check(lib.gode_gn_time_gemm_bwd_f32(ctypes.byref(lc), n_rows, d_in, groups, float(eps), ptr(gamma), ptr(W),
                                        d_out, 1 if has_time else 0, ptr(dS), float(out_scale),
                                        ctypes.byref(pre) if pre is not None else None, ptr(out),
                                        ptr(dg), ptr(db), stream_ptr()), "gode_gn_time_gemm_bwd_f32")
    return out, dg, db


def bwd_wgrad_supported(n_rows, d, groups):
    return bool(_lib.load().gode_bwd_wgrad_supported(n_rows, d, d, groups))


def gn_time_gemm_bwd_wgrad(x_terms, n_rows, d, groups, eps, gamma, beta, W, has_time, dS, out_scale=1.0, out=None,
                           pre_terms=None, parts=None, wpart=None):
    """The VJP w.r.t. x and the weight gradient in one pass (csrc/gemm_pc.hip): returns (dx, dgamma_part, dbeta_part,
    dW_part); dW_part is [gode_bwd_wgrad_parts(n_rows), (d + has_time) * d].  Raises where bwd_wgrad_supported is false."""
    lib = _lib.load()
    _need(W, "W"); _need(gamma, "gamma"); _need(beta, "beta"); _need(dS, "dS")
    if _need_terms(x_terms, "x") != n_rows * d:
        raise ValueError("gn_time_gemm_bwd_wgrad: x terms have wrong size")
    if out is None:
        out = torch.empty(n_rows, d, dtype=torch.float32, device=W.device)
    _need(out, "out")
    dg = db = None
    if groups > 0:
        n_part = lib.gode_gemm_bwd_parts(n_rows)
        if parts is not None:
            dg, db = parts
        else:
            dg = torch.empty(n_part, d, dtype=torch.float32, device=W.device)
            db = torch.empty(n_part, d, dtype=torch.float32, device=W.device)
    K = d + (1 if has_time else 0)
    npw = lib.gode_bwd_wgrad_parts(n_rows)
    if wpart is None:
        wpart = torch.empty(npw, K * d, dtype=torch.float32, device=W.device)
    elif wpart.numel() < npw * K * d:
        raise ValueError("gn_time_gemm_bwd_wgrad: weight-gradient partial buffer too small")
    lc = lincomb(x_terms)
    pre = lincomb(pre_terms) if pre_terms is not None else None
    check(lib.gode_gn_time_gemm_bwd_wgrad_f32(ctypes.byref(lc), n_rows, d, groups, float(eps), ptr(gamma), ptr(beta), ptr(W), d,
                                              1 if has_time else 0, ptr(dS), float(out_scale),
                                              ctypes.byref(pre) if pre is not None else None, ptr(out), ptr(dg), ptr(db),
                                              ptr(wpart), stream_ptr()), "gode_gn_time_gemm_bwd_wgrad_f32")
    return out, dg, db, wpart.view(-1)[:npw * K * d].view(npw, K * d)


def wgrad(x_terms, n_rows, d_in, groups, eps, gamma, beta, dS, has_time, part=None):
    """Block partials [n_part, (d_in+has_time)*d_out] of dW = [1 | GN(x)]^T dS (row 0 = colsum(dS))."""
    lib = _lib.load()
    _need(dS, "dS"); _need(gamma, "gamma"); _need(beta, "beta")
    if _need_terms(x_terms, "x") != n_rows * d_in:
        raise ValueError("wgrad: x terms have wrong size")
    d_out = dS.shape[1]
    n_part = lib.gode_wgrad_parts(n_rows)
    K = d_in + (1 if has_time else 0)
    if part is None:
        part = torch.empty(n_part, K * d_out, dtype=torch.float32, device=dS.device)
    elif part.numel() != n_part * K * d_out:
        raise ValueError("wgrad: part buffer has wrong size")
    _need(part, "part")
    lc = lincomb(x_terms)
    check(lib.gode_wgrad_f32(ctypes.byref(lc), n_rows, d_in, groups, float(eps), ptr(gamma), ptr(beta), ptr(dS),
                             d_out, 1 if has_time else 0, ptr(part), stream_ptr()), "gode_wgrad_f32")
    return part


def _need_rows(t, name):
    """A row-major matrix view: unit column stride, any leading dimension (a column block of a padded matrix)."""
    if not t.is_cuda:
        raise RuntimeError("graph_odenet_amd: %s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or t.stride(0) < t.shape[1]:
        raise ValueError("graph_odenet_amd: %s must be a float32 row-major matrix (unit column stride)" % name)
    return t.stride(0)


def rect_gemm(x, W, pad_to=None):
    """S = x @ W for a rectangular W (K x M) on the fp32 MFMA kernel of csrc/rect.hip (GCN/layers.py:32).  pad_to:
    leading dimension of the result (>= M): returns an n x pad_to tensor whose columns M.. are zero."""
    lib = _lib.load()
    _need(W, "W")
    ldx = _need_rows(x, "x")
    n, K = x.shape
    M = W.shape[1]
    if W.shape[0] != K:
        raise ValueError("rect_gemm: x is %s, W is %s" % (tuple(x.shape), tuple(W.shape)))
    lds = M if pad_to is None else int(pad_to)
    out = torch.empty(n, lds, dtype=torch.float32, device=x.device)
    check(lib.gode_rect_gemm_f32(ptr(x), ldx, n, K, ptr(W), M, ptr(out), lds, stream_ptr()), "gode_rect_gemm_f32")
    return out


def rect_gemm_nt(dS, W):
    """dX = dS[:, :M] @ W^T for W (K x M); dS may be a column block of a wider (padded) matrix."""
    lib = _lib.load()
    _need(W, "W")
    ldds = _need_rows(dS, "dS")
    n = dS.shape[0]
    K, M = W.shape
    if dS.shape[1] != M:
        raise ValueError("rect_gemm_nt: dS is %s, W is %s" % (tuple(dS.shape), tuple(W.shape)))
    out = torch.empty(n, K, dtype=torch.float32, device=dS.device)
    check(lib.gode_rect_gemm_nt_f32(ptr(dS), ldds, n, M, ptr(W), K, ptr(out), K, stream_ptr()), "gode_rect_gemm_nt_f32")
    return out


def rect_wgrad(x, dS):
    """dW = x^T @ dS (K x M): block partials by csrc/rect.hip, summed by reduce_parts_ (fixed order)."""
    lib = _lib.load()
    ldx, ldds = _need_rows(x, "x"), _need_rows(dS, "dS")
    n, K = x.shape
    M = dS.shape[1]
    if dS.shape[0] != n:
        raise ValueError("rect_wgrad: x is %s, dS is %s" % (tuple(x.shape), tuple(dS.shape)))
    out = torch.empty(K, M, dtype=torch.float32, device=x.device)
    if n == 0:
        return out.zero_()
    n_part = lib.gode_rect_wgrad_parts(n)
    for m0 in range(0, M, 128):                       # the kernel takes up to 128 output columns per call
        mw = min(128, M - m0)
        part = torch.empty(n_part, K * mw, dtype=torch.float32, device=x.device)
        if mw == M:                                   # one column block: partials and their sum in ONE call
            check(lib.gode_rect_wgrad_sum_f32(ptr(x), ldx, n, K, ptr(dS), ldds, mw, ptr(part), ptr(out), stream_ptr()),
                  "gode_rect_wgrad_sum_f32")
            return out
        check(lib.gode_rect_wgrad_f32(ptr(x), ldx, n, K, dS.data_ptr() + 4 * m0, ldds, mw, ptr(part), stream_ptr()),
              "gode_rect_wgrad_f32")
        if mw == M:
            reduce_parts_(out.view(-1), part)
        else:
            blk = torch.empty(K * mw, dtype=torch.float32, device=x.device)
            reduce_parts_(blk, part)
            out[:, m0:m0 + mw] = blk.view(K, mw)
    return out


def gemm(A, B, trans_a=False, trans_b=False, bias=None, relu=False, mask=None, out=None):
    """C = op(A) @ op(B) (+ bias) (relu) (* [mask > 0]) on csrc/mlp.hip's fp32-MFMA GEMM (row-major matrices with unit
    column stride; any leading dimension); large products (pgemm_pays) are cut and go to csrc/pgemm.hip."""
    lib = _lib.load()
    lda, ldb = _need_rows(A, "A"), _need_rows(B, "B")
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if K != K2:
        raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, K2))
    _need(bias, "bias")
    if bias is not None and bias.numel() != N:
        raise ValueError("gemm: bias must have %d elements" % N)
    ldm = 0
    if mask is not None:
        ldm = _need_rows(mask, "mask")
        if tuple(mask.shape) != (M, N):
            raise ValueError("gemm: mask must be %d x %d" % (M, N))
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    ldc = _need_rows(out, "out")
    if tuple(out.shape) != (M, N):
        raise ValueError("gemm: out must be %d x %d" % (M, N))
    if K == 0:
        out.zero_()
        return out if bias is None else out.add_(bias)
    if pgemm_pays(M, N, K):
        return pgemm(cut3(A), cut3(B), trans_a, trans_b, bias, relu, mask, out)
    if bias is None and not relu and mask is None and out.is_contiguous():
        parts = lib.gode_gemm_splitk_parts(M, N, K)
        if parts > 1:                                  # tall contraction, few output tiles (weight gradients of small layers)
            part = torch.empty(parts, M * N, dtype=torch.float32, device=A.device)
            check(lib.gode_gemm_splitk_f32(1 if trans_a else 0, 1 if trans_b else 0, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(part),
                                           ptr(out), stream_ptr()), "gode_gemm_splitk_f32")
            return out
    check(lib.gode_gemm_f32(1 if trans_a else 0, 1 if trans_b else 0, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc,
                            ptr(bias), 1 if relu else 0, ptr(mask), ldm, stream_ptr()), "gode_gemm_f32")
    return out


class Cut3:
    """The three bf16 piece planes of an fp32 matrix (gode_cut_bf16x3_f32): x = hi + mid + lo exactly; rows and columns
    zero-padded to multiples of 128.  Cut once, used by every product the matrix enters, in either operand role."""

    __slots__ = ("planes", "rows", "cols")

    def __init__(self, planes, rows, cols):
        self.planes, self.rows, self.cols = planes, rows, cols


def cut3(X):
    lib = _lib.load()
    ld = _need_rows(X, "X")
    R, C = X.shape
    Rp, Cp = lib.gode_cut_pad(R), lib.gode_cut_pad(C)
    planes = torch.empty(3, Rp, Cp, dtype=torch.bfloat16, device=X.device)
    check(lib.gode_cut_bf16x3_f32(ptr(X), ld, R, C, ptr(planes), stream_ptr()), "gode_cut_bf16x3_f32")
    return Cut3(planes, R, C)


PGEMM_MIN_FLOP = 2.0e9          # products below this stay on the exact-fp32 MFMA kernel (the cuts are two extra passes)


def pgemm_pays(M, N, K):
    """Whether C[M x N] = A B with inner dimension K goes to the bf16-piece kernel: enough work to pay for cutting both
    operands, and no dimension so small that the 128 x 128 x 32 tiles are mostly padding."""
    return 2.0 * M * N * K >= PGEMM_MIN_FLOP and min(M, N) >= 256 and K >= 256


def pgemm(A, B, trans_a=False, trans_b=False, bias=None, relu=False, mask=None, out=None, products=8):
    """C = op(A) @ op(B) from Cut3 operands on the bf16 matrix cores (csrc/pgemm.hip); arguments as gemm()."""
    lib = _lib.load()
    M, K = (A.cols, A.rows) if trans_a else (A.rows, A.cols)
    K2, N = (B.cols, B.rows) if trans_b else (B.rows, B.cols)
    if K != K2:
        raise ValueError("pgemm: inner dimensions differ (%d vs %d)" % (K, K2))
    _need(bias, "bias")
    if bias is not None and bias.numel() != N:
        raise ValueError("pgemm: bias must have %d elements" % N)
    ldm = 0
    if mask is not None:
        ldm = _need_rows(mask, "mask")
        if tuple(mask.shape) != (M, N):
            raise ValueError("pgemm: mask must be %d x %d" % (M, N))
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.planes.device)
    ldc = _need_rows(out, "out")
    if tuple(out.shape) != (M, N):
        raise ValueError("pgemm: out must be %d x %d" % (M, N))
    wb = lib.gode_pgemm_workspace_bytes(M, N, K)
    ws = torch.empty(wb // 4, dtype=torch.float32, device=out.device) if wb else None
    check(lib.gode_pgemm_bf16x3(1 if trans_a else 0, 1 if trans_b else 0, M, N, K, ptr(A.planes), ptr(B.planes), ptr(out), ldc,
                                ptr(bias), 1 if relu else 0, ptr(mask), ldm, int(products), ptr(ws), stream_ptr()),
          "gode_pgemm_bf16x3")
    return out


def reduce_parts_(out, part, scale=1.0, accumulate=False):
    """out (+)= scale * part.sum(0)   with `part` of shape [n_part, out.numel()]."""
    lib = _lib.load()
    _need(out, "out"); _need(part, "part")
    n_part = part.shape[0]
    if part.numel() != n_part * out.numel():
        raise ValueError("reduce_parts: size mismatch")
    check(lib.gode_reduce_parts_f32(ptr(out), ptr(part), n_part, out.numel(), float(scale), 1 if accumulate else 0,
                                    stream_ptr()), "gode_reduce_parts_f32")
    return out


def reduce_parts2_(out_a, part_a, out_b, part_b):
    """Two reductions of equal shape in one launch: out_a = part_a.sum(0), out_b = part_b.sum(0)."""
    lib = _lib.load()
    for t, nm in ((out_a, "out_a"), (part_a, "part_a"), (out_b, "out_b"), (part_b, "part_b")):
        _need(t, nm)
    n_part = part_a.shape[0]
    if part_b.shape != part_a.shape or out_b.numel() != out_a.numel() or part_a.numel() != n_part * out_a.numel():
        raise ValueError("reduce_parts2: size mismatch")
    check(lib.gode_reduce_parts2_f32(ptr(out_a), ptr(part_a), ptr(out_b), ptr(part_b), n_part, out_a.numel(), 1.0, 0,
                                     stream_ptr()), "gode_reduce_parts2_f32")


def colsum_parts(X, scratch):
    """First half of colsum_: block partials of X.sum(0) into `scratch` (uint8, gode_colsum_scratch_bytes); returns the
    number of partial rows (each X.shape[1] floats), to be closed by a segment of reduce_segments_."""
    lib = _lib.load()
    _need(X, "X")
    n, d = X.shape
    if scratch.numel() < lib.gode_colsum_scratch_bytes(n, d):
        raise ValueError("colsum_parts: scratch too small")
    npart = ctypes.c_int64(0)
    check(lib.gode_colsum_parts_f32(ptr(X), n, d, ptr(scratch), ctypes.byref(npart), stream_ptr()), "gode_colsum_parts_f32")
    return int(npart.value)


def reduce_segments_(segs, t=0.0, at=None):
    """Several block-partial reductions in one launch.  segs: tuples (out, part_tensor_or_ptr, n_part, ld, col0, col_stride,
    length, w_row0 or None, time_len); see gode_reduce_segments_f32."""
    lib = _lib.load()
    arr = (_lib.ReduceSeg * len(segs))()
    for k, (out, part, n_part, ld, col0, cs, length, w0, tl) in enumerate(segs):
        _need(out, "out"); _need(w0, "w_row0")
        if out.numel() < length:
            raise ValueError("reduce_segments: output %d too short" % k)
        q = arr[k]
        q.out, q.part = out.data_ptr(), (part.data_ptr() if torch.is_tensor(part) else int(part))
        q.n_part, q.ld, q.col0, q.col_stride, q.len = int(n_part), int(ld), int(col0), int(cs), int(length)
        q.w_row0, q.time_len = (w0.data_ptr() if w0 is not None else None), int(tl)
    _need(at, "at")
    check(lib.gode_reduce_segments_f32(arr, len(segs), float(t), ptr(at), stream_ptr()), "gode_reduce_segments_f32")


def colsum_(out, X, scale=1.0, accumulate=False):
    """out (+)= scale * X.sum(0)."""
    lib = _lib.load()
    _need(out, "out"); _need(X, "X")
    n, d = X.shape
    if out.numel() != d:
        raise ValueError("colsum: size mismatch")
    sc = _scratch(X.device, lib.gode_colsum_scratch_bytes(n, d))
    check(lib.gode_colsum_f32(ptr(out), ptr(X), n, d, float(scale), 1 if accumulate else 0, ptr(sc), stream_ptr()),
          "gode_colsum_f32")
    return out


# ---- GAT-style edge attention ---------------------------------------------------------------
# (projections stored by role + fused VJP pieces; gat_layers.py, gat_ode.py)
def gat_proj(Ps, Pt, A2):
    """gode_gat_proj_t over Ps, Pt (n x o each) and A2 (n x 2: logit parts by source / by target)."""
    _need(Ps, "Ps"); _need(Pt, "Pt"); _need(A2, "A2")
    if Ps.shape != Pt.shape or A2.shape != (Ps.shape[0], 2):
        raise ValueError("gat_proj: shapes %s %s %s" % (tuple(Ps.shape), tuple(Pt.shape), tuple(A2.shape)))
    p = _lib.GatProj()
    p.ps, p.ld_s, p.pt, p.ld_t = Ps.data_ptr(), Ps.shape[1], Pt.data_ptr(), Pt.shape[1]
    p.as_, p.at, p.ld_a = A2.data_ptr(), A2.data_ptr() + 4, 2
    return p


def gat_logits(proj, bw, src, tgt, a, amax):
    lib = _lib.load()
    _need(bw, "bw"); _need(a, "a"); _need(amax, "amax")
    E = src.numel()
    sc = _scratch(a.device, lib.gode_gat_logits_scratch_bytes(E))
    check(lib.gode_gat_logits_f32(ctypes.byref(proj), ptr(bw), ptr(src), ptr(tgt), E, ptr(a), ptr(amax), ptr(sc),
                                  stream_ptr()), "gode_gat_logits_f32")


def _edge_csr(eg, width):
    """gode_graph_t of the CSR-by-target of an EdgeGraph (col = NULL when the edges are target-sorted)."""
    mt = eg.Mt
    gs = _lib.Graph()
    gs.rowptr = mt.rowptr.data_ptr()
    gs.col = None if eg.canonical else mt.col.data_ptr()
    gs.val = mt.val.data_ptr() if mt.val is not None else None
    gs.items, gs.n_items = (mt.items.data_ptr() if mt.items is not None else None), mt.n_items
    gs.long_rows = mt.long_rows.data_ptr() if mt.long_rows is not None else None
    gs.n_long = mt.n_long
    part = mt.partial(width)
    gs.partial = part.data_ptr() if part is not None else None
    gs.n_rows, gs.nnz = mt.n_rows, mt.nnz
    return gs


def gat_agg_fwd(eg, proj, o, bf, a, amax, eps, out, w, den):
    lib = _lib.load()
    _need(out, "out"); _need(w, "w"); _need(den, "den"); _need(bf, "bf")
    gs = _edge_csr(eg, o + 4)
    check(lib.gode_gat_agg_f32_fwd(ctypes.byref(gs), ptr(eg.src), ptr(eg.tgt), ctypes.byref(proj), o, ptr(bf), ptr(a),
                                   ptr(amax), float(eps), ptr(out), ptr(w), ptr(den), stream_ptr()), "gode_gat_agg_f32_fwd")


def gat_logits_heads(proj, src, tgt, heads, a, hmax=None, bw=None):
    """Logits of the H-fold graph (gat_heads.py) + bw[head], shifted by the maximum of their head (tgt % heads)."""
    lib = _lib.load()
    _need(a, "a"); _need(hmax, "hmax"); _need(bw, "bw")
    if bw is not None and bw.numel() != heads:
        raise ValueError("gat_logits_heads: bw must hold one bias per head")
    E = src.numel()
    sc = _scratch(a.device, lib.gode_gat_heads_scratch_bytes(E, heads))
    check(lib.gode_gat_logits_heads_f32(ctypes.byref(proj), ptr(bw), ptr(src), ptr(tgt), E, int(heads), ptr(a), ptr(hmax), ptr(sc),
                                        stream_ptr()), "gode_gat_logits_heads_f32")


def gat_logits_heads_raw(proj, src, tgt, heads, a, scratch, bw=None):
    """Raw logits of the H-fold graph (+ bw[head]) and per-block partial maxima in `scratch` (caller-owned, at least
    gode_gat_heads_scratch_bytes; it must stay untouched until the stage's gat_vjp(..., raw_scratch=scratch))."""
    lib = _lib.load()
    _need(a, "a"); _need(bw, "bw"); _need(scratch, "scratch", torch.uint8)
    E = src.numel()
    if scratch.numel() * scratch.element_size() < lib.gode_gat_heads_scratch_bytes(E, heads):
        raise ValueError("gat_logits_heads_raw: scratch too small")
    check(lib.gode_gat_logits_heads_raw_f32(ctypes.byref(proj), ptr(bw), ptr(src), ptr(tgt), E, int(heads), ptr(a), ptr(scratch),
                                            stream_ptr()), "gode_gat_logits_heads_raw_f32")


def gat_agg_heads_fwd(eg, proj, o, bf, a, scratch, heads, eps, out, w, den):
    """gat_agg_fwd on the H-fold graph with every row shifted by its head's maximum (reduced in the kernel from the partial
    maxima gat_logits_heads_raw left in `scratch`)."""
    lib = _lib.load()
    _need(out, "out"); _need(w, "w"); _need(den, "den"); _need(bf, "bf"); _need(scratch, "scratch", torch.uint8)
    gs = _edge_csr(eg, o + 4)
    check(lib.gode_gat_agg_heads_f32_fwd(ctypes.byref(gs), ptr(eg.src), ptr(eg.tgt), ctypes.byref(proj), o, ptr(bf), ptr(a),
                                         ptr(scratch), eg.E, int(heads), float(eps), ptr(out), ptr(w), ptr(den), stream_ptr()),
          "gode_gat_agg_heads_f32_fwd")


def gat_vjp(eg, proj, o, bf, a, amax, w, den, out, dz, da, dPs, dPt, dA2, dout=None, cot_terms=None, cot_scale=1.0,
            heads=1, raw_scratch=None, defer_maxpath=False):
    """Vector-Jacobian product of the edge-attention aggregation w.r.t. the projections: fills dz[E, o], da[E] (with the
    path through the global maximum folded in), dPs, dPt (N x o) and dA2 (N x 2).  The cotangent is `dout`, or
    cot_scale * (sum cot_terms) masked by out > 0.  heads > 1: `eg` is the H-fold graph, `a` holds logits shifted
    per head and the path through the maximum is applied per head."""
    lib = _lib.load()
    for t, nm in ((out, "out"), (dz, "dz"), (da, "da"), (dout, "dout"), (dPs, "dPs"), (dPt, "dPt"), (dA2, "dA2")):
        _need(t, nm)
    lc = None
    if cot_terms is not None:
        if _need_terms(cot_terms, "cot") != out.numel():
            raise ValueError("gat_vjp: cotangent terms have wrong size")
        lc = lincomb(cot_terms)
    elif dout is None:
        raise ValueError("gat_vjp: dout or cot_terms required")
    gs = _edge_csr(eg, o + 4)
    did = ctypes.c_int32(0)
    at_ptr = ctypes.c_void_p(dA2.data_ptr() + 4)
    check(lib.gode_gat_agg_f32_bwd(ctypes.byref(gs), ptr(eg.src), ptr(eg.tgt), ctypes.byref(proj), o, ptr(bf), ptr(w),
                                   ptr(den), ptr(out), ptr(dout), ctypes.byref(lc) if lc is not None else None,
                                   float(cot_scale), ptr(dz), ptr(da), ptr(dPt), o, at_ptr, 2, ctypes.byref(did),
                                   stream_ptr()), "gode_gat_agg_f32_bwd")
    if eg.E > 0 and heads > 1 and raw_scratch is not None and defer_maxpath and not did.value:
        # first half only; gat_dense_vjp_small(..., maxfix=(raw_scratch, eg.src, eg.tgt)) closes the step
        check(lib.gode_gat_maxpath_heads_part_f32(ptr(a), ptr(da), eg.E, int(heads), ptr(eg.tgt), ptr(raw_scratch), stream_ptr()),
              "gode_gat_maxpath_heads_part_f32")
    elif eg.E > 0 and heads > 1 and raw_scratch is not None:       # raw logits + partial maxima (gat_logits_heads_raw)
        big = bool(did.value)
        check(lib.gode_gat_maxpath_heads_raw_f32(ptr(a), ptr(da), eg.E, int(heads), ptr(eg.tgt), at_ptr if big else None, 2,
                                                 ptr(raw_scratch), stream_ptr()), "gode_gat_maxpath_heads_raw_f32")
    elif eg.E > 0 and heads > 1:
        big = bool(did.value)
        sc = _scratch(a.device, lib.gode_gat_heads_scratch_bytes(eg.E, heads))
        check(lib.gode_gat_maxpath_heads_f32(ptr(a), ptr(da), eg.E, int(heads), ptr(eg.tgt), at_ptr if big else None, 2,
                                             ptr(sc), stream_ptr()), "gode_gat_maxpath_heads_f32")
    elif eg.E > 0:                                                 # path through the global max (GAT/layers.py:47)
        big = bool(did.value)
        check(lib.gode_gat_maxpath_f32(ptr(a), ptr(amax), ptr(da), eg.E, ptr(eg.tgt) if big else None,
                                       at_ptr if big else None, 2, ptr(eg.maxpath_scratch()), stream_ptr()),
              "gode_gat_maxpath_f32")
    if did.value:
        # the record kernel already summed per target; the source side is two incidence products
        spmm(eg.Ms_inc, dz, out=dPs)
        spmm(eg.Ms_inc, da.view(-1, 1), out=dA2[:, 0:1])
    else:
        gat_scatter(eg.Ms_inc, eg.Mt_inc, dz, da, dPs, dPt, dA2)


def gat_maxpath_(a, amax, da):
    """da[first argmax(a)] -= da.sum()   (stand-alone form; gat_vjp applies it itself)."""
    lib = _lib.load()
    sc = _scratch(a.device, lib.gode_gat_maxpath_scratch_bytes(a.numel()))
    check(lib.gode_gat_maxpath_f32(ptr(a), ptr(amax), ptr(da), a.numel(), None, None, 0, ptr(sc), stream_ptr()),
          "gode_gat_maxpath_f32")


def gat_scatter(Ms_inc, Mt_inc, dz, da, dPs, dPt, dA2):
    """Edge cotangents summed per source node into dPs / dA2[:,0] and per target node into dPt / dA2[:,1]."""
    lib = _lib.load()
    _need(dz, "dz"); _need(da, "da"); _need(dPs, "dPs"); _need(dPt, "dPt"); _need(dA2, "dA2")
    n, o = dPs.shape
    check(lib.gode_gat_scatter_f32(ptr(Ms_inc.rowptr), ptr(Ms_inc.col), ptr(Mt_inc.rowptr), ptr(Mt_inc.col), ptr(dz),
                                   ptr(da), o, n, ptr(dPs), o, ptr(dPt), o, ctypes.c_void_p(dA2.data_ptr()),
                                   ctypes.c_void_p(dA2.data_ptr() + 4), 2, stream_ptr()), "gode_gat_scatter_f32")


def time_row_fixup_(g_row0, w_row0, t, at, accumulate):
    """at (+)= <g_row0, w_row0>; g_row0 *= t."""
    lib = _lib.load()
    _need(g_row0, "g_row0"); _need(w_row0, "w_row0"); _need(at, "at")
    check(lib.gode_time_row_fixup_f32(ptr(g_row0), ptr(w_row0), g_row0.numel(), float(t), ptr(at),
                                      1 if accumulate else 0, stream_ptr()), "gode_time_row_fixup_f32")


def time_row_fixup3_(rows, w_rows, t, at):
    """at = sum_b <rows[b], w_rows[b]>; rows[b] *= t   (three weight blocks, one launch)."""
    lib = _lib.load()
    for r, w in zip(rows, w_rows):
        _need(r, "g_row0"); _need(w, "w_row0")
    _need(at, "at")
    a = []
    for r, w in zip(rows, w_rows):
        a += [ptr(r), ptr(w), r.numel()]
    check(lib.gode_time_row_fixup3_f32(*a, float(t), ptr(at), stream_ptr()), "gode_time_row_fixup3_f32")


# ---- QC edge-conditioned messages ---------------------------------------------------------------
# from this many edges on, the message step runs as a block per edge + the per-target sum as an SpMM (two launches); below,
# one launch whose blocks walk a target's edges one after the other.  Round 4: 256 (was 4 096) - on a 20-molecule batch
# (760 edges, in-degree up to 8) the serial walk of the busiest target sets the fused launch's duration: 22-25 us against
# 15-17 us for the two launches (tools/dev/edge_path_ab.py; the results are bit-identical)
EDGE_MSG_MIN_EDGES = 256


def edge_matvec_fwd(Mt, src, A, X):
    """M[v] = sum_{(e,val) in row v} val * A[e] @ X[src[e]]."""
    lib = _lib.load()
    _need(A, "edge_data"); _need(X, "x"); _need(src, "Esrc", torch.int32)
    h = X.shape[1]
    if A.dim() != 3 or A.shape[1] != h or A.shape[2] != h or A.shape[0] != src.numel():
        raise ValueError("edge_matvec: edge_data must be E x h x h with h = %d, got %s" % (h, tuple(A.shape)))
    E = src.numel()
    if E >= EDGE_MSG_MIN_EDGES:
        # large batches: a workgroup per edge streams its matrix, the per-target sum is an SpMM over Etgt
        msg = torch.empty(E, h, dtype=torch.float32, device=X.device)
        check(lib.gode_edge_matvec_msg_f32(ptr(src), ptr(A), ptr(X), h, h, E, ptr(msg), stream_ptr()),
              "gode_edge_matvec_msg_f32")
        return spmm(Mt, msg)
    out = torch.empty(Mt.n_rows, h, dtype=torch.float32, device=X.device)
    check(lib.gode_edge_matvec_f32_fwd(ptr(Mt.rowptr), ptr(Mt.col), ptr(Mt.val), ptr(src), ptr(A), ptr(X), h, h,
                                       Mt.n_rows, ptr(out), h, stream_ptr()), "gode_edge_matvec_f32_fwd")
    return out


def edge_matvec_bwd(edge_row, edge_val, src, A, X, dM, want_dA=True, want_dx=True):
    lib = _lib.load()
    _need(dM, "dM")
    E, h = src.numel(), X.shape[1]
    dA = torch.empty_like(A) if want_dA else None
    dxe = torch.empty(E, h, dtype=torch.float32, device=X.device) if want_dx else None
    check(lib.gode_edge_matvec_f32_bwd(ptr(edge_row), ptr(edge_val), ptr(src), ptr(A), ptr(X), h, ptr(dM), h, h, E,
                                       ptr(dA), ptr(dxe), stream_ptr()), "gode_edge_matvec_f32_bwd")
    return dA, dxe


def dense_first_nonzero(M):
    """index[c] = first row with M[r, c] != 0 (0 for an all-zero column) of a dense 2-D fp32 matrix: what
    `(M != 0).to(torch.uint8).argmax(0)` returns, in one launch (csrc/convert.hip)."""
    lib = _lib.load()
    ld = _need_rows(M, "M")
    n, e = M.shape
    out = torch.empty(e, dtype=torch.int64, device=M.device)
    if e:
        check(lib.gode_dense_first_nonzero_f32(ptr(M), ld, n, e, ptr(out), stream_ptr()), "gode_dense_first_nonzero_f32")
    return out


EDGE_OUTER_MAX_TERMS = 8        # GODE_MAX_TERMS: message steps whose edge-matrix gradient one launch sums


def edge_outer_sum_supported(h):
    return h <= 1024


def edge_outer_sum(edge_row, edge_val, src, pairs, like):
    """dA[e] = sum_t (val_e dM_t[tgt_e]) (x) x_t[src_e] over `pairs` = [(dM_t, x_t), ...] (csrc/edge.hip): the edge-matrix
    gradient of several message steps on the same edge matrices in one pass."""
    lib = _lib.load()
    k = len(pairs)
    if not 1 <= k <= EDGE_OUTER_MAX_TERMS:
        raise ValueError("edge_outer_sum: 1 .. %d steps" % EDGE_OUTER_MAX_TERMS)
    E, h = src.numel(), pairs[0][1].shape[1]
    dms, xs = (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)()
    for t, (dM, x) in enumerate(pairs):
        _need(dM, "dM"); _need(x, "x")
        if dM.shape[1] != h or x.shape[1] != h:
            raise ValueError("edge_outer_sum: every step must have width %d" % h)
        dms[t], xs[t] = dM.data_ptr(), x.data_ptr()
    dA = torch.empty_like(like)
    check(lib.gode_edge_outer_sum_f32(ptr(edge_row), ptr(edge_val), ptr(src), k, dms, xs, h, h, h, E, ptr(dA), stream_ptr()),
          "gode_edge_outer_sum_f32")
    return dA


# ---- QC node update: fused GRU cell ---------------------------------------------------------------
def lstm_cell_supported(B, I, H):
    return bool(_lib.load().gode_lstm_cell_supported(B, I, H))


def lstm_cell_fwd(x, h, c, w_ih, w_hh, b_ih, b_hh, want_gates=True):
    """(h', c') = LSTMCell(x, (h, c)) with torch.nn.LSTM's parameter layout; returns (h', c', gates[B, 4H] or None)."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (h, "h"), (c, "c"), (w_ih, "weight_ih"), (w_hh, "weight_hh"), (b_ih, "bias_ih"), (b_hh, "bias_hh")):
        _need(t, nm)
    B, I = x.shape
    H = h.shape[1]
    if tuple(h.shape) != (B, H) or tuple(c.shape) != (B, H) or tuple(w_ih.shape) != (4 * H, I) or tuple(w_hh.shape) != (4 * H, H):
        raise ValueError("lstm_cell: x B x I, h, c B x H, weight_ih 4H x I, weight_hh 4H x H (got %s %s %s %s %s)"
                         % (tuple(x.shape), tuple(h.shape), tuple(c.shape), tuple(w_ih.shape), tuple(w_hh.shape)))
    f = dict(dtype=torch.float32, device=x.device)
    h_out, c_out = torch.empty(B, H, **f), torch.empty(B, H, **f)
    gates = torch.empty(B, 4 * H, **f) if want_gates else None
    check(lib.gode_lstm_cell_f32_fwd(ptr(x), ptr(h), ptr(c), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), B, I, H, ptr(h_out),
                                     ptr(c_out), ptr(gates), stream_ptr()), "gode_lstm_cell_f32_fwd")
    return h_out, c_out, gates


def lstm_cell_bwd(x, h, c, w_ih, w_hh, gates, c_out, dh_out, dc_out, has_bias=True, want_dx=True, want_dh=True):
    """Returns (dx, dh, dc, dw_ih, dw_hh, db_ih, db_hh)."""
    lib = _lib.load()
    for t, nm in ((gates, "gates"), (c_out, "c_out"), (dh_out, "dh_out"), (dc_out, "dc_out")):
        _need(t, nm)
    B, I = x.shape
    H = h.shape[1]
    f = dict(dtype=torch.float32, device=x.device)
    dx = torch.empty(B, I, **f) if want_dx else None
    dh = torch.empty(B, H, **f) if want_dh else None
    dc = torch.empty(B, H, **f)
    dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
    db_ih = torch.empty(4 * H, **f) if has_bias else None
    db_hh = torch.empty(4 * H, **f) if has_bias else None
    check(lib.gode_lstm_cell_f32_bwd(ptr(x), ptr(h), ptr(c), ptr(w_ih), ptr(w_hh), ptr(gates), ptr(c_out), ptr(dh_out), ptr(dc_out),
                                     B, I, H, ptr(dx), ptr(dh), ptr(dc), ptr(dw_ih), ptr(dw_hh), ptr(db_ih), ptr(db_hh),
                                     stream_ptr()), "gode_lstm_cell_f32_bwd")
    return dx, dh, dc, dw_ih, dw_hh, db_ih, db_hh


def gru_cell_fwd(x, m, w_ih, w_hh, b_ih, b_hh, want_gates=True):
    """out = GRUCell([x | m], x) with torch.nn.GRUCell's parameter layout; returns (out, gates[n, 4h] or None)."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (m, "m"), (w_ih, "weight_ih"), (w_hh, "weight_hh"), (b_ih, "bias_ih"), (b_hh, "bias_hh")):
        _need(t, nm)
    n, h = x.shape
    if m.shape != x.shape or tuple(w_ih.shape) != (3 * h, 2 * h) or tuple(w_hh.shape) != (3 * h, h):
        raise ValueError("gru_cell: x, m must be n x h, weight_ih 3h x 2h, weight_hh 3h x h (got %s %s %s %s)"
                         % (tuple(x.shape), tuple(m.shape), tuple(w_ih.shape), tuple(w_hh.shape)))
    out = torch.empty_like(x)
    gates = torch.empty(n, 4 * h, dtype=torch.float32, device=x.device) if want_gates else None
    check(lib.gode_gru_cell_f32_fwd(ptr(x), ptr(m), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), n, h, ptr(out), ptr(gates),
                                    stream_ptr()), "gode_gru_cell_f32_fwd")
    return out, gates


def gru_wgrad_part_len(n, h):
    """Floats of the weight-gradient partials one gru_cell_bwd call writes."""
    return int(_lib.load().gode_gru_wgrad_parts(n)) * 3 * h * (3 * h + 2)


def gru_cell_bwd(x, m, w_ih, w_hh, gates, dout, has_bias=True, part_out=None):
    """Returns (dx, dm, dw_ih, dw_hh, db_ih, db_hh).  part_out (a float32 view of gru_wgrad_part_len(n, h) elements): the
    call writes its weight-gradient partials there and returns None for the four parameter gradients - the caller sums the
    partials of several applications of the cell with gru_wreduce."""
    lib = _lib.load()
    _need(dout, "dout"); _need(gates, "gates")
    n, h = x.shape
    f = dict(dtype=torch.float32, device=x.device)
    dx, dm = torch.empty_like(x), torch.empty_like(x)
    dgi, dgh = torch.empty(n, 3 * h, **f), torch.empty(n, 3 * h, **f)
    if part_out is not None:
        _need(part_out, "part_out")
        if part_out.numel() != gru_wgrad_part_len(n, h) or n == 0:
            raise ValueError("gru_cell_bwd: part_out must hold gru_wgrad_part_len(n, h) floats")
        check(lib.gode_gru_cell_f32_bwd(ptr(x), ptr(m), ptr(w_ih), ptr(w_hh), ptr(gates), ptr(dout), n, h, ptr(dx), ptr(dm),
                                        ptr(dgi), ptr(dgh), ptr(part_out), None, None, None, None, stream_ptr()),
              "gode_gru_cell_f32_bwd")
        return dx, dm, None, None, None, None
    part = torch.empty(gru_wgrad_part_len(n, h), **f)
    dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
    db_ih = torch.empty(3 * h, **f) if has_bias else None
    db_hh = torch.empty(3 * h, **f) if has_bias else None
    check(lib.gode_gru_cell_f32_bwd(ptr(x), ptr(m), ptr(w_ih), ptr(w_hh), ptr(gates), ptr(dout), n, h, ptr(dx), ptr(dm),
                                    ptr(dgi), ptr(dgh), ptr(part), ptr(dw_ih), ptr(dw_hh), ptr(db_ih), ptr(db_hh),
                                    stream_ptr()), "gode_gru_cell_f32_bwd")
    return dx, dm, dw_ih, dw_hh, db_ih, db_hh


def gru_wreduce(part, n_part, w_ih, w_hh, has_bias=True):
    """(dw_ih, dw_hh, db_ih, db_hh) = the sum of the first n_part partial rows of `part` (gode_gru_wreduce_f32)."""
    lib = _lib.load()
    _need(part, "part")
    h = w_hh.shape[1]
    f = dict(dtype=torch.float32, device=part.device)
    if part.numel() < n_part * 3 * h * (3 * h + 2):
        raise ValueError("gru_wreduce: part is too small")
    dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
    db_ih = torch.empty(3 * h, **f) if has_bias else None
    db_hh = torch.empty(3 * h, **f) if has_bias else None
    check(lib.gode_gru_wreduce_f32(ptr(part), n_part, h, ptr(dw_ih), ptr(dw_hh), ptr(db_ih), ptr(db_hh), stream_ptr()),
          "gode_gru_wreduce_f32")
    return dw_ih, dw_hh, db_ih, db_hh


# ---- Set2Set attention readout ---------------------------------------------------------------------
def segment_attention_fwd(segptr, perm, x, q):
    """a = per-graph softmax(<x_i, q_b>), r_b = sum_i a_i x_i; returns (a[N], r[B, h])."""
    lib = _lib.load()
    _need(x, "x"); _need(q, "q"); _need(segptr, "segptr", torch.int32)
    if perm is not None:
        _need(perm, "perm", torch.int32)
    nb, h = q.shape
    if x.dim() != 2 or x.shape[1] != h or segptr.numel() != nb + 1:
        raise ValueError("segment_attention: x must be N x %d and segptr have %d entries" % (h, nb + 1))
    if int(segptr.numel()) and x.shape[0] == 0:
        return x.new_zeros(0), x.new_zeros(nb, h)
    # every node id must occur exactly once in the segments (callers build them from a batch vector, which
    # guarantees it), so a[] and, in the backward, dx[] are fully written by the kernels
    a = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    r = torch.empty(nb, h, dtype=torch.float32, device=x.device)
    check(lib.gode_segment_attention_f32_fwd(ptr(segptr), ptr(perm), ptr(x), x.stride(0), ptr(q), nb, h, ptr(a), ptr(r),
                                             stream_ptr()), "gode_segment_attention_f32_fwd")
    return a, r


def segment_attention_bwd(segptr, perm, x, q, a, dr):
    lib = _lib.load()
    _need(dr, "dr")
    nb, h = q.shape
    dx = torch.empty_like(x, memory_format=torch.contiguous_format)
    dq = torch.empty_like(q)
    check(lib.gode_segment_attention_f32_bwd(ptr(segptr), ptr(perm), ptr(x), x.stride(0), ptr(q), ptr(a), ptr(dr), nb, h,
                                             ptr(dx), ptr(dq), stream_ptr()), "gode_segment_attention_f32_bwd")
    return dx, dq


# ---- stand-alone GroupNorm on 2-D node features ---------------------------------------------------
def group_norm_fwd(x, groups, eps, gamma, beta):
    lib = _lib.load()
    _need(x, "x"); _need(gamma, "gamma"); _need(beta, "beta")
    n, d = x.shape
    y = torch.empty_like(x)
    check(lib.gode_group_norm_f32_fwd(ptr(x), n, d, groups, float(eps), ptr(gamma), ptr(beta), ptr(y), stream_ptr()),
          "gode_group_norm_f32_fwd")
    return y


def group_norm_bwd(x, groups, eps, gamma, dy, want_affine_grads=True):
    lib = _lib.load()
    _need(x, "x"); _need(gamma, "gamma"); _need(dy, "dy")
    n, d = x.shape
    dx = torch.empty_like(x)
    dg = db = None
    if want_affine_grads:
        n_part = lib.gode_group_norm_parts(n)
        dg = torch.empty(n_part, d, dtype=torch.float32, device=x.device)
        db = torch.empty(n_part, d, dtype=torch.float32, device=x.device)
    check(lib.gode_group_norm_f32_bwd(ptr(x), n, d, groups, float(eps), ptr(gamma), ptr(dy), ptr(dx), ptr(dg), ptr(db),
                                      stream_ptr()), "gode_group_norm_f32_bwd")
    return dx, dg, db


# ---- the whole Set2Set readout loop (QC/set2set.py:50-75): one launch per direction (csrc/set2set.hip) -------------
def set2set_supported(H):
    return bool(_lib.load().gode_set2set_supported(int(H)))


def set2set_fwd(segptr, perm, x, w_ih, w_hh, b_ih, b_hh, steps, n_graphs):
    """Returns (qs, cs, gates, att): qs[steps] is q_star; the rest is what set2set_bwd reads."""
    lib = _lib.load()
    for t_, nm in ((x, "x"), (w_ih, "weight_ih"), (w_hh, "weight_hh"), (b_ih, "bias_ih"), (b_hh, "bias_hh")):
        _need(t_, nm)
    N, H = x.shape
    B = int(n_graphs)
    if tuple(w_ih.shape) != (4 * H, 2 * H) or tuple(w_hh.shape) != (4 * H, H):
        raise ValueError("set2set: weight_ih 4H x 2H, weight_hh 4H x H for H = %d (got %s, %s)" % (H, tuple(w_ih.shape), tuple(w_hh.shape)))
    f = dict(dtype=torch.float32, device=x.device)
    Wt = torch.cat([w_ih, w_hh], 1).t().contiguous()        # (3H) x (4H): gate rows contiguous for the cell's threads
    qs, cs = torch.empty(steps + 1, B, 2 * H, **f), torch.empty(steps + 1, B, H, **f)
    gates, att = torch.empty(steps, B, 4 * H, **f), torch.empty(steps, N, **f)
    check(lib.gode_set2set_f32_fwd(ptr(segptr), ptr(perm), ptr(x), x.stride(0), ptr(Wt), ptr(b_ih), ptr(b_hh), B, H, steps, N,
                                   ptr(qs), ptr(cs), ptr(gates), ptr(att), stream_ptr()), "gode_set2set_f32_fwd")
    return qs, cs, gates, att


def set2set_bwd(segptr, perm, x, w_ih, w_hh, has_bias, saved, dq_star):
    """(dx, dw_ih, dw_hh, db_ih, db_hh) from dq_star (B x 2H): the loop kernel, one small product for the weight gradient
    (dW_ih = DG^T QS; dW_hh is its first H columns), one column sum for the biases."""
    lib = _lib.load()
    qs, cs, gates, att = saved
    steps, B, H = gates.shape[0], qs.shape[1], cs.shape[2]
    N = x.shape[0]
    f = dict(dtype=torch.float32, device=x.device)
    dx = torch.zeros(N, H, **f) if B == 0 else torch.empty(N, H, **f)
    DG = torch.empty(steps, B, 4 * H, **f)
    check(lib.gode_set2set_f32_bwd(ptr(segptr), ptr(perm), ptr(x), x.stride(0), ptr(w_ih), ptr(w_hh), B, H, steps, N, ptr(qs),
                                   ptr(cs), ptr(gates), ptr(att), ptr(dq_star.contiguous()), ptr(dx), ptr(DG), stream_ptr()),
          "gode_set2set_f32_bwd")
    dg2 = DG.view(steps * B, 4 * H)
    dw_ih = gemm(dg2, qs[:steps].reshape(steps * B, 2 * H), trans_a=True)
    dw_hh = dw_ih[:, :H].contiguous()
    db = colsum_(torch.empty(4 * H, **f), dg2) if has_bias else None
    return dx, dw_ih, dw_hh, db, db
