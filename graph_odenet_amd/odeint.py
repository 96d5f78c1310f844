"""`odeint` / `odeint_adjoint` with the public torchdiffeq signature (the solver seam of
GCN/models.py:5,192, GAT/models.py:5,192 in the reference):

    odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None)
        -> Tensor[len(t), *y0.shape]

`func` is an nn.Module called as func(t: 0-dim tensor, y).  When it offers
`gode_fields(y0)` (our ODEfunc does) the whole f-eval and its VJP run as fused HIP kernels
and stage inputs are never materialised; any other module runs through autograd with the
RK arithmetic still in the HIP kernels.
"""
import weakref

import torch

from . import ops
from .solver import (Dopri5Stats, Field, integrate_dopri5_inplace, integrate_rk4, uniform_grid)


NATIVE_RK4 = True      # fused fields: issue a whole rk4 solve from one C-ABI call (False: per-stage Python driver)
# Fixed-grid solves of fused fields on launch-bound sizes (state of at most this many elements) are captured into a
# HIP graph the second time the same solve is requested and replayed afterwards; 0 turns the capture off.
GRAPH_CAPTURE_MAX_ELEMS = 1 << 21
# `nfe` after a backward pass is an integer the reference's harness prints (GCN/train_res.py:100-101).  torchdiffeq's
# adjoint evaluates func once more per output time for dL/dt before it integrates backwards; under fixed-grid rk4 that
# value can reach nothing the seam returns (the time gradient is discarded and a fixed grid never looks at it), so the
# evaluation is NOT executed here - but with this flag on (default) it is COUNTED, so that nfe_b reads 65 for 64 stage
# evaluations exactly as with torchdiffeq (asserted against the oracle's own count in tests/test_gpu_gcn.py).  Off:
# nfe counts the evaluations actually launched (64).  Adaptive dopri5 executes and counts the evaluation either way.
NFE_COUNTS_SKIPPED_DLDT_EVAL = True


def _materialise(terms):
    if len(terms) == 1 and terms[0][0] == 1.0:
        return terms[0][1]
    out = torch.empty_like(terms[0][1])
    return ops.lincomb_(out, terms)


class AutogradField(Field):
    """Forward field of an arbitrary module: one component."""
    n_components = 1

    def __init__(self, func, like):
        self.func = func
        self.like = like

    def eval(self, t, terms, out):
        y = _materialise(terms[0])
        tt = torch.tensor(t, dtype=self.like.dtype, device=self.like.device)
        with torch.no_grad():
            out[0].copy_(self.func(tt, y))


class AutogradAdjointField(Field):
    """Augmented field (y, a, a_t, theta...) of an arbitrary module via torch.autograd."""

    def __init__(self, func, params, like):
        self.func = func
        self.params = tuple(params)
        self.like = like
        self.n_components = 3 + len(self.params)
        self.ratio_groups = [[0], [1], [2]] + ([list(range(3, 3 + len(self.params)))] if self.params else [])

    def eval(self, t, terms, out):
        y = _materialise(terms[0])
        a = _materialise(terms[1])
        with torch.enable_grad():
            tt = torch.tensor(t, dtype=self.like.dtype, device=self.like.device, requires_grad=True)
            y_ = y.detach().requires_grad_(True)
            fe = self.func(tt, y_)
            vj = torch.autograd.grad(fe, (tt, y_) + self.params, -a, allow_unused=True)
        out[0].copy_(fe.detach())
        out[1].copy_(vj[1]) if vj[1] is not None else out[1].zero_()
        out[2].copy_(vj[0].reshape(out[2].shape)) if vj[0] is not None else out[2].zero_()
        for i, _p in enumerate(self.params):
            g = vj[2 + i]
            out[3 + i].copy_(g) if g is not None else out[3 + i].zero_()


def _check_state(y0):
    if not torch.is_tensor(y0):
        raise TypeError("graph_odenet_amd.odeint: y0 must be a tensor (tuple states are not part of the hot path)")
    if not y0.is_cuda:
        raise RuntimeError("graph_odenet_amd.odeint: y0 must live on the GPU (got %s); there is no CPU path" % y0.device)
    if y0.dtype != torch.float32:
        raise TypeError("graph_odenet_amd.odeint: y0 must be float32")


def _times(t):
    # The reference's ODEBlock keeps `integration_time` on the device (GCN/models.py:195 `type_as(x)`), so reading it
    # is a device->host copy = a host synchronisation in the middle of every forward pass (measured at C5: the GPU
    # then idles ~0.3 ms while the host catches up).  The values are remembered on the tensor object together with its
    # storage address and version counter, so only the first call (and any call after an in-place change or a
    # `set_` / `.data =` re-pointing) pays for it.  NOT seen: a write through an alias that bumps no version counter of
    # this tensor (`t.data.copy_(...)`, a raw kernel): pass the end points as Python floats or a CPU tensor then
    # (models.ODEBlock re-creates `integration_time` from its own floats on every forward, as the reference does).
    if torch.is_tensor(t) and t.is_cuda:
        hit = getattr(t, "_gode_times", None)
        key = (t.data_ptr(), t._version, tuple(t.shape))
        if hit is not None and hit[0] == key:
            tl = hit[1]
        else:
            tl = [float(v) for v in t.tolist()]
            try:
                t._gode_times = (key, tl)
            except Exception:
                pass
        tl = list(tl)
    else:
        tl = [float(v) for v in (t.tolist() if torch.is_tensor(t) else t)]
    if len(tl) < 2:
        raise ValueError("odeint: t must hold at least two time points")
    return tl


def _method(method):
    m = "dopri5" if method is None else method
    if m not in ("dopri5", "rk4"):
        raise ValueError("odeint: unsupported method %r (supported: dopri5, rk4)" % (method,))
    return m


def _integrate(field, comps, t0, t1, rtol, atol, method, options, stats):
    if method == "rk4":
        stats.nfe += _run_rk4(field, comps, t0, t1, uniform_grid(t0, t1, (options or {}).get("step_size")))
    else:
        if getattr(field, "fixed_grid_only", False):
            raise NotImplementedError("odeint: this field supports the fixed-grid method only (method='rk4')")
        if hasattr(field, "adaptive"):
            field.adaptive = True                  # row-partitioned fields: keep the small components global
        prep = getattr(field, "prepare", None)
        if prep is not None:
            prep()
        integrate_dopri5_inplace(field, comps, t0, t1, rtol, atol, stats)


class _GraphedSolve:
    """One captured fixed-grid solve: static input components, the HIP graph, and the components holding the result."""

    def __init__(self, field, comps, t0, t1, n_steps):
        from . import _lib
        lib = _lib.load()
        self.field = field
        self.inputs = list(comps)
        self.nfe = 4 * n_steps
        work = list(comps)
        overlap = lib.gode_get_option(b"overlap")
        lib.gode_set_option(b"overlap", 0)           # launch-bound sizes gain nothing from the second stream
        try:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                _run_rk4(field, work, t0, t1, n_steps)
        finally:
            lib.gode_set_option(b"overlap", overlap)
        self.outputs = work

    def run(self, values):
        """values[c]: tensor copied into input component c, or None for zero.  Returns clones of the result."""
        for dst, src in zip(self.inputs, values):
            if src is None:
                dst.zero_()
            else:
                dst.copy_(src)
        self.graph.replay()
        return [o.clone() for o in self.outputs]


class _Plan:
    """Per-(func, problem) record: fields kept alive across training steps and their captured solves."""

    def __init__(self, fwd, mk_adj, plist):
        self.fwd, self.mk_adj, self.plist = fwd, mk_adj, plist
        self.adj = None
        self.seen_f = self.seen_b = 0
        self.gf = self.gb = None
        self.no_capture = False          # set when a capture attempt failed: the plan stays on the eager path


_PLANS = weakref.WeakKeyDictionary()      # func module -> {key: _Plan}; kept off the module so that deepcopy / pickling
                                          # of a model never meets a HIP graph


def plans_of(func):
    """The captured-solve plans of an ODE function module (created on first use)."""
    plans = _PLANS.get(func)
    if plans is None:
        plans = _PLANS[func] = {}
    return plans


def _try_capture(plan, field, comps, t0, t1):
    """A captured solve, or None (and the plan switched to the eager path for good) when the capture fails."""
    try:
        return _GraphedSolve(field, comps, t0, t1, plan.n_steps)
    except RuntimeError as exc:
        import warnings
        plan.no_capture = True
        warnings.warn("graph_odenet_amd: HIP-graph capture of a fixed-grid solve failed (%s); staying on the eager path"
                      % (str(exc).splitlines()[0],))
        return None


def _plan_for(func, y0, tl, method, options, params):
    """(plan, fields) of this solve; plan is None when the solve is not eligible for capture.  A func opts in by
    offering gode_plan_token(y0): a cheap hashable identity of everything the fields depend on besides the state
    shape and the parameter storage (i.e. the graph)."""
    tok = getattr(func, "gode_plan_token", None)
    token = None
    if (tok is not None and GRAPH_CAPTURE_MAX_ELEMS > 0 and method == "rk4" and len(tl) == 2
            and y0.numel() <= GRAPH_CAPTURE_MAX_ELEMS and not torch.cuda.is_current_stream_capturing()):
        token = tok(y0)
    if token is None:
        return None, _fields(func, y0)
    n = uniform_grid(tl[0], tl[1], (options or {}).get("step_size"))
    # the hook's identity is part of the key: a func whose gode_fields was swapped (e.g. to force the autograd path)
    # must not be served the fused fields of an earlier plan
    key = (tuple(y0.shape), y0.device.index, tl[0], tl[1], n, token, tuple(p.data_ptr() for p in params), NATIVE_RK4,
           id(getattr(func, "gode_fields", None).__func__) if hasattr(getattr(func, "gode_fields", None), "__func__") else None)
    plans = plans_of(func)
    plan = plans.get(key)
    if plan is None:
        fields = _fields(func, y0)
        if not getattr(fields[0], "fused", False):
            return None, fields
        if len(plans) >= 4:
            plans.clear()                       # a func that keeps changing graphs / shapes: start over
        plan = plans[key] = _Plan(*fields)
        plan.n_steps = n
    return plan, (plan.fwd, plan.mk_adj, plan.plist)


def _run_rk4(field, comps, t0, t1, n):
    prep = getattr(field, "prepare", None)
    if prep is not None:
        prep()
    native = getattr(field, "rk4_native", None)
    if native is not None and NATIVE_RK4:
        return native(comps, t0, t1, n)          # whole solve issued from C (csrc/ode_driver.hip)
    return integrate_rk4(field, comps, t0, t1, n)


def _fields(func, y0):
    mk = getattr(func, "gode_fields", None)
    if mk is not None:
        pair = mk(y0)
        if pair is not None:
            return pair
    params = tuple(p for p in func.parameters() if p.requires_grad) if isinstance(func, torch.nn.Module) else ()
    return AutogradField(func, y0), (lambda: AutogradAdjointField(func, params, y0)), params


def _rows(field):
    """(order, inverse) when the field integrates on renumbered nodes (state rows y' = y[order]), else (None, None)."""
    return getattr(field, "row_order", None), getattr(field, "row_inverse", None)


def _bump_nfe(func, n):
    # fused fields do not call func.forward; keep the reference's counter (GCN/models.py:173) alive
    if getattr(func, "_gode_counts_nfe", False):
        func.nfe += n


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None):
    """Forward solve without gradient support through the solver (use odeint_adjoint to train)."""
    _check_state(y0)
    tl = _times(t)
    method = _method(method)
    fwd, _, _ = _fields(func, y0)
    stats = Dopri5Stats()
    order, inverse = _rows(fwd)
    ys = [y0.detach().contiguous().clone() if order is None else y0.detach().index_select(0, order)]
    outs = [y0.detach().contiguous().clone()]
    with torch.no_grad():
        for i in range(1, len(tl)):
            _integrate(fwd, ys, tl[i - 1], tl[i], rtol, atol, method, options, stats)
            outs.append(ys[0].clone() if order is None else ys[0].index_select(0, inverse))
    _bump_nfe(func, stats.nfe if getattr(fwd, "fused", False) else 0)
    return torch.stack(outs)


class _OdeintAdjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, func, tl, rtol, atol, method, options, last_only, y0, *params):
        plan, (fwd, mk_adj, plist) = _plan_for(func, y0, tl, method, options, params)
        stats = Dopri5Stats()
        order, inverse = _rows(fwd)
        y0c = y0.detach().contiguous()
        # ans_p[i] = state at tl[i] in the field's row order (what the adjoint starts from); every slice is written in
        # place - the start by the renumbering gather, each later one by the integrator working on that slice - so a
        # solve moves the state through memory twice around the integration instead of once per clone and stack
        # (at 2^20 x 128 a pass is 0.2 ms; the clone-and-stack form made ten of them per forward pass)
        ans_p = torch.empty((len(tl),) + tuple(y0c.shape), dtype=y0c.dtype, device=y0c.device)
        if order is None:
            ans_p[0].copy_(y0c)
        else:
            torch.index_select(y0c, 0, order, out=ans_p[0])     # state rows in the renumbered graph's order
        if plan is not None and plan.gf is None and plan.seen_f >= 1 and not plan.no_capture:
            plan.gf = _try_capture(plan, fwd, [ans_p[0].clone()], tl[0], tl[1])
        if plan is not None and plan.gf is not None:
            (y_end,) = plan.gf.run([ans_p[0]])
            stats.nfe += plan.gf.nfe
            ans_p[1].copy_(y_end)
        else:
            for i in range(1, len(tl)):
                ans_p[i].copy_(ans_p[i - 1])
                ys = [ans_p[i]]
                _integrate(fwd, ys, tl[i - 1], tl[i], rtol, atol, method, options, stats)
                if ys[0].data_ptr() != ans_p[i].data_ptr():       # an integrator that returns a new tensor
                    ans_p[i].copy_(ys[0])
            if plan is not None:
                plan.seen_f += 1
        _bump_nfe(func, stats.nfe if getattr(fwd, "fused", False) else 0)
        ctx.last_only = bool(last_only)
        if last_only:
            # only y(t[-1]) leaves (what the reference's ODEBlock keeps, GCN/models.py:200 `out[1]`): no copy of y0 into
            # a stacked result, and the backward pass gets the cotangent of that one state instead of a zero-filled stack
            ans = ans_p[-1].clone() if order is None else torch.index_select(ans_p[-1], 0, inverse)
        elif order is None:
            ans = ans_p
        else:
            ans = torch.empty_like(ans_p)
            ans[0].copy_(y0c)
            for i in range(1, len(tl)):
                torch.index_select(ans_p[i], 0, inverse, out=ans[i])
        ctx.func, ctx.tl, ctx.rtol, ctx.atol, ctx.method, ctx.options = func, tl, rtol, atol, method, options
        ctx.mk_adj = mk_adj
        ctx.fwd = fwd
        ctx.plan = plan
        ctx.n_params = len(params)
        ctx.rows = (order, inverse)
        ctx.save_for_backward(ans_p)
        return ans

    @staticmethod
    def backward(ctx, grad_out):
        (ans,) = ctx.saved_tensors
        func, tl = ctx.func, ctx.tl
        grad_out = grad_out.contiguous()
        order, inverse = ctx.rows
        back = (lambda g: g) if order is None else (lambda g: g.index_select(0, inverse))
        last_only = ctx.last_only
        n_t = len(tl)

        def g_raw(i):                                 # dL/dy(tl[i]) in the caller's row order, None = zero
            if last_only:
                return grad_out if i in (n_t - 1, -1) else None
            return grad_out[i]

        def g_at(i):                                  # the same in the field's row order (one gather, when asked for)
            g = g_raw(i)
            return g if (g is None or order is None) else g.index_select(0, order)

        def add_start(g):                             # + cotangent of the start state, rows already in the caller's order
            g0 = g_raw(0) if n_t > 1 else None
            return g if g0 is None else g.add_(g0)
        plan = ctx.plan
        if plan is not None and plan.seen_b >= 1 and plan.gb is None and not plan.no_capture:
            with torch.no_grad():
                plan.adj = plan.mk_adj()
                plan.gb = _try_capture(plan, plan.adj, plan.adj.new_state(ans[1]), tl[1], tl[0])
        if plan is not None and plan.gb is not None:
            with torch.no_grad():
                vals = [ans[1], g_at(1)] + [None] * (len(plan.gb.inputs) - 2)
                comps = plan.gb.run(vals)
                gy0 = add_start(back(comps[1]))
            _bump_nfe(func, plan.gb.nfe + ((n_t - 1) if NFE_COUNTS_SKIPPED_DLDT_EVAL else 0))
            return (None, None, None, None, None, None, None, gy0, *plan.adj.param_grads(comps))
        if plan is not None:
            plan.seen_b += 1
        adj = ctx.mk_adj()
        fwd = ctx.fwd
        ctx_tmp = torch.empty_like(ans[0])
        stats = Dopri5Stats()
        with torch.no_grad():
            comps = adj.new_state(ans[-1]) if hasattr(adj, "new_state") else None
            if comps is None:
                comps = [ans[-1].clone(), torch.zeros_like(ans[-1]),
                         torch.zeros(1, dtype=ans.dtype, device=ans.device)]
                comps += [torch.zeros_like(p) for p in adj.params]
            if order is None:
                comps[1].copy_(g_raw(-1))
            else:
                torch.index_select(g_raw(-1), 0, order, out=comps[1])
            for i in range(len(tl) - 1, 0, -1):
                comps[0].copy_(ans[i])
                if ctx.method != "rk4":
                    # dL/dt at the output time enters the adaptive error control (torchdiffeq evaluates
                    # func once more here); a fixed grid never looks at it, so rk4 skips the eval.
                    fwd.eval(tl[i], [[(1.0, ans[i])]], [ctx_tmp])
                    stats.nfe += 1
                    gi = g_at(i)
                    if gi is not None:
                        comps[2].sub_((ctx_tmp * gi).sum().reshape(1))
                    if getattr(adj, "adaptive", None) is not None:
                        adj.adaptive = True
                        adj.reduce_small(comps[2])      # row-partitioned: a_t is a sum over all rows
                _integrate(adj, comps, tl[i], tl[i - 1], ctx.rtol, ctx.atol, ctx.method, ctx.options, stats)
                if i > 1 and g_raw(i - 1) is not None:
                    comps[1].add_(g_at(i - 1))
            # the cotangent of the start state is added after the rows are back in the caller's order: one pass, and
            # no gather of a slice that is all zeros whenever the loss only looks at the end state
            gy0 = add_start(back(comps[1]))
        skipped = (n_t - 1) if (ctx.method == "rk4" and NFE_COUNTS_SKIPPED_DLDT_EVAL) else 0
        _bump_nfe(func, (stats.nfe if getattr(adj, "fused", False) else 0) + skipped)
        pg = adj.param_grads(comps) if hasattr(adj, "param_grads") else comps[3:]
        return (None, None, None, None, None, None, None, gy0, *pg)


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None, _last_only=False):
    """torchdiffeq's odeint_adjoint for the two solvers of the hot path.  `_last_only=True` (not part of the reference
    API; used by models.ODEBlock, which keeps `out[1]` only) returns y(t[-1]) instead of the stack over t."""
    _check_state(y0)
    tl = _times(t)
    method = _method(method)
    if not isinstance(func, torch.nn.Module):
        raise ValueError("odeint_adjoint: func must be an nn.Module")
    params = tuple(p for p in func.parameters() if p.requires_grad)
    return _OdeintAdjoint.apply(func, tl, float(rtol), float(atol), method, options, bool(_last_only), y0, *params)
