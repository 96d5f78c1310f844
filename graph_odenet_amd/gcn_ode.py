"""Fused GCN ODE function: f(t, x) = relu( A @ ([t | GroupNorm(x)] @ W) + b )
(reference: ODEfunc.forward, GCN/models.py:172-179 -> FixedGraphConvolution.forward,
GCN/layers.py:69-75) and its vector-Jacobian products, as two / five HIP kernel launches:

  forward   : gode_gn_time_gemm_f32 (stage combine + GroupNorm + time column + MFMA GEMM)
              gode_spmm_csr_f32     (gather-aggregate + bias + relu)
  adjoint   : + relu-masked cotangent in the SpMM epilogue, gode_spmm_csr_f32 on A^T,
              gode_gn_time_gemm_bwd_f32 (dS W^T + GroupNorm backward), gode_wgrad_f32,
              gode_colsum_f32 (bias), gode_reduce_parts_f32 (block partials).

Stage inputs arrive as (coef, tensor) term lists and are combined inside the kernels.
"""
import ctypes

import torch

from . import _lib, ops
from .graph import as_graph
from .solver import Field


def _graph_struct(g, d):
    gs = _lib.Graph()
    gs.rowptr, gs.col = g.rowptr.data_ptr(), g.col.data_ptr()
    gs.val = g.val.data_ptr() if g.val is not None else None
    gs.items, gs.n_items = (g.items.data_ptr() if g.items is not None else None), g.n_items
    gs.long_rows = g.long_rows.data_ptr() if g.long_rows is not None else None
    gs.n_long = g.n_long
    part = g.partial(d)
    gs.partial = part.data_ptr() if part is not None else None
    gs.n_rows, gs.nnz = g.n_rows, g.nnz
    return gs


class _Shared:
    """Per-(graph, d) workspaces shared by the forward and adjoint fields."""

    def __init__(self, graph, d, device):
        n = graph.n_rows
        self.S = torch.empty(n, d, dtype=torch.float32, device=device)
        self.dZ = None
        self.dS = None
        self.n = n
        self.d = d
        self.device = device
        self._ky = self._ka = self._kt = None
        self._parts = None
        # rows of the per-block column sums the forward-recompute SpMM of an adjoint stage can leave (0: not on this shape)
        self.y2_rows = 0                       # (a row-partitioned graph runs the per-stage Python path: no per-block sums)
        if not getattr(graph, "is_partitioned", False) and hasattr(graph, "n_items"):
            self.y2_rows = _lib.load().gode_spmm_y2_colsum_rows(graph.n_items if graph.items is not None else n, graph.n_long, d)
        self._y2_colsum = None

    def bwd(self):
        if self.dZ is None:
            self.dZ = torch.empty_like(self.S)
            self.dS = torch.empty_like(self.S)
        return self.dZ, self.dS

    # ---- buffers of the C-level rk4 driver (csrc/ode_driver.hip) --------------------------------
    def stage_buffers(self, adjoint):
        if self._ky is None:
            self._ky = [torch.empty_like(self.S) for _ in range(4)]
        if adjoint and self._ka is None:
            lib = _lib.load()
            P = lib.gode_gcn_ode_theta_len(self.d)
            self._ka = [torch.empty_like(self.S) for _ in range(4)]
            self._kt = [torch.empty(P, dtype=torch.float32, device=self.device) for _ in range(4)]
            nW = (self.d + 1) * self.d
            self._parts = (torch.empty(lib.gode_wgrad_parts(self.n) * nW, dtype=torch.float32, device=self.device),
                           torch.empty(lib.gode_gemm_bwd_parts(self.n) * self.d, dtype=torch.float32, device=self.device),
                           torch.empty(lib.gode_gemm_bwd_parts(self.n) * self.d, dtype=torch.float32, device=self.device),
                           torch.empty(max(lib.gode_colsum_scratch_bytes(self.n, self.d), 16), dtype=torch.uint8,
                                       device=self.device))
        return self._ky, self._ka, self._kt, self._parts

    def workspace_struct(self, adjoint, groups=0):
        ky, ka, kt, parts = self.stage_buffers(adjoint)
        ws = _lib.Rk4Workspace()
        ws.S = self.S.data_ptr()
        for i in range(4):
            ws.ky[i] = ky[i].data_ptr()
        if not adjoint and _lib.load().gode_gcn_small_supported(self.n, self.d, int(groups)):
            # launch-bound graphs: the dopri5 step driver chains the stage inputs through this pair (csrc/ode_driver.hip)
            if getattr(self, "X2", None) is None:
                self.X2 = [torch.empty_like(self.S), torch.empty_like(self.S)]
            ws.X[0], ws.X[1] = self.X2[0].data_ptr(), self.X2[1].data_ptr()
        if adjoint:
            dZ, dS = self.bwd()
            ws.dZ, ws.dS = dZ.data_ptr(), dS.data_ptr()
            if getattr(self, "S2", None) is None:
                self.S2 = torch.empty_like(self.S)
            ws.S2 = self.S2.data_ptr()
            for i in range(4):
                ws.ka[i] = ka[i].data_ptr()
                ws.ktheta[i] = kt[i].data_ptr()
            ws.wpart, ws.gpart, ws.bpart, ws.colsum_scratch = (p.data_ptr() for p in parts)
            if getattr(self, "X2", None) is None:
                self.X2 = [torch.empty_like(self.S), torch.empty_like(self.S)]
            ws.X[0], ws.X[1] = self.X2[0].data_ptr(), self.X2[1].data_ptr()
            sp = self.small_part(groups)
            ws.small_part = sp.data_ptr() if sp is not None else None
            if self.y2_rows > 0:
                if self._y2_colsum is None:
                    self._y2_colsum = torch.empty(self.y2_rows, self.d, dtype=torch.float32, device=self.device)
                ws.y2_colsum = self._y2_colsum.data_ptr()
        return ws

    def small_part(self, groups):
        """Block partials of the fused launch-bound VJP (csrc/small.hip); None when the shape is outside that path."""
        lib = _lib.load()
        if not lib.gode_gcn_small_supported(self.n, self.d, int(groups)):
            return None                       # e.g. 2^20 x 128: 277 MB that nothing would ever read
        if getattr(self, "_small_part", None) is None:
            self._small_part = torch.empty(6 * lib.gode_gcn_small_parts(self.n) * lib.gode_gcn_small_part_len(self.d),
                                           dtype=torch.float32, device=self.device)
        return self._small_part


def _func_struct(spec):
    fs = _lib.GcnOdeFunc()
    fs.A = _graph_struct(spec.graph, spec.d)
    fs.AT = _graph_struct(spec.graph.transpose(), spec.d)
    fs.n, fs.d, fs.groups, fs.eps = spec.graph.n_rows, spec.d, spec.groups, spec.eps
    fs.W, fs.b = spec.W.data_ptr(), spec.b.data_ptr()
    fs.gamma, fs.beta = spec.gamma.data_ptr(), spec.beta.data_ptr()
    return fs


def _by_ptr(ptr_value, candidates):
    for t in candidates:
        if t.data_ptr() == ptr_value:
            return t
    raise RuntimeError("rk4 driver returned an unknown buffer")


# ---- node renumbering of large graphs ("Node order of large graphs") --------------------------
# The aggregation gathers d*4-byte operand rows; which rows are hot is a property of the graph, where they sit in
# memory is ours to choose.  On the R-MAT benchmark graph the hubs are the ids with few one-bits (0, 1, 2, 4, ...): their
# rows share their low address bits, and the same product runs 13-15 % faster with the nodes renumbered hubs first
# (tools/dev/relabel_probe.py; bit-identical state either way: CSRGraph.relabel).
#
# Round 2 decided this with a stopwatch (three timed launches each way, 4 % margin, 1 GB of scratch, a host
# synchronisation) - the same inputs could then integrate on different node orders on different boxes, and parameter
# gradients (sums over the nodes in row order) differ by 2e-5 between the orders.  Since round 3 the decision is a
# function of the graph alone, in integer arithmetic:
#
#   imbalance(order) = max over residues r of  #{gathers whose operand row id = r  (mod m)}  /  (gathers / m),
#                      m = 32 KiB / row bytes (64 at d = 128), over the gathers of A and of A^T
#
# i.e. how unevenly the gathered rows fall on the address bits just above a row.  R-MAT as generated: 10.8; the same
# graph hubs-first: 1.14; a uniform random graph: 1.005.  `node_order`:
#   "auto"   - renumber hubs-first when the operand is past the caches (>= 64 MB, >= 2^21 non-zeros), the given order's
#              imbalance is >= RELABEL_MIN_IMBALANCE and the hubs-first order at least halves it;
#   "given"  - never;   "degree" - always (any size).
# The decision is cached on the graph per (d, node_order) and reported by bench.py.
RELABEL_MIN_OPERAND_BYTES = 64 << 20      # below this the operand sits in the L2s / the Infinity Cache anyway
RELABEL_MIN_NNZ = 1 << 21
RELABEL_MIN_IMBALANCE = 2.0
NODE_ORDERS = ("auto", "given", "degree")


def gather_imbalance(graph, d, inverse=None):
    """The statistic above for `graph` (square CSR) at row width d, optionally under a renumbering (inverse[old] = new).
    Exact integer counts -> the same number on every box.  One device->host copy of 2 m counters."""
    m = max(8, min(4096, (32 << 10) // (4 * d)))
    col = graph.col.to(torch.int64)
    rp = graph.rowptr.to(torch.int64)
    deg = rp[1:] - rp[:-1]                                  # row r is gathered deg(r) times by the product with A^T
    rows = torch.arange(graph.n_rows, device=col.device)
    if inverse is not None:
        col, rows = inverse[col], inverse
    h_a = torch.bincount(col % m, minlength=m)
    h_t = torch.zeros(m, dtype=torch.int64, device=col.device).index_add_(0, rows % m, deg)
    h = torch.stack([h_a, h_t]).tolist()
    return max(max(hh) * m / max(sum(hh), 1) for hh in h)


def tuned_graph(graph, d, node_order="auto"):
    """(graph the ODE block should integrate on, order, inverse) - order is None when the graph is used as given.
    y' = y[order] are the state rows in the renumbered graph; y = y'[inverse].  Deterministic in (graph, d, node_order)."""
    if node_order not in NODE_ORDERS:
        raise ValueError("node_order must be one of %s, got %r" % (NODE_ORDERS, node_order))
    cache = graph.__dict__.setdefault("_tuned", {})
    hit = cache.get((d, node_order))
    if hit is not None:
        return hit
    res = (graph, None, None)
    info = {"node_order": node_order, "renumbered": False}
    square = graph.n_rows == graph.n_cols and not getattr(graph, "is_partitioned", False)
    big = graph.n_rows * d * 4 >= RELABEL_MIN_OPERAND_BYTES and graph.nnz >= RELABEL_MIN_NNZ
    if square and node_order != "given" and (node_order == "degree" or big) \
            and not torch.cuda.is_current_stream_capturing():
        order = graph.degree_order()
        inverse = torch.empty_like(order)
        inverse[order] = torch.arange(graph.n_rows, device=graph.device)
        take = node_order == "degree"
        if not take:
            given, hubs = gather_imbalance(graph, d), gather_imbalance(graph, d, inverse)
            info.update(imbalance_given=round(given, 4), imbalance_hubs_first=round(hubs, 4))
            take = given >= RELABEL_MIN_IMBALANCE and hubs <= 0.5 * given
        if take:
            res = (graph.relabel(order), order, inverse)
            info["renumbered"] = True
    graph.__dict__.setdefault("_tuned_info", {})[(d, node_order)] = info
    cache[(d, node_order)] = res
    return res


class GcnOdeSpec:
    """Plain description of one ODEfunc instance (tensors are the live parameters)."""

    def __init__(self, graph, W, b, gamma, beta, groups, eps):
        self.graph, self.W, self.b, self.gamma, self.beta = graph, W, b, gamma, beta
        self.groups, self.eps = int(groups), float(eps)
        self.d = W.shape[1]
        if W.shape[0] != self.d + 1:
            raise ValueError("GcnOdeSpec: weight must be (d+1) x d, got %s" % (tuple(W.shape),))
        if graph.n_rows != graph.n_cols:
            raise ValueError("GcnOdeSpec: adjacency must be square")


def small_fused(spec):
    """True when the launch-bound one-launch f-eval / VJP of csrc/small.hip applies (and is switched on)."""
    lib = _lib.load()
    return bool(lib.gode_get_option(b"small_fused") == 1 and not getattr(spec.graph, "is_partitioned", False)
                and lib.gode_gcn_small_supported(spec.graph.n_rows, spec.d, spec.groups))


def _feval_small(spec, t, terms, out, pre=None, alpha=1.0, cot=None, out2=None, next_terms=None, x_next=None):
    """next_terms / x_next: the launch also writes the next stage's combined input (a term that is `out` itself is this
    launch's result) - what the C dopri5 step driver does between its stages."""
    lib = _lib.load()
    fs = _func_struct(spec)
    lx = _lib.lincomb(terms)
    lp = _lib.lincomb(pre) if pre is not None else None
    lc = _lib.lincomb(cot) if cot is not None else None
    ln = _lib.lincomb(next_terms) if next_terms is not None else None
    _lib.check(lib.gode_gcn_feval_small_next_f32(ctypes.byref(fs), ctypes.byref(lx), float(t), float(alpha),
                                                 ctypes.byref(lp) if lp is not None else None,
                                                 ctypes.byref(lc) if lc is not None else None,
                                                 _lib.ptr(out2) if cot is not None else None, _lib.ptr(out),
                                                 ctypes.byref(ln) if ln is not None else None,
                                                 _lib.ptr(x_next) if ln is not None else None, _lib.stream_ptr()),
               "gode_gcn_feval_small_next_f32")


class GcnOdeField(Field):
    n_components = 1
    fused = True

    def __init__(self, spec, shared):
        self.s, self.w = spec, shared
        self.token = ("gcn", id(spec.graph))     # identity of the problem besides shapes and parameters (odeint plans)

    def eval(self, t, terms, out):
        s, w = self.s, self.w
        if small_fused(s):
            return _feval_small(s, t, terms[0], out[0])
        ops.gn_time_gemm(terms[0], w.n, s.d, s.groups, s.eps, s.gamma, s.beta, s.W, True, t, out=w.S)
        ops.spmm(s.graph, w.S, bias=s.b, relu=True, out=out[0])

    def rk4_native(self, comps, t0, t1, n_steps):
        """Whole fixed-grid solve in one C call (csrc/ode_driver.hip); comps[0] is re-bound to the result."""
        lib = _lib.load()
        s, w = self.s, self.w
        ky = w.stage_buffers(False)[0]
        fs, ws = _func_struct(s), w.workspace_struct(False, s.groups)
        res = ctypes.c_void_p()
        _lib.check(lib.gode_gcn_ode_rk4_forward(ctypes.byref(fs), _lib.ptr(comps[0]), ctypes.byref(res), ctypes.byref(ws),
                                                float(t0), float(t1), int(n_steps), _lib.stream_ptr()),
                   "gode_gcn_ode_rk4_forward")
        out = _by_ptr(res.value, [comps[0]] + ky)
        if out is not comps[0]:
            # keep the driver's buffers private to the workspace: hand back a copy in the caller's tensor
            comps[0].copy_(out)
        return 4 * n_steps

    def eval_combine(self, t, terms, pre, coef, out):
        """Last RK stage: out[0] = (sum pre[0]) + coef * f(t, sum terms[0]) without materialising f."""
        s, w = self.s, self.w
        if small_fused(s):
            _feval_small(s, t, terms[0], out[0], pre=pre[0], alpha=coef)
            return (0,)
        ops.gn_time_gemm(terms[0], w.n, s.d, s.groups, s.eps, s.gamma, s.beta, s.W, True, t, out=w.S)
        ops.spmm(s.graph, w.S, bias=s.b, relu=True, out=out[0], pre_terms=pre[0], alpha=coef)
        return (0,)

    def dopri5_step_native(self, y, kk, y1, t, h, rtol, atol):
        """One adaptive step in one C call (csrc/ode_driver.hip); returns the error sums as an fp64 device tensor."""
        lib = _lib.load()
        s, w = self.s, self.w
        fs, ws = _func_struct(s), w.workspace_struct(False, s.groups)
        kptr = (ctypes.c_void_p * 7)(*[kk[i][0].data_ptr() for i in range(7)])
        sums = torch.empty(1, dtype=torch.float64, device=y[0].device)
        sc = ops._scratch(y[0].device, lib.gode_rk_errnorm_scratch_bytes())
        _lib.check(lib.gode_gcn_ode_dopri5_step_forward(ctypes.byref(fs), _lib.ptr(y[0]), kptr, _lib.ptr(y1[0]),
                                                        ctypes.byref(ws), float(t), float(h), float(rtol), float(atol),
                                                        _lib.ptr(sums), _lib.ptr(sc), _lib.stream_ptr()),
                   "gode_gcn_ode_dopri5_step_forward")
        return sums


class _PartMixin:
    """Row-partitioned state (partition.py): hooks the adaptive solver uses to see global error norms."""
    rk4_native = None
    dopri5_step_native = None
    adaptive = False                # set by odeint when the method is adaptive

    def _part(self):
        return self.s.graph.part

    def global_numel(self, c, t):
        # the big components are row slices (padding rows of a ragged partition count: they are integrated too)
        return t.numel() * self._part().world if c in self.big_components else t.numel()

    def reduce_error_sums(self, sums):
        """Per-component sums of squares -> sums over the ranks for the row-sliced components; the small components
        are already global in adaptive mode."""
        from .partition import global_sum
        p = self._part()
        if p.world == 1:
            return sums
        big = self.big_components
        t = torch.tensor([sums[c] for c in big], dtype=torch.float64, device=self.w.S.device)
        t = global_sum(t, p.group).tolist()
        out = list(sums)
        for c, v in zip(big, t):
            out[c] = v
        return out

    def reduce_small(self, t):
        from .partition import global_sum
        return global_sum(t, self._part().group)


class GcnOdePartField(_PartMixin, GcnOdeField):
    """The same field on a row-partitioned graph (partition.py): ops.spmm gathers the operand rows between the two
    launches of an f-eval, so the whole-solve C drivers are not offered and the solver takes the per-stage path."""
    big_components = (0,)

    def __init__(self, spec, shared):
        GcnOdeField.__init__(self, spec, shared)
        self.token = None


class GcnOdeAdjointField(Field):
    """Components: [y, a, a_t, W, b, gamma, beta] (b may be absent -> never, FixedGC always has bias here)."""
    fused = True

    def __init__(self, spec, shared, params_order):
        self.s, self.w = spec, shared
        self.params_order = params_order      # list of 'gamma','beta','W','b' in func.parameters() order
        self.n_components = 7
        self.ratio_groups = [[0], [1], [2], [3, 4, 5, 6]]

    def new_state(self, y_end):
        """[y, a, a_t, W, b, gamma, beta]; the small components are views of ONE packed buffer laid out as the
        C driver expects: [W | b | gamma | beta | a_t]."""
        s = self.s
        d = s.d
        nW = (d + 1) * d
        self.theta = torch.zeros(nW + 3 * d + 1, dtype=torch.float32, device=y_end.device)
        th = self.theta
        comps = [y_end.clone(), torch.zeros_like(y_end), th[nW + 3 * d:], th[:nW].view(d + 1, d), th[nW:nW + d],
                 th[nW + d:nW + 2 * d], th[nW + 2 * d:nW + 3 * d]]
        comps[2]._gode_packed = th
        return comps

    def rk4_native(self, comps, t0, t1, n_steps):
        lib = _lib.load()
        s, w = self.s, self.w
        ky, ka, _, _ = w.stage_buffers(True)
        fs, ws = _func_struct(s), w.workspace_struct(True, s.groups)
        yr, ar = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.gode_gcn_ode_rk4_adjoint(ctypes.byref(fs), _lib.ptr(comps[0]), _lib.ptr(comps[1]), _lib.ptr(self.theta),
                                                ctypes.byref(yr), ctypes.byref(ar), ctypes.byref(ws),
                                                float(t0), float(t1), int(n_steps), _lib.stream_ptr()),
                   "gode_gcn_ode_rk4_adjoint")
        yo = _by_ptr(yr.value, [comps[0]] + ky)
        ao = _by_ptr(ar.value, [comps[1]] + ka)
        if yo is not comps[0]:
            comps[0].copy_(yo)
        if ao is not comps[1]:
            comps[1].copy_(ao)
        return 4 * n_steps

    def param_grads(self, comps):
        m = {"W": comps[3], "b": comps[4], "gamma": comps[5], "beta": comps[6]}
        return [m[k] for k in self.params_order]

    def _packed_like(self, y):
        """A work copy of the state with the small components as views of ONE packed buffer, as new_state lays it out."""
        d = self.s.d
        nW = (d + 1) * d
        th = torch.empty(nW + 3 * d + 1, dtype=torch.float32, device=y[0].device)
        comps = [torch.empty_like(y[0]), torch.empty_like(y[1]), th[nW + 3 * d:], th[:nW].view(d + 1, d), th[nW:nW + d],
                 th[nW + d:nW + 2 * d], th[nW + 2 * d:nW + 3 * d]]
        comps[2]._gode_packed = th            # keeps the base alive and lets the native step find it
        return comps

    def alloc_like(self, y, n):
        return [self._packed_like(y) for _ in range(n)]

    @staticmethod
    def _theta_of(comps):
        th = getattr(comps[2], "_gode_packed", None)
        if th is None:
            raise RuntimeError("adjoint state is not in the packed layout")
        return th

    def dopri5_step_native(self, y, kk, y1, t, h, rtol, atol):
        lib = _lib.load()
        s, w = self.s, self.w
        fs, ws = _func_struct(s), w.workspace_struct(True, s.groups)
        arr = lambda idx: (ctypes.c_void_p * 7)(*[kk[i][idx].data_ptr() for i in range(7)])      # noqa: E731
        kth = (ctypes.c_void_p * 7)(*[self._theta_of(kk[i]).data_ptr() for i in range(7)])
        sums = torch.empty(4, dtype=torch.float64, device=y[0].device)
        sc = ops._scratch(y[0].device, lib.gode_rk_errnorm_scratch_bytes())
        _lib.check(lib.gode_gcn_ode_dopri5_step_adjoint(
            ctypes.byref(fs), _lib.ptr(y[0]), _lib.ptr(y[1]), _lib.ptr(self._theta_of(y)), arr(0), arr(1), kth,
            _lib.ptr(y1[0]), _lib.ptr(y1[1]), _lib.ptr(self._theta_of(y1)), ctypes.byref(ws), float(t), float(h),
            float(rtol), float(atol), _lib.ptr(sums), _lib.ptr(sc), _lib.stream_ptr()), "gode_gcn_ode_dopri5_step_adjoint")
        return sums

    def eval(self, t, terms, out):
        self._stage(t, terms, out, None, 0.0)

    def eval_combine(self, t, terms, pre, coef, out):
        """Last RK stage: the new y and a are written by the producing launches (SpMM epilogue / GEMM-VJP
        epilogue); the small components get their k in `out` and are combined by the caller."""
        self._stage(t, terms, out, pre, coef)
        return (0, 1)

    def _stage(self, t, terms, out, pre, coef):
        s, w = self.s, self.w
        dZ, dS = w.bwd()
        n, d = w.n, s.d
        y_terms = terms[0]
        packed = getattr(out[2], "_gode_packed", None)
        if packed is not None and small_fused(s):
            # launch-bound graphs: f-eval (+ masked cotangent), VJP (+ block partials), their reduction - three launches,
            # the same ones the C drivers issue (csrc/small.hip)
            lib = _lib.load()
            last = pre is not None
            _feval_small(s, t, y_terms, out[0], pre=pre[0] if last else None, alpha=coef if last else 1.0,
                         cot=[(-c, x) for (c, x) in terms[1]], out2=dZ)
            fs = _func_struct(s)
            lx = _lib.lincomb(y_terms)
            lp = _lib.lincomb(pre[1]) if last else None
            part = w.small_part(s.groups)
            _lib.check(lib.gode_gcn_vjp_small_f32(ctypes.byref(fs), ctypes.byref(lx), _lib.ptr(dZ), float(coef if last else 1.0),
                                                  ctypes.byref(lp) if lp is not None else None, _lib.ptr(out[1]), _lib.ptr(part),
                                                  _lib.stream_ptr()), "gode_gcn_vjp_small_f32")
            _lib.check(lib.gode_gcn_small_finish_f32(ctypes.byref(fs), _lib.ptr(part), _lib.ptr(packed), float(t),
                                                     _lib.stream_ptr()), "gode_gcn_small_finish_f32")
            return
        ops.gn_time_gemm(y_terms, n, d, s.groups, s.eps, s.gamma, s.beta, s.W, True, t, out=w.S)
        # k_y = relu(A S + b);  dZ = (-a) * mask
        ops.spmm(s.graph, w.S, bias=s.b, relu=True, out=out[0],
                 cot_terms=[(-c, x) for (c, x) in terms[1]], out2=dZ,
                 pre_terms=pre[0] if pre is not None else None, alpha=coef if pre is not None else 1.0)
        ops.spmm(s.graph.transpose(), dZ, out=dS)                       # dS = A^T dZ
        if ops.bwd_wgrad_supported(n, d, s.groups):
            # large graphs at d = 128: k_a and the weight-gradient partials from ONE read of y and dS (csrc/gemm_pc.hip)
            _, dgp, dbp, part = ops.gn_time_gemm_bwd_wgrad(y_terms, n, d, s.groups, s.eps, s.gamma, s.beta, s.W, True, dS,
                                                           out_scale=coef if pre is not None else 1.0, out=out[1],
                                                           pre_terms=pre[1] if pre is not None else None)
        else:
            _, dgp, dbp = ops.gn_time_gemm_bwd(y_terms, n, d, s.groups, s.eps, s.gamma, s.W, True, dS,
                                               out_scale=coef if pre is not None else 1.0, out=out[1],
                                               pre_terms=pre[1] if pre is not None else None)   # k_a = -a^T df/dy
            part = ops.wgrad(y_terms, n, d, s.groups, s.eps, s.gamma, s.beta, dS, True)
        ops.reduce_parts_(out[3].view(-1), part)                        # row 0 = colsum(dS)
        out[2].copy_((out[3][0] * s.W[0]).sum().reshape(1))             # a_t' = -a^T df/dt
        out[3][0].mul_(t)                                               # dW[0,:] = t * colsum(dS)
        ops.colsum_(out[4], dZ)
        if dgp is not None:
            ops.reduce_parts2_(out[5], dgp, out[6], dbp)
        else:
            out[5].zero_(); out[6].zero_()


class GcnOdePartAdjointField(_PartMixin, GcnOdeAdjointField):
    """Adjoint on a row-partitioned graph: y and a are this rank's rows.  Fixed grid: the small components (a_t and the
    parameter gradients) stay PARTIAL sums over the local rows and the caller sums them over the ranks once per step
    (GradBucket.allreduce_sum) - the adjoint ODE is linear in them, so the sum commutes with the integration.
    Adaptive method: the step-size controller looks at their error norms, so every stage's small derivatives are summed
    over the ranks (one 66 KB all-reduce per stage at d = 128) and every rank integrates the GLOBAL small components;
    param_grads then hands back 1/world of them, so that the caller's allreduce_sum is right in both modes."""
    big_components = (0, 1)

    def _stage(self, t, terms, out, pre, coef):
        GcnOdeAdjointField._stage(self, t, terms, out, pre, coef)
        if self.adaptive and self._part().world > 1:
            packed = getattr(out[2], "_gode_packed", None)
            if packed is not None:
                self.reduce_small(packed)
            else:
                for c in range(2, 7):
                    self.reduce_small(out[c])

    def param_grads(self, comps):
        gs = GcnOdeAdjointField.param_grads(self, comps)
        w = self._part().world
        return [g / w for g in gs] if self.adaptive and w > 1 else gs


class _OdeFuncFn(torch.autograd.Function):
    """Stand-alone differentiable f(t, x) (used when ODEfunc is called outside our solver)."""

    @staticmethod
    def forward(ctx, spec, t, x, W, b, gamma, beta):
        x = x.contiguous()
        n, d = x.shape
        S = ops.gn_time_gemm([(1.0, x)], n, d, spec.groups, spec.eps, gamma, beta, W, True, t)
        out = ops.spmm(spec.graph, S, bias=b, relu=True)
        ctx.spec, ctx.t = spec, t
        ctx.save_for_backward(x, W, b, gamma, beta, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, b, gamma, beta, out = ctx.saved_tensors
        spec, t = ctx.spec, ctx.t
        n, d = x.shape
        dZ = g.contiguous() * (out > 0).to(g.dtype)
        dS = ops.spmm(spec.graph.transpose(), dZ)
        dx, dgp, dbp = ops.gn_time_gemm_bwd([(1.0, x)], n, d, spec.groups, spec.eps, gamma, W, True, dS)
        part = ops.wgrad([(1.0, x)], n, d, spec.groups, spec.eps, gamma, beta, dS, True)
        gW = torch.empty_like(W)
        ops.reduce_parts_(gW.view(-1), part)
        gW[0].mul_(t)
        gb = torch.empty_like(b)
        ops.colsum_(gb, dZ)
        gg = torch.zeros_like(gamma)
        gbe = torch.zeros_like(beta)
        if dgp is not None:
            ops.reduce_parts2_(gg, dgp, gbe, dbp)
        return None, None, dx, gW, gb, gg, gbe


def odefunc_apply(adj, t, x, W, b, gamma, beta, groups, eps):
    spec = GcnOdeSpec(as_graph(adj), W, b, gamma, beta, groups, eps)
    return _OdeFuncFn.apply(spec, float(t), x, W, b, gamma, beta)
