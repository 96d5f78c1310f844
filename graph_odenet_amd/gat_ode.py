"""Fused GAT ODE function  f(t, x) = relu(EdgeAttention([t | GroupNorm(x)]))  (reference: GAT/models.py:172-179 ->
GAT/layers.py:95-122) and its vector-Jacobian product as plain kernel sequences over the C ABI - no autograd
graph, no per-edge E x 2i tensor.

The two Linear layers of the reference act on h_e = [x[src_e] | x[tgt_e]] (f: o x 2i, w: 1 x 2i, i = d+1 with the
time column first).  They are applied at NODE level and split by role, so that the two d x d blocks run on the square
MFMA kernels and only the two logit columns take the generic path:

  Ps = [t|xn] Wsrc   (Wsrc = Wf[:, :i]^T, i x o)      As = [t|xn] ww[:, :i]^T     z_e = Ps[src_e] + Pt[tgt_e] + bf
  Pt = [t|xn] Wtgt   (Wtgt = Wf[:, i:]^T, i x o)      At = [t|xn] ww[:, i:]^T     a_e = As[src_e] + At[tgt_e] + bw

  forward : 3 x gode_gn_time_gemm_f32 (GroupNorm + time column fused; a multi-term stage input is combined once and
            written out by the first launch), gode_gat_logits_f32, gode_gat_agg_f32_fwd
  adjoint : + gode_gat_agg_f32_bwd (stage cotangent combined and relu-masked inside), gode_gat_maxpath_f32,
            gode_gat_scatter_f32 (all four incidence sums in one launch), 3 x gode_gn_time_gemm_bwd_f32 (accumulating
            into k_a), 3 x gode_wgrad_f32 + reductions, gode_time_row_fixup_f32, column sums for the biases.

The adjoint integrates [y, a, a_t, theta] with theta = [Wsrc | Wtgt | Wlog | bf | bw | gamma | beta] packed in ONE
buffer (one RK combine launch for all parameters); the gradient is converted back to the parameters' layout once,
at the end of the solve.
"""
import ctypes

import torch

from . import _lib, ops
from .gat_layers import edge_graph
from .solver import Field


# launch-bound graphs: the reduction launches that close an adjoint stage run as one (ops.reduce_segments_); larger graphs
# keep the separate launches (the weight-gradient reduction has a 16-byte form that matters there)
MERGED_FINISH_MAX_ROWS = 1 << 16


def _repack(spec, heads):
    """Keep spec.Wpacked (csrc/gat_small.hip: gode_gat_small_pack_f32) in step with the weights.  Worth a launch per solve
    from d = 32 on (at d = 16 the blocks are 2.4 KB and the kernels lay them out themselves)."""
    if spec.d < 32 or not spec.Wsrc.is_cuda:
        return
    lib = _lib.load()
    # spec.n is the BASE node count (what _HeadsWork and the fields' small() test): with heads spec.eg is the H-fold graph
    if not lib.gode_gat_small_supported(spec.n, spec.d, int(spec.groups), heads):
        return
    spec.Wpacked = ops.gat_small_pack(spec.Wsrc, spec.Wtgt, spec.Wlog, heads, out=spec.Wpacked)


class GatOdeSpec:
    def __init__(self, eg, layer, norm):
        self.eg, self.layer, self.norm = eg, layer, norm
        self.d = layer.out_features
        self.i = layer.in_features                    # d + 1 (time column first)
        if self.i != self.d + 1:
            raise ValueError("GatOdeSpec: the ODE layer maps d+1 -> d features")
        self.groups, self.eps_gn = int(norm.num_groups), float(norm.eps)
        self.eps = float(layer.eps)
        dev = layer.f.weight.device
        self.Wsrc = torch.empty(self.i, self.d, dtype=torch.float32, device=dev)
        self.Wtgt = torch.empty(self.i, self.d, dtype=torch.float32, device=dev)
        self.Wlog = torch.empty(self.i, 2, dtype=torch.float32, device=dev)
        self.Wpacked = None                           # LDS images of the three blocks for the one-launch kernels (d >= 32)
        self.n = eg.n
        self.refresh()
        self.bf, self.bw = layer.f.bias.detach(), layer.w.bias.detach()
        self.gamma, self.beta = norm.weight.detach(), norm.bias.detach()
        self.n = eg.n
        i, o, d = self.i, self.d, self.d
        # offsets inside the packed parameter-gradient buffer
        self.off = {}
        p = 0
        for name, ln in (("Wsrc", i * o), ("Wtgt", i * o), ("Wlog", i * 2), ("bf", o), ("bw", 1), ("gamma", d), ("beta", d)):
            self.off[name] = (p, p + ln)
            p += ln
        self.n_theta = p

    def refresh(self):
        """Re-pack the two Linear weights by role, in place (at the start of every solve: the parameters move between
        solves, the buffers - and a HIP graph captured over them - do not)."""
        Wf, ww = self.layer.f.weight.detach(), self.layer.w.weight.detach()
        i = self.i
        self.Wsrc.copy_(Wf[:, :i].t())
        self.Wtgt.copy_(Wf[:, i:].t())
        self.Wlog[:, 0].copy_(ww[0, :i])
        self.Wlog[:, 1].copy_(ww[0, i:])
        _repack(self, 1)


    def views(self, theta):
        o, i = self.d, self.i
        v = {k: theta[a:b] for k, (a, b) in self.off.items()}
        v["Wsrc"], v["Wtgt"], v["Wlog"] = v["Wsrc"].view(i, o), v["Wtgt"].view(i, o), v["Wlog"].view(i, 2)
        return v


class _Work:
    """Buffers of one (graph, d): allocated once, so that neither the eager path nor a captured graph allocates."""

    def __init__(self, spec, device):
        n, o, E = spec.n, spec.d, spec.eg.E
        lib = _lib.load()
        f = dict(dtype=torch.float32, device=device)
        self.X = torch.empty(n, o, **f)
        self.Ps, self.Pt, self.A2 = torch.empty(n, o, **f), torch.empty(n, o, **f), torch.empty(n, 2, **f)
        self.dPs, self.dPt, self.dA2 = torch.empty(n, o, **f), torch.empty(n, o, **f), torch.empty(n, 2, **f)
        self.a, self.amax = torch.empty(max(E, 1), **f)[:E], torch.empty(1, **f)
        self.wgt, self.den = torch.zeros(max(E, 1), **f)[:E], torch.empty(n, **f)
        self.dz, self.da = torch.zeros(max(E, 1), o, **f)[:E], torch.zeros(max(E, 1), **f)[:E]
        self.proj = ops.gat_proj(self.Ps, self.Pt, self.A2)
        self.np_b = lib.gode_gemm_bwd_parts(n)
        self.gp, self.bp = torch.empty(3 * self.np_b, o, **f), torch.empty(3 * self.np_b, o, **f)
        npw = lib.gode_wgrad_parts(n)
        self.wp = [torch.empty(npw, spec.i * o, **f), torch.empty(npw, spec.i * o, **f), torch.empty(npw, spec.i * 2, **f)]
        # extras of the C-level dopri5 step (csrc/gat_driver.hip)
        self.pair = torch.empty(2, **f)
        u8 = dict(dtype=torch.uint8, device=device)
        self.logits_scratch = torch.empty(max(lib.gode_gat_logits_scratch_bytes(E), 16), **u8)
        self.colsum_scratch = torch.empty(max(lib.gode_colsum_scratch_bytes(n, o), 16), **u8)
        self.colsum_scratch2 = torch.empty(max(lib.gode_colsum_scratch_bytes(n, 2), 16), **u8)
        self.err_scratch = torch.empty(lib.gode_rk_errnorm_scratch_bytes(), **u8)
        # launch-bound graphs: block partials of the one-launch dense VJP (csrc/gat_small.hip)
        self.small_part = ops.gat_small_part(n, o, 1, device) if lib.gode_gat_small_supported(n, o, spec.groups, 1) else None
        self.step_parts = None                 # four such buffers, one per stage of a fixed-grid step (allocated on first use)


class GatOdeField(Field):
    n_components = 1
    fused = True

    def __init__(self, spec, work):
        self.s, self.w = spec, work
        self.token = ("gat", id(spec.eg))

    heads = 1

    def prepare(self):
        self.s.refresh()

    def small(self):
        """The dense half runs on the one-launch kernels of csrc/gat_small.hip (launch-bound graphs; option small_fused)."""
        return self.w.small_part is not None and ops.gat_small_supported(self.s.n, self.s.d, self.s.groups, self.heads)

    # ---- one adaptive step per C call (csrc/gat_driver.hip) -------------------------------------------------------
    def _structs(self, adjoint):
        s, w, eg = self.s, self.w, self.s.eg
        fs = _lib.GatOdeFunc()
        fs.mt = ops._edge_csr(eg, s.d + 4)
        for name, gph in (("ms_inc", eg.Ms_inc), ("mt_inc", eg.Mt_inc)):
            gs = _lib.Graph()
            gs.rowptr, gs.col, gs.val = gph.rowptr.data_ptr(), gph.col.data_ptr(), None
            gs.items, gs.n_items = (gph.items.data_ptr() if gph.items is not None else None), gph.n_items
            gs.long_rows = gph.long_rows.data_ptr() if gph.long_rows is not None else None
            gs.n_long = gph.n_long
            part = gph.partial(s.d) if adjoint else None
            gs.partial = part.data_ptr() if part is not None else None
            gs.n_rows, gs.nnz = gph.n_rows, gph.nnz
            setattr(fs, name, gs)
        p = lambda t: (t.data_ptr() or None) if t is not None else None      # noqa: E731  (struct fields take ints)
        fs.src, fs.tgt, fs.n_edges = p(eg.src), p(eg.tgt), eg.E
        fs.n, fs.d, fs.groups, fs.eps_gn, fs.eps = s.n, s.d, s.groups, s.eps_gn, s.eps
        fs.Wsrc, fs.Wtgt, fs.Wlog = s.Wsrc.data_ptr(), s.Wtgt.data_ptr(), s.Wlog.data_ptr()
        fs.bf, fs.bw, fs.gamma, fs.beta = s.bf.data_ptr(), s.bw.data_ptr(), s.gamma.data_ptr(), s.beta.data_ptr()
        fs.Wpacked = s.Wpacked.data_ptr() if s.Wpacked is not None else None
        ws = _lib.GatWorkspace()
        for k in ("X", "Ps", "Pt", "A2", "a", "amax", "wgt", "den", "logits_scratch"):
            setattr(ws, k, p(getattr(w, k)))
        if adjoint:
            for k in ("dz", "da", "dPs", "dPt", "dA2", "pair", "gp", "bp", "colsum_scratch", "colsum_scratch2"):
                setattr(ws, k, p(getattr(w, k)))
            for j in range(3):
                ws.wp[j] = w.wp[j].data_ptr()
            ws.maxpath_scratch = p(eg.maxpath_scratch())
            ws.small_part = p(w.small_part)
        return fs, ws

    def dopri5_step_native(self, y, kk, y1, t, h, rtol, atol):
        lib = _lib.load()
        fs, ws = self._structs(False)
        kptr = (ctypes.c_void_p * 7)(*[kk[i][0].data_ptr() for i in range(7)])
        sums = torch.empty(1, dtype=torch.float64, device=y[0].device)
        _lib.check(lib.gode_gat_ode_dopri5_step_forward(ctypes.byref(fs), _lib.ptr(y[0]), kptr, _lib.ptr(y1[0]), ctypes.byref(ws),
                                                        float(t), float(h), float(rtol), float(atol), _lib.ptr(sums),
                                                        _lib.ptr(self.w.err_scratch), _lib.stream_ptr()),
                   "gode_gat_ode_dopri5_step_forward")
        return sums

    def _project(self, t, y_terms):
        """Ps, Pt, A2 of the stage input; returns the term list later launches of the stage should read."""
        s, w = self.s, self.w
        x_out = w.X if len(y_terms) > 1 else None
        terms = [(1.0, w.X)] if x_out is not None else y_terms
        if self.small():
            ops.gat_project_small(y_terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wsrc, s.Wtgt, s.Wlog, 1, None, t,
                                  w.Ps, w.Pt, w.A2, x_out=x_out, packed=s.Wpacked)
            return terms
        ops.gn_time_gemm_pair(y_terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wsrc, s.Wtgt, True, t, w.Ps, w.Pt,
                              x_out=x_out)
        ops.gn_time_gemm(terms, s.n, s.d, s.groups, s.eps_gn, s.gamma, s.beta, s.Wlog, True, t, out=w.A2)
        return terms

    def _forward(self, t, y_terms, out):
        s, w, eg = self.s, self.w, self.s.eg
        terms = self._project(t, y_terms)
        ops.gat_logits(w.proj, s.bw, eg.src, eg.tgt, w.a, w.amax)
        ops.gat_agg_fwd(eg, w.proj, s.d, s.bf, w.a, w.amax, s.eps, out, w.wgt, w.den)
        return terms                 # the outer relu of ODEfunc is the identity on a weighted mean of relu's

    def eval(self, t, terms, out):
        self._forward(t, terms[0], out[0])


class GatOdeAdjointField(GatOdeField):
    """Components: [y, a, a_t, theta] with theta = [Wsrc | Wtgt | Wlog | bf | bw | gamma | beta]."""

    def __init__(self, spec, work, order):
        super().__init__(spec, work)
        self.order = order
        self.n_components = 4
        self.ratio_groups = [[0], [1], [2], [3]]

    def new_state(self, y_end):
        s = self.s
        return [y_end.clone(), torch.zeros_like(y_end), torch.zeros(1, dtype=torch.float32, device=y_end.device),
                torch.zeros(s.n_theta, dtype=torch.float32, device=y_end.device)]

    def param_grads(self, comps):
        s = self.s
        i = s.i
        v = s.views(comps[3])
        gWf = torch.cat([v["Wsrc"].t(), v["Wtgt"].t()], 1).contiguous()                          # o x 2i
        gww = torch.cat([v["Wlog"][:, 0], v["Wlog"][:, 1]]).view(1, 2 * i).contiguous()           # 1 x 2i
        m = {"gamma": v["gamma"].clone(), "beta": v["beta"].clone(), "Wf": gWf, "bf": v["bf"].clone(),
             "ww": gww, "bw": v["bw"].clone()}
        return [m[k] for k in self.order]

    def dopri5_step_native(self, y, kk, y1, t, h, rtol, atol):
        lib = _lib.load()
        fs, ws = self._structs(True)
        arr = lambda c: (ctypes.c_void_p * 7)(*[kk[i][c].data_ptr() for i in range(7)])      # noqa: E731
        sums = torch.empty(4, dtype=torch.float64, device=y[0].device)
        _lib.check(lib.gode_gat_ode_dopri5_step_adjoint(
            ctypes.byref(fs), _lib.ptr(y[0]), _lib.ptr(y[1]), _lib.ptr(y[2]), _lib.ptr(y[3]), arr(0), arr(1), arr(2), arr(3),
            _lib.ptr(y1[0]), _lib.ptr(y1[1]), _lib.ptr(y1[2]), _lib.ptr(y1[3]), ctypes.byref(ws), float(t), float(h),
            float(rtol), float(atol), _lib.ptr(sums), _lib.ptr(self.w.err_scratch), _lib.stream_ptr()),
            "gode_gat_ode_dopri5_step_adjoint")
        return sums

    # ---- fixed-grid steps on launch-bound graphs: the small components (a_t, theta) are advanced once per RK step ----
    deferred_components = (2, 3)
    DEFER_SMALL = True            # False: every stage closes its own partials (round 3)

    def begin_rk4_step(self):
        if not (self.DEFER_SMALL and self.small()):
            self._slots = None
            return False
        w = self.w
        if w.step_parts is None:
            w.step_parts = torch.empty(4 * w.small_part.numel(), dtype=torch.float32, device=w.small_part.device)
        self._slots = []              # evaluation times of the stages seen so far in this step
        return True

    def finish_rk4_step(self, weights, y):
        s, ts = self.s, self._slots
        self._slots = None
        ops.gat_small_finish_step(self.w.step_parts, s.n, s.d, self.heads, ts, weights[:len(ts)], y[3], y[2])
        return self.deferred_components

    def _stage_part(self, t):
        """Partial buffer of the stage being evaluated, and whether its closing launch is deferred to the end of the step."""
        slots = getattr(self, "_slots", None)
        if slots is None:
            return self.w.small_part, False
        k = len(slots)
        slots.append(t)
        n = self.w.small_part.numel()
        return self.w.step_parts[k * n:(k + 1) * n], True

    def eval(self, t, terms, out):
        s, w = self.s, self.w
        eg, n, o = s.eg, s.n, s.d
        xt = self._forward(t, terms[0], out[0])
        g = s.views(out[3])
        # cotangent -a of the VJP, masked by the outer relu, is formed inside the kernel
        ops.gat_vjp(eg, w.proj, o, s.bf, w.a, w.amax, w.wgt, w.den, out[0], w.dz, w.da, w.dPs, w.dPt, w.dA2,
                    cot_terms=terms[1], cot_scale=-1.0)
        if self.small():
            # launch-bound graphs: k_a and the partials of every parameter gradient in one launch, one more to close
            part, deferred = self._stage_part(t)
            ops.gat_dense_vjp_small(xt, n, o, s.groups, s.eps_gn, s.gamma, s.beta, s.Wsrc, s.Wtgt, s.Wlog, 1, w.dPs, w.dPt, w.dA2,
                                    out[1], part, packed=s.Wpacked)
            if not deferred:
                ops.gat_small_finish(part, n, o, 1, t, out[3], out[2])
            return
        # bias gradients: sum over edges of dz / da = sum over nodes of the per-target sums just formed (every edge has
        # exactly one target) - N rows instead of E
        merged = n <= MERGED_FINISH_MAX_ROWS and s.groups > 0        # launch-bound: ONE reduction launch closes the stage
        if merged:
            n_a = ops.colsum_parts(w.dPt, w.colsum_scratch)
            n_b = ops.colsum_parts(w.dA2, w.colsum_scratch2)
        else:
            ops.colsum_(g["bf"], w.dPt)
            ops.colsum_(g["bw"], w.dA2[:, 1:2].contiguous())
        nb = w.np_b
        affine = s.groups > 0
        for j, (Wj, dPj) in enumerate(((s.Wsrc, w.dPs), (s.Wtgt, w.dPt), (s.Wlog, w.dA2))):
            ops.gn_time_gemm_bwd(xt, n, o, s.groups, s.eps_gn, s.gamma, Wj, True, dPj, out=out[1],
                                 pre_terms=[(1.0, out[1])] if j else None,
                                 parts=(w.gp[j * nb:(j + 1) * nb], w.bp[j * nb:(j + 1) * nb]) if affine else None)
        if affine and not merged:
            ops.reduce_parts2_(g["gamma"], w.gp, g["beta"], w.bp)
        elif not affine:
            g["gamma"].zero_(); g["beta"].zero_()
        for j, dPj in enumerate((w.dPs, w.dPt, w.dA2)):
            ops.wgrad(xt, n, o, s.groups, s.eps_gn, s.gamma, s.beta, dPj, True, part=w.wp[j])
        if merged:
            i, npw = s.i, w.wp[0].shape[0]
            ops.reduce_segments_([
                (g["Wsrc"], w.wp[0], npw, i * o, 0, 1, i * o, s.Wsrc[0], o),          # row 0 of each block = its time row
                (g["Wtgt"], w.wp[1], npw, i * o, 0, 1, i * o, s.Wtgt[0], o),
                (g["Wlog"], w.wp[2], npw, i * 2, 0, 1, i * 2, s.Wlog[0], 2),
                (g["bf"], w.colsum_scratch, n_a, o, 0, 1, o, None, 0),
                (g["bw"], w.colsum_scratch2, n_b, 2, 1, 2, 1, None, 0),               # column 1 of the n x 2 sums
                (g["gamma"], w.gp, 3 * nb, o, 0, 1, o, None, 0),
                (g["beta"], w.bp, 3 * nb, o, 0, 1, o, None, 0)], t, out[2])
            return
        ops.reduce_parts2_(g["Wsrc"].view(-1), w.wp[0], g["Wtgt"].view(-1), w.wp[1])
        ops.reduce_parts_(g["Wlog"].view(-1), w.wp[2])
        # a_t' = -a^T df/dt over the three time rows; each row 0 *= t
        ops.time_row_fixup3_([g["Wsrc"][0], g["Wtgt"][0], g["Wlog"][0]], [s.Wsrc[0], s.Wtgt[0], s.Wlog[0]], t, out[2])


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
    work = getattr(eg, "_ode_work", {}).get((spec.d, y0.device))
    if work is None:
        work = _Work(spec, y0.device)
        eg.__dict__.setdefault("_ode_work", {})[(spec.d, y0.device)] = work
    order = [names[id(p)] for p in plist]
    return GatOdeField(spec, work), (lambda: GatOdeAdjointField(spec, work, order)), tuple(plist)
