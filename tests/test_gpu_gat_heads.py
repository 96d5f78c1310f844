"""H-head edge attention (graph_odenet_amd/gat_heads.py; BASELINE.json configs[2] "Citeseer GAT 8-head ODEBlock rk4"):
the product path through the C ABI against H instances of the reference's layer (golden), and against the oracle
solver driving the oracle H-head ODE function on Citeseer's real edge list."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def dev():
    return torch.device("cuda:0")


def close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), (b.detach().cpu() if torch.is_tensor(b) else T(b)).double()
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), "%s: max abs err %.3e" % (what, err)


def _load_heads(layer, g, H):
    for h in range(H):
        hd = layer.heads[h]
        hd.f.weight.data.copy_(T(g["h%d__f__weight" % h])); hd.f.bias.data.copy_(T(g["h%d__f__bias" % h]))
        hd.w.weight.data.copy_(T(g["h%d__w__weight" % h])); hd.w.bias.data.copy_(T(g["h%d__w__bias" % h]))


def test_multihead_layer_vs_reference_golden(golden):
    from graph_odenet_amd.gat_heads import MultiHeadGraphConvolution
    g = golden("gat_heads.npz")
    n, H = int(g["n"]), 4
    src, tgt = T(g["src"]).long(), T(g["tgt"]).long()
    E = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    layer = MultiHeadGraphConvolution(12, 20, heads=H)
    _load_heads(layer, g, H)
    layer = layer.to(dev())
    x = T(g["x"]).to(dev()).requires_grad_(True)
    out = layer(x, src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    close(out, g["out"], 1e-5, "out")
    out.backward(T(g["gout"]).to(dev()))
    close(x.grad, g["gx"], 1e-5, "gx")
    for h in range(H):
        hd = layer.heads[h]
        for p, k in ((hd.f.weight, "f__weight"), (hd.f.bias, "f__bias"), (hd.w.weight, "w__weight"), (hd.w.bias, "w__bias")):
            close(p.grad, g["h%d__grad__%s" % (h, k)], 1e-5, "head %d %s" % (h, k))


def test_one_head_equals_the_single_head_layer(golden):
    from graph_odenet_amd.gat_heads import MultiHeadGraphConvolution
    from graph_odenet_amd.gat_layers import GraphConvolution
    g = golden("gat_heads.npz")
    n = int(g["n"])
    src, tgt = T(g["src"]).long().to(dev()), T(g["tgt"]).long().to(dev())
    E = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E, device=dev())]), torch.ones(E, device=dev()), (n, E))
    torch.manual_seed(0)
    one = GraphConvolution(12, 7).to(dev())
    multi = MultiHeadGraphConvolution(12, 7, heads=1).to(dev())
    multi.heads[0].load_state_dict(one.state_dict())
    x = T(g["x"]).to(dev())
    close(multi(x, src, tgt, Mtgt), one(x, src, tgt, Mtgt), 1e-6, "H=1")


def _citeseer(golden):
    ge = golden("citeseer_gat_edges.npz")
    n = int(ge["n"])
    src, tgt = T(ge["src"]).long(), T(ge["tgt"]).long()
    Mtgt = torch.sparse_coo_tensor(torch.stack([T(ge["m_rows"]).long(), T(ge["m_cols"]).long()]), T(ge["m_vals"]),
                                   (n, src.numel()))
    return n, src, tgt, Mtgt


@pytest.mark.parametrize("method,kw,small", [("rk4", dict(step_size=0.25), False), ("dopri5", {}, True)])
def test_eight_head_ode_block_vs_oracle_on_citeseer_edges(golden, method, kw, small):
    """configs[2]: Citeseer's edge list, 8 heads x 16 features, ODEBlock; forward and parameter gradients against
    the oracle's adjoint solver driving the oracle heads (same algorithm end to end)."""
    from graph_odenet_amd.gat_heads import ODEfunc
    from graph_odenet_amd.models import ODEBlock
    from oracle import layers_ref as R, solver_ref as S
    if small:                 # the adaptive solver on the 50-node golden graph: the CPU oracle takes every step too
        g = golden("gat_heads.npz")
        n, src, tgt = int(g["n"]), T(g["src"]).long(), T(g["tgt"]).long()
        Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(src.numel())]), torch.ones(src.numel()), (n, src.numel()))
        d, H, tol = 32, 4, 1e-5
    else:
        n, src, tgt, Mtgt = _citeseer(golden)
        d, H, tol = 128, 8, 1e-6
    torch.manual_seed(5)
    blk = ODEBlock(ODEfunc(d, H), method=method, tol=tol, **kw)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    x0 = torch.randn(n, d) * 0.5

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gn = torch.nn.ParameterList([torch.nn.Parameter(sd["odefunc.norm1." + k].clone()) for k in ("weight", "bias")])
            self.hp = torch.nn.ParameterList([torch.nn.Parameter(sd["odefunc.gc1.heads.%d.%s" % (h, k)].clone())
                                              for h in range(H) for k in ("f.weight", "f.bias", "w.weight", "w.bias")])

        def forward(self, t, x):
            heads = [list(self.hp[4 * h:4 * h + 4]) for h in range(H)]
            return R.gat_multihead_odefunc(t, x, src, tgt, Mtgt, self.gn[0], self.gn[1], heads)
    f = F()
    opts = dict(options={"step_size": 0.25}) if method == "rk4" else {}
    ref = S.odeint_adjoint(f, x0, torch.tensor([0., 1.]), tol, tol, method, opts.get("options"))[1]
    gout = torch.randn(n, d, generator=torch.Generator().manual_seed(6))
    ref.backward(gout)
    blk = blk.to(dev())
    xg = x0.to(dev()).requires_grad_(True)
    out = blk(xg, src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    close(out, ref, 2e-5, "8-head %s forward" % method)
    out.backward(gout.to(dev()))
    gtol = 2e-4 if method == "rk4" else 5e-3      # adaptive steps: accept / reject decisions in fp32 on both sides
    close(blk.odefunc.norm1.weight.grad, f.gn[0].grad, gtol, "dgamma")
    close(blk.odefunc.norm1.bias.grad, f.gn[1].grad, gtol, "dbeta")
    for h in (0, H // 2, H - 1):
        hd = blk.odefunc.gc1.heads[h]
        for j, p in enumerate((hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias)):
            close(p.grad, f.hp[4 * h + j].grad, gtol, "head %d param %d" % (h, j))
    # a second and third call replay the captured solve (rk4): same numbers
    if method == "rk4":
        for _ in range(2):
            out2 = blk(xg, src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
        close(out2, out, 1e-6, "captured replay")


@pytest.mark.parametrize("big", [False, True])
def test_fused_fields_agree_with_autograd_through_the_layer(golden, big):
    """The fused adjoint (kernel sequence on packed parameters) against autograd through ODEfunc.forward; `big`: from
    4096 nodes on the 2H logit columns ride the square MFMA kernels as a zero-padded block."""
    from graph_odenet_amd.gat_heads import ODEfunc
    from graph_odenet_amd.models import ODEBlock
    if big:
        n, E = 5000, 24000
        g = torch.Generator().manual_seed(12)
        src, tgt = torch.randint(0, n, (E,), generator=g), torch.randint(0, n, (E,), generator=g)
        Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
        d, H = 32, 8
    else:
        n, src, tgt, Mtgt = _citeseer(golden)
        d, H = 32, 4
    src, tgt, Mtgt = src.to(dev()), tgt.to(dev()), Mtgt.to(dev())
    torch.manual_seed(8)
    blk = ODEBlock(ODEfunc(d, H), method="rk4", step_size=0.5).to(dev())
    x = torch.randn(n, d, device=dev())
    gout = torch.randn(n, d, device=dev())
    out = blk(x.clone().requires_grad_(True), src, tgt, Mtgt)
    out.backward(gout)
    fused = [p.grad.clone() for p in blk.parameters()]
    assert blk.odefunc.gode_fields(x)[0].s.pad_logits == big
    for p in blk.parameters():
        p.grad = None
    type(blk.odefunc).gode_fields_saved = type(blk.odefunc).gode_fields
    try:
        type(blk.odefunc).gode_fields = lambda self, y0: None        # generic autograd fields
        out2 = blk(x.clone().requires_grad_(True), src, tgt, Mtgt)
        out2.backward(gout)
    finally:
        type(blk.odefunc).gode_fields = type(blk.odefunc).gode_fields_saved
        del type(blk.odefunc).gode_fields_saved
    close(out2, out, 1e-5, "forward")
    for (nm, p), gf in zip(blk.named_parameters(), fused):
        close(p.grad, gf, 1e-4, nm)


def _finite_grads(model, step):
    for nm, p in model.named_parameters():
        assert p.grad is not None, "step %d: %s received no gradient" % (step, nm)
        assert bool(torch.isfinite(p.grad).all()), "step %d: non-finite gradient in %s" % (step, nm)


def test_eight_head_model_trains_on_citeseer_edges(golden):
    """Eight Adam steps of the 8-head ODEGCN3 (step 2 captures the solves, 3.. replay them): every parameter gradient
    finite at every step.  Round 1's red run: replayed memset nodes left stale GroupNorm-gradient partials
    (DESIGN.md section 2); the library launches no memset any more (tests/test_abi.py checks the sources)."""
    from graph_odenet_amd import gat_heads
    n, src, tgt, Mtgt = _citeseer(golden)
    src, tgt, Mtgt = src.to(dev()), tgt.to(dev()), Mtgt.to(dev())
    zoo = gat_heads.zoo(8)
    torch.manual_seed(1)
    m = zoo.ODEGCN3(nfeat=50, nhid=64, nclass=6, dropout=0.0, method="rk4", step_size=0.25).to(dev())
    assert type(m.gc1).__name__ == "MultiHeadGraphConvolution" and type(m.gc3).__name__ == "GraphConvolution"
    x = torch.randn(n, 50, device=dev())
    y = torch.randint(0, 6, (n,), device=dev())
    opt = torch.optim.Adam(m.parameters(), lr=0.01)
    losses = []
    for step in range(8):
        opt.zero_grad()
        loss = torch.nn.functional.nll_loss(m(x, src, tgt, Mtgt), y)
        loss.backward()
        _finite_grads(m, step)
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("d,H", [(64, 8), (32, 4)])
def test_captured_heads_solves_match_eager(golden, d, H):
    """The H-head counterpart of test_gpu_gcn.py::test_hip_graph_captured_solves_match_eager: five optimiser steps
    with HIP-graph capture on (step 2 captures forward and adjoint, steps 3-5 replay with moved parameters) and off;
    outputs, input gradients and every parameter gradient agree bit for bit, step by step.  d = 64, H = 8 puts the 2H
    logit columns on the generic dense kernels (the path whose memsets broke the replays in round 1)."""
    from graph_odenet_amd import odeint as OI
    from graph_odenet_amd.gat_heads import ODEfunc
    from graph_odenet_amd.models import ODEBlock
    n, src, tgt, Mtgt = _citeseer(golden)
    src, tgt, Mtgt = src.to(dev()), tgt.to(dev()), Mtgt.to(dev())
    g = torch.Generator().manual_seed(21)
    xs = [torch.randn(n, d, generator=g).to(dev()) for _ in range(5)]
    res = {}
    old = OI.GRAPH_CAPTURE_MAX_ELEMS
    for capture in (True, False):
        OI.GRAPH_CAPTURE_MAX_ELEMS = old if capture else 0
        try:
            torch.manual_seed(4)
            blk = ODEBlock(ODEfunc(d, H), method="rk4", step_size=0.25).to(dev())
            opt = torch.optim.SGD(blk.parameters(), lr=0.02)
            log = []
            for step, x in enumerate(xs):
                opt.zero_grad()
                xi = x.clone().requires_grad_(True)
                out = blk(xi, src, tgt, Mtgt)
                out.square().mean().backward()
                _finite_grads(blk, step)
                log.append((out.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()]))
                opt.step()
            res[capture] = log
            if capture:
                plans = list(OI.plans_of(blk.odefunc).values())
                assert len(plans) == 1 and plans[0].gf is not None and plans[0].gb is not None
        finally:
            OI.GRAPH_CAPTURE_MAX_ELEMS = old
    names = [nm for nm, _ in blk.named_parameters()]
    for step, ((o1, g1, p1), (o2, g2, p2)) in enumerate(zip(res[True], res[False])):
        assert torch.equal(o1, o2), "step %d: outputs differ" % step
        assert torch.equal(g1, g2), "step %d: input gradients differ" % step
        for nm, a, b in zip(names, p1, p2):
            assert torch.equal(a, b), "step %d: gradient of %s differs (max %.3e)" % (step, nm, (a - b).abs().max().item())


def test_heads_on_the_record_path():
    """Above 65 536 virtual targets the aggregation runs on the nnz-balanced record kernels (target sums formed in
    the kernel, per-head maximum correction applied to them): against the fp64 oracle."""
    from graph_odenet_amd.gat_heads import MultiHeadGraphConvolution
    from oracle import layers_ref as R
    n, E, H, nin, o = 20000, 90000, 4, 16, 16
    g = torch.Generator().manual_seed(2)
    src = torch.randint(0, n, (E,), generator=g)
    tgt = torch.cat([torch.randint(0, n, (E - 3000,), generator=g), torch.zeros(3000, dtype=torch.int64)])   # one hub
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    torch.manual_seed(3)
    layer = MultiHeadGraphConvolution(nin, H * o, heads=H)
    x = torch.randn(n, nin, generator=g)
    gout = torch.randn(n, H * o, generator=g)
    xd = x.double().requires_grad_(True)
    heads = [[p.detach().double().requires_grad_(True) for p in (hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias)]
             for hd in layer.heads]
    ref = R.gat_multihead_layer(xd, src, tgt, Mtgt.double(), heads)
    ref.backward(gout.double())
    layer = layer.to(dev())
    xg = x.to(dev()).requires_grad_(True)
    out = layer(xg, src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    close(out, ref, 1e-5, "out")
    out.backward(gout.to(dev()))
    close(xg.grad, xd.grad, 2e-5, "gx")
    for h in range(H):
        hd = layer.heads[h]
        for p, q in zip((hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias), heads[h]):
            close(p.grad, q.grad, 1e-4, "head %d" % h)


@pytest.mark.parametrize("H,o,n,E,weighted", [(3, 4, 287, 2861, False), (8, 2, 163, 2934, True), (16, 16, 233, 1414, False),
                                              (5, 3, 18, 202, True), (1, 1, 16, 1058, True), (16, 1, 13, 567, False)])
def test_head_counts_and_widths_vs_fp64_oracle(H, o, n, E, weighted):
    """Head counts that are not powers of two, one-feature heads, weighted Mtgt entries, nodes without edges; edge
    lists on both sides of the single-block threshold (tools/dev/fuzz_heads.py runs many more)."""
    from graph_odenet_amd.gat_heads import MultiHeadGraphConvolution
    from oracle import layers_ref as R
    g = torch.Generator().manual_seed(H * 1000 + o)
    nin = 7
    src, tgt = torch.randint(0, n, (E,), generator=g), torch.randint(0, max(1, n - 2), (E,), generator=g)
    vals = torch.rand(E, generator=g) + 0.25 if weighted else torch.ones(E)
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), vals, (n, E))
    torch.manual_seed(o)
    layer = MultiHeadGraphConvolution(nin, H * o, heads=H)
    with torch.no_grad():
        for hd in layer.heads:
            hd.w.bias.add_(torch.randn(1) * 3)
    x, gout = torch.randn(n, nin, generator=g), torch.randn(n, H * o, generator=g)
    xd = x.double().requires_grad_(True)
    heads = [[p.detach().double().requires_grad_(True) for p in (hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias)]
             for hd in layer.heads]
    ref = R.gat_multihead_layer(xd, src, tgt, Mtgt.double(), heads)
    ref.backward(gout.double())
    layer = layer.to(dev())
    xg = x.to(dev()).requires_grad_(True)
    out = layer(xg, src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    out.backward(gout.to(dev()))
    close(out, ref, 1e-5, "out")
    close(xg.grad, xd.grad, 2e-5, "gx")
    for hd, ps in zip(layer.heads, heads):
        for p, q in zip((hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias), ps):
            close(p.grad, q.grad, 2e-5, "param")


@pytest.mark.parametrize("act_name", ["tanh", "elu", "identity"])
@pytest.mark.parametrize("H", [1, 4])
def test_non_relu_activation_vs_fp64_oracle(act_name, H):
    """The `act=` constructor argument of the layer (GAT/layers.py:16) with something other than relu: the general path
    (messages as the reference forms them, the two per-target sums on the SpMM kernel), one head and H heads, forward
    and every gradient against the fp64 oracle."""
    import torch.nn.functional as F
    from graph_odenet_amd.gat_heads import MultiHeadGraphConvolution
    from graph_odenet_amd.gat_layers import GraphConvolution
    from oracle import layers_ref as R
    act = {"tanh": torch.tanh, "elu": F.elu, "identity": (lambda v: v)}[act_name]
    g = torch.Generator().manual_seed(31 + H)
    n, E, nin, o = 211, 1900, 9, 6
    src, tgt = torch.randint(0, n, (E,), generator=g), torch.randint(0, n - 3, (E,), generator=g)
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    torch.manual_seed(H)
    layer = GraphConvolution(nin, o, act=act) if H == 1 else MultiHeadGraphConvolution(nin, H * o, heads=H, act=act)
    hds = [layer] if H == 1 else list(layer.heads)
    x, gout = torch.randn(n, nin, generator=g), torch.randn(n, H * o, generator=g)
    xd = x.double().requires_grad_(True)
    ps = [[p.detach().double().requires_grad_(True) for p in (hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias)] for hd in hds]
    ref = torch.cat([R.gat_layer(xd, src, tgt, Mtgt.double(), *p4, act=act) for p4 in ps], 1)
    ref.backward(gout.double())
    layer = layer.to(dev())
    xg = x.to(dev()).requires_grad_(True)
    out = layer(xg, src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    close(out, ref, 1e-5, "out")
    out.backward(gout.to(dev()))
    close(xg.grad, xd.grad, 2e-5, "gx")
    for hd, p4 in zip(hds, ps):
        for p, q in zip((hd.f.weight, hd.f.bias, hd.w.weight, hd.w.bias), p4):
            close(p.grad, q.grad, 2e-5, "param")


def test_multihead_odefunc2_and_odek2(golden):
    """The two-layer ODE function with H-head layers (the GAT variant's ODEfunc2, GAT/models.py:551-575) and the ODEK2
    model built on it: against the oracle heads composed the same way, and one training step of the zoo's ODEK2."""
    from graph_odenet_amd import gat_heads
    from oracle import layers_ref as R
    g = golden("gat_heads.npz")
    n, src, tgt = int(g["n"]), T(g["src"]).long(), T(g["tgt"]).long()
    E = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    d, H = 128, 4                      # 4 channels per GroupNorm group: well conditioned
    torch.manual_seed(9)
    f = gat_heads.ODEfunc2(d, 0.0, heads=H)
    x = torch.randn(n, d)
    t = 0.3

    def heads_of(layer):
        return [[hd.f.weight.detach().double(), hd.f.bias.detach().double(), hd.w.weight.detach().double(),
                 hd.w.bias.detach().double()] for hd in layer.heads]

    def gn(v, norm):
        return torch.nn.functional.group_norm(v, norm.num_groups, norm.weight.detach().double(), norm.bias.detach().double(), norm.eps)
    xd = x.double()
    tt = torch.full((n, 1), t, dtype=torch.float64)
    h1 = gn(torch.relu(R.gat_multihead_layer(torch.cat([tt, xd], 1), src, tgt, Mtgt.double(), heads_of(f.gc1))), f.norm1)
    ref = gn(torch.relu(R.gat_multihead_layer(torch.cat([tt, h1], 1), src, tgt, Mtgt.double(), heads_of(f.gc2))), f.norm2)
    f = f.to(dev())
    f.set_adj(src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    out = f(torch.tensor(t), x.to(dev()))
    # GroupNorm after relu: groups whose four messages are all near zero are normalised with rstd up to 1/sqrt(eps) =
    # 316, which amplifies the 1e-6 rounding of the layer output (measured 4.5e-5 on outputs of magnitude 1.7)
    close(out, ref, 1e-4, "ODEfunc2 with heads")
    zoo = gat_heads.zoo(H)
    torch.manual_seed(2)
    m = zoo.ODEK2(nfeat=12, nhid=d, nclass=5, dropout=0.3, nlayers=4, method="rk4", step_size=0.5).to(dev())
    assert type(m.gcs[1].odefunc).__name__ == "ODEfunc2" and type(m.gcs[1].odefunc.gc1).__name__ == "FixedMultiHeadGraphConvolution"
    xin = torch.randn(n, 12, device=dev())
    y = torch.randint(0, 5, (n,), device=dev())
    loss = torch.nn.functional.nll_loss(m(xin, src.to(dev()), tgt.to(dev()), Mtgt.to(dev())), y)
    loss.backward()
    _finite_grads(m, 0)


@pytest.mark.parametrize("d,H", [(32, 4), (64, 8)])
def test_native_dopri5_step_for_heads_matches_python_driver(golden, d, H):
    """The H-head field's adaptive step as ONE C call (csrc/gat_driver.hip, heads = H) against the per-stage Python
    driver: same kernels in the same order, so outputs, gradients and nfe agree bit for bit."""
    from graph_odenet_amd import solver as PS
    from graph_odenet_amd.gat_heads import ODEfunc
    from graph_odenet_amd.models import ODEBlock
    n, src, tgt, Mtgt = _citeseer(golden)
    src, tgt, Mtgt = src.to(dev()), tgt.to(dev()), Mtgt.to(dev())
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(3)).to(dev()) * 0.5
    gout = torch.randn(n, d, generator=torch.Generator().manual_seed(4)).to(dev())
    res = {}
    for native in (True, False):
        PS.DOPRI5_NATIVE = native
        try:
            torch.manual_seed(6)
            blk = ODEBlock(ODEfunc(d, H), tol=1e-4).to(dev())
            xi = x.clone().requires_grad_(True)
            out = blk(xi, src, tgt, Mtgt)
            nfe_f = blk.nfe
            out.backward(gout)
            res[native] = (out.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()], nfe_f, blk.nfe)
            fld = blk.odefunc.gode_fields(x)[0]
            assert (fld.dopri5_step_native is not None) and not fld.s.pad_logits
        finally:
            PS.DOPRI5_NATIVE = True
    assert res[True][3:] == res[False][3:]
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n", [1, 37, 3327, 4500])
@pytest.mark.parametrize("d,H", [(16, 1), (32, 1), (64, 1), (16, 2), (32, 4), (64, 8), (16, 8), (64, 3)])
def test_one_launch_dense_half_vs_float64_autograd(n, d, H):
    """csrc/gat_small.hip (launch-bound graphs): the node-level products of GAT/layers.py:43,45 behind GroupNorm and the
    time column (GAT/models.py:175-177) as ONE launch - Ps, Pt (+ per-head bias), A2 - and their whole autograd as one
    launch plus the closing reduction: k_a, dWsrc, dWtgt, dWlog (time rows scaled by t), bf = colsum(dPt), bw = odd
    columns of colsum(dA2), dgamma, dbeta, and a_t' = colsums . time rows.  Against float64 autograd; a multi-term stage
    input is combined in the kernel and written out."""
    from graph_odenet_amd import ops
    if not ops.gat_small_supported(n, d, min(32, d), H):
        pytest.skip("shape outside the one-launch kernels")
    g = torch.Generator().manual_seed(n + 7 * d + H)
    groups, eps, t = min(32, d), 1e-5, 0.37
    xa, xb = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
    terms = [(1.0, xa), (0.25, xb)]
    gamma, beta = 1 + 0.3 * torch.randn(d, generator=g), 0.3 * torch.randn(d, generator=g)
    Wsrc, Wtgt = torch.randn(d + 1, d, generator=g) / d ** 0.5, torch.randn(d + 1, d, generator=g) / d ** 0.5
    Wlog, bf = torch.randn(d + 1, 2 * H, generator=g) / d ** 0.5, torch.randn(d, generator=g)
    dPs, dPt, dA2 = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g), torch.randn(n, 2 * H, generator=g)
    pre = torch.randn(n, d, generator=g)
    # float64 reference
    P = [v.double().requires_grad_(True) for v in (xa + 0.25 * xb, gamma, beta, Wsrc, Wtgt, Wlog)]
    x64, g64, b64, ws64, wt64, wl64 = P
    xg = x64.view(n, groups, d // groups)                         # GroupNorm(groups, d) on n x d (GAT/models.py:165,175)
    xn = ((xg - xg.mean(2, keepdim=True)) / torch.sqrt(xg.var(2, unbiased=False, keepdim=True) + eps)).reshape(n, d) * g64 + b64
    ttx = torch.cat([torch.full((n, 1), t, dtype=torch.float64), xn], 1)
    rPs, rPt, rA2 = ttx @ ws64, ttx @ wt64 + bf.double(), ttx @ wl64
    ((rPs * dPs.double()).sum() + (rPt * dPt.double()).sum() + (rA2 * dA2.double()).sum()).backward()
    D = dev()
    f = dict(dtype=torch.float32, device=D)
    dt = [(c, v.to(D)) for c, v in terms]
    Ps, Pt, A2, X = torch.empty(n, d, **f), torch.empty(n, d, **f), torch.empty(n, 2 * H, **f), torch.empty(n, d, **f)
    ops.gat_project_small(dt, n, d, groups, eps, gamma.to(D), beta.to(D), Wsrc.to(D), Wtgt.to(D), Wlog.to(D), H, bf.to(D), t, Ps, Pt, A2,
                          x_out=X)
    assert torch.equal(X.cpu(), torch.addcmul(xa, xb, torch.tensor(0.25))) or (X.cpu() - (xa + 0.25 * xb)).abs().max().item() <= 1e-6
    # conditioning (as tests/test_gpu_kernels.py::test_gn_time_gemm_fwd_bwd_wgrad): one channel per group
    # -> GroupNorm returns beta + rounding noise * 316 and its x-gradient is 0 in real arithmetic; two channels per group
    # -> rstd reaches 316 on rows whose two values nearly coincide
    cg = d // groups
    ftol, xtol, wtol = {1: (2e-4, 2e-3, 2e-4), 2: (2e-5, 1e-4, 2e-5)}.get(cg, (3e-6, 2e-5, 4e-6))
    for got, ref, nm in ((Ps, rPs, "Ps"), (Pt, rPt, "Pt"), (A2, rA2, "A2")):
        assert (got.cpu().double() - ref.detach()).abs().max().item() <= ftol * max(1.0, ref.abs().max().item()), nm
    part = ops.gat_small_part(n, d, H, D)
    part.fill_(float("nan"))                                   # every entry that is read must have been written
    ka = torch.empty(n, d, **f)
    ops.gat_dense_vjp_small([(1.0, X)], n, d, groups, eps, gamma.to(D), beta.to(D), Wsrc.to(D), Wtgt.to(D), Wlog.to(D), H,
                            dPs.to(D), dPt.to(D), dA2.to(D), ka, part, out_scale=-0.5, pre_terms=[(2.0, pre.to(D))])
    want_ka = -0.5 * x64.grad + 2.0 * pre.double()
    assert (ka.cpu().double() - want_ka).abs().max().item() <= xtol * max(1.0, want_ka.abs().max().item())
    nth = 2 * (d + 1) * d + (d + 1) * 2 * H + d + H + 2 * d
    kth, kat = torch.full((nth,), float("nan"), **f), torch.full((1,), float("nan"), **f)
    ops.gat_small_finish(part, n, d, H, t, kth, kat)
    nW, nL = (d + 1) * d, (d + 1) * 2 * H
    cs = [dPs.double().sum(0), dPt.double().sum(0), dA2.double().sum(0)]
    want = torch.cat([ws64.grad.reshape(-1), wt64.grad.reshape(-1), wl64.grad.reshape(-1), cs[1], cs[2][1::2], g64.grad, b64.grad])
    got = kth.cpu().double()
    assert got.shape == want.shape
    i_gamma = 2 * nW + nL + d + H                          # dgamma multiplies the cotangent by the normalised (noisy) value
    for lo, hi, tol, nm in ((0, i_gamma, wtol, "weights and biases"), (i_gamma, i_gamma + d, 2e-3 if cg == 1 else 2e-5, "dgamma"),
                            (i_gamma + d, nth, 2e-5, "dbeta")):
        scale = max(1.0, want[lo:hi].abs().max().item())
        bad = (got[lo:hi] - want[lo:hi]).abs().max().item()
        assert bad <= tol * scale * max(1.0, n ** 0.5), (nm, bad, scale)
    # the same launches with the weights' LDS images formed beforehand (gode_gat_small_pack_f32: what the ODE fields do once
    # per solve from d = 32 on): a different way of staging the same numbers - every output bit for bit the same
    packed = ops.gat_small_pack(Wsrc.to(D), Wtgt.to(D), Wlog.to(D), H)
    Ps2, Pt2, A22 = torch.empty_like(Ps), torch.empty_like(Pt), torch.empty_like(A2)
    X2 = torch.full_like(X, float("nan"))
    ops.gat_project_small(dt, n, d, groups, eps, gamma.to(D), beta.to(D), Wsrc.to(D), Wtgt.to(D), Wlog.to(D), H, bf.to(D), t, Ps2, Pt2, A22,
                          x_out=X2, packed=packed)
    assert torch.equal(X2, X)
    if d < 64:
        assert torch.equal(Ps2, Ps) and torch.equal(Pt2, Pt) and torch.equal(A22, A2)
    else:       # d = 64 with the packed image: 16-row tiles on the fp32 matrix instruction (gat_project_d64_kernel) - same bars
        for got, ref, nm in ((Ps2, rPs, "Ps"), (Pt2, rPt, "Pt"), (A22, rA2, "A2")):
            assert (got.cpu().double() - ref.detach()).abs().max().item() <= ftol * max(1.0, ref.abs().max().item()), (nm, "matrix-instruction form")
    part2, ka2 = ops.gat_small_part(n, d, H, D), torch.empty_like(ka)
    ops.gat_dense_vjp_small([(1.0, X)], n, d, groups, eps, gamma.to(D), beta.to(D), Wsrc.to(D), Wtgt.to(D), Wlog.to(D), H,
                            dPs.to(D), dPt.to(D), dA2.to(D), ka2, part2, out_scale=-0.5, pre_terms=[(2.0, pre.to(D))], packed=packed)
    kth2, kat2 = torch.full_like(kth, float("nan")), torch.full_like(kat, float("nan"))
    ops.gat_small_finish(part2, n, d, H, t, kth2, kat2)
    want_at = (cs[0] * Wsrc[0].double()).sum() + (cs[1] * Wtgt[0].double()).sum() + (cs[2] * Wlog[0].double()).sum()
    assert abs(kat.item() - want_at.item()) <= 4e-6 * max(1.0, abs(want_at.item())) * max(1.0, n ** 0.5)
    if d < 64:
        assert torch.equal(ka2, ka) and torch.equal(kth2, kth) and torch.equal(kat2, kat)
    else:
        # d = 64 with the packed images: tiles of 16 rows on the fp32 matrix instruction (gat_dense_vjp_d64_kernel; n = 4500:
        # more tiles than blocks) - another summation order, the same bars against float64
        assert (ka2.cpu().double() - want_ka).abs().max().item() <= xtol * max(1.0, want_ka.abs().max().item())
        got2 = kth2.cpu().double()
        for lo, hi, tol, nm in ((0, i_gamma, wtol, "weights and biases"), (i_gamma, i_gamma + d, 2e-3 if cg == 1 else 2e-5, "dgamma"),
                                (i_gamma + d, nth, 2e-5, "dbeta")):
            scale = max(1.0, want[lo:hi].abs().max().item())
            bad = (got2[lo:hi] - want[lo:hi]).abs().max().item()
            assert bad <= tol * scale * max(1.0, n ** 0.5), (nm, "matrix-instruction form", bad, scale)
        assert abs(kat2.item() - want_at.item()) <= 4e-6 * max(1.0, abs(want_at.item())) * max(1.0, n ** 0.5)


@pytest.mark.parametrize("heads,d", [(1, 16), (1, 64), (8, 64), (4, 32)])
@pytest.mark.parametrize("method", ["rk4", "dopri5"])
def test_one_launch_dense_half_matches_multi_launch_path(golden, heads, d, method):
    """The same ODE block with the option small_fused off (three products, three VJPs, three weight-gradient launches,
    two column sums per stage) and on (csrc/gat_small.hip), on Citeseer's edge list: forward and every parameter gradient."""
    from graph_odenet_amd import _lib, gat_heads, gat_models
    from graph_odenet_amd.models import ODEBlock
    lib = _lib.load()
    n, src, tgt, Mtgt = _citeseer(golden)
    D = dev()
    src, tgt, Mtgt = src.to(D), tgt.to(D), Mtgt.to(D)
    x0 = (torch.randn(n, d, generator=torch.Generator().manual_seed(3)) * 0.5).to(D)
    gout = torch.randn(n, d, generator=torch.Generator().manual_seed(4)).to(D)
    res = {}
    try:
        for fused in (0, 1):
            lib.gode_set_option(b"small_fused", fused)
            torch.manual_seed(9)
            fn = gat_models.ODEfunc(d) if heads == 1 else gat_heads.ODEfunc(d, heads)
            kw = dict(method="rk4", step_size=0.25) if method == "rk4" else dict(method="dopri5", tol=1e-5)
            blk = ODEBlock(fn, **kw).to(D)
            xg = x0.clone().requires_grad_(True)
            out = blk(xg, src, tgt, Mtgt)
            out.backward(gout)
            res[fused] = (out.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    finally:
        lib.gode_set_option(b"small_fused", 1)
    ftol = 1e-5 if method == "rk4" else 1e-4
    gtol = 1e-4 if method == "rk4" else 5e-3            # adaptive: accept / reject decisions move with rounding
    close(res[1][0], res[0][0], 2e-3 if d == 64 else ftol, "forward")
    if d == 64:
        # two channels per GroupNorm group: rstd reaches 316 on rows whose two values nearly coincide, and the summation
        # order of the products (MFMA tiles vs sub-group sums) then shows at 1e-4 in the forward pass - BOTH paths sit
        # 2.4e-4 .. 8e-4 from the float64 oracle on this problem and as far from each other (tools/dev/gat_cond_probe.py,
        # profiles/r03_gat_conditioning.txt) - and at the per-cent level in the gradients, which those rows dominate
        # (|dx| ~ 1e4; tools/dev/gat_grad_probe.py).  The kernels themselves are held to float64 at this width by
        # test_one_launch_dense_half_vs_float64_autograd; here the two solves must agree as two fp32 runs can.
        def rel(a, b):
            return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

        def typical(a, b):                                  # the median entry, not the few rows that carry the norm
            return ((a - b).abs() / (b.abs() + 1e-3)).median().item()
        assert typical(res[1][1], res[0][1]) <= 1e-2, "dx (median entry)"
        if method == "rk4":                                 # adaptive steps on those rows: accept / reject moves with rounding
            # norm-wise bar: 0.25.  It was 0.1 (and held) while the one-launch path summed its products in sub-group order
            # (rounds 2-3); with both of its products on the matrix instruction (round 4: gat_project_d64_kernel,
            # gat_dense_vjp_d64_kernel - each within the same 2e-5 of float64 as the kernels they replace, test above) the
            # one-head case measures 0.18: the norm is carried by the handful of rows with rstd ~ 316, where any change of
            # summation order moves the result by this much (the median entry, asserted above, agrees to 1e-2)
            assert rel(res[1][1], res[0][1]) <= 0.25, "dx"
            for a, b in zip(res[1][2], res[0][2]):
                assert b.abs().max().item() < 1e-3 or rel(a, b) <= 0.25, "parameter gradient"
        return
    if method == "dopri5":
        # two fp32 runs of an adaptive solve take accept / reject decisions that move with rounding (and at one channel per
        # group GroupNorm returns beta + noise x 316): the gradients agree in norm, not entry by entry
        def rel(a, b):
            return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        assert rel(res[1][1], res[0][1]) <= 5e-2, "dx"
        for a, b in zip(res[1][2], res[0][2]):
            assert b.abs().max().item() < 1e-3 or rel(a, b) <= 5e-2, "parameter gradient"
        return
    close(res[1][1], res[0][1], gtol, "dx")
    for a, b in zip(res[1][2], res[0][2]):
        close(a, b, gtol, "parameter gradient")


@pytest.mark.parametrize("d,H", [(64, 8), (32, 4), (16, 2)])
def test_dense_vjp_closes_the_max_path_step_bit_for_bit(d, H):
    """The path of the gradient through each head's maximum (GAT/layers.py:47) on the raw-logit route: the max-path launch
    leaves per-block sums of da and arg-max candidates, and gode_gat_dense_vjp_small_f32 takes head h's sum T_h off
    dA2[src(e*), 2h] and dA2[tgt(e*), 2h+1] while it loads those rows.  Against the same launch on a dA2 that was corrected
    beforehand: every output identical bit for bit (the term is of size eps / (sum + eps) in a real solve and would hide
    under any parity tolerance)."""
    from graph_odenet_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(d + H)
    n, E = 700, 9000                                            # virtual edges E * 1: indices are node * H + head
    D = dev()
    f = dict(dtype=torch.float32, device=D)
    x = torch.randn(n, d, generator=g).to(D)
    gamma, beta = (1 + 0.3 * torch.randn(d, generator=g)).to(D), (0.3 * torch.randn(d, generator=g)).to(D)
    Wsrc, Wtgt = (torch.randn(d + 1, d, generator=g) / d ** 0.5).to(D), (torch.randn(d + 1, d, generator=g) / d ** 0.5).to(D)
    Wlog = (torch.randn(d + 1, 2 * H, generator=g) / d ** 0.5).to(D)
    dPs, dPt, dA2 = torch.randn(n, d, generator=g).to(D), torch.randn(n, d, generator=g).to(D), torch.randn(n, 2 * H, generator=g).to(D)
    heads_of = torch.arange(E) % H
    esrc = (torch.randint(0, n, (E,), generator=g) * H + heads_of).to(torch.int32)
    etgt = (torch.randint(0, n, (E,), generator=g) * H + heads_of).to(torch.int32)
    cap, n_part = lib.gode_gat_heads_block_cap(), lib.gode_gat_heads_parts(E)
    psum = torch.zeros(cap * H)
    psum[:n_part * H] = torch.randn(n_part * H, generator=g)
    pidx = torch.full((cap * H,), 2 ** 31 - 1, dtype=torch.int32)
    first = {}
    for h in range(H if H > 2 else 1):                          # (at H = 2 one head has no arg-max candidate: nothing to take off)
        cand = torch.randint(0, E // H, (3,), generator=g) * H + h
        for b, e in zip(torch.randint(0, n_part, (3,), generator=g).tolist(), cand.tolist()):
            pidx[b * H + h] = min(int(pidx[b * H + h]), e)
        first[h] = int(pidx.view(cap, H)[:n_part, h].min())
    scratch = torch.zeros(lib.gode_gat_heads_scratch_bytes(E, H), dtype=torch.uint8)
    scratch[:cap * H * 4] = psum.view(torch.uint8)
    scratch[cap * H * 4:cap * H * 8] = pidx.view(torch.uint8)
    scratch = scratch.to(D)
    fixed = dA2.clone()
    for h, e in first.items():
        T = torch.zeros((), dtype=torch.float32)
        for b in range(n_part):                                  # the kernel's order of summation
            T = T + psum[b * H + h]
        assert heads_of[e] == h
        fixed[int(esrc[e]) // H, 2 * h] -= T.to(D)
        fixed[int(etgt[e]) // H, 2 * h + 1] -= T.to(D)
    assert not torch.equal(fixed, dA2)
    # without and with the weights' packed images (at d = 64 the latter is the matrix-instruction form of the launch)
    for packed in (None, ops.gat_small_pack(Wsrc, Wtgt, Wlog, H)):
        outs = []
        for da2, mfx in ((fixed, None), (dA2, (scratch, esrc.to(D), etgt.to(D)))):
            part = ops.gat_small_part(n, d, H, D)
            ka = torch.empty(n, d, **f)
            ops.gat_dense_vjp_small([(1.0, x)], n, d, min(32, d), 1e-5, gamma, beta, Wsrc, Wtgt, Wlog, H, dPs, dPt, da2, ka, part,
                                    maxfix=mfx, packed=packed)
            kth = torch.empty(2 * (d + 1) * d + (d + 1) * 2 * H + d + H + 2 * d, **f)
            kat = torch.empty(1, **f)
            ops.gat_small_finish(part, n, d, H, 0.3, kth, kat)
            outs.append((ka, kth, kat))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
