"""GPU parity of the GCN-family path (layers -> ODEfunc -> ODEBlock -> solver) against the golden
vectors captured from the reference's classes and against the oracle solver on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=TOL, what=""):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


def coo(g, n):
    return torch.sparse_coo_tensor(torch.stack([T(g["rows"]).long(), T(g["cols"]).long()]), T(g["vals"]), (n, n))


def test_graph_convolution_vs_reference_golden(golden):
    from graph_odenet_amd.layers import GraphConvolution, FixedGraphConvolution
    g = golden("gcn_layer.npz")
    n = int(g["n"])
    adj = coo(g, n).to(dev())
    for cls in (GraphConvolution, FixedGraphConvolution):
        layer = cls(g["x"].shape[1], g["weight"].shape[1]).to(dev())
        layer.load_state_dict({"weight": T(g["weight"]), "bias": T(g["bias"])})
        x = T(g["x"]).to(dev()).requires_grad_(True)
        if cls is GraphConvolution:
            out = layer(x, adj)
        else:
            layer.set_adj(adj)
            out = layer(x)
        close(out, g["out"], what="fwd")
        out.backward(T(g["gout"]).to(dev()))
        close(x.grad, g["gx"], what="gx"); close(layer.weight.grad, g["gw"], what="gw")
        close(layer.bias.grad, g["gb"], what="gb")
    assert repr(layer) == "FixedGraphConvolution (40 -> 16)"


def test_dense_adjacency_accepted(golden):
    """GCN-dense-paper passes a dense N x N tensor (GCN-dense-paper/utils.py:87)."""
    from graph_odenet_amd.layers import GraphConvolution
    g = golden("gcn_layer.npz")
    n = int(g["n"])
    layer = GraphConvolution(40, 16).to(dev())
    layer.load_state_dict({"weight": T(g["weight"]), "bias": T(g["bias"])})
    out = layer(T(g["x"]).to(dev()), coo(g, n).to_dense().to(dev()))
    close(out, g["out"])


@pytest.mark.parametrize("d", [16, 64, 128])
def test_odefunc_fwd_vjp_vs_reference_golden(golden, d):
    from graph_odenet_amd.models import ODEfunc
    g = golden("gcn_odefunc_%d.npz" % d)
    n = int(g["n"])
    f = ODEfunc(d).to(dev())
    f.load_state_dict({"norm1.weight": T(g["gn_w"]), "norm1.bias": T(g["gn_b"]),
                       "gc1.weight": T(g["W"]), "gc1.bias": T(g["b"])})
    f.set_adj(coo(g, n).to(dev()))
    x = T(g["x"]).to(dev()).requires_grad_(True)
    out = f(torch.tensor(float(g["t"])), x)
    assert f.nfe == 1
    # d=16: one channel per group -> GroupNorm output is beta + rounding noise * 316
    tol = 2e-4 if d == 16 else TOL
    close(out, g["out"], tol, "odefunc fwd")
    out.backward(T(g["gout"]).to(dev()))
    close(x.grad, g["gx"], max(tol, 2e-5), "gx")
    close(f.gc1.weight.grad, g["gW"], 1e-4 if d > 16 else 1e-3, "gW")
    close(f.gc1.bias.grad, g["gb"], 5e-5, "gb")
    close(f.norm1.bias.grad, g["g_gn_b"], 5e-5, "g beta")
    if d > 16:
        close(f.norm1.weight.grad, g["g_gn_w"], 5e-5, "g gamma")


def test_odefunc2_fwd_vjp_vs_reference_golden(golden):
    """A4: ODEfunc2.forward (GCN/models.py:565-575) against the vector captured from the reference class."""
    from graph_odenet_amd.models import ODEfunc2
    g = golden("gcn_odefunc2.npz")
    n, d = int(g["n"]), g["x"].shape[1]
    f = ODEfunc2(d, 0.5).to(dev())
    f.load_state_dict({k.replace("__", "."): T(v) for k, v in g.items() if k.startswith(("norm", "gc"))})
    f.set_adj(coo(g, n).to(dev()))
    x = T(g["x"]).to(dev()).requires_grad_(True)
    out = f(torch.tensor(float(g["t"])), x)
    # d = 64: two channels per group.  x_hat = (a-b)/2 / sqrt((a-b)^2/4 + eps) has slope up to
    # 1/(2 sqrt(eps)) = 158 where the two relu outputs nearly coincide, and f ENDS with such a norm, so
    # 1e-7-level input differences surface as ~1e-5..1e-4 in the output (the reference's own property).
    close(out, g["out"], 2e-4, "odefunc2 fwd")
    out.backward(T(g["gout"]).to(dev()))
    close(x.grad, g["gx"], 5e-3, "gx")            # ill-conditioned GroupNorm backward (see test_oracle_golden)
    close(f.gc2.bias.grad, g["g__gc2__bias"], 5e-3, "g gc2.bias")
    close(f.gc1.weight.grad, g["g__gc1__weight"], 5e-3, "g gc1.weight")
    close(f.norm2.weight.grad, g["g__norm2__weight"], 5e-3, "g norm2.weight")
    close(f.norm2.bias.grad, g["g__norm2__bias"], 5e-3, "g norm2.bias")
    close(f.norm1.weight.grad, g["g__norm1__weight"], 5e-3, "g norm1.weight")


def test_odek_models_run():
    """ODEK1 / ODEK2 compose the same blocks (GCN/models.py:524-600); ODEK2 reproduces quirk Q1 (tol = dropout)."""
    from graph_odenet_amd import models
    torch.manual_seed(0)
    n = 300
    r = torch.randint(0, n, (2000,)); c = torch.randint(0, n, (2000,))
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), torch.rand(2000) / 8, (n, n)).to(dev())
    x = torch.randn(n, 20, device=dev())
    m1 = models.ODEK1(nfeat=20, nhid=64, nclass=4, dropout=0.0, nlayers=4, method="rk4", step_size=0.25).to(dev())
    m2 = models.ODEK2(nfeat=20, nhid=64, nclass=4, dropout=0.5, nlayers=5, method="rk4", step_size=0.5).to(dev())
    assert m2.gcs[1].tol == 0.5
    for m in (m1, m2):
        out = m(x, adj)
        assert out.shape == (n, 4) and torch.isfinite(out).all()
        out.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def cora(golden):
    gr = golden("cora_graph.npz")
    n = int(gr["n"])
    adj = torch.sparse_coo_tensor(torch.stack([T(gr["rows"].astype(np.int64)), T(gr["cols"].astype(np.int64))]),
                                  T(gr["vals"]), (n, n))
    feats = torch.zeros(n, int(gr["n_feat"]))
    feats[T(gr["feat_rows"].astype(np.int64)), T(gr["feat_cols"].astype(np.int64))] = T(gr["feat_vals"])
    return adj, feats, T(gr["labels"].astype(np.int64)), T(gr["idx_train"].astype(np.int64))


def test_gcn3_cora_logits_vs_reference_golden(golden):
    from graph_odenet_amd import models
    adj, feats, _, _ = cora(golden)
    g = golden("gcn3_cora.npz")
    for name in ("GCN3", "RGCN3"):
        m = getattr(models, name)(nfeat=feats.shape[1], nhid=16, nclass=7, dropout=0.5).to(dev())
        m.load_state_dict({k: T(g["%s__%s" % (name, k.replace(".", "__"))]) for k in m.state_dict()})
        m.eval()
        with torch.no_grad():
            out = m(feats.to(dev()), adj.to(dev()))
        close(out, g[name + "__out"], what=name)


def oracle_odegcn3(sd, feats, adj, method, options, tol, labels=None, idx=None, dtype=torch.float32):
    """ODEGCN3.forward (GCN/models.py:213-218) on the oracle: reference layer math + oracle solver.
    dtype=float64 gives the ground truth used to measure the fp32 noise floor of this computation."""
    from oracle import layers_ref as R, solver_ref as S
    sd = {k: v.to(dtype) for k, v in sd.items()}
    feats, adj = feats.to(dtype), adj.to(dtype)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gn_w = torch.nn.Parameter(sd["gc2.odefunc.norm1.weight"].clone())
            self.gn_b = torch.nn.Parameter(sd["gc2.odefunc.norm1.bias"].clone())
            self.W = torch.nn.Parameter(sd["gc2.odefunc.gc1.weight"].clone())
            self.b = torch.nn.Parameter(sd["gc2.odefunc.gc1.bias"].clone())
            self.nfe = 0

        def forward(self, t, x):
            self.nfe += 1
            return R.odefunc(t.to(dtype), x, adj, self.gn_w, self.gn_b, self.W, self.b)
    f = F()
    w1 = sd["gc1.weight"].clone().requires_grad_(True); b1 = sd["gc1.bias"].clone().requires_grad_(True)
    w3 = sd["gc3.weight"].clone().requires_grad_(True); b3 = sd["gc3.bias"].clone().requires_grad_(True)
    x = torch.relu(R.graph_convolution(feats, adj, w1, b1))
    x = S.odeint_adjoint(f, x, torch.tensor([0., 1.], dtype=dtype), tol, tol, method, options)[1]
    x = R.graph_convolution(x, adj, w3, b3)
    out = torch.log_softmax(x, 1)
    grads = None
    if labels is not None:
        loss = torch.nn.functional.nll_loss(out[idx], labels[idx])
        loss.backward()
        grads = {"gc1.weight": w1.grad, "gc1.bias": b1.grad, "gc3.weight": w3.grad, "gc3.bias": b3.grad,
                 "gc2.odefunc.norm1.weight": f.gn_w.grad, "gc2.odefunc.norm1.bias": f.gn_b.grad,
                 "gc2.odefunc.gc1.weight": f.W.grad, "gc2.odefunc.gc1.bias": f.b.grad}
    return out.detach(), grads, f.nfe


def noise_floor_check(got, ref32, ref64, what, slack=4.0, floor=1e-5):
    """|got - exact| must stay within `slack` x the fp32 oracle's own distance to the fp64 ground truth
    (plus 1e-5 of the magnitude): parity to the noise floor of the fp32 computation itself."""
    got = got.detach().cpu().double()
    e_ref = (ref32.double() - ref64).abs().max().item()
    e_got = (got - ref64).abs().max().item()
    scale = max(1.0, ref64.abs().max().item())
    assert e_got <= slack * e_ref + floor * scale, "%s: err %.3e vs fp32-oracle err %.3e (scale %.2e)" % (what, e_got, e_ref, scale)


@pytest.mark.parametrize("nhid", [16, 64, 128])
def test_odegcn3_rk4_forward_backward_vs_oracle_on_cora(golden, nhid):
    """The north-star step (fwd + adjoint bwd) at NFE=64 on real Cora, product vs oracle.
    Logits: 1e-5 against the fp32 oracle.  Gradients: the adjoint chains 128 f-evals through relu
    kinks and (at nhid=64, two channels per group) an ill-conditioned GroupNorm backward, so the fp32
    oracle itself sits ~1e-3 from the fp64 ground truth; the product must be as close to the ground
    truth as the fp32 oracle is (noise_floor_check)."""
    from graph_odenet_amd import models
    adj, feats, labels, idx = cora(golden)
    torch.manual_seed(42)
    m = models.ODEGCN3(nfeat=feats.shape[1], nhid=nhid, nclass=7, dropout=0.0, method="rk4", step_size=1 / 16)
    with torch.no_grad():
        m.gc2.odefunc.norm1.weight.uniform_(0.5, 1.5)
        m.gc2.odefunc.norm1.bias.uniform_(-0.5, 0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    opts = {"step_size": 1 / 16}
    ref_out, ref_g, ref_nfe = oracle_odegcn3(sd, feats, adj, "rk4", opts, 1e-5, labels, idx)
    out64, g64, _ = oracle_odegcn3(sd, feats, adj, "rk4", opts, 1e-5, labels, idx, dtype=torch.float64)
    m = m.to(dev())
    m.train()
    m.nfe = 0
    out = m(feats.to(dev()), adj.to(dev()))
    assert m.nfe == 64
    if nhid == 128:
        close(out, ref_out, what="logits")
    else:                      # 16: the reference's own default width (--hidden 16), one channel per GroupNorm group
        noise_floor_check(out, ref_out, out64, "logits")
    m.nfe = 0
    torch.nn.functional.nll_loss(out[idx.to(dev())], labels.to(dev())[idx.to(dev())]).backward()
    # nfe is an integer the reference's harness prints (GCN/train_res.py:100-101): the adjoint's count includes the one
    # evaluation per output time that torchdiffeq spends on dL/dt (skipped here under rk4, but counted:
    # odeint.NFE_COUNTS_SKIPPED_DLDT_EVAL) - asserted against the oracle's own counter, which runs that evaluation
    assert m.nfe == 65 and ref_nfe == 64 + m.nfe
    for k, p in m.named_parameters():
        noise_floor_check(p.grad, ref_g[k], g64[k], "grad " + k, slack=4.0 if nhid == 128 else 20.0)
    if nhid == 16:
        from graph_odenet_amd import odeint as OI
        OI.NFE_COUNTS_SKIPPED_DLDT_EVAL = False
        try:
            m.nfe = 0
            out = m(feats.to(dev()), adj.to(dev()))
            m.nfe = 0
            out.sum().backward()
            assert m.nfe == 64                    # evaluations actually launched
        finally:
            OI.NFE_COUNTS_SKIPPED_DLDT_EVAL = True


MARGIN = 0.15      # error ratios this close to 1 are ties: the ratio itself is only reproducible to several per cent
                   # deep in the adjoint solve (relu masks of pre-activations next to zero flip with fp32 rounding)


def _same_steps(got, ref, what, dt_tol=(2e-3, 2e-2)):
    """How far product and oracle controllers walk together: per solve, the number of leading attempts with the same
    accept / reject decision and the same step size (forward solve: to 2e-3, the rounding of the fp32 error norms;
    adjoint solve: to 2e-2).  The walk ends where the step sizes have drifted apart or a decision differs; a differing
    decision is only tolerated at a tie (both error ratios within MARGIN of 1) - anywhere else it is a controller
    bug.  From that point on the two runs integrate different grids and only the replayed comparison applies."""
    assert len(got) == len(ref), "%s: %d solves vs %d" % (what, len(got), len(ref))
    compared = []
    for k, (a, b) in enumerate(zip(got, ref)):
        n = 0
        for (da, acc_a, ra), (db, acc_b, rb) in zip(a, b):
            if abs(da - db) > dt_tol[min(k, 1)] * abs(db):
                break
            if acc_a != acc_b:
                # forward solve: only a tie may be decided differently.  The adjoint solve's error estimate is not
                # reproducible beyond its first ~20 attempts (measured: ratios 1.62 vs 0.76 at attempt 25 with the step
                # sizes still within 2 %) - the walk just ends there.
                assert k > 0 or (abs(rb - 1.0) <= MARGIN and abs(ra - 1.0) <= MARGIN), \
                    "%s solve %d attempt %d: decisions differ at error ratios %.4f / %.4f" % (what, k, n, ra, rb)
                break
            n += 1
        compared.append(n)
    return compared


def free_running_check(grads, sd, feats, adj, labels, idx, ref_g, g64_replay, what, slack=4.0, floor=1e-5):
    """Gradients of the FREE-RUNNING adaptive solves (the product's own controller - the reference's default path)
    against the fp64 oracle.  Two adaptive runs do not share a grid, so the yardstick is what the grid itself is
    worth: the fp64 oracle at rtol = atol = 1e-5 against the fp64 oracle at 1e-6 (discretisation noise of the
    tolerance the reference runs at), plus the fp32 oracle's own distance from the fp64 run of its own steps (rounding
    noise, as in noise_floor_check).  |product - fp64 oracle(1e-6)| <= slack * (discretisation + rounding) + floor."""
    _, g5, _ = oracle_odegcn3(sd, feats, adj, None, None, 1e-5, labels, idx, dtype=torch.float64)
    _, g6, _ = oracle_odegcn3(sd, feats, adj, None, None, 1e-6, labels, idx, dtype=torch.float64)
    report = {}
    for k in grads:
        got = grads[k].detach().cpu().double()
        e_disc = (g5[k] - g6[k]).abs().max().item()
        e_round = (ref_g[k].double() - g64_replay[k]).abs().max().item()
        e_got = (got - g6[k]).abs().max().item()
        scale = max(1.0, g6[k].abs().max().item())
        report[k] = (e_got, e_disc, e_round)
        assert e_got <= slack * (e_disc + e_round) + floor * scale, \
            "%s free-running grad %s: err %.3e vs discretisation %.3e + rounding %.3e" % (what, k, e_got, e_disc, e_round)
    return report


def _dopri5_three_ways(sd, feats, adj, labels, idx, run_product):
    """(1) the fp32 oracle, its step sequences recorded; (2) the fp64 oracle REPLAYING those steps (the ground truth of
    the same discretisation); (3) the product twice: free-running (its own controller, sequences recorded) and
    replaying the oracle's steps on its own kernels."""
    from graph_odenet_amd import solver as PS
    from oracle import solver_ref as S
    S.TRACE = []
    try:
        ref_out, ref_g, _ = oracle_odegcn3(sd, feats, adj, None, None, 1e-5, labels, idx)
        ref_seq = S.TRACE
    finally:
        S.TRACE = None
    S.REPLAY = [list(q) for q in ref_seq]
    try:
        out64, g64, _ = oracle_odegcn3(sd, feats, adj, None, None, 1e-5, labels, idx, dtype=torch.float64)
        assert S.REPLAY == []
    finally:
        S.REPLAY = None
    PS.TRACE = []
    try:
        free = run_product()
        got_seq = PS.TRACE
    finally:
        PS.TRACE = None
    PS.REPLAY = [list(q) for q in ref_seq]
    try:
        replayed = run_product()
        assert PS.REPLAY == []
    finally:
        PS.REPLAY = None
    return (ref_out, ref_g, ref_seq), (out64, g64), (free, got_seq), replayed


def test_odegcn3_dopri5_vs_oracle_on_cora(golden):
    """C1 with the reference's DEFAULT solver (no method=: dopri5, rtol = atol = 1e-5; GCN/models.py:192).  The
    product's controller takes exactly the oracle's decisions (same accept / reject sequence, same step sizes, forward
    and adjoint solve, up to the first accept / reject TIE - _same_steps); logits to 1e-4; and with the oracle's step
    sequence replayed on the product's kernels, logits to 1e-5 and every gradient as close to the fp64 ground truth of
    that step sequence as the fp32 oracle is (noise_floor_check).  Solver parity vs torchdiffeq itself is unpinned
    (oracle/solver_ref.py)."""
    from graph_odenet_amd import models
    adj, feats, labels, idx = cora(golden)
    torch.manual_seed(7)
    m = models.ODEGCN3(nfeat=feats.shape[1], nhid=128, nclass=7, dropout=0.0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev())

    def run():
        m.zero_grad(set_to_none=True)
        m.nfe = 0
        out = m(feats.to(dev()), adj.to(dev()))
        nfe_f = m.nfe
        m.nfe = 0
        torch.nn.functional.nll_loss(out[idx.to(dev())], labels.to(dev())[idx.to(dev())]).backward()
        return out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, (nfe_f, m.nfe)
    (ref_out, ref_g, ref_seq), (out64, g64), ((out, grads, nfe), got_seq), (out_r, grads_r, _) = \
        _dopri5_three_ways(sd, feats, adj, labels, idx, run)
    compared = _same_steps(got_seq, ref_seq, "cora dopri5")
    assert compared[0] == len(ref_seq[0]) and compared[1] >= 8          # whole forward solve; adjoint up to a tie
    close(out, ref_out, 1e-4, "dopri5 logits")
    # the oracle's steps on the product's kernels: arithmetic parity of the adaptive solve and its adjoint
    close(out_r, ref_out, 1e-5, "dopri5 logits, replayed steps")
    for k in grads_r:
        noise_floor_check(grads_r[k], ref_g[k], g64[k], "grad (replayed steps) " + k)
    assert 8 <= nfe[0] <= 400 and nfe[1] >= 8
    # the product's OWN controller, forward and adjoint: gradients against the fp64 oracle
    free_running_check(grads, sd, feats, adj, labels, idx, ref_g, g64, "cora dopri5")


def test_pubmed_dense_paper_dopri5_vs_oracle(golden):
    """C2: Pubmed's real topology, symmetric normalisation, adjacency passed DENSE as GCN-dense-paper does
    (utils.py:87), through dense_paper.ODEGCN3 (the variant's own model class), ODEBlock with the reference's default dopri5 (rtol=atol=1e-5), d=16 (--hidden default).
    Features are synthetic (the reference checkout lacks ind.pubmed.allx): row-normalised sparse Bernoulli,
    density 10 %, F=500, seed 0.  Forward AND adjoint: same controller decisions as the oracle, logits 1e-4,
    gradients at the noise floor of the fp32 computation (fp64 replay of the same steps as ground truth)."""
    from graph_odenet_amd import models
    g = golden("pubmed_graph_sym.npz")
    n = int(g["n"])
    assert n == 19717 and g["rows"].shape[0] == 108365
    ii = torch.stack([T(g["rows"].astype(np.int64)), T(g["cols"].astype(np.int64))])
    adj_sp = torch.sparse_coo_tensor(ii, T(g["vals"]), (n, n))
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(n, 500, generator=gen) < 0.1).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1)
    labels = torch.randint(0, 3, (n,), generator=gen)
    idx = torch.randperm(n, generator=gen)[:60]                 # Planetoid's 20 labelled nodes per class
    torch.manual_seed(1)
    # the GCN-dense-paper variant's own class: Glorot-uniform weights with the relu gain, zero
    # biases, dropout on the input features (p = 0 here: the oracle has no dropout), dense adjacency
    from graph_odenet_amd import dense_paper
    m = dense_paper.ODEGCN3(nfeat=500, nhid=16, nclass=3, dropout=0.0)
    assert type(m.gc2.odefunc) is dense_paper.ODEfunc and float(m.gc1.bias.abs().max()) == 0.0
    with torch.no_grad():                            # zero biases make relu'(0) ties: give them the GCN variant's spread
        for b in (m.gc1.bias, m.gc3.bias, m.gc2.odefunc.gc1.bias):
            b.uniform_(-0.25, 0.25)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev())
    adj_dense = adj_sp.to(dev()).to_dense()          # 19717^2 fp32 = 1.55 GB, as the reference holds it

    def run():
        m.zero_grad(set_to_none=True)
        m.nfe = 0
        out = m(x.to(dev()), adj_dense)
        nfe_f = m.nfe
        m.nfe = 0
        torch.nn.functional.nll_loss(out[idx.to(dev())], labels.to(dev())[idx.to(dev())]).backward()
        return out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, (nfe_f, m.nfe)
    (ref_out, ref_g, ref_seq), (out64, g64), ((out, grads, nfe), got_seq), (out_r, grads_r, _) = \
        _dopri5_three_ways(sd, x, adj_sp, labels, idx, run)
    compared = _same_steps(got_seq, ref_seq, "pubmed dopri5")
    assert compared[0] == len(ref_seq[0]) and compared[1] >= 4
    close(out, ref_out, 1e-4, "pubmed dopri5 logits")
    close(out_r, ref_out, 1e-5, "pubmed dopri5 logits, replayed steps")
    for k in grads_r:
        noise_floor_check(grads_r[k], ref_g[k], g64[k], "grad (replayed steps) " + k, slack=8.0)
    assert nfe[0] >= 8 and nfe[1] >= 8
    free_running_check(grads, sd, x, adj_sp, labels, idx, ref_g, g64, "pubmed dopri5", slack=8.0)


def test_generic_module_through_solver():
    """Any nn.Module goes through the autograd field; RK arithmetic still in HIP kernels."""
    from graph_odenet_amd.odeint import odeint_adjoint
    from oracle import solver_ref as S

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.W = torch.nn.Parameter(torch.randn(8, 8) * 0.3)

        def forward(self, t, y):
            return torch.tanh(y @ self.W) * (1 + t)
    torch.manual_seed(0)
    f = F()
    y0 = torch.randn(50, 8)
    t = torch.tensor([0., 1.])
    for method, opt, tol in (("rk4", {"step_size": 0.125}, 1e-5), (None, None, 2e-4)):
        f.zero_grad()
        y0c = y0.clone().requires_grad_(True)
        S.odeint_adjoint(f, y0c, t, 1e-5, 1e-5, method, opt)[1].pow(2).sum().backward()
        gW, gy = f.W.grad.clone(), y0c.grad.clone()
        fg = F().to(dev())
        fg.load_state_dict(f.state_dict())
        y0g = y0.clone().to(dev()).requires_grad_(True)
        out = odeint_adjoint(fg, y0g, t.to(dev()), 1e-5, 1e-5, method, opt)
        assert out.shape == (2, 50, 8)
        out[1].pow(2).sum().backward()
        close(fg.W.grad, gW, tol * 10, "gW"); close(y0g.grad, gy, tol * 10, "gy0")


@pytest.mark.parametrize("norm", ["per_tensor", "pooled"])
def test_initial_step_norm_option_matches_oracle(norm):
    """the initial-step norm of the 4-tensor adjoint solve is an explicit, documented option in
    BOTH solver.py and oracle/solver_ref.py (default "per_tensor").  Under either setting the product's first attempted
    adjoint step is the oracle's, and the two settings give different first steps on this problem."""
    from graph_odenet_amd import solver as PS
    from graph_odenet_amd.odeint import odeint_adjoint
    from oracle import solver_ref as S

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.W = torch.nn.Parameter(torch.randn(8, 8) * 0.3)

        def forward(self, t, y):
            return torch.tanh(y @ self.W) * (1 + 3 * t)
    torch.manual_seed(0)
    f = F()
    y0 = torch.randn(50, 8)
    t = torch.tensor([0., 1.])
    first = {}
    for nm in (norm, "per_tensor" if norm == "pooled" else "pooled"):
        S.INITIAL_STEP_NORM = PS.INITIAL_STEP_NORM = nm
        S.TRACE, PS.TRACE = [], []
        try:
            f.zero_grad()
            S.odeint_adjoint(f, y0.clone().requires_grad_(True), t, 1e-5, 1e-5)[1].pow(2).sum().backward()
            fg = F().to(dev())
            fg.load_state_dict(f.state_dict())
            odeint_adjoint(fg, y0.clone().to(dev()).requires_grad_(True), t.to(dev()), 1e-5, 1e-5)[1].pow(2).sum().backward()
            ref_seq, got_seq = S.TRACE, PS.TRACE
        finally:
            S.INITIAL_STEP_NORM = PS.INITIAL_STEP_NORM = "per_tensor"
            S.TRACE = PS.TRACE = None
        assert len(ref_seq) == len(got_seq) == 2
        for k in (0, 1):                                # forward solve (one tensor) and adjoint solve (four tensors)
            assert abs(got_seq[k][0][0] - ref_seq[k][0][0]) <= 2e-3 * ref_seq[k][0][0], (nm, k, got_seq[k][0], ref_seq[k][0])
        first[nm] = (ref_seq[0][0][0], ref_seq[1][0][0])
    a, b = first["per_tensor"], first["pooled"]
    assert a[0] == b[0]                                  # a one-tensor solve does not see the option
    assert abs(a[1] - b[1]) > 0.02 * b[1]                # the adjoint solve does


def test_cpu_tensor_is_refused():
    from graph_odenet_amd.odeint import odeint_adjoint
    with pytest.raises(RuntimeError):
        odeint_adjoint(torch.nn.Linear(2, 2), torch.zeros(3, 2), torch.tensor([0., 1.]))


def test_native_rk4_driver_matches_python_driver():
    """The one-call C driver (csrc/ode_driver.hip, two-stream adjoint schedule) and the per-stage Python
    driver issue the same kernels: identical logits, gradients equal to rounding of the partial sums."""
    from graph_odenet_amd import models, odeint as OI
    torch.manual_seed(0)
    n, d = 3000, 128
    r = torch.randint(0, n, (30000,)); c = torch.randint(0, n, (30000,))
    v = torch.rand(30000)
    v = v / torch.zeros(n).index_add_(0, r, v)[r]
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).to(dev())
    x = torch.randn(n, d, device=dev())
    res = {}
    for native in (True, False):
        OI.NATIVE_RK4 = native
        try:
            torch.manual_seed(1)
            blk = models.ODEBlock(models.ODEfunc(d), method="rk4", step_size=0.125).to(dev())
            xi = x.clone().requires_grad_(True)
            out = blk(xi, adj)
            out.square().sum().backward()
            res[native] = (out.detach(), xi.grad, [p.grad.clone() for p in blk.parameters()], blk.nfe)
        finally:
            OI.NATIVE_RK4 = True
    assert res[True][3] == res[False][3] == 32 + 33      # 32 forward + 32 adjoint evaluations + torchdiffeq's dL/dt one (odeint.py)
    close(res[True][0], res[False][0], 1e-6, "state")
    close(res[True][1], res[False][1], 1e-6, "gx")
    for a, b in zip(res[True][2], res[False][2]):
        close(a, b, 1e-5, "param grad")


def test_shape_errors_are_loud():
    from graph_odenet_amd import graph as G, ops
    g = G.from_coo(torch.tensor([0, 1], device=dev()), torch.tensor([1, 0], device=dev()), None, 2, 2)
    with pytest.raises(ValueError):
        ops.spmm(g, torch.zeros(3, 4, device=dev()))                 # wrong number of rows
    with pytest.raises(RuntimeError):
        ops.spmm(g, torch.zeros(2, 4))                                # CPU tensor: no fallback
    with pytest.raises(TypeError):
        ops.spmm(g, torch.zeros(2, 4, device=dev(), dtype=torch.float64))


def test_fullnorm_model_grads_vs_cpu_on_cora(golden):
    """RGCN3fullnorm (GCN/models.py:140-159) at nhid=128 on real Cora: logits and every gradient, including the
    GroupNorm affine parameters, against the same composition on the CPU (oracle layer + torch CPU GroupNorm)."""
    from graph_odenet_amd import models
    from oracle import layers_ref as R
    import torch.nn.functional as F
    adj, feats, labels, idx = cora(golden)
    torch.manual_seed(11)
    m = models.RGCN3fullnorm(nfeat=feats.shape[1], nhid=128, nclass=7, dropout=0.0)
    with torch.no_grad():
        for nm in (m.norm1, m.norm2):
            nm.weight.uniform_(0.5, 1.5); nm.bias.uniform_(-0.5, 0.5)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = F.relu(R.graph_convolution(feats, adj, p["gc1.weight"], p["gc1.bias"]))
    x = F.group_norm(x, 32, p["norm1.weight"], p["norm1.bias"], 1e-5)
    r = x
    x = F.relu(R.graph_convolution(x, adj, p["gc2.weight"], p["gc2.bias"]))
    x = F.group_norm(x, 32, p["norm2.weight"], p["norm2.bias"], 1e-5) + r
    ref = torch.log_softmax(R.graph_convolution(x, adj, p["gc3.weight"], p["gc3.bias"]), 1)
    F.nll_loss(ref[idx], labels[idx]).backward()
    m = m.to(dev())
    out = m(feats.to(dev()), adj.to(dev()))
    close(out, ref, what="fullnorm logits")
    F.nll_loss(out[idx.to(dev())], labels.to(dev())[idx.to(dev())]).backward()
    for k, q in m.named_parameters():
        close(q.grad, p[k].grad, 5e-5, "grad " + k)


@pytest.mark.parametrize("native,d", [(True, 16), (True, 128), (False, 16)])
def test_hip_graph_captured_solves_match_eager(native, d):
    """Launch-bound fixed-grid solves are captured into a HIP graph on their second occurrence and replayed afterwards
    (odeint.py: _GraphedSolve).  Four training steps with changing weights and inputs: the replayed solves return the
    bits of the eager ones (same kernels, same order), forward and adjoint, for the C driver and the Python driver."""
    from graph_odenet_amd import models, odeint as OI
    torch.manual_seed(0)
    n = 1500
    r = torch.randint(0, n, (9000,)); c = torch.randint(0, n, (9000,))
    v = torch.rand(9000)
    v = v / torch.zeros(n).index_add_(0, r, v)[r]
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).to(dev())
    xs = [torch.randn(n, d, device=dev()) for _ in range(4)]
    res = {}
    old = OI.GRAPH_CAPTURE_MAX_ELEMS
    for capture in (True, False):
        OI.NATIVE_RK4 = native
        OI.GRAPH_CAPTURE_MAX_ELEMS = old if capture else 0
        try:
            torch.manual_seed(1)
            blk = models.ODEBlock(models.ODEfunc(d), method="rk4", step_size=0.25).to(dev())
            opt = torch.optim.SGD(blk.parameters(), lr=0.05)
            log = []
            for x in xs:
                opt.zero_grad()
                xi = x.clone().requires_grad_(True)
                out = blk(xi, adj)
                out.square().mean().backward()
                log.append((out.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()]))
                opt.step()
            res[capture] = (log, blk.nfe)
            if capture:
                plans = list(OI.plans_of(blk.odefunc).values())
                assert len(plans) == 1 and plans[0].gf is not None and plans[0].gb is not None
        finally:
            OI.NATIVE_RK4 = True
            OI.GRAPH_CAPTURE_MAX_ELEMS = old
    assert res[True][1] == res[False][1] == 4 * (16 + 17)     # 17: the adjoint's count includes torchdiffeq's dL/dt evaluation (odeint.py)
    for (o1, g1, p1), (o2, g2, p2) in zip(res[True][0], res[False][0]):
        assert torch.equal(o1, o2) and torch.equal(g1, g2)
        for a, b in zip(p1, p2):
            assert torch.equal(a, b)


@pytest.mark.parametrize("method", ["rk4"])
def test_renumbered_ode_block_is_bit_identical(method):
    """Graphs whose hot rows crowd a few address residues are integrated on a hubs-first renumbering of their nodes
    (gcn_ode.tuned_graph; the solver permutes the state rows on entry and exit).  The choice is a deterministic function
    of the graph and of ODEBlock(node_order=...) - no stopwatch - so the same inputs give
    BIT-IDENTICAL parameter gradients in every process: run twice here per setting, on fresh graph objects.  Across
    the settings, outputs and the input gradient equal the unrenumbered run bit for bit (the SpMM on the renumbered
    graph is the row permutation of the SpMM on the given one, also bit for bit); parameter gradients are sums over the
    nodes in row order and agree to rounding.  (Fixed grid only: the adaptive controller's error norms are sums over
    the nodes too, so under a renumbering its accept / reject ties may fall differently.)"""
    from graph_odenet_amd import gcn_ode, graph as G, models, ops
    n, d = 3000, 64
    rs = np.random.RandomState(9)
    deg = rs.zipf(1.7, n).clip(1, 400)
    r = torch.from_numpy(np.repeat(np.arange(n), deg))
    c = torch.from_numpy(rs.randint(0, n, r.numel()) // rs.randint(1, 40, r.numel()))          # skewed towards low ids
    key = torch.unique(r * n + c)
    r, c = key // n, key % n
    v = 1.0 / torch.bincount(r, minlength=n).float()[r]

    def fresh():
        return torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).to(dev())
    adj = fresh()
    g = G.as_graph(adj)
    order = g.degree_order()
    h = g.relabel(order)
    X = torch.randn(n, d, device=dev())
    assert torch.equal(ops.spmm(h, X[order]), ops.spmm(g, X)[order])
    assert torch.equal(ops.spmm(h.transpose(), X[order]), ops.spmm(g.transpose(), X)[order])
    x = torch.randn(n, d, device=dev())
    gout = torch.randn(n, d, device=dev())
    with pytest.raises(ValueError):
        models.ODEBlock(models.ODEfunc(d), node_order="fastest")

    def run(node_order, a):
        torch.manual_seed(3)
        blk = models.ODEBlock(models.ODEfunc(d), method=method, step_size=0.25, node_order=node_order).to(dev())
        xi = x.clone().requires_grad_(True)
        out = blk(xi, a)
        out.backward(gout)
        took = gcn_ode.tuned_graph(G.as_graph(a), d, blk.odefunc.node_order)[1] is not None
        return (out.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()], blk.nfe), took
    res = {}
    for node_order, want in (("given", False), ("degree", True), (None, False)):      # None = "auto": small graph, kept
        (a1, took), (a2, _) = run(node_order, fresh()), run(node_order, fresh())
        assert took == want, node_order
        assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1])
        assert all(torch.equal(p, q) for p, q in zip(a1[2], a2[2])), "parameter gradients must not depend on the run"
        res[node_order] = a1
    assert res["degree"][3] == res["given"][3]
    assert torch.equal(res["degree"][0], res["given"][0]) and torch.equal(res["degree"][1], res["given"][1])
    assert all(torch.equal(p, q) for p, q in zip(res[None][2], res["given"][2]))
    # parameter gradients are sums over the nodes, taken in row order: the same numbers added in another order
    for a, b in zip(res["degree"][2], res["given"][2]):
        close(a, b, 2e-5, "parameter gradient")
    # the "auto" rule itself, with its size gate lifted, on graphs large enough for the statistic to mean something:
    # R-MAT as generated crowds the ids with few one-bits (imbalance ~10 at m = 64) -> hubs first (~1.4); a uniform
    # random graph (~1.03) stays as given.  The numbers are integer counts: asserted exactly reproducible.
    from graph_odenet_amd.synth import rmat_graph
    saved = (gcn_ode.RELABEL_MIN_OPERAND_BYTES, gcn_ode.RELABEL_MIN_NNZ)
    gcn_ode.RELABEL_MIN_OPERAND_BYTES = gcn_ode.RELABEL_MIN_NNZ = 0
    try:
        infos = []
        for _ in range(2):
            ga = rmat_graph(16, 600_000, seed=0, device=dev())
            assert gcn_ode.tuned_graph(ga, 128)[1] is not None
            infos.append(ga.__dict__["_tuned_info"][(128, "auto")])
        assert infos[0] == infos[1] and infos[0]["renumbered"]
        assert infos[0]["imbalance_given"] > 5.0 and infos[0]["imbalance_hubs_first"] < 2.0
        nu = 1 << 16
        gen = torch.Generator().manual_seed(0)
        ku = torch.unique(torch.randint(0, nu, (600_000,), generator=gen) * nu + torch.randint(0, nu, (600_000,), generator=gen))
        gu = G.as_graph(torch.sparse_coo_tensor(torch.stack([ku // nu, ku % nu]), torch.ones(ku.numel()), (nu, nu)).to(dev()))
        assert gcn_ode.tuned_graph(gu, 128)[1] is None
        iu = gu.__dict__["_tuned_info"][(128, "auto")]
        assert not iu["renumbered"] and iu["imbalance_given"] < 1.2
        assert gcn_ode.tuned_graph(gu, 128, "degree")[1] is not None and gcn_ode.tuned_graph(gu, 128, "given")[1] is None
    finally:
        gcn_ode.RELABEL_MIN_OPERAND_BYTES, gcn_ode.RELABEL_MIN_NNZ = saved


@pytest.mark.parametrize("name,nl", [("GCNK", 2), ("GCNK", 4), ("GCNKnorm", 2), ("GCNKnorm", 5), ("RESK1", 3), ("RESK1", 5),
                                     ("RESK2", 4), ("RESK2", 7), ("RESK1norm", 3), ("RESK1norm", 6), ("RESK2norm", 4),
                                     ("RESK2norm", 7), ("RESK", 5), ("RESK", 6), ("RESKnorm", 5), ("RESKnorm", 6)])
def test_depth_sweep_models_vs_reference_golden(golden, name, nl):
    """The model family GCN/train_layers.py sweeps (GCN/models.py:255-522): same state_dict keys, eval outputs and a
    gradient equal to what the reference classes produced."""
    from graph_odenet_amd import models
    g = golden("gcn_depth_models.npz")
    n = int(g["n"])
    adj = coo(g, n).to(dev())
    key = "%s_%d" % (name, nl)
    kw = dict(residue_layers=3) if name in ("RESK", "RESKnorm") else {}
    m = getattr(models, name)(nfeat=12, nhid=8, nclass=4, dropout=0.5, nlayers=nl, **kw)
    pre = key + "__sd__"
    sd = {k[len(pre):].replace("__", "."): T(v) for k, v in g.items() if k.startswith(pre)}
    assert set(sd) == set(m.state_dict().keys())
    m.load_state_dict(sd)
    m = m.to(dev()).eval()
    out = m(T(g["x"]).to(dev()), adj)
    close(out, g[key + "__out"], TOL, key + " out")
    out.backward(T(g["gout"]).to(dev()))
    close(m.gcs[0].bias.grad, g[key + "__gbias0"], 2e-5, key + " grad")


def test_depth_sweep_models_refuse_too_few_layers():
    from graph_odenet_amd import models
    for name, nl in (("GCNK", 1), ("GCNKnorm", 1), ("RESK1", 2), ("RESK2", 3), ("RESK1norm", 2), ("RESK2norm", 3), ("ODEK1", 2),
                     ("ODEK2", 3)):
        with pytest.raises(ValueError):
            getattr(models, name)(nfeat=4, nhid=4, nclass=2, dropout=0.5, nlayers=nl)


def test_failed_graph_capture_falls_back_to_eager(monkeypatch):
    """A capture attempt that raises leaves the plan on the eager path (one warning), with the same results."""
    from graph_odenet_amd import models, odeint as OI
    n, d = 400, 16
    adj = (torch.rand(n, n) < 0.02).float() + torch.eye(n)
    adj = (adj / adj.sum(1, keepdim=True)).to(dev())
    x = torch.randn(n, d, device=dev())

    def boom(*a, **k):
        raise RuntimeError("capture refused (test)")
    torch.manual_seed(0)
    blk = models.ODEBlock(models.ODEfunc(d), method="rk4", step_size=0.5).to(dev())
    ref = blk(x, adj).detach().clone()                       # first call: eager by design
    monkeypatch.setattr(OI, "_GraphedSolve", boom)
    with pytest.warns(UserWarning, match="capture of a fixed-grid solve failed"):
        out = blk(x, adj)
    assert torch.equal(out, ref)
    plan = list(OI.plans_of(blk.odefunc).values())[0]
    assert plan.no_capture and plan.gf is None
    assert torch.equal(blk(x, adj), ref)                     # and no further attempts


@pytest.mark.parametrize("d", [16, 128])
def test_native_dopri5_step_matches_python_driver(d):
    """Adaptive solves of the fused GCN field take one C-ABI call per step (csrc/ode_driver.hip:
    gode_gcn_ode_dopri5_step_{forward,adjoint}); the per-stage Python driver issues the same kernels, so both walk the
    same steps: equal nfe, states and gradients equal up to the summation order of the error norm / a_t."""
    from graph_odenet_amd import models, solver as SV
    torch.manual_seed(0)
    n = 2000
    r = torch.randint(0, n, (12000,)); c = torch.randint(0, n, (12000,))
    v = torch.rand(12000)
    v = v / torch.zeros(n).index_add_(0, r, v)[r]
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).to(dev())
    x = torch.randn(n, d, device=dev()).relu()
    res = {}
    for native in (True, False):
        SV.DOPRI5_NATIVE = native
        try:
            torch.manual_seed(1)
            blk = models.ODEBlock(models.ODEfunc(d), tol=1e-4).to(dev())          # default method: dopri5
            xi = x.clone().requires_grad_(True)
            out = blk(xi, adj)
            nf = blk.nfe; blk.nfe = 0
            out.square().mean().backward()
            res[native] = (out.detach(), xi.grad, [p.grad.clone() for p in blk.parameters()], nf, blk.nfe)
        finally:
            SV.DOPRI5_NATIVE = True
    assert res[True][3] == res[False][3] and res[True][4] == res[False][4] and res[True][3] >= 14
    if d == 16:     # launch-bound path: the C driver chains the stage inputs through a buffer pair, the Python driver combines
        assert torch.equal(res[True][0], res[False][0])          # them while gathering - the same bits (csrc/small.hip)
    close(res[True][0], res[False][0], 1e-6, "state")
    close(res[True][1], res[False][1], 1e-5, "gx")
    tol = 1e-4 if d == 16 else 1e-5          # d = 16: one channel per GroupNorm group (noise-floor gradients)
    for a, b in zip(res[True][2], res[False][2]):
        close(a, b, tol, "param grad")


@pytest.mark.parametrize("d,n", [(16, 2000), (32, 9000)])
def test_small_feval_writes_the_next_stage_input_bit_for_bit(d, n):
    """gode_gcn_feval_small_next_f32 (what the C dopri5 step driver does between the stages of a launch-bound step): a
    stage's launch also writes the NEXT stage's combined input y + h sum_j a_j k_j row by row, the term it has just computed
    taken from the register; the next evaluation then gathers that one array.  Same multiply-adds in the same order as
    combining the term list while gathering: the next stage's output is bit for bit the same either way, and the
    stage's own outputs do not change."""
    from graph_odenet_amd import gcn_ode, graph as G
    gen = torch.Generator().manual_seed(d + n)
    r = torch.randint(0, n, (5 * n,), generator=gen); c = torch.randint(0, n, (5 * n,), generator=gen)
    key = torch.unique(r * n + c)
    r, c = key // n, key % n
    v = torch.rand(key.numel(), generator=gen) + 0.1
    g = G.from_coo(r.to(dev()), c.to(dev()), v.to(dev()), n, n)
    f = dict(dtype=torch.float32, device=dev())
    W = (torch.randn(d + 1, d, generator=gen) / d ** 0.5).to(dev())
    b, gam, bet = (torch.randn(d, generator=gen) * 0.1).to(dev()), (torch.rand(d, generator=gen) + 0.5).to(dev()), (torch.rand(d, generator=gen) - 0.5).to(dev())
    spec = gcn_ode.GcnOdeSpec(g, W, b, gam, bet, min(32, d), 1e-5)
    y = torch.randn(n, d, generator=gen).to(dev()).relu()
    k = [torch.randn(n, d, generator=gen).to(dev()) for _ in range(3)]
    h = 0.37
    terms_s = [(1.0, y), (h * 0.3, k[0]), (h * -0.9, k[1])]                 # stage s: three terms
    k_s_plain = torch.empty(n, d, **f)
    gcn_ode._feval_small(spec, 0.4, terms_s, k_s_plain)
    # stage s + 1 names the result of stage s among its terms
    k_s, x_next = torch.empty(n, d, **f), torch.full((n, d), float("nan"), **f)
    terms_next = [(1.0, y), (h * 0.2, k[0]), (h * 1.1, k_s), (h * -0.4, k[2])]
    cot = [(-1.0, k[2])]
    dz_plain, dz = torch.empty(n, d, **f), torch.empty(n, d, **f)
    gcn_ode._feval_small(spec, 0.4, terms_s, k_s_plain, cot=cot, out2=dz_plain)
    gcn_ode._feval_small(spec, 0.4, terms_s, k_s, cot=cot, out2=dz, next_terms=terms_next, x_next=x_next)
    assert torch.equal(k_s, k_s_plain) and torch.equal(dz, dz_plain)
    want = y + (h * 0.2) * k[0] + (h * 1.1) * k_s + (h * -0.4) * k[2]
    assert (x_next - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    out_terms, out_chain = torch.empty(n, d, **f), torch.empty(n, d, **f)
    gcn_ode._feval_small(spec, 0.6, terms_next, out_terms)                  # term list combined while gathering
    gcn_ode._feval_small(spec, 0.6, [(1.0, x_next)], out_chain)             # the one array
    assert torch.equal(out_chain, out_terms)
    # refused: writing the next input over an array this launch gathers from, or over its own output
    with pytest.raises(Exception):
        gcn_ode._feval_small(spec, 0.4, terms_s, k_s, next_terms=terms_next, x_next=y)
    with pytest.raises(Exception):
        gcn_ode._feval_small(spec, 0.4, terms_s, k_s, next_terms=terms_next, x_next=k_s)


@pytest.mark.parametrize("d", [16, 128])
def test_square_graph_convolution_on_mfma_kernels(d):
    """A square GraphConvolution on >= 4096 rows takes the fused kernels for X W and its two gradients; same layer
    formula (GCN/layers.py:31-37), checked against torch CPU ops."""
    from graph_odenet_amd.layers import GraphConvolution
    gen = torch.Generator().manual_seed(d)
    n = 6000
    r = torch.randint(0, n, (40000,), generator=gen); c = torch.randint(0, n, (40000,), generator=gen)
    v = torch.rand(40000, generator=gen)
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    lay = GraphConvolution(d, d)
    x = torch.randn(n, d, generator=gen, requires_grad=True)
    W, b = lay.weight.detach().clone().requires_grad_(True), lay.bias.detach().clone().requires_grad_(True)
    ref = torch.sparse.mm(adj, x @ W) + b
    gout = torch.randn(n, d, generator=gen)
    ref.backward(gout)
    lay = lay.to(dev())
    xd = x.detach().to(dev()).requires_grad_(True)
    out = lay(xd, adj.to(dev()))
    close(out, ref, 1e-5, "out")
    out.backward(gout.to(dev()))
    close(xd.grad, x.grad, 2e-5, "gx")
    close(lay.weight.grad, W.grad, 2e-5, "gW")
    close(lay.bias.grad, b.grad, 2e-5, "gb")


def test_model_with_captured_solves_can_be_deep_copied_and_pickled():
    import copy
    import io
    from graph_odenet_amd import models
    n, d = 300, 16
    adj = (torch.rand(n, n) < 0.03).float() + torch.eye(n)
    adj = (adj / adj.sum(1, keepdim=True)).to(dev())
    x = torch.randn(n, d, device=dev())
    blk = models.ODEBlock(models.ODEfunc(d), method="rk4", step_size=0.5).to(dev())
    ref = [blk(x, adj).detach().clone() for _ in range(3)][-1]        # third call replays a captured graph
    twin = copy.deepcopy(blk)
    assert torch.equal(twin(x, adj), ref)
    buf = io.BytesIO()
    torch.save(blk, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    assert torch.equal(again(x, adj), ref)


def _citation(golden, name):
    """Citeseer: the reference loader's captured adjacency / features / labels; Pubmed: the real topology
    (sym-normalised as GCN-dense-paper holds it) with synthetic features and labels (ind.pubmed.allx is absent)."""
    if name == "citeseer":
        gr = golden("citeseer_graph.npz")
        n = int(gr["n"])
        adj = torch.sparse_coo_tensor(torch.stack([T(gr["rows"].astype(np.int64)), T(gr["cols"].astype(np.int64))]),
                                      T(gr["vals"]), (n, n))
        feats = torch.zeros(n, int(gr["n_feat"]))
        feats[T(gr["feat_rows"].astype(np.int64)), T(gr["feat_cols"].astype(np.int64))] = T(gr["feat_vals"])
        return adj, feats, T(gr["labels"].astype(np.int64)), T(gr["idx_train"].astype(np.int64)), 6
    gr = golden("pubmed_graph_sym.npz")
    n = int(gr["n"])
    adj = torch.sparse_coo_tensor(torch.stack([T(gr["rows"].astype(np.int64)), T(gr["cols"].astype(np.int64))]),
                                  T(gr["vals"]), (n, n))
    gen = torch.Generator().manual_seed(0)
    feats = (torch.rand(n, 500, generator=gen) < 0.1).float()
    feats = feats / feats.sum(1, keepdim=True).clamp_min(1)
    return adj, feats, torch.randint(0, 3, (n,), generator=gen), torch.arange(60), 3


@pytest.mark.parametrize("name", ["citeseer", "pubmed"])
def test_odegcn3_rk4_forward_backward_vs_oracle_on_citeseer_and_pubmed(golden, name):
    """The north-star step (64 f-evals forward, adjoint backward) on the other two citation graphs of north_star's
    parity clause, product vs oracle: logits to 1e-5, gradients to the fp32 noise floor (as on Cora above)."""
    from graph_odenet_amd import models
    adj, feats, labels, idx, ncls = _citation(golden, name)
    torch.manual_seed(7)
    m = models.ODEGCN3(nfeat=feats.shape[1], nhid=128, nclass=ncls, dropout=0.0, method="rk4", step_size=1 / 16)
    with torch.no_grad():
        m.gc2.odefunc.norm1.weight.uniform_(0.5, 1.5)
        m.gc2.odefunc.norm1.bias.uniform_(-0.5, 0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    opts = {"step_size": 1 / 16}
    ref_out, ref_g, _ = oracle_odegcn3(sd, feats, adj, "rk4", opts, 1e-5, labels, idx)
    out64, g64, _ = oracle_odegcn3(sd, feats, adj, "rk4", opts, 1e-5, labels, idx, dtype=torch.float64)
    m = m.to(dev())
    m.train()
    m.nfe = 0
    out = m(feats.to(dev()), adj.to(dev()))
    assert m.nfe == 64
    noise_floor_check(out, ref_out, out64, "logits", slack=2.0)
    close(out, ref_out, 2e-5, what="logits")
    torch.nn.functional.nll_loss(out[idx.to(dev())], labels.to(dev())[idx.to(dev())]).backward()
    # Pubmed: the ODE function's weight gradient is BIMODAL under a 1e-7 relative perturbation of the first layer's output
    # (one relu mask of a pre-activation next to zero, on an entry with a large cotangent, decides): 2.1e-5 or 1.08e-4 from
    # the fp64 oracle, three trials each way, with the library GEMM in the first layer as with csrc/rect.hip
    # (tools/dev/pubmed_rect_probe.py, profiles/r03_pubmed_conditioning.txt; the fp32 oracle sits at 1.1e-5).  The bar
    # covers both branches; on Citeseer no such tie exists and the 1e-5 floor stays.
    floor = 2e-4 if name == "pubmed" else 1e-5
    for k, p in m.named_parameters():
        noise_floor_check(p.grad, ref_g[k], g64[k], "grad " + k, slack=4.0, floor=floor)


def test_integration_times_read_once_per_version():
    """odeint keeps the host values of a device-resident `integration_time` (the reference's ODEBlock moves it to the
    GPU, GCN/models.py:195) on the tensor, keyed by its storage address, version counter and shape: no device->host copy
    - a host synchronisation - per forward pass, and an in-place change or a re-pointing (`set_`) is still seen."""
    from graph_odenet_amd import odeint as OI
    t = torch.tensor([0.0, 1.0], device=dev())
    assert OI._times(t) == [0.0, 1.0]
    assert t._gode_times == ((t.data_ptr(), t._version, (2,)), [0.0, 1.0])
    got = OI._times(t)
    got[1] = 5.0                                   # callers get their own list
    assert OI._times(t) == [0.0, 1.0]
    t[1] = 2.0
    assert OI._times(t) == [0.0, 2.0]
    t.set_(torch.tensor([0.0, 3.0], device=dev()))          # same object, other storage: no version bump
    assert OI._times(t) == [0.0, 3.0]
    assert OI._times(torch.tensor([0.0, 0.5])) == [0.0, 0.5] and OI._times([0, 1]) == [0.0, 1.0]


@pytest.mark.parametrize("method", ["rk4", "dopri5"])
def test_last_state_only_equals_stacked_result(method):
    """odeint_adjoint(..., _last_only=True) - what models.ODEBlock asks for, since the reference keeps `out[1]` - returns
    the last slice of the stacked result and the same gradients, bit for bit (same kernels in the same order; only the
    copies around the solve differ), also with a cotangent on the start state folded in by the caller."""
    from graph_odenet_amd import models
    from graph_odenet_amd.odeint import odeint_adjoint
    n, d = 700, 32
    torch.manual_seed(5)
    r = torch.randint(0, n, (6000,)); c = torch.randint(0, n, (6000,))
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), torch.rand(6000) / 8, (n, n)).coalesce().to(dev())
    f = models.ODEfunc(d).to(dev())
    f.set_adj(adj)
    t = torch.tensor([0.0, 1.0], device=dev())
    opts = {"step_size": 0.25} if method == "rk4" else None
    x = torch.randn(n, d, device=dev())
    gout = torch.randn(n, d, device=dev())
    res = []
    for last in (False, True):
        f.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        out = odeint_adjoint(f, xi, t, rtol=1e-5, atol=1e-5, method=method, options=opts, _last_only=last)
        y1 = out if last else out[1]
        assert y1.shape == (n, d)
        (y1 * gout).sum().backward()
        res.append((y1.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in f.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)


def test_graph_convolution_sparse_features_and_padded_classes():
    """A1 without a library GEMM: the rectangular X W / dY W^T / X^T dY of GraphConvolution run on
    csrc/rect.hip; a mostly-zero feature matrix takes the CSR(X) path from the second time the same tensor object
    arrives (graph_odenet_amd/layers.py: X W on the aggregation kernel with W as the dense operand); 7 output classes
    are padded to 8 columns so that the aggregation runs on the 16-byte kernel.  All three routes against the oracle
    layer (GCN/layers.py:31-37) on a Cora-shaped problem: 2708 x 1433 features with 1.3 % non-zeros, 16 and 7 outputs."""
    from graph_odenet_amd import layers as L
    from oracle import layers_ref as R
    torch.manual_seed(5)
    n, f = 2708, 1433
    x = (torch.rand(n, f) < 0.013).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1)
    r, c = torch.randint(0, n, (13264,)), torch.randint(0, n, (13264,))
    v = torch.rand(13264) + 0.1
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    for cout, needs_gx in ((16, False), (7, False), (7, True)):
        lay = L.GraphConvolution(f, cout)
        w, b = lay.weight.detach().clone().requires_grad_(True), lay.bias.detach().clone().requires_grad_(True)
        xr = x.clone().requires_grad_(needs_gx)
        ref = R.graph_convolution(xr, adj, w, b)
        gout = torch.randn(n, cout)
        ref.backward(gout)
        lay = lay.to(dev())
        xd, ad = x.to(dev()).requires_grad_(needs_gx), adj.to(dev())
        routes = []
        for sighting in range(3):                        # 1st: rectangular MFMA kernels; 2nd, 3rd: CSR(X) (unless x needs a gradient)
            lay.zero_grad()
            if xd.grad is not None:
                xd.grad = None
            out = lay(xd, ad)
            out.backward(gout.to(dev()))
            routes.append(L.sparse_features(xd) is not None if sighting else False)
            close(out, ref, 1e-5, "out (%d classes, sighting %d)" % (cout, sighting))
            close(lay.weight.grad, w.grad, 2e-5, "dW"); close(lay.bias.grad, b.grad, 2e-5, "db")
            if needs_gx:
                close(xd.grad, xr.grad, 2e-5, "dx")
        assert routes[1:] == [not needs_gx, not needs_gx], routes
    # a dense feature matrix stays on the rectangular kernels however often it arrives
    xdense = torch.randn(300, 200, device=dev())
    lay = L.GraphConvolution(200, 16).to(dev())
    a2 = torch.eye(300).to_sparse().to(dev())
    for _ in range(3):
        lay(xdense, a2)
    assert L.sparse_features(xdense) is None
    # the explicit switches: sparse_input=True takes CSR(X) from the FIRST call, False never does,
    # and a sparse tensor as the input is the CSR route whatever the switch - all against the oracle layer
    lay = L.GraphConvolution(f, 7)
    w, b = lay.weight.detach().clone().requires_grad_(True), lay.bias.detach().clone().requires_grad_(True)
    ref = R.graph_convolution(x, adj, w, b)
    gout = torch.randn(n, 7)
    ref.backward(gout)
    lay, ad = lay.to(dev()), adj.to(dev())
    for mode, inp, want in ((True, x.to(dev()), True), (False, x.to(dev()), False), (None, x.to_sparse().to(dev()), True),
                            (None, x.to_sparse_csr().to(dev()), True)):
        lay.sparse_input = mode
        lay.zero_grad()
        out = lay(inp, ad)
        out.backward(gout.to(dev()))
        close(out, ref, 1e-5, "out (sparse_input=%r, %s)" % (mode, inp.layout))
        close(lay.weight.grad, w.grad, 2e-5, "dW"); close(lay.bias.grad, b.grad, 2e-5, "db")
        if inp.layout == torch.strided:
            assert (L.sparse_features(inp, mode) is not None) == want
    # inside a capture nothing is decided: an unseen tensor runs dense, no host synchronisation is attempted
    fresh = x.to(dev())
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        L.sparse_features(fresh)                              # first sighting outside
        with torch.cuda.graph(g, stream=st):
            assert L.sparse_features(fresh) is None           # the second sighting would decide: not while capturing
    L.invalidate_sparse_features(fresh)
    assert L.sparse_features(fresh) is None                   # forgotten: a first sighting again


@pytest.mark.parametrize("n", [1500, 9000])          # a wave per row / four rows per wave (from 8 192 rows on)
@pytest.mark.parametrize("d", [16, 32])
@pytest.mark.parametrize("method", ["rk4", "dopri5"])
def test_fused_small_graph_path_matches_multi_launch_path(d, method, n):
    """csrc/small.hip (launch-bound graphs: one launch per f-eval, one per VJP, by re-association - everything after the
    gather is row-local) against the multi-launch path of the large graphs (option small_fused 0) through the same ODE
    block, C drivers and Python drivers: states to 1e-5, gradients to the noise floor of the width (one channel per
    GroupNorm group at d = 16 / 32), on a graph with a 300-neighbour hub, isolated rows and duplicate-free random edges;
    the fused path also against the fp64 oracle on Cora in test_odegcn3_rk4_forward_backward_vs_oracle_on_cora."""
    from graph_odenet_amd import _lib, models, odeint as OI
    lib = _lib.load()
    torch.manual_seed(d)
    r = torch.cat([torch.randint(0, n - 10, (4 * n,)), torch.zeros(300, dtype=torch.long)])     # rows n-10.. have no entry
    c = torch.cat([torch.randint(0, n, (4 * n,)), torch.randperm(n)[:300]])
    key = torch.unique(r * n + c)
    r, c = key // n, key % n
    v = torch.rand(key.numel()) + 0.1
    v = v / torch.zeros(n).index_add_(0, r, v)[r]
    adj = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n)).to(dev())
    x = torch.randn(n, d, device=dev()).relu()
    gout = torch.randn(n, d, device=dev())
    res = {}
    try:
        for fused in (1, 0):
            for native in (True, False):
                assert lib.gode_set_option(b"small_fused", fused) == 0 and lib.gode_get_option(b"small_fused") == fused
                OI.NATIVE_RK4 = native
                old_cap = OI.GRAPH_CAPTURE_MAX_ELEMS
                OI.GRAPH_CAPTURE_MAX_ELEMS = 0
                try:
                    torch.manual_seed(1)
                    blk = models.ODEBlock(models.ODEfunc(d), tol=1e-4, method=None if method == "dopri5" else "rk4",
                                          step_size=0.25 if method == "rk4" else None).to(dev())
                    with torch.no_grad():
                        blk.odefunc.norm1.weight.uniform_(0.5, 1.5); blk.odefunc.norm1.bias.uniform_(-0.5, 0.5)
                    xi = x.clone().requires_grad_(True)
                    out = blk(xi, adj)
                    out.backward(gout)
                    res[(fused, native)] = (out.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()])
                finally:
                    OI.NATIVE_RK4 = True
                    OI.GRAPH_CAPTURE_MAX_ELEMS = old_cap
    finally:
        lib.gode_set_option(b"small_fused", 1)
    ref = res[(0, True)]
    # two fp32 summation orders through 16-64 relu / GroupNorm stages: the same bars as the Cora test gives the widths
    # with 1 or 2 channels per group (noise_floor_check slack 20: rstd up to 316 amplifies every rounding)
    gtol = 5e-3 if (method == "rk4" or n < 8192) else 1e-2       # adaptive steps on 9 000 rows: 5.5e-3 measured (accept / reject noise)
    for key_ in ((1, True), (1, False)):
        got = res[key_]
        close(got[0], ref[0], 1e-5 if method == "rk4" else 1e-4, "state %s" % (key_,))
        close(got[1], ref[1], gtol, "gx %s" % (key_,))
        for a, b in zip(got[2], ref[2]):
            close(a, b, gtol, "param grad %s" % (key_,))
    # C driver and Python driver issue the same fused kernels
    if method == "rk4":
        assert torch.equal(res[(1, True)][0], res[(1, False)][0])
