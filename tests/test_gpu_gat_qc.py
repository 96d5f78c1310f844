"""GPU parity of the GAT edge-attention path and the QC edge-message path against the golden
vectors captured from the reference's classes (tests/golden/make_golden.py)."""
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
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


def gat_inputs(g):
    n = int(g["n"])
    src, tgt = T(g["src"]).long().to(dev()), T(g["tgt"]).long().to(dev())
    e = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e, device=dev())]),
                                   torch.ones(e, device=dev()), (n, e))
    return n, src, tgt, Mtgt


def test_gat_layer_vs_reference_golden(golden):
    from graph_odenet_amd.gat_layers import GraphConvolution, FixedGraphConvolution
    g = golden("gat_layer.npz")
    n, src, tgt, Mtgt = gat_inputs(g)
    for cls in (GraphConvolution, FixedGraphConvolution):
        layer = cls(g["x"].shape[1], g["f_w"].shape[0]).to(dev())
        layer.load_state_dict({"f.weight": T(g["f_w"]), "f.bias": T(g["f_b"]), "w.weight": T(g["w_w"]), "w.bias": T(g["w_b"])})
        x = T(g["x"]).to(dev()).requires_grad_(True)
        if cls is GraphConvolution:
            out = layer(x, src, tgt, Mtgt)
        else:
            layer.set_adj(src, tgt, Mtgt)
            out = layer(x)
        close(out, g["out"], what="gat fwd")
        assert (out[-10:] == 0).all()          # nodes that are never a target: 0 / eps = 0
        out.backward(T(g["gout"]).to(dev()))
        close(x.grad, g["gx"], 2e-5, "gx")
        close(layer.f.weight.grad, g["g_f_w"], 2e-5, "g f.weight")
        close(layer.f.bias.grad, g["g_f_b"], 2e-5, "g f.bias")
        close(layer.w.weight.grad, g["g_w_w"], 2e-5, "g w.weight")
        close(layer.w.bias.grad, g["g_w_b"], 2e-5, "g w.bias")


def test_gat_odefunc_vs_reference_golden(golden):
    from graph_odenet_amd.gat_models import ODEfunc
    g = golden("gat_odefunc.npz")
    n, src, tgt, Mtgt = gat_inputs(g)
    d = g["x"].shape[1]
    f = ODEfunc(d).to(dev())
    f.load_state_dict({"norm1.weight": T(g["gn_w"]), "norm1.bias": T(g["gn_b"]), "gc1.f.weight": T(g["f_w"]),
                       "gc1.f.bias": T(g["f_b"]), "gc1.w.weight": T(g["w_w"]), "gc1.w.bias": T(g["w_b"])})
    f.set_adj(src, tgt, Mtgt)
    x = T(g["x"]).to(dev()).requires_grad_(True)
    out = f(torch.tensor(float(g["t"]), device=dev()), x)
    close(out, g["out"], what="gat odefunc fwd")
    out.backward(T(g["gout"]).to(dev()))
    # d = 64: two channels per GroupNorm group (ill-conditioned backward)
    close(x.grad, g["gx"], 1e-4, "gx")
    close(f.gc1.f.weight.grad, g["g_f_w"], 1e-4, "g f.weight")
    close(f.gc1.w.weight.grad, g["g_w_w"], 1e-4, "g w.weight")
    close(f.norm1.weight.grad, g["g_gn_w"], 1e-3, "g norm1.weight")      # reference computed these on the CPU
    close(f.norm1.bias.grad, g["g_gn_b"], 1e-3, "g norm1.bias")


def test_gat_ode_block_rk4_vs_oracle_on_citeseer_edges(golden):
    """Citeseer's real edge list (GAT/utils.py:187-196 semantics), GAT ODEBlock with rk4: product vs
    the oracle solver driving the oracle layer."""
    from graph_odenet_amd.gat_models import ODEBlock, ODEfunc
    from oracle import layers_ref as R, solver_ref as S
    ge = golden("citeseer_gat_edges.npz")
    n = int(ge["n"])
    src, tgt = T(ge["src"]).long(), T(ge["tgt"]).long()
    e = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([T(ge["m_rows"]).long(), T(ge["m_cols"]).long()]), T(ge["m_vals"]), (n, e))
    d = 128
    torch.manual_seed(3)
    blk = ODEBlock(ODEfunc(d), method="rk4", step_size=0.25)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    x0 = torch.randn(n, d)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.ParameterList([torch.nn.Parameter(sd["odefunc." + k].clone()) for k in
                                             ("norm1.weight", "norm1.bias", "gc1.f.weight", "gc1.f.bias", "gc1.w.weight", "gc1.w.bias")])

        def forward(self, t, x):
            return R.gat_odefunc(t, x, src, tgt, Mtgt, *self.p)
    ref = S.odeint(F(), x0, torch.tensor([0., 1.]), method="rk4", options={"step_size": 0.25})[1]
    blk = blk.to(dev())
    xg = x0.to(dev()).requires_grad_(True)
    out = blk(xg, src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    assert blk.nfe == 16
    close(out, ref, 2e-5, "GAT rk4")
    out.sum().backward()
    assert torch.isfinite(xg.grad).all()


def qc_inputs(g):
    n = int(g["n"])
    Esrc, etgt = T(g["Esrc"]).long(), T(g["etgt"]).long()
    e = Esrc.numel()
    Etgt = torch.zeros(n, e)
    Etgt[etgt, torch.arange(e)] = 1.0
    return n, Esrc.to(dev()), Etgt.to(dev())


def test_qc_mpnn_and_edge_gcn_vs_reference_golden(golden):
    from graph_odenet_amd.qc_layers import MPNN_enn_edge, EdgeGraphConvolution
    g = golden("qc_layers.npz")
    n, Esrc, Etgt = qc_inputs(g)
    h = g["x"].shape[1]
    for TT in (1, 3):
        m = MPNN_enn_edge(5, h).to(dev())
        m.set_T(TT)
        m.update_net.load_state_dict({k: T(g["T%d__gru__%s" % (TT, k)]) for k in m.update_net.state_dict()})
        x = T(g["x"]).to(dev()).requires_grad_(True)
        ed = T(g["edge_data"]).to(dev()).requires_grad_(True)
        out = m(x, Esrc, Etgt, ed)
        close(out, g["T%d__out" % TT], what="mpnn fwd T=%d" % TT)
        out.backward(T(g["T%d__gout" % TT]).to(dev()))
        close(x.grad, g["T%d__gx" % TT], 2e-5, "mpnn gx")
        close(ed.grad, g["T%d__gedge" % TT], 2e-5, "mpnn gedge")
        for k, p in m.update_net.named_parameters():
            close(p.grad, g["T%d__ggru__%s" % (TT, k)], 2e-5, "gru " + k)
    egc = EdgeGraphConvolution(h, h).to(dev())
    egc.load_state_dict({"weight": T(g["egc__weight"]), "bias": T(g["egc__bias"])})
    x = T(g["x"]).to(dev()).requires_grad_(True)
    ed = T(g["edge_data"]).to(dev()).requires_grad_(True)
    out = egc(x, Esrc, Etgt, ed)
    close(out, g["egc__out"], what="egc fwd")
    out.backward(T(g["egc__gout"]).to(dev()))
    close(x.grad, g["egc__gx"], 2e-5, "egc gx"); close(ed.grad, g["egc__gedge"], 2e-5, "egc gedge")
    close(egc.weight.grad, g["egc__gw"], 2e-5, "egc gw"); close(egc.bias.grad, g["egc__gb"], 2e-5, "egc gb")


@pytest.mark.parametrize("n,h,bias", [(360, 73, True), (37, 16, True), (1000, 200, True), (5, 8, False), (700, 73, True),
                                      (0, 12, True)])
def test_fused_gru_update_vs_torch_cpu(n, h, bias):
    """SURVEY 8(f) N2, GRU half: x' = update_net(cat([x, m], 1), x) of QC/mpnn.py:30 as the fused cell of csrc/gru.hip,
    on the nn.GRUCell's own parameters, against torch's CPU GRUCell: output, input gradients, parameter gradients."""
    from graph_odenet_amd.qc_layers import gru_update
    torch.manual_seed(n + h)
    cell = torch.nn.GRUCell(2 * h, h, bias=bias)
    x = torch.randn(n, h, requires_grad=True)
    m = torch.randn(n, h, requires_grad=True)
    gout = torch.randn(n, h)
    ref = cell(torch.cat([x, m], 1), x)
    ref.backward(gout)
    want = [x.grad.clone(), m.grad.clone()] + [p.grad.clone() for p in cell.parameters()]
    cell_g = torch.nn.GRUCell(2 * h, h, bias=bias)
    cell_g.load_state_dict(cell.state_dict())
    cell_g = cell_g.to(dev())
    xg, mg = x.detach().to(dev()).requires_grad_(True), m.detach().to(dev()).requires_grad_(True)
    out = gru_update(cell_g, xg, mg)
    close(out, ref, 1e-5, "GRU output")
    out.backward(gout.to(dev()))
    got = [xg.grad, mg.grad] + [p.grad for p in cell_g.parameters()]
    names = ["dx", "dm"] + [k for k, _ in cell_g.named_parameters()]
    for nm, a, b in zip(names, got, want):
        close(a, b, 2e-5, nm)


@pytest.mark.parametrize("name", ["EdgeGCN_K_Sum", "MPNN_ENN_K_Set2Set"])
def test_c4_full_size_training_steps_match_reference(golden, name):
    """BASELINE.json configs[3] / SURVEY 8(d) C4 at its real size: 20 synthetic molecules, h = 73, T = 3, the
    reference's 14.4 M-parameter shapes; three Adam steps (MSE, lr 1e-3, QC/train_egcn.py:122) of the reference's own
    classes (tests/golden/train_traj_qc_c4.npz, weights regenerated by name) against ours: loss sequence and outputs.
    Batch-20 sizes run the fused < 4096-edge message kernel, the fused GRU cell and the Set2Set segment kernels."""
    import os
    import sys
    import torch.nn.functional as F
    from graph_odenet_amd import qc_models
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from named_init import fill_by_name
    g = golden("train_traj_qc_c4.npz")
    n = int(g["n"])
    x, ef, tgt = T(g["x"]).to(dev()), T(g["ef"]).to(dev()), T(g["target"]).to(dev())
    Esrc, batch = T(g["Esrc"]).long().to(dev()), T(g["batch"]).long().to(dev())
    E = Esrc.numel()
    assert E < 4096 and n == x.shape[0]
    Etgt = torch.zeros(n, E)
    Etgt[T(g["etgt"]).long(), torch.arange(E)] = 1.0
    Etgt = Etgt.to(dev())
    m = getattr(qc_models, name)(node_features=13, edge_features=5, target_features=12, hidden_features=73, num_layers=3,
                                 s2s_processing_steps=12, dropout=0.0)
    assert sum(p.numel() for p in m.parameters()) == int(g[name + "__n_params"])
    m = fill_by_name(m).to(dev()).train()
    from graph_odenet_amd.optim import Adam
    opt = Adam(m.parameters(), lr=1e-3)                              # the product's one-launch Adam against the reference's torch.optim.Adam steps
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = F.mse_loss(m(x, ef, Esrc, Etgt, batch), tgt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    ref = np.asarray(g[name + "__losses"])
    assert np.abs(np.asarray(losses) - ref).max() < 5e-5 * max(1.0, float(ref.max())), (losses, ref)
    out = m(x, ef, Esrc, Etgt, batch).detach()
    close(out, g[name + "__out"], 2e-4, "outputs after three steps")


@pytest.mark.parametrize("name", ["EdgeGCN_K_Sum", "MPNN_ENN_K_Set2Set"])
def test_shape_bucket_padding_leaves_the_batch_unchanged(name):
    """qc_batch.pad_batch: a batch rounded up to its shape bucket (one extra graph of isolated dummy atoms and dummy
    edges) gives the original outputs in its first n_graphs rows and the original parameter gradients."""
    import torch.nn.functional as F
    from graph_odenet_amd import qc_models
    from graph_odenet_amd.qc_batch import pad_batch
    from graph_odenet_amd.synth import qm9_like_batch
    x, ef, Esrc, Etgt, batch = qm9_like_batch(7, seed=4, device=dev())
    tgt = torch.randn(7, 12, generator=torch.Generator().manual_seed(1)).to(dev())
    torch.manual_seed(2)
    m = getattr(qc_models, name)(node_features=13, edge_features=5, target_features=12, hidden_features=24, num_layers=2,
                                 s2s_processing_steps=3, dropout=0.0).to(dev())
    out = m(x, ef, Esrc, Etgt, batch)
    F.mse_loss(out, tgt).backward()
    want = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    xp, efp, srcp, Etp, bp, nb = pad_batch(x, ef, Esrc, Etgt, batch, node_multiple=64, edge_multiple=128)
    assert nb == 7 and xp.shape[0] % 64 == 0 and srcp.numel() % 128 == 0 and xp.shape[0] > x.shape[0]
    outp = m(xp, efp, srcp, Etp, bp)
    assert outp.shape[0] == 8
    close(outp[:nb], out, 1e-5, "outputs of the real graphs")
    F.mse_loss(outp[:nb], tgt).backward()
    for (nm, p), w in zip(m.named_parameters(), want):
        close(p.grad, w, 2e-5, nm)


@pytest.mark.parametrize("name", ["EdgeGCN_K_Sum", "MPNN_ENN_K_Set2Set"])
def test_captured_qc_step_matches_eager(name):
    """qc_step.CapturedQCStep: the whole training step of a shape bucket (graph conversion, forward, MSE, backward, Adam)
    captured into one HIP graph per bucket and replayed on new batches, two buckets interleaved - losses and final
    parameters equal the eager run of the same padded batches."""
    import torch.nn.functional as F
    from graph_odenet_amd import hipgraph, qc_models
    from graph_odenet_amd.qc_batch import pad_batch
    from graph_odenet_amd.qc_step import CapturedQCStep
    from graph_odenet_amd.synth import qm9_like_batch
    if not hipgraph.memset_nodes_ok():
        pytest.skip("this process runs HIP's graph fast path, on which replayed memset nodes are unreliable "
                    "(hipgraph.py): CapturedQCStep stays eager by design")
    batches = []
    for b in range(60):
        x, ef, Esrc, Etgt, batch = qm9_like_batch(6, seed=100 + b, device=dev())
        tgt = torch.randn(6, 12, generator=torch.Generator().manual_seed(b)).to(dev())
        batches.append(pad_batch(x, ef, Esrc, Etgt, batch, 64, 128)[:5] + (tgt,))
    shapes = {}
    for i, bt in enumerate(batches):
        shapes.setdefault((bt[0].shape, bt[1].shape), []).append(i)
    big = sorted(shapes.values(), key=len, reverse=True)
    assert len(big) >= 2 and len(big[1]) >= 3, "need two buckets with several batches each"
    # batches of the two most frequent buckets, interleaved: every switch re-points the parameters' gradients to the
    # other graph's tensors
    idx = [i for pair in zip(big[0][:4], big[1][:4]) for i in pair]
    res = {}
    for captured in (False, True):
        torch.manual_seed(3)
        m = getattr(qc_models, name)(node_features=13, edge_features=5, target_features=12, hidden_features=24,
                                     num_layers=2, s2s_processing_steps=3, dropout=0.0).to(dev())
        from graph_odenet_amd.optim import Adam
        opt = Adam(m.parameters(), lr=1e-3)
        losses = []
        if captured:
            step = CapturedQCStep(m, opt, F.mse_loss, warmup=1)
            for i in idx:
                losses.append(float(step(*batches[i])))
            assert len(step.buckets) == 2 and all(b.graph is not None for b in step.buckets.values())
        else:
            for i in idx:
                x, ef, Esrc, Etgt, batch, tgt = batches[i]
                opt.zero_grad(set_to_none=True)
                loss = F.mse_loss(m(x, ef, Esrc, Etgt, batch)[:6], tgt)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        res[captured] = (losses, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone())
    assert np.abs(np.asarray(res[True][0]) - np.asarray(res[False][0])).max() < 1e-5 * max(1.0, max(res[False][0]))
    close(res[True][1], res[False][1], 1e-5, "parameters after the steps")


@pytest.mark.parametrize("name", ["EdgeGCN_K_Sum", "MPNN_ENN_K_Set2Set"])
def test_loader_prepared_batch_equals_the_dense_route(name):
    """qc_batch.prepare: edges handed over as index vectors (no dense N x E matrix, no host synchronisation) give the
    outputs and gradients of the reference's (Esrc, dense Etgt, batch) call."""
    import torch.nn.functional as F
    from graph_odenet_amd import qc_models
    from graph_odenet_amd.qc_batch import prepare
    from graph_odenet_amd.synth import qm9_like_batch
    x, ef, Esrc, Etgt, batch = qm9_like_batch(9, seed=13, device=dev())
    tgt = torch.randn(9, 12, generator=torch.Generator().manual_seed(4)).to(dev())
    torch.manual_seed(5)
    m = getattr(qc_models, name)(node_features=13, edge_features=5, target_features=12, hidden_features=24, num_layers=2,
                                 s2s_processing_steps=3, dropout=0.0).to(dev())
    out = m(x, ef, Esrc, Etgt, batch)
    F.mse_loss(out, tgt).backward()
    want = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    edges, b2 = prepare(Esrc, Etgt.argmax(0), batch.clone(), x.shape[0], 9)
    out2 = m(x, ef, Esrc, edges, b2)
    assert torch.equal(out2, out)
    F.mse_loss(out2, tgt).backward()
    for (nm, p), w in zip(m.named_parameters(), want):
        close(p.grad, w, 1e-6, nm)


def test_message_chain_forms_the_edge_matrix_gradient_once():
    """qc_layers.MessageChain: T message steps on the same edge matrices (QC/mpnn.py:27-30).  With a chain the backward of a
    step records (dM, x) and the first step's backward - the last to run - forms sum_t val dM_t[tgt] (x) x_t[src] in one pass
    (gode_edge_outer_sum_f32); without it autograd adds T arrays.  Same gradients to fp32 summation accuracy, against
    float64 too; a step that runs its backward after the flush (two independent branches on one chain) still contributes."""
    from graph_odenet_amd import qc_layers as QL
    from graph_odenet_amd.synth import qm9_like_batch
    x0, ef, Esrc, Etgt, batch = qm9_like_batch(6, seed=3)
    n, E, h = x0.shape[0], Esrc.numel(), 21
    g = torch.Generator().manual_seed(0)
    A = torch.randn(E, h, h, generator=g) / h ** 0.5
    x = torch.randn(n, h, generator=g)
    W = [torch.randn(h, h, generator=g) / h ** 0.5 for _ in range(3)]
    gout = torch.randn(n, h, generator=g)
    D = dev()

    def run(chain_cls, dtype=torch.float32, device=D):
        Ad = A.to(device=device, dtype=dtype).requires_grad_(True)
        xd = x.to(device=device, dtype=dtype).requires_grad_(True)
        if device == "cpu":                                     # float64 reference: the reference's formula
            Et = Etgt.to(dtype)
            cur = xd
            for t in range(3):
                m = Et @ torch.bmm(Ad, cur[Esrc].unsqueeze(2)).squeeze(2)
                cur = torch.tanh(m @ W[t].to(dtype)) + cur
        else:
            chain = chain_cls() if chain_cls else None
            cur = xd
            for t in range(3):
                m = QL.edge_message(cur, Esrc.to(D), Etgt.to(D), Ad, chain)
                cur = torch.tanh(m @ W[t].to(D)) + cur
        cur.backward(gout.to(device=device, dtype=dtype))
        return Ad.grad.detach().cpu().double(), xd.grad.detach().cpu().double()

    ref = run(None, torch.float64, "cpu")
    plain = run(None)
    chained = run(QL.MessageChain)
    for got in (plain, chained):
        for a, b in zip(got, ref):
            assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    assert (chained[0] - plain[0]).abs().max().item() <= 2e-6 * max(1.0, plain[0].abs().max().item())
    assert torch.equal(chained[1], plain[1])                    # dx does not go through the chain
    # two INDEPENDENT uses on one chain: the second branch's backward may run after the first step flushed
    chain = QL.MessageChain()
    Ad = A.to(D).requires_grad_(True)
    xa, xb = x.to(D).requires_grad_(True), (0.5 * x).to(D).requires_grad_(True)
    ma = QL.edge_message(xa, Esrc.to(D), Etgt.to(D), Ad, chain)
    mb = QL.edge_message(xb, Esrc.to(D), Etgt.to(D), Ad, chain)
    (ma * gout.to(D)).sum().backward(retain_graph=True)
    (mb * gout.to(D)).sum().backward()
    got = Ad.grad.clone()
    Ad.grad = None
    ma2 = QL.edge_message(xa, Esrc.to(D), Etgt.to(D), Ad)
    mb2 = QL.edge_message(xb, Esrc.to(D), Etgt.to(D), Ad)
    ((ma2 + mb2) * gout.to(D)).sum().backward()
    assert (got - Ad.grad).abs().max().item() <= 2e-6 * max(1.0, Ad.grad.abs().max().item())


def test_gru_chain_sums_the_weight_gradient_partials_once():
    """qc_layers.GruChain: T applications of one nn.GRUCell(2h, h) in a message-passing loop (QC/mpnn.py:30).  With a chain
    every step's backward writes its weight-gradient partials into one buffer and the first step's backward sums all partial
    rows once (gode_gru_wreduce_f32); without it autograd adds T gradients per parameter.  Same numbers to fp32 summation
    accuracy, and against torch's own GRUCell on the CPU in float64."""
    from graph_odenet_amd import qc_layers as QL
    h, n, T = 29, 333, 3
    g = torch.Generator().manual_seed(5)
    cell = torch.nn.GRUCell(2 * h, h)
    x0 = torch.randn(n, h, generator=g)
    ms = [torch.randn(n, h, generator=g) for _ in range(T)]
    gout = torch.randn(n, h, generator=g)
    c64 = torch.nn.GRUCell(2 * h, h).double()
    c64.load_state_dict({k: v.double() for k, v in cell.state_dict().items()})
    xr = x0.double().requires_grad_(True)
    cur = xr
    for t in range(T):
        cur = c64(torch.cat([cur, ms[t].double()], 1), cur)
    cur.backward(gout.double())
    want = [xr.grad] + [p.grad for p in c64.parameters()]
    D = dev()
    cell = cell.to(D)
    res = {}
    for use_chain in (False, True):
        for p in cell.parameters():
            p.grad = None
        xd = x0.to(D).requires_grad_(True)
        chain = QL.GruChain() if use_chain else None
        cur = xd
        for t in range(T):
            cur = QL.gru_update(cell, cur, ms[t].to(D), chain)
        cur.backward(gout.to(D))
        res[use_chain] = [xd.grad.cpu().double()] + [p.grad.cpu().double() for p in cell.parameters()]
        if use_chain:
            assert chain.flushed and chain.written == T
    for got in res.values():
        for a, b in zip(got, want):
            assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    assert torch.equal(res[True][0], res[False][0])                # dx does not go through the chain
    for a, b in zip(res[True][1:], res[False][1:]):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())


def test_qc_colliding_indices_and_weighted_incidence():
    """Q5 of SURVEY.md: the reference's batches do not offset node ids, so many edges collide on the
    first nodes; Etgt is a dense float matrix whose values are used as weights."""
    from graph_odenet_amd.qc_layers import edge_message
    from oracle import layers_ref as R
    torch.manual_seed(0)
    n, e, h = 60, 400, 20
    Esrc = torch.randint(0, 9, (e,))                       # everything indexes the first 9 nodes
    etgt = torch.randint(0, 9, (e,))
    Etgt = torch.zeros(n, e)
    Etgt[etgt, torch.arange(e)] = torch.rand(e) + 0.5      # weighted incidence
    x = torch.randn(n, h, requires_grad=True)
    A = (torch.randn(e, h, h) * 0.2).requires_grad_(True)
    ref = R.edge_message_aggregate(x, Esrc, Etgt, A)
    gout = torch.randn(n, h)
    ref.backward(gout)
    xg = x.detach().to(dev()).requires_grad_(True)
    Ag = A.detach().to(dev()).requires_grad_(True)
    out = edge_message(xg, Esrc.to(dev()), Etgt.to(dev()), Ag)
    close(out, ref, 2e-5, "colliding fwd")
    out.backward(gout.to(dev()))
    close(xg.grad, x.grad, 2e-5, "colliding gx"); close(Ag.grad, A.grad, 2e-5, "colliding gA")


def test_segment_sum_kat(golden):
    """QC/torch_scatter.py:207-218 known answer, through the SpMM kernel as a segment-sum."""
    from graph_odenet_amd import graph as G, ops
    g = golden("scatter_kat.npz")
    src, index, want = T(g["src"]), T(g["index"]), T(g["out"])
    for r in range(2):
        inc = G.incidence_from_index(index[r].to(dev()), 6)
        got = ops.spmm(inc, src[r].view(-1, 1).contiguous().to(dev())).view(-1)
        assert torch.equal(got.cpu(), want[r])


def test_qc_synthetic_batch_mpnn_vs_oracle():
    """C4: synthetic QM9-like batch (20 graphs, h=73, T=3, correctly offset indices) - product vs oracle,
    forward and gradients w.r.t. node states and edge matrices."""
    from graph_odenet_amd.qc_layers import MPNN_enn_edge
    from graph_odenet_amd.synth import qm9_like_batch
    from oracle import layers_ref as R
    torch.manual_seed(0)
    x, ef, Esrc, Etgt, batch = qm9_like_batch(20, seed=1)
    h = 73
    hx = torch.randn(x.shape[0], h, requires_grad=True)
    A = (torch.randn(Esrc.numel(), h, h) * 0.1).requires_grad_(True)
    m = MPNN_enn_edge(5, h)
    m.set_T(3)
    ref = R.mpnn_enn_edge(hx, Esrc, Etgt, A, m.update_net, 3)
    gout = torch.randn_like(ref)
    ref.backward(gout)
    gref = {k: p.grad.clone() for k, p in m.update_net.named_parameters()}
    mg = MPNN_enn_edge(5, h).to(dev())
    mg.set_T(3)
    mg.load_state_dict(m.state_dict())
    hg = hx.detach().to(dev()).requires_grad_(True)
    Ag = A.detach().to(dev()).requires_grad_(True)
    out = mg(hg, Esrc.to(dev()), Etgt.to(dev()), Ag)
    close(out, ref, 2e-5, "qc batch fwd")
    out.backward(gout.to(dev()))
    close(hg.grad, hx.grad, 5e-5, "qc batch gx"); close(Ag.grad, A.grad, 5e-5, "qc batch gA")
    for k, p in mg.update_net.named_parameters():
        close(p.grad, gref[k], 5e-5, "gru " + k)


def test_gat_empty_and_isolated():
    """Edge cases: a graph with no edges gives all-zero output (0/eps); duplicate (src,tgt) pairs are
    distinct edges (the reference's edge list may repeat pairs)."""
    from graph_odenet_amd.gat_layers import GraphConvolution
    from oracle import layers_ref as R
    torch.manual_seed(0)
    n, fi, fo = 30, 6, 8
    layer = GraphConvolution(fi, fo)
    x = torch.randn(n, fi)
    src = torch.tensor([0, 0, 3, 3, 5], dtype=torch.int64)
    tgt = torch.tensor([1, 1, 2, 2, 5], dtype=torch.int64)        # repeated pairs + a self loop
    e = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e)]), torch.ones(e), (n, e))
    ref = R.gat_layer(x, src, tgt, Mtgt, layer.f.weight, layer.f.bias, layer.w.weight, layer.w.bias)
    lg = GraphConvolution(fi, fo).to(dev())
    lg.load_state_dict(layer.state_dict())
    out = lg(x.to(dev()), src.to(dev()), tgt.to(dev()), Mtgt.to(dev()))
    close(out, ref, what="gat small")
    src0 = torch.zeros(0, dtype=torch.int64, device=dev())
    M0 = torch.sparse_coo_tensor(torch.zeros(2, 0, dtype=torch.int64), torch.zeros(0), (n, 0)).to(dev())
    out0 = lg(x.to(dev()), src0, src0, M0)
    assert out0.shape == (n, fo) and (out0 == 0).all()


@pytest.mark.parametrize("method,opts", [("rk4", {"step_size": 0.25}), (None, None)])
def test_gat_fused_field_matches_autograd_path(golden, method, opts):
    """The fused GAT ODE field (gat_ode.py) against the same module driven through torch.autograd
    (generic field), on Citeseer's edge list: states equal to rounding, all six parameter gradients equal."""
    from graph_odenet_amd import gat_models, odeint as OI
    ge = golden("citeseer_gat_edges.npz")
    n = int(ge["n"])
    src, tgt = T(ge["src"]).long().to(dev()), T(ge["tgt"]).long().to(dev())
    e = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(e, device=dev())]), torch.ones(e, device=dev()), (n, e))
    d = 128
    torch.manual_seed(5)
    f = gat_models.ODEfunc(d).to(dev())
    f.set_adj(src, tgt, Mtgt)
    x0 = torch.randn(n, d, device=dev())
    t = torch.tensor([0., 1.], device=dev())
    res = {}
    for fused in (True, False):
        hook = gat_models.ODEfunc.gode_fields
        if not fused:
            gat_models.ODEfunc.gode_fields = lambda self, y0: None
        try:
            f.zero_grad(); f.nfe = 0
            xi = x0.clone().requires_grad_(True)
            out = OI.odeint_adjoint(f, xi, t, 1e-5, 1e-5, method, opts)[1]
            out.square().sum().backward()
            res[fused] = (out.detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in f.named_parameters()}, f.nfe)
        finally:
            gat_models.ODEfunc.gode_fields = hook
    tol = 2e-5 if method == "rk4" else 2e-3
    close(res[True][0], res[False][0], tol, "state")
    close(res[True][1], res[False][1], tol * 5, "gx")
    for k in res[True][2]:
        close(res[True][2][k], res[False][2][k], tol * 5, "grad " + k)
    if method == "rk4":
        assert res[True][3] == res[False][3] == 16 + 17         # the adjoint count includes torchdiffeq's dL/dt evaluation (odeint.py)


@pytest.mark.parametrize("name", ["MPNN_ENN_K_Sum", "MPNN_ENN_K_Set2Set", "EdgeGCN_K_Sum", "EdgeGCN_K_Set2Set",
                                  "EdgeRES1_K_Set2Set", "MPNN_ENN_Sum", "MPNN_ENN_Set2Set", "EdgeGCN3_Sum",
                                  "EdgeGCN3_Set2Set"])
def test_qc_model_zoo_vs_reference_golden(golden, name):
    """QC/layer_models.py classes: same state_dict keys (checkpoint-compatible), outputs and the captured
    gradients of the reference classes on a synthetic 4-molecule batch."""
    from graph_odenet_amd import qc_models
    g = golden("qc_models.npz")
    n = int(g["n"])
    x, ef = T(g["x"]).to(dev()), T(g["ef"]).to(dev())
    Esrc, etgt, batch = T(g["Esrc"]).long().to(dev()), T(g["etgt"]).long(), T(g["batch"]).long().to(dev())
    e = Esrc.numel()
    Etgt = torch.zeros(n, e)
    Etgt[etgt, torch.arange(e)] = 1.0
    Etgt = Etgt.to(dev())
    if "_K_" in name:        # QC/layer_models.py
        m = getattr(qc_models, name)(node_features=13, edge_features=5, target_features=12, hidden_features=16,
                                     num_layers=3, s2s_processing_steps=3, dropout=0.0).to(dev())
    else:                    # QC/models.py: (node_features, edge_features, hidden_features, out_features, ...)
        kw = dict(processing_steps=3) if "Set2Set" in name else {}
        m = getattr(qc_models, name)(13, 5, 16, 12, **kw).to(dev())
    pre = name + "__sd__"
    sd = {k[len(pre):].replace("__", "."): T(v) for k, v in g.items() if k.startswith(pre)}
    assert set(sd) == set(m.state_dict().keys())
    m.load_state_dict(sd)
    m.train()          # dropout is 0.0 here; MIOpen's LSTM backward refuses eval mode
    out = m(x, ef, Esrc, Etgt, batch)
    close(out, g[name + "__out"], 2e-5, name + " out")
    out.backward(T(g[name + "__gout"]).to(dev()))
    pre = name + "__g__"
    checked = 0
    params = dict(m.named_parameters())
    for k, v in g.items():
        if k.startswith(pre):
            # hidden 16 -> one channel per GroupNorm group in RESKnorm (noise-floor gradients)
            tol = 2e-3 if "RES1" in name else 5e-5
            close(params[k[len(pre):].replace("__", ".")].grad, v, tol, name + " grad " + k[len(pre):])
            checked += 1
    assert checked >= 4


@pytest.mark.parametrize("h,sorted_batch", [(16, True), (73, True), (73, False), (200, False), (1024, True)])
def test_segment_attention_vs_oracle(h, sorted_batch):
    """Set2Set's per-graph softmax readout (QC/set2set.py:63-74) through the C ABI against the oracle's restatement of
    the reference loop, forward and backward; graphs of 1..40 nodes, one graph id without nodes, unsorted batch vector."""
    from graph_odenet_amd import ops
    from graph_odenet_amd.qc_models import _Segments
    from oracle import layers_ref as R
    gen = torch.Generator().manual_seed(h + int(sorted_batch))
    sizes = [1, 40, 0, 7, 23, 2, 64, 5]
    batch = torch.cat([torch.full((n,), b, dtype=torch.int64) for b, n in enumerate(sizes)])
    if not sorted_batch:
        batch = batch[torch.randperm(batch.numel(), generator=gen)]
    nb, n = len(sizes), batch.numel()
    x = torch.randn(n, h, generator=gen)
    q = torch.randn(nb, h, generator=gen) * (2.0 / h ** 0.5)
    dr = torch.randn(nb, h, generator=gen)
    xo, qo = x.clone().requires_grad_(True), q.clone().requires_grad_(True)
    a_ref, r_ref = R.set2set_attention(xo, qo, batch, nb)
    r_ref.backward(dr)
    seg = _Segments(batch.to(dev()))
    assert (seg.perm is None) == sorted_batch
    xd, qd = x.to(dev()), q.to(dev())
    a, r = ops.segment_attention_fwd(seg.segptr, seg.perm, xd, qd)
    close(a, a_ref, TOL, "attention weights")
    close(r, r_ref, TOL, "readout")
    dx, dq = ops.segment_attention_bwd(seg.segptr, seg.perm, xd, qd, a, dr.to(dev()))
    close(dx, xo.grad, 2e-5, "dx")
    close(dq, qo.grad, 2e-5, "dq")
    assert r[2].abs().max().item() == 0.0 and dq[2].abs().max().item() == 0.0      # the empty graph


def test_segment_sum_matches_index_add():
    from graph_odenet_amd.qc_models import segment_sum
    gen = torch.Generator().manual_seed(3)
    batch = torch.randint(0, 9, (300,), generator=gen)
    batch[0] = 8
    x = torch.randn(300, 12, generator=gen)
    ref = torch.zeros(9, 12).index_add_(0, batch, x)
    xd = x.to(dev()).requires_grad_(True)
    out = segment_sum(xd, batch.to(dev()))
    close(out, ref, TOL, "segment sum")
    g = torch.randn(9, 12, generator=gen)
    out.backward(g.to(dev()))
    close(xd.grad, g[batch], 0.0, "segment sum backward")


def test_set2set_module_vs_reference_golden(golden):
    """graph_odenet_amd.qc_models.Set2Set (segment-attention kernel + fused LSTM cell on the reference's `lstm.*`
    parameters) against q_star and the gradients captured from the reference's Set2Set (QC/set2set.py:6-75)."""
    from graph_odenet_amd.qc_models import Set2Set
    g = golden("set2set.npz")
    m = Set2Set(24, 4, 1)
    sd = {k[len("sd__"):].replace("__", "."): T(v) for k, v in g.items() if k.startswith("sd__")}
    assert set(sd) == set(m.state_dict().keys())
    m.load_state_dict(sd)
    m = m.to(dev())
    x = T(g["x"]).to(dev()).requires_grad_(True)
    out = m(x, T(g["batch"]).long().to(dev()))
    close(out, g["out"], 1e-5, "q_star")
    out.backward(T(g["gout"]).to(dev()))
    close(x.grad, g["gx"], 2e-5, "gx")
    for k, p in m.named_parameters():
        close(p.grad, g["g__" + k.replace(".", "__")], 2e-5, "grad " + k)


@pytest.mark.parametrize("sorted_batch", [True, False])
def test_set2set_one_node_matches_per_step_path(sorted_batch):
    """The readout loop as ONE autograd node (ops.set2set_fwd / set2set_bwd: the LSTM cell writes q_t into q_star, the
    weight gradients of the processing steps collect inside the cell's backward kernel, the node cotangent inside the
    attention's) against the chain of per-step nodes: q_star and every gradient, a sorted and an unsorted batch vector, a
    graph without nodes, 12 processing steps."""
    from graph_odenet_amd import qc_models as Q
    g = torch.Generator().manual_seed(3)
    n, h, nb = 230, 73, 9
    batch = torch.randint(0, nb, (n,), generator=g)
    batch[batch == 4] = 5                                        # graph 4 has no nodes
    batch[0] = nb - 1                                            # the largest id is present: nb graphs
    if sorted_batch:
        batch = batch.sort().values
    x0 = torch.randn(n, h, generator=g)
    gout = torch.randn(nb, 2 * h, generator=g)
    res = []
    for one in (True, False):
        Q.SET2SET_ONE_NODE = one
        try:
            torch.manual_seed(1)
            m = Q.Set2Set(h, 12, 1).to(dev())
            x = x0.to(dev()).requires_grad_(True)
            b = batch.to(dev())
            out = m(x, b)
            (out * gout.to(dev())).sum().backward()
            res.append((out.detach().cpu(), x.grad.cpu(), [p.grad.cpu() for p in m.parameters()]))
        finally:
            Q.SET2SET_ONE_NODE = True
    close(res[0][0], res[1][0], 1e-6, "q_star")
    close(res[0][1], res[1][1], 1e-5, "dx")
    for a, b_ in zip(res[0][2], res[1][2]):
        close(a, b_, 1e-5, "lstm parameter gradient")


def test_gat_hip_graph_captured_solves_match_eager(golden):
    """Same as the GCN capture test for the fused GAT field (Python-driven stages, weights re-packed inside the graph)."""
    from graph_odenet_amd import gat_models, odeint as OI
    ge = golden("citeseer_gat_edges.npz")
    n, src, tgt, Mtgt = gat_inputs(ge)
    d = 16
    xs = [torch.randn(n, d, generator=torch.Generator().manual_seed(i)).to(dev()) for i in range(4)]
    res = {}
    old = OI.GRAPH_CAPTURE_MAX_ELEMS
    for capture in (True, False):
        OI.GRAPH_CAPTURE_MAX_ELEMS = old if capture else 0
        try:
            torch.manual_seed(1)
            blk = gat_models.ODEBlock(gat_models.ODEfunc(d), method="rk4", step_size=0.25).to(dev())
            opt = torch.optim.SGD(blk.parameters(), lr=0.05)
            log = []
            for x in xs:
                opt.zero_grad()
                xi = x.clone().requires_grad_(True)
                out = blk(xi, src, tgt, Mtgt)
                out.square().mean().backward()
                log.append((out.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()]))
                opt.step()
            res[capture] = (log, blk.nfe)
            if capture:
                plans = list(OI.plans_of(blk.odefunc).values())
                assert len(plans) == 1 and plans[0].gf is not None and plans[0].gb is not None
        finally:
            OI.GRAPH_CAPTURE_MAX_ELEMS = old
    assert res[True][1] == res[False][1]
    for (o1, g1, p1), (o2, g2, p2) in zip(res[True][0], res[False][0]):
        close(o1, o2, 1e-6, "state"); close(g1, g2, 1e-6, "gx")
        for a, b in zip(p1, p2):
            close(a, b, 1e-6, "param grad")


@pytest.mark.parametrize("o", [7, 16, 128])
def test_gat_vjp_pieces_vs_torch(o):
    """gode_gat_scatter_f32 / gode_gat_maxpath_f32 / gode_time_row_fixup_f32 against index_add / argmax / dot on the
    CPU; random multigraph with a hub node (in-degree 300), isolated nodes and a tie for the maximum logit."""
    from graph_odenet_amd import ops
    from graph_odenet_amd.graph import incidence_from_index
    gen = torch.Generator().manual_seed(o)
    n, E = 200, 1500
    src = torch.randint(0, n - 10, (E,), generator=gen)
    tgt = torch.randint(0, n - 10, (E,), generator=gen)
    tgt[:300] = 3
    dz, da = torch.randn(E, o, generator=gen), torch.randn(E, generator=gen)
    a = torch.randn(E, generator=gen)
    a[700] = a[900] = a.max() + 1.0                      # two edges attain the maximum: the first one takes the path
    da_ref = da.clone()
    da_ref[700] -= da.sum()
    dad = da.to(dev())
    ops.gat_maxpath_(a.to(dev()), a.max().reshape(1).to(dev()), dad)
    close(dad, da_ref, 2e-5, "max path")
    Ms, Mt = incidence_from_index(src.to(dev()).to(torch.int32), n), incidence_from_index(tgt.to(dev()).to(torch.int32), n)
    dPs, dPt, dA2 = (torch.full((n, o), 7.0, device=dev()), torch.full((n, o), 7.0, device=dev()),
                     torch.full((n, 2), 7.0, device=dev()))
    ops.gat_scatter(Ms, Mt, dz.to(dev()), dad, dPs, dPt, dA2)
    close(dPs, torch.zeros(n, o).index_add_(0, src, dz), 2e-5, "dPs")
    close(dPt, torch.zeros(n, o).index_add_(0, tgt, dz), 2e-5, "dPt")
    close(dA2[:, 0], torch.zeros(n).index_add_(0, src, da_ref), 2e-5, "das")
    close(dA2[:, 1], torch.zeros(n).index_add_(0, tgt, da_ref), 2e-5, "dat")
    g0, w0 = torch.randn(o, generator=gen), torch.randn(o, generator=gen)
    at = torch.tensor([0.25]).to(dev())
    g0d = g0.to(dev())
    ops.time_row_fixup_(g0d, w0.to(dev()), 0.3, at, accumulate=True)
    close(at, torch.tensor([0.25 + float(g0 @ w0)]), 1e-5, "a_t")
    close(g0d, g0 * 0.3, 1e-6, "time row")


@pytest.mark.parametrize("name,kw", [("GCN", {}), ("RGCN2", {}), ("GCN3", {}), ("GCN3norm", {}), ("RGCN3", {}), ("RGCN3norm", {}),
                                     ("RGCN3fullnorm", {}), ("GCNK", dict(nlayers=4)), ("RESK2", dict(nlayers=5)),
                                     ("RESK1norm", dict(nlayers=4))])
def test_gat_model_zoo_vs_reference_golden(golden, name, kw):
    """GAT/models.py holds the whole zoo over (x, src, tgt, Mtgt) (GAT/train_res.py's model_dict names seven of them);
    gat_models re-binds the kit-generic classes of models.py to the edge-attention layers."""
    from graph_odenet_amd import gat_models
    g = golden("gat_zoo.npz")
    n = int(g["n"])
    src, tgt = T(g["src"]).long().to(dev()), T(g["tgt"]).long().to(dev())
    E = src.numel()
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E, device=dev())]), torch.ones(E, device=dev()), (n, E))
    m = getattr(gat_models, name)(nfeat=10, nhid=8, nclass=3, dropout=0.5, **kw)
    pre = name + "__sd__"
    sd = {k[len(pre):].replace("__", "."): T(v) for k, v in g.items() if k.startswith(pre)}
    assert set(sd) == set(m.state_dict().keys())
    m.load_state_dict(sd)
    m = m.to(dev()).eval()
    out = m(T(g["x"]).to(dev()), src, tgt, Mtgt)
    # hidden 8 -> one channel per GroupNorm group: y = x*scale + (beta - x*scale) with scale = 316*gamma is beta plus
    # ~2e-5 of rounding noise on either side; the norm variants are compared at that floor
    tol = 1e-4 if "norm" in name else 2e-5
    close(out, g[name + "__out"], tol, name + " out")
    out.backward(T(g["gout"]).to(dev()))
    first = m.gcs[0] if hasattr(m, "gcs") else m.gc1
    close(first.f.bias.grad, g[name + "__gbias0"], 5 * tol, name + " grad")


def test_gat_zoo_has_every_reference_class():
    from graph_odenet_amd import gat_models, models
    for name in models.ZOO + ("ODEfunc", "ODEfunc2", "ODEBlock"):
        assert hasattr(gat_models, name), name
    m = gat_models.ODEK2(nfeat=6, nhid=8, nclass=2, dropout=0.5, nlayers=5)
    assert type(m.gcs[1].odefunc) is gat_models.ODEfunc2 and type(m.gcs[2].odefunc) is gat_models.ODEfunc


@pytest.mark.parametrize("o", [16, 128])
def test_gat_layer_record_path_vs_oracle_large(o):
    """Above 65 536 targets the attention kernels run over nnz-balanced records (csrc/edge.hip: gat_agg_*_rec_kernel)
    on the target-sorted edge list.  70 000 nodes / 300 000 edges in random order, a 3 000-edge hub (split into
    records with a partial slab), nodes without incoming edges; forward and all gradients against the oracle's
    restatement of GAT/layers.py:31-58 evaluated in float64 on the CPU (in float32 torch's own CPU backward is 1e-2
    off at the hub for o = 128, tools/dev/gat_layer_check.py)."""
    from graph_odenet_amd.gat_layers import GraphConvolution, edge_graph
    from oracle import layers_ref as R
    gen = torch.Generator().manual_seed(o)
    n, E, i = 70000, 300000, 8
    src = torch.randint(0, n, (E,), generator=gen)
    tgt = torch.randint(0, n - 100, (E,), generator=gen)
    tgt[:3000] = 5
    p = torch.randperm(E, generator=gen)
    src, tgt = src[p], tgt[p]
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E)]), torch.ones(E), (n, E))
    x = torch.randn(n, i, generator=gen)
    gout = torch.randn(n, o, generator=gen)
    torch.manual_seed(1)
    lay = GraphConvolution(i, o)
    ref_p = [q.detach().double().requires_grad_(True) for q in (lay.f.weight, lay.f.bias, lay.w.weight, lay.w.bias)]
    xr = x.double().requires_grad_(True)
    ref = R.gat_layer(xr, src, tgt, Mtgt.coalesce().double(), *ref_p)
    ref.backward(gout.double())
    lay = lay.to(dev())
    xd = x.to(dev()).requires_grad_(True)
    srcd, tgtd, Md = src.to(dev()), tgt.to(dev()), Mtgt.to(dev())
    out = lay(xd, srcd, tgtd, Md)
    eg = edge_graph(srcd, tgtd, Md)
    assert eg.canonical and eg.Mt.n_long >= 1 and eg.n > 65536
    close(out, ref, 2e-5, "out")
    out.backward(gout.to(dev()))
    close(xd.grad, xr.grad, 2e-5, "gx")
    for q, r, nm in zip((lay.f.weight, lay.f.bias, lay.w.weight, lay.w.bias), ref_p, ("Wf", "bf", "ww", "bw")):
        close(q.grad, r.grad, 1e-4 if nm == "bw" else 1e-5, "grad " + nm)      # bw: analytically 0, sum of 3e5 roundings


def test_edge_matvec_large_batch_path_matches_fused_path():
    """From 4096 edges on the QC message step is a per-edge matvec kernel plus an SpMM over Etgt; same sums as the
    fused per-target kernel (both visit a target's edges in edge-id order) and as the oracle."""
    from graph_odenet_amd import ops
    from graph_odenet_amd.graph import incidence_from_index
    from oracle import layers_ref as R
    gen = torch.Generator().manual_seed(7)
    n, E, h = 1500, 5000, 73
    src, tgt = torch.randint(0, n, (E,), generator=gen), torch.randint(0, n - 20, (E,), generator=gen)
    A, X = torch.randn(E, h, h, generator=gen) / h ** 0.5, torch.randn(n, h, generator=gen)
    Mt = incidence_from_index(tgt.to(dev()).to(torch.int32), n)
    srcd = src.to(dev()).to(torch.int32)
    big = ops.edge_matvec_fwd(Mt, srcd, A.to(dev()), X.to(dev()))
    old = ops.EDGE_MSG_MIN_EDGES
    ops.EDGE_MSG_MIN_EDGES = 1 << 30
    try:
        fused = ops.edge_matvec_fwd(Mt, srcd, A.to(dev()), X.to(dev()))
    finally:
        ops.EDGE_MSG_MIN_EDGES = old
    Etgt = torch.zeros(n, E)
    Etgt[tgt, torch.arange(E)] = 1.0
    ref = R.edge_message_aggregate(X, src, Etgt, A)
    close(big, ref, 1e-5, "per-edge + SpMM path")
    close(fused, ref, 1e-5, "fused path")
    close(big, fused, 2e-6, "paths agree")


def test_gat_ode_block_on_edgeless_graph():
    """No edges: every aggregation is 0 / eps = 0, so f = 0 and the block is the identity; all gradients are defined."""
    from graph_odenet_amd import gat_models
    n, d = 50, 16
    src = torch.zeros(0, dtype=torch.int64, device=dev())
    tgt = torch.zeros(0, dtype=torch.int64, device=dev())
    Mtgt = torch.sparse_coo_tensor(torch.zeros(2, 0, dtype=torch.int64, device=dev()), torch.zeros(0, device=dev()), (n, 0))
    blk = gat_models.ODEBlock(gat_models.ODEfunc(d), method="rk4", step_size=0.5).to(dev())
    x = torch.randn(n, d, device=dev(), requires_grad=True)
    out = blk(x, src, tgt, Mtgt)
    assert torch.equal(out, x.detach())
    out.sum().backward()
    close(x.grad, torch.ones(n, d), 0.0, "identity gradient")
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) == 0.0 for p in blk.parameters())


@pytest.mark.parametrize("d", [16, 128])
def test_gat_native_dopri5_step_matches_python_driver(golden, d):
    """Adaptive steps of the fused GAT field as one C-ABI call (csrc/gat_driver.hip) against the per-stage Python driver
    on Citeseer's edge list: same kernels in the same order, so the same accepted / rejected steps."""
    from graph_odenet_amd import gat_models, solver as SV
    ge = golden("citeseer_gat_edges.npz")
    n, src, tgt, Mtgt = gat_inputs(ge)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(d)).relu().to(dev())
    res = {}
    for native in (True, False):
        SV.DOPRI5_NATIVE = native
        try:
            torch.manual_seed(1)
            blk = gat_models.ODEBlock(gat_models.ODEfunc(d), tol=1e-4).to(dev())          # default method: dopri5
            xi = x.clone().requires_grad_(True)
            out = blk(xi, src, tgt, Mtgt)
            nf = blk.nfe; blk.nfe = 0
            out.square().mean().backward()
            res[native] = (out.detach(), xi.grad, [p.grad.clone() for p in blk.parameters()], nf, blk.nfe)
        finally:
            SV.DOPRI5_NATIVE = True
    assert res[True][3] == res[False][3] and res[True][4] == res[False][4] and res[True][3] >= 14
    close(res[True][0], res[False][0], 1e-6, "state")
    close(res[True][1], res[False][1], 1e-5, "gx")
    tol = 1e-4 if d == 16 else 1e-5
    for a, b in zip(res[True][2], res[False][2]):
        close(a, b, tol, "param grad")


@pytest.mark.parametrize("B,I,H", [(20, 146, 73), (1, 8, 4), (64, 146, 73), (7, 300, 100), (3, 5, 3)])
@pytest.mark.parametrize("bias", [True, False])
def test_fused_lstm_cell_vs_float64(B, I, H, bias):
    """csrc/lstm.hip: the LSTM cell of the Set2Set readout (QC/set2set.py:44-47,61; torch.nn.LSTM's layout and gate order)
    as one launch forward and one backward, against torch.lstm_cell in float64 on the CPU: h', c', and the gradients of
    x, h, c, both weight matrices and both biases under cotangents on BOTH outputs (the next processing step reads h' and
    c') and on h' alone."""
    from graph_odenet_amd import ops
    from graph_odenet_amd.qc_models import _LstmCellFn
    if not ops.lstm_cell_supported(B, I, H):
        pytest.skip("outside the fused cell's limits")
    g = torch.Generator().manual_seed(B + 3 * I + 7 * H)
    mk = lambda *s: torch.randn(*s, generator=g)                                                     # noqa: E731
    x, h, c = mk(B, I), mk(B, H), mk(B, H)
    w_ih, w_hh = mk(4 * H, I) / I ** 0.5, mk(4 * H, H) / H ** 0.5
    b_ih, b_hh = (mk(4 * H), mk(4 * H)) if bias else (None, None)
    gh, gc = mk(B, H), mk(B, H)
    for use_gc in (True, False):
        ref_in = [t.double().requires_grad_(True) if t is not None else None for t in (x, h, c, w_ih, w_hh, b_ih, b_hh)]
        rh, rc = torch.lstm_cell(ref_in[0], (ref_in[1], ref_in[2]), ref_in[3], ref_in[4], ref_in[5], ref_in[6])
        ((rh * gh.double()).sum() + ((rc * gc.double()).sum() if use_gc else 0.0)).backward()
        dev_in = [t.to(dev()).requires_grad_(True) if t is not None else None for t in (x, h, c, w_ih, w_hh, b_ih, b_hh)]
        oh, oc = _LstmCellFn.apply(*dev_in)
        close(oh, rh, 2e-6, "h'"); close(oc, rc, 2e-6, "c'")
        ((oh * gh.to(dev())).sum() + ((oc * gc.to(dev())).sum() if use_gc else 0.0)).backward()
        for a, b, nm in zip(dev_in, ref_in, ("dx", "dh", "dc", "dw_ih", "dw_hh", "db_ih", "db_hh")):
            if a is not None:
                close(a.grad, b.grad, 4e-6 * max(1.0, B ** 0.5), nm)


@pytest.mark.parametrize("o", [8, 16, 64])
def test_gat_layer_with_an_aggregation_matrix_that_disagrees_with_tgt(o):
    """`Mtgt` is an INPUT of the reference's layer (GAT/layers.py:31,53-55), not derived from `tgt`: the messages and logits
    read x[tgt_e], the sums run over the rows of Mtgt.  Its loaders always pass the incidence of tgt, but a caller may not
    (EdgeGraph.canonical False: edge-id indirection, per-edge target projections).  Rows of Mtgt here are a random
    re-assignment of the edges, values not 1, one 90-edge row; widths on both sides of the prefetched-index kernels
    (o >= 16) and the plain wave kernels (o = 8).  Forward and all gradients against the oracle layer in float64."""
    from graph_odenet_amd.gat_layers import GraphConvolution
    from graph_odenet_amd.gat_layers import edge_graph
    from oracle import layers_ref as R
    gen = torch.Generator().manual_seed(40 + o)
    n, E, i = 300, 1500, 12
    src, tgt = torch.randint(0, n, (E,), generator=gen), torch.randint(0, n, (E,), generator=gen)
    rows = torch.randint(0, n - 10, (E,), generator=gen)             # where each edge is summed: NOT its tgt
    rows[:90] = 7                                                     # one long row (two 64-slot chunks)
    vals = torch.rand(E, generator=gen) + 0.5
    Mtgt = torch.sparse_coo_tensor(torch.stack([rows, torch.arange(E)]), vals, (n, E)).coalesce()
    x = torch.randn(n, i, generator=gen)
    layer = GraphConvolution(i, o)
    P = [p.detach().double().requires_grad_(True) for p in (layer.f.weight, layer.f.bias, layer.w.weight, layer.w.bias)]
    x64 = x.double().requires_grad_(True)
    ref = R.gat_layer(x64, src, tgt, Mtgt.double(), *P)
    gout = torch.randn(n, o, generator=gen)
    ref.backward(gout.double())
    layer = layer.to(dev())
    xd = x.to(dev()).requires_grad_(True)
    sd, td, Md = src.to(dev()), tgt.to(dev()), Mtgt.to(dev())
    assert not edge_graph(sd, td, Md).canonical
    out = layer(xd, sd, td, Md)
    close(out, ref, 2e-5, "forward")
    out.backward(gout.to(dev()))
    close(xd.grad, x64.grad, 5e-5, "dx")
    for p, q, nm in zip((layer.f.weight, layer.f.bias, layer.w.weight, layer.w.bias), P, ("f.weight", "f.bias", "w.weight", "w.bias")):
        close(p.grad, q.grad, 5e-5, nm)
