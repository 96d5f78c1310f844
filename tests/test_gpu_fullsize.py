"""Size-independent properties at BASELINE's full size (2^20 nodes, 10.8 M edges, d = 128) for the kernels whose oracle
comparison runs at small sizes only: the fused dense kernels, the whole ODE block, the GAT record kernels and the
large-batch QC message path.  Everything goes through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu
N, D = 1 << 20, 128


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def rmat():
    from graph_odenet_amd.synth import rmat_graph
    g = rmat_graph(20, 10_000_000, seed=0, device=dev())
    g.transpose()
    return g


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(1.0, float(b.double().abs().max())))


def test_dense_kernels_full_size_properties():
    """gn_time_gemm: linear in W and in t; its VJP and weight gradient satisfy <S(W), dS> = <W, dW> (S is linear in W)
    and the first-order change of S along a direction v equals <VJP, v>; wgrad is additive over row blocks."""
    from graph_odenet_amd import ops
    gen = torch.Generator(device=dev()).manual_seed(0)
    x, k1 = torch.randn(N, D, generator=gen, device=dev()), torch.randn(N, D, generator=gen, device=dev())
    terms = [(1.0, x), (0.3, k1)]
    gam, bet = torch.rand(D, generator=gen, device=dev()) + 0.5, torch.rand(D, generator=gen, device=dev()) - 0.5
    W1 = torch.randn(D + 1, D, generator=gen, device=dev()) / D ** 0.5
    W2 = torch.randn(D + 1, D, generator=gen, device=dev()) / D ** 0.5
    S1 = ops.gn_time_gemm(terms, N, D, 32, 1e-5, gam, bet, W1, True, 0.4)
    S2 = ops.gn_time_gemm(terms, N, D, 32, 1e-5, gam, bet, W2, True, 0.4)
    S12 = ops.gn_time_gemm(terms, N, D, 32, 1e-5, gam, bet, 2.0 * W1 - 0.5 * W2, True, 0.4)
    assert rel(S12, 2.0 * S1 - 0.5 * S2) < 2e-5
    xo = torch.empty(N, D, device=dev())
    S1b = ops.gn_time_gemm(terms, N, D, 32, 1e-5, gam, bet, W1, True, 0.4, x_out=xo)
    # two terms without x_out: the bf16-piece kernel (gemm_split 2); with x_out: the fp32-MFMA kernel - same fp32 result
    assert rel(S1b, S1) < 2e-6 and rel(xo, x + 0.3 * k1) < 1e-6
    t3 = terms + [(-0.2, torch.randn(N, D, generator=gen, device=dev()))]         # three terms: one kernel either way
    S3 = ops.gn_time_gemm(t3, N, D, 32, 1e-5, gam, bet, W1, True, 0.4)
    xo3 = torch.empty(N, D, device=dev())
    S3b = ops.gn_time_gemm(t3, N, D, 32, 1e-5, gam, bet, W1, True, 0.4, x_out=xo3)
    assert torch.equal(S3b, S3)
    del S3, S3b, t3, xo3
    # S is linear in W:  <S(W1), dS> = <W1, dW(dS)>  with dW from the weight-gradient kernel (time row scaled by t)
    dS = torch.randn(N, D, generator=gen, device=dev())
    part = ops.wgrad(terms, N, D, 32, 1e-5, gam, bet, dS, True)
    dW = part.double().sum(0).view(D + 1, D)
    dW[0] *= 0.4
    lhs = (S1.double() * dS.double()).sum().item()
    rhs = (W1.double() * dW).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * max(1.0, abs(lhs))
    # additivity of the weight gradient over row blocks (first half + second half)
    h = N // 2
    pa = ops.wgrad([(1.0, xo[:h])], h, D, 32, 1e-5, gam, bet, dS[:h].contiguous(), True).double().sum(0)
    pb = ops.wgrad([(1.0, xo[h:])], N - h, D, 32, 1e-5, gam, bet, dS[h:].contiguous(), True).double().sum(0)
    assert rel(pa + pb, part.double().sum(0)) < 1e-5
    # VJP: directional derivative of <S, dS> along v by central differences in fp64 on a 4096-row slice
    dx, _, _ = ops.gn_time_gemm_bwd(terms, N, D, 32, 1e-5, gam, W1, True, dS)
    sl = slice(1000, 1000 + 4096)
    xs = xo[sl].double().cpu().requires_grad_(True)
    xn = torch.nn.functional.group_norm(xs, 32, gam.double().cpu(), bet.double().cpu(), 1e-5)
    Sd = torch.cat([torch.full((4096, 1), 0.4, dtype=torch.float64), xn], 1) @ W1.double().cpu()
    (Sd * dS[sl].double().cpu()).sum().backward()
    assert rel(dx[sl].cpu(), xs.grad) < 2e-5


def test_ode_block_full_size_semigroup(rmat):
    """Fixed-grid rk4 through the one-call C driver: integrating [0, 1] in four steps equals integrating [0, 0.5] and
    then [0.5, 1] in two steps each (same stage times, same kernels), and the Python per-stage driver agrees."""
    from graph_odenet_amd import models, odeint as OI
    torch.manual_seed(0)
    f = models.ODEfunc(D).to(dev())
    f.set_adj(rmat)
    x = torch.randn(N, D, device=dev()).relu()
    opt = {"step_size": 0.25}
    with torch.no_grad():
        y1 = OI.odeint(f, x, torch.tensor([0.0, 1.0]), method="rk4", options=opt)[1]
        ya = OI.odeint(f, x, torch.tensor([0.0, 0.5]), method="rk4", options=opt)[1]
        yb = OI.odeint(f, ya, torch.tensor([0.5, 1.0]), method="rk4", options=opt)[1]
        OI.NATIVE_RK4 = False
        try:
            yp = OI.odeint(f, x, torch.tensor([0.0, 1.0]), method="rk4", options=opt)[1]
        finally:
            OI.NATIVE_RK4 = True
    assert torch.isfinite(y1).all() and rel(y1, x) > 1e-2          # the state moved
    assert rel(yb, y1) < 1e-6 and rel(yp, y1) < 1e-6
    assert f.nfe == 16 + 8 + 8 + 16


def test_adjoint_bias_gradient_from_the_spmm_column_sums_full_size(rmat):
    """The adjoint rk4 driver at the benchmark size: the bias gradient of every stage reduced from the per-block column
    sums its forward-recompute SpMM leaves (option y2_colsum, the default) against the same solve with a column-sum pass
    over dZ (option off): every other gradient bit for bit the same (nothing else changes), the bias gradient to fp32
    summation accuracy (2^20 rows in another order)."""
    from graph_odenet_amd import _lib, models, odeint as OI
    lib = _lib.load()
    torch.manual_seed(1)
    f = models.ODEfunc(D).to(dev())
    f.set_adj(rmat)
    x0 = torch.randn(N, D, device=dev()).relu()
    gout = torch.randn(N, D, generator=torch.Generator(device=dev()).manual_seed(2), device=dev())
    res = {}
    try:
        for on in (1, 0):
            assert lib.gode_set_option(b"y2_colsum", on) == 0 and lib.gode_get_option(b"y2_colsum") == on
            x = x0.clone().requires_grad_(True)
            for p in f.parameters():
                p.grad = None
            y = OI.odeint_adjoint(f, x, torch.tensor([0.0, 1.0]), method="rk4", options={"step_size": 0.5})[1]
            y.backward(gout)
            res[on] = (x.grad.clone(), {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None})
    finally:
        lib.gode_set_option(b"y2_colsum", 1)
    assert torch.equal(res[1][0], res[0][0])
    differ = []
    for n in res[1][1]:
        a, b = res[1][1][n], res[0][1][n]
        if not torch.equal(a, b):
            differ.append(n)
            assert rel(a, b) < 1e-5, n
    assert set(differ) <= {"gc1.bias"}, differ


def test_gat_record_kernels_full_size_properties(rmat):
    """Attention aggregation on the R-MAT edge list (record path): with constant projections every edge carries the
    same message, so out = relu(c) * s / (s + eps) row by row; the weights sum to the denominator; the VJP's node sums
    equal the column sums of the edge cotangents."""
    from graph_odenet_amd import ops
    from graph_odenet_amd.gat_layers import EdgeGraph
    rp = rmat.rowptr.to(torch.int64)
    tgt = torch.repeat_interleave(torch.arange(N, device=dev()), rp[1:] - rp[:-1])
    src = rmat.col.to(torch.int64)
    E = src.numel()
    perm = torch.randperm(E, device=dev())
    src, tgt = src[perm], tgt[perm]
    Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E, device=dev())]), torch.ones(E, device=dev()), (N, E))
    eg = EdgeGraph(src, tgt, Mtgt)
    assert eg.canonical and eg.Mt.n_long > 0
    f = dict(dtype=torch.float32, device=dev())
    Ps, Pt = torch.full((N, D), 0.25, **f), torch.full((N, D), 0.5, **f)
    A2 = torch.randn(N, 2, **f)
    bf, bw = torch.linspace(-1, 1, D, device=dev()), torch.zeros(1, **f)
    proj = ops.gat_proj(Ps, Pt, A2)
    a, amax = torch.empty(E, **f), torch.empty(1, **f)
    out, w, den = torch.empty(N, D, **f), torch.empty(E, **f), torch.empty(N, **f)
    ops.gat_logits(proj, bw, eg.src, eg.tgt, a, amax)
    ops.gat_agg_fwd(eg, proj, D, bf, a, amax, 1e-6, out, w, den)
    s = torch.zeros(N, dtype=torch.float64, device=dev()).index_add_(0, eg.tgt.long(), w.double())
    assert rel(den, s.float() + 1e-6) < 1e-5
    expect = torch.relu(0.75 + bf)[None, :] * (s / (s + 1e-6)).float()[:, None]
    assert rel(out, expect) < 1e-5
    dz, da = torch.empty(E, D, **f), torch.empty(E, **f)
    dPs, dPt, dA2 = torch.empty(N, D, **f), torch.empty(N, D, **f), torch.empty(N, 2, **f)
    dout = torch.randn(N, D, **f)
    ops.gat_vjp(eg, proj, D, bf, a, amax, w, den, out, dz, da, dPs, dPt, dA2, dout=dout)
    col = dz.double().sum(0)
    assert rel(dPs.double().sum(0), col) < 1e-5 and rel(dPt.double().sum(0), col) < 1e-5
    tot = da.double().sum()                          # after the max-path correction the logit cotangents sum to ~0
    assert abs(float(dA2[:, 0].double().sum() - tot)) < 1e-3 and abs(float(dA2[:, 1].double().sum() - tot)) < 1e-3
    assert abs(float(tot)) < 1e-2 * float(da.abs().double().sum())


def test_qc_large_batch_message_properties():
    """76 000 edges, h = 73 (2 000 molecules' worth): identity edge matrices turn the message step into a plain
    gather-sum, and the step is linear in x."""
    from graph_odenet_amd import ops
    from graph_odenet_amd.graph import incidence_from_index
    gen = torch.Generator(device=dev()).manual_seed(1)
    n, E, h = 36000, 76000, 73
    src = torch.randint(0, n, (E,), generator=gen, device=dev())
    tgt = torch.randint(0, n, (E,), generator=gen, device=dev())
    Mt = incidence_from_index(tgt.to(torch.int32), n)
    X = torch.randn(n, h, generator=gen, device=dev())
    eye = torch.eye(h, device=dev()).expand(E, h, h).contiguous()
    out = ops.edge_matvec_fwd(Mt, src.to(torch.int32), eye, X)
    ref = torch.zeros(n, h, device=dev()).index_add_(0, tgt, X[src])
    assert rel(out, ref) < 1e-5
    A = torch.randn(E, h, h, generator=gen, device=dev()) / h ** 0.5
    X2 = torch.randn(n, h, generator=gen, device=dev())
    lhs = ops.edge_matvec_fwd(Mt, src.to(torch.int32), A, 2.0 * X - 0.5 * X2)
    rhs = 2.0 * ops.edge_matvec_fwd(Mt, src.to(torch.int32), A, X) - 0.5 * ops.edge_matvec_fwd(Mt, src.to(torch.int32), A, X2)
    assert rel(lhs, rhs) < 2e-5
