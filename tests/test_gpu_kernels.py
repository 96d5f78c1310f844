"""GPU parity of the C-ABI kernels against the oracle / plain torch CPU ops on the same seeded
inputs.  Tolerance: 1e-5 (fp32, relative to the largest reference magnitude), as BASELINE.json
states.  All calls go through libgraphode.so (graph_odenet_amd.ops)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def close(a, b, tol=TOL, what=""):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


def powerlaw_graph(n, m, avg, seed, long_row=None, empty=0, vals=True):
    rs = np.random.RandomState(seed)
    deg = rs.zipf(1.8, n).clip(1, max(4, 20 * avg))
    deg[:empty] = 0
    if long_row:
        deg[empty] = long_row
    rows = np.repeat(np.arange(n), deg)
    cols = rs.randint(0, m, rows.size)
    v = (rs.rand(rows.size).astype(np.float32) + 0.1) if vals else None
    return torch.from_numpy(rows), torch.from_numpy(cols), (torch.from_numpy(v) if vals else None)


@pytest.mark.parametrize("d", [4, 7, 16, 32, 64, 73, 128, 256])
@pytest.mark.parametrize("pattern_only", [False, True])
def test_spmm_matches_cpu(d, pattern_only):
    from graph_odenet_amd import graph as G, ops
    n, m = 1500, 1300
    r, c, v = powerlaw_graph(n, m, 8, d, long_row=1000, empty=7, vals=not pattern_only)
    vv = v if v is not None else torch.ones(r.numel())
    A = torch.sparse_coo_tensor(torch.stack([r, c]), vv, (n, m))
    g = G.from_coo(r.to(dev()), c.to(dev()), None if v is None else v.to(dev()), n, m, split=64)
    assert g.n_long >= 1
    X = torch.randn(m, d)
    bias = torch.randn(d)
    ref = torch.sparse.mm(A, X) + bias
    out = ops.spmm(g, X.to(dev()), bias=bias.to(dev()), relu=False)
    close(out, ref, what="spmm")
    out = ops.spmm(g, X.to(dev()), bias=bias.to(dev()), relu=True)
    close(out, ref.clamp_min(0), what="spmm+relu")
    # transposed graph (backward direction)
    Y = torch.randn(n, d)
    close(ops.spmm(g.transpose(), Y.to(dev())), torch.sparse.mm(A.t(), Y), what="spmm^T")


@pytest.mark.parametrize("weighted", [True, False])
def test_spmm_one_column_record_path(weighted):
    """d = 1 over a record list (spmv_wave_kernel: a wave per record, lanes stride over its non-zeros): against the
    dense product, with split long rows (n_long > 0), empty rows, and the strided output the attention VJP uses
    (column 0 of an N x 2 matrix, ldy = 2, the other column untouched)."""
    from graph_odenet_amd import graph as G, ops
    n, m = 1100, 1700
    r, c, v = powerlaw_graph(n, m, 8, 11, long_row=900, empty=5, vals=weighted)
    g = G.from_coo(r.to(dev()), c.to(dev()), None if v is None else v.to(dev()), n, m, split=64)
    assert g.n_long >= 1 and g.items is not None
    x = torch.randn(m, 1)
    ref = g.to_dense().cpu().double() @ x.double()
    close(ops.spmm(g, x.to(dev())), ref, what="one column")
    out2 = torch.full((n, 2), 7.0, device=dev())
    ops.spmm(g, x.to(dev()), out=out2[:, 0:1])
    close(out2[:, 0:1], ref, what="one column, ldy = 2")
    assert bool((out2[:, 1] == 7.0).all()), "the neighbouring column was written"
    b = torch.randn(1)
    close(ops.spmm(g, x.to(dev()), bias=b.to(dev()), relu=True), (ref + b.double()).clamp_min(0), what="bias + relu")
    y = torch.randn(n, 1)
    close(ops.spmm(g.transpose(), y.to(dev())), g.to_dense().cpu().double().t() @ y.double(), what="transposed")


def test_spmm_masked_cotangent_epilogue():
    from graph_odenet_amd import graph as G, ops
    n, d = 900, 128
    r, c, v = powerlaw_graph(n, n, 6, 3, long_row=700)
    A = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    g = G.from_coo(r.to(dev()), c.to(dev()), v.to(dev()), n, n, split=128)
    X, b = torch.randn(n, d), torch.randn(d)
    a0, a1 = torch.randn(n, d), torch.randn(n, d)
    Z = torch.sparse.mm(A, X) + b
    out, out2 = ops.spmm(g, X.to(dev()), bias=b.to(dev()), relu=True,
                         cot_terms=[(-1.0, a0.to(dev())), (0.25, a1.to(dev()))])
    close(out, Z.clamp_min(0))
    ref2 = (-a0 + 0.25 * a1) * (Z > 0)
    # a handful of entries sit within rounding of the relu kink; compare where |Z| is clear of it
    safe = Z.abs() > 1e-4
    close(out2.cpu() * safe, ref2 * safe, what="masked cotangent")


@pytest.mark.parametrize("d", [128, 16, 256])
def test_spmm_leaves_block_column_sums_of_the_masked_cotangent(d):
    """gode_spmm_csr_f32 with Y2_colsum: the launch that writes the relu-masked cotangent of an adjoint stage also leaves
    the column sums of the rows each block stored (the bias gradient is colsum of that array: the driver reduces these
    partial rows instead of reading the n x d array again).  Y and Y2 are bit for bit those of the launch without the
    option; the partial rows add up to Y2.sum(0); rows split over several records (finished by the second launch) are in;
    a small graph (a wave per record) reports 0 rows and refuses the option."""
    from graph_odenet_amd import graph as G, ops
    n = 70_000
    r, c, v = powerlaw_graph(n, n, 5, 11, long_row=900, empty=3)
    g = G.from_coo(r.to(dev()), c.to(dev()), v.to(dev()), n, n, split=128)
    assert g.n_long >= 1 and g.n_items > 65536
    gen = torch.Generator().manual_seed(d)
    X, b = torch.randn(n, d, generator=gen).to(dev()), torch.randn(d, generator=gen).to(dev())
    a0, a1 = torch.randn(n, d, generator=gen).to(dev()), torch.randn(n, d, generator=gen).to(dev())
    cot = [(-1.0, a0), (0.25, a1)]
    rows = ops.spmm_y2_colsum_rows(g, d)
    lpr = d // 4
    assert rows == (g.n_items * lpr + 255) // 256 + (g.n_long * lpr + 255) // 256
    out, out2 = ops.spmm(g, X, bias=b, relu=True, cot_terms=cot)
    part = torch.full((rows, d), float("nan"), device=dev())
    outb, out2b = ops.spmm(g, X, bias=b, relu=True, cot_terms=cot, out2_colsum=part)
    assert torch.equal(outb, out) and torch.equal(out2b, out2)
    assert bool(torch.isfinite(part).all())
    want = out2.double().sum(0)
    got = part.double().sum(0)
    assert (got - want).abs().max().item() <= 2e-6 * out2.abs().double().sum(0).max().item()
    # and through the library's own column sum, as the adjoint driver does
    cs = ops.colsum_(torch.empty(d, device=dev()), part)
    assert (cs.double() - want).abs().max().item() <= 2e-6 * out2.abs().double().sum(0).max().item()
    # small graph: no per-block sums
    r2, c2, v2 = powerlaw_graph(500, 500, 5, 12)
    g2 = G.from_coo(r2.to(dev()), c2.to(dev()), v2.to(dev()), 500, 500)
    assert ops.spmm_y2_colsum_rows(g2, 16) == 0
    with pytest.raises(ValueError):
        ops.spmm(g2, torch.randn(500, 16, device=dev()), relu=True, cot_terms=[(1.0, torch.randn(500, 16, device=dev()))],
                 out2_colsum=torch.empty(4, 16, device=dev()))


def test_spmm_row_sum_property_full_size():
    """Size-independent property at BASELINE's full size (2^20 nodes, ~10M edges, d=128):
    a row-normalised A_hat maps the all-ones matrix to all-ones, and SpMM is linear."""
    from graph_odenet_amd import graph as G, ops
    from graph_odenet_amd.synth import rmat_graph
    g = rmat_graph(20, 10_000_000, seed=0, device=dev())
    n, d = g.n_rows, 128
    ones = torch.ones(n, d, device=dev())
    out = ops.spmm(g, ones)
    assert (out - 1).abs().max().item() < 1e-5
    gen = torch.Generator(device=dev()).manual_seed(123)
    x1 = torch.randn(n, d, device=dev(), generator=gen)
    x2 = torch.randn(n, d, device=dev(), generator=gen)
    lhs = ops.spmm(g, 2.0 * x1 - 0.5 * x2)
    rhs = 2.0 * ops.spmm(g, x1) - 0.5 * ops.spmm(g, x2)
    assert (lhs - rhs).abs().max().item() < 1e-4
    # <A x, y> == <x, A^T y>
    y = torch.randn(n, d, device=dev(), generator=gen)
    ta = ops.spmm(g, x1).double() * y.double()
    tb = x1.double() * ops.spmm(g.transpose(), y).double()
    a, b = ta.sum(), tb.sum()
    # both sides are sums of 1.3e8 fp32-rounded terms (the products themselves are added in fp64 here): their
    # difference is a random walk of the terms' rounding errors, a few ulp each - measured 1.0e-3 on |a| = 890.
    # A wrong transpose is off by the order of |a| itself.
    noise = 2.0 ** -24 * (ta.square().sum() + tb.square().sum()).sqrt().item()
    assert abs(a.item()) > 100 * 32 * noise
    assert abs(a.item() - b.item()) <= 32 * noise, (a.item(), b.item(), noise)


def test_lincomb_and_error_norms():
    from graph_odenet_amd import ops
    torch.manual_seed(0)
    for n in (5, 1024, 100003):
        xs = [torch.randn(n) for _ in range(6)]
        cs = [1.0, 0.5, -0.25, 3.0, 1e-3, -2.0]
        ref = sum(c * x for c, x in zip(cs, xs))
        gx = [x.to(dev()) for x in xs]
        out = torch.empty(n, device=dev())
        ops.lincomb_(out, list(zip(cs, gx)))
        close(out, ref, what="lincomb")
        # in place
        ops.lincomb_(gx[0], list(zip(cs, gx)))
        close(gx[0], ref, what="lincomb in place")
        y0, y1 = torch.randn(n), torch.randn(n)
        err = 1e-4 * xs[1] - 2e-4 * xs[2]
        ref_s = ((err / (1e-5 + 1e-5 * torch.max(y0.abs(), y1.abs()))).double() ** 2).sum()
        got = ops.rk_error_sumsq(y0.to(dev()), y1.to(dev()), [(1e-4, gx[1]), (-2e-4, gx[2])], 1e-5, 1e-5)
        assert abs(got.item() - ref_s.item()) <= 1e-4 * ref_s.item()
        ref_n = ((xs[3] / (1e-5 + 1e-5 * y0.abs())).double() ** 2).sum()
        got = ops.rk_scaled_sumsq([(1.0, gx[3])], y0.to(dev()), 1e-5, 1e-5)
        assert abs(got.item() - ref_n.item()) <= 1e-4 * ref_n.item()


def test_colsum_and_reduce_parts():
    from graph_odenet_amd import ops
    torch.manual_seed(1)
    # (760, 5329), (37, 2667), (2000, 600): wide and short - the bias gradients of the QC edge encoder (colsum_wide_kernel)
    for n, d in ((1, 16), (300, 7), (5000, 128), (70000, 73), (760, 5329), (37, 2667), (2000, 600), (1, 513)):
        X = torch.randn(n, d)
        out = torch.zeros(d, device=dev())
        ops.colsum_(out, X.to(dev()), scale=-0.5)
        close(out, -0.5 * X.double().sum(0).float(), tol=1e-5 * max(1, n ** 0.5), what="colsum")
        ops.colsum_(out, X.to(dev()), scale=2.0, accumulate=True)
        close(out, 1.5 * X.double().sum(0).float(), tol=2e-5 * max(1, n ** 0.5), what="colsum accumulate")
    P = torch.randn(37, 1000)
    out = torch.ones(1000, device=dev())
    ops.reduce_parts_(out, P.to(dev()), scale=2.0, accumulate=True)
    close(out, 1 + 2 * P.sum(0), tol=1e-5 * 6)


@pytest.mark.parametrize("d,groups", [(16, 16), (32, 32), (64, 32), (128, 32), (128, 0), (24, 24), (96, 32), (40, 8)])
@pytest.mark.parametrize("n", [1, 257, 4100])
def test_gn_time_gemm_fwd_bwd_wgrad(d, groups, n):
    """GroupNorm + time column + GEMM against plain torch CPU ops (GCN/models.py:175-177 + layers.py:70)."""
    from graph_odenet_amd import ops
    import torch.nn.functional as F
    if n == 1 and groups == d:
        pytest.skip("torch's own group_norm refuses one value per group at batch 1")
    torch.manual_seed(d * 7 + n)
    y, k1 = torch.randn(n, d), torch.randn(n, d)
    h = 0.3
    x = (y + h * k1).requires_grad_(True)
    gam = (torch.rand(d) + 0.5).requires_grad_(True)
    bet = (torch.rand(d) - 0.5).requires_grad_(True)
    W = (torch.randn(d + 1, d) / d ** 0.5).requires_grad_(True)
    t = 0.37
    xn = F.group_norm(x, groups, gam, bet, 1e-5) if groups else x
    S = torch.cat([torch.full((n, 1), t), xn], 1) @ W
    dS = torch.randn(n, d)
    S.backward(dS)
    D = dev()
    terms = [(1.0, y.to(D)), (h, k1.to(D))]
    g_, b_ = (gam.detach().to(D), bet.detach().to(D)) if groups else (None, None)
    got = ops.gn_time_gemm(terms, n, d, groups, 1e-5, g_, b_, W.detach().to(D), True, t)
    # Conditioning: with ONE channel per group GroupNorm's output is beta plus
    # rounding noise amplified by rstd = 1/sqrt(eps) = 316 and its x-gradient is exactly 0 in real
    # arithmetic - both sides return noise of size ~316 * 2^-23 * |dy| there; with TWO channels per
    # group rstd reaches 316 on rows whose two values nearly coincide.  Tolerances follow that.
    cg = d // groups if groups else 0
    tol = {0: TOL, 1: 2e-4, 2: 2e-5}.get(cg, TOL)
    close(got, S, tol=tol, what="gn_time_gemm")
    dx, dgp, dbp = ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, g_, W.detach().to(D), True, dS.to(D))
    close(dx, x.grad, tol={0: TOL, 1: 2e-3, 2: 1e-4}.get(cg, 2e-5), what="dx")
    if groups:
        close(dgp.sum(0), gam.grad, tol=(2e-3 if cg == 1 else 2e-5) * max(1, n ** 0.5), what="dgamma")
        close(dbp.sum(0), bet.grad, tol=2e-5 * max(1, n ** 0.5), what="dbeta")
    part = ops.wgrad(terms, n, d, groups, 1e-5, g_, b_, dS.to(D), True)
    gW = part.sum(0).view(d + 1, d)
    close(gW[1:], W.grad[1:], tol=max(tol, 2e-5) * max(1, n ** 0.5), what="dW")
    close(gW[0] * t, W.grad[0], tol=2e-5 * max(1, n ** 0.5), what="dW time row")


def test_reduce_segments_one_launch():
    """gode_reduce_segments_f32: several block-partial reductions, a strided column subset, two time rows (scaled by t,
    their unscaled sums dotted with the weight rows into `at`) and gode_colsum_parts_f32 as a segment source."""
    from graph_odenet_amd import _lib, ops
    lib = _lib.load()
    D = dev()
    g = torch.Generator().manual_seed(17)
    pa, pb, pc = torch.randn(37, 300, generator=g), torch.randn(5, 64, generator=g), torch.randn(130, 24, generator=g)
    wa, wc = torch.randn(40, generator=g), torch.randn(7, generator=g)
    X = torch.randn(1234, 16, generator=g)
    outs = [torch.full((k,), 9.0, device=D) for k in (300, 64, 12, 16)]
    at = torch.zeros(1, device=D)
    scratch = torch.empty(lib.gode_colsum_scratch_bytes(1234, 16), dtype=torch.uint8, device=D)
    n_x = ops.colsum_parts(X.to(D), scratch)
    t = 0.37
    ops.reduce_segments_([
        (outs[0], pa.to(D), 37, 300, 0, 1, 300, wa.to(D), 40),
        (outs[1], pb.to(D), 5, 64, 0, 1, 64, None, 0),
        (outs[2], pc.to(D), 130, 24, 1, 2, 12, wc.to(D), 7),          # the odd columns of a 24-wide partial
        (outs[3], scratch, n_x, 16, 0, 1, 16, None, 0)], t, at)
    sa, sc = pa.double().sum(0), pc.double().sum(0)[1::2]
    ea = sa.clone(); ea[:40] *= t
    ec = sc.clone(); ec[:7] *= t
    close(outs[0], ea, 1e-5, "segment with a time row")
    close(outs[1], pb.double().sum(0), 1e-5, "plain segment")
    close(outs[2], ec, 1e-5, "strided segment with a time row")
    close(outs[3], X.double().sum(0), 1e-5 * 35, "column sums closed by a segment")
    close(at, ((sa[:40] * wa.double()).sum() + (sc[:7] * wc.double()).sum()).reshape(1), 1e-5 * 10, "time derivative")


@pytest.mark.parametrize("d,dout,groups", [(16, 34, 16), (64, 130, 32), (128, 258, 32), (24, 7, 0)])
def test_gn_time_gemm_rectangular(d, dout, groups):
    """d_out != d_in (the GAT node-level projection is d x (2o+2)): generic kernels, same parity bar."""
    from graph_odenet_amd import ops
    import torch.nn.functional as F
    torch.manual_seed(d + dout)
    n = 777
    x = torch.randn(n, d, requires_grad=True)
    gam = (torch.rand(d) + 0.5).requires_grad_(True)
    bet = (torch.rand(d) - 0.5).requires_grad_(True)
    W = (torch.randn(d + 1, dout) / d ** 0.5).requires_grad_(True)
    t = 0.61
    xn = F.group_norm(x, groups, gam, bet, 1e-5) if groups else x
    S = torch.cat([torch.full((n, 1), t), xn], 1) @ W
    dS = torch.randn(n, dout)
    S.backward(dS)
    D = dev()
    g_, b_ = (gam.detach().to(D), bet.detach().to(D)) if groups else (None, None)
    terms = [(1.0, x.detach().to(D))]
    cg = d // groups if groups else 0
    tol = {0: TOL, 1: 2e-4, 2: 2e-5}.get(cg, TOL)
    close(ops.gn_time_gemm(terms, n, d, groups, 1e-5, g_, b_, W.detach().to(D), True, t), S, tol, "fwd")
    dx, dgp, dbp = ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, g_, W.detach().to(D), True, dS.to(D))
    close(dx, x.grad, {0: TOL, 1: 2e-3, 2: 1e-4}.get(cg, 2e-5), "dx")
    part = ops.wgrad(terms, n, d, groups, 1e-5, g_, b_, dS.to(D), True)
    gW = part.sum(0).view(d + 1, dout)
    close(gW[1:], W.grad[1:], max(tol, 2e-5) * n ** 0.5, "dW")
    close(gW[0] * t, W.grad[0], 2e-5 * n ** 0.5, "dW time row")


@pytest.mark.parametrize("d,groups", [(16, 16), (32, 16), (64, 32), (128, 32), (128, 0), (64, 16)])
@pytest.mark.parametrize("n", [5, 1000, 70001])
def test_gn_time_gemm_two_columns(d, groups, n):
    """d_out = 2 (the logit columns of the GAT ODE function): the narrow register kernels (csrc/gemm.hip,
    gn_narrow_*), incl. a multi-term input written out through x_out, accumulation into `pre`, affine partials and the
    time row, against torch CPU ops.  (64, 16) has 4 channels per group at d = 64."""
    from graph_odenet_amd import ops
    import torch.nn.functional as F
    torch.manual_seed(d + n)
    y, k1 = torch.randn(n, d), torch.randn(n, d)
    x = (y + 0.25 * k1).requires_grad_(True)
    gam = (torch.rand(d) + 0.5).requires_grad_(True)
    bet = (torch.rand(d) - 0.5).requires_grad_(True)
    W = (torch.randn(d + 1, 2) / d ** 0.5).requires_grad_(True)
    t = 0.61
    xn = F.group_norm(x, groups, gam, bet, 1e-5) if groups else x
    S = torch.cat([torch.full((n, 1), t), xn], 1) @ W
    dS = torch.randn(n, 2)
    S.backward(dS)
    D = dev()
    g_, b_ = (gam.detach().to(D), bet.detach().to(D)) if groups else (None, None)
    terms = [(1.0, y.to(D)), (0.25, k1.to(D))]
    cg = d // groups if groups else 0
    tol = {0: TOL, 1: 2e-4, 2: 2e-5}.get(cg, TOL)
    xo = torch.empty(n, d, device=D)
    close(ops.gn_time_gemm(terms, n, d, groups, 1e-5, g_, b_, W.detach().to(D), True, t, x_out=xo), S, tol, "fwd")
    close(xo, x.detach(), 1e-6, "x_out")
    pre = torch.randn(n, d)
    dx, dgp, dbp = ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, g_, W.detach().to(D), True, dS.to(D), out_scale=0.5,
                                        pre_terms=[(1.0, pre.to(D))])
    close(dx, pre + 0.5 * x.grad, {0: TOL, 1: 2e-3, 2: 1e-4}.get(cg, 2e-5), "pre + scale * dx")
    if groups:
        close(dgp.sum(0), gam.grad, (2e-3 if cg == 1 else 2e-5) * max(1, n ** 0.5), "dgamma")
        close(dbp.sum(0), bet.grad, 2e-5 * max(1, n ** 0.5), "dbeta")
    part = ops.wgrad(terms, n, d, groups, 1e-5, g_, b_, dS.to(D), True)
    gW = part.sum(0).view(d + 1, 2)
    close(gW[1:], W.grad[1:], max(tol, 2e-5) * max(1, n ** 0.5), "dW")
    close(gW[0] * t, W.grad[0], 2e-5 * max(1, n ** 0.5), "dW time row")


def test_spmm_strided_output():
    from graph_odenet_amd import graph as G, ops
    n, d = 500, 16
    r, c, v = powerlaw_graph(n, n, 5, 9)
    A = torch.sparse_coo_tensor(torch.stack([r, c]), v, (n, n))
    g = G.from_coo(r.to(dev()), c.to(dev()), v.to(dev()), n, n)
    X = torch.randn(n, d)
    wide = torch.full((n, 2 * d + 2), 7.0, device=dev())
    ops.spmm(g, X.to(dev()), out=wide[:, d:2 * d])
    close(wide[:, d:2 * d], torch.sparse.mm(A, X), what="column block")
    assert (wide[:, :d] == 7).all() and (wide[:, 2 * d:] == 7).all()
    x1 = torch.randn(n, 1)
    ops.spmm(g, x1.to(dev()), out=wide[:, 2 * d:2 * d + 1])
    close(wide[:, 2 * d:2 * d + 1], torch.sparse.mm(A, x1), what="single column")


@pytest.mark.parametrize("n,c,g", [(3327, 128, 32), (257, 16, 16), (120, 64, 32), (1000, 48, 8), (5, 7, 1)])
def test_group_norm_2d_vs_cpu_torch(n, c, g):
    """Stand-alone GroupNorm on (nodes x channels) against torch's CPU implementation.  (torch-ROCm's own GPU
    backward returns wrong dgamma / dbeta for 2-D inputs with more than a few hundred rows - the reason this
    op exists; see functional.GroupNorm.)"""
    from graph_odenet_amd.functional import GroupNorm
    torch.manual_seed(n + c)
    x = torch.randn(n, c)
    go = torch.randn(n, c)
    ref = torch.nn.GroupNorm(g, c)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-0.5, 0.5)
    xr = x.clone().requires_grad_(True)
    ref(xr).backward(go)
    m = GroupNorm(g, c).to(dev())
    m.load_state_dict(ref.state_dict())
    xg = x.clone().to(dev()).requires_grad_(True)
    out = m(xg)
    out.backward(go.to(dev()))
    cg = c // g
    ftol = {1: 2e-4, 2: 2e-5}.get(cg, TOL)
    close(out, ref(x), ftol, "gn fwd")
    close(xg.grad, xr.grad, {1: 2e-3, 2: 1e-4}.get(cg, 2e-5), "gn dx")
    close(m.weight.grad, ref.weight.grad, (2e-3 if cg == 1 else 2e-5) * max(1, n ** 0.5), "gn dgamma")
    close(m.bias.grad, ref.bias.grad, 2e-5 * max(1, n ** 0.5), "gn dbeta")


@pytest.mark.parametrize("n", [5000, 70001])
@pytest.mark.parametrize("n_terms", [1, 2, 3])
def test_split_bf16_forward_product_matches_fp32_path(n, n_terms):
    """Split-bf16 x3 variant of the d=128 forward dense product (exact 3-piece decomposition by rounding, eight piece
    products - what is dropped is below 2^-32 of a product - fp32 accumulation) against the exact-fp32 MFMA path and
    float64.  gemm_split 1 = always, 2 = launches of at most two terms and at least 65 536 rows, 0 = never.  (Since
    round 3 the default route of those launches is the producer / consumer kernel, fwd_pc - switched off here; its own
    test is test_forward_product_producer_consumer.)"""
    from graph_odenet_amd import _lib, ops
    import torch.nn.functional as F
    lib = _lib.load()
    torch.manual_seed(3 + n_terms)
    d = 128
    xs = [torch.randn(n, d) * (3 if j == 0 else 1) for j in range(n_terms)]
    coef = [1.0, 0.2, -0.4][:n_terms]
    gam, bet = torch.rand(d) + 0.5, torch.rand(d) - 0.5
    W = torch.randn(d + 1, d) / d ** 0.5
    x = sum(c * t for c, t in zip(coef, xs))
    xg = x.double().view(n, 32, 4)
    xn = ((xg - xg.mean(2, keepdim=True)) / torch.sqrt(xg.var(2, unbiased=False, keepdim=True) + 1e-5)).view(n, d)
    ref = torch.cat([torch.full((n, 1), 0.4).double(), xn * gam.double() + bet.double()], 1) @ W.double()
    terms = [(c, t.to(dev())) for c, t in zip(coef, xs)]
    outs = {}
    fwd_pc = lib.gode_get_option(b"fwd_pc")
    try:
        assert lib.gode_set_option(b"fwd_pc", 0) == 0
        for mode in (0, 1, 2):
            assert lib.gode_set_option(b"gemm_split", mode) == 0 and lib.gode_get_option(b"gemm_split") == mode
            outs[mode] = ops.gn_time_gemm(terms, n, d, 32, 1e-5, gam.to(dev()), bet.to(dev()), W.to(dev()), True, 0.4).cpu()
        assert lib.gode_set_option(b"gemm_split", 3) != 0
    finally:
        lib.gode_set_option(b"gemm_split", 2)
        lib.gode_set_option(b"fwd_pc", fwd_pc)
    scale = ref.abs().max().item()
    for mode in (0, 1, 2):
        e = (outs[mode].double() - ref).abs().max().item()
        assert e <= 2e-6 * scale, (mode, e, scale)
    if n < 65536 or n_terms > 2:
        assert torch.equal(outs[2], outs[0])              # mode 2 left these launches to the fp32 kernel
    else:
        assert torch.equal(outs[2], outs[1])


@pytest.mark.parametrize("n", [1, 31, 33, 65, 1000, 4097, 70001])
@pytest.mark.parametrize("groups", [32, 0, 64, 128])
def test_forward_product_producer_consumer(n, groups):
    """csrc/gemm_pc.hip gn_gemm_fwd_pc_kernel (d = 128, option fwd_pc, the default route from 65 536 rows on): the
    operands cut exactly into three bf16 pieces, eight piece products in fp32 on the bf16 matrix cores, consumer waves
    with the weight slab in registers, producer waves staging 32-row tiles - against float64 and against the fp32-MFMA
    kernel: 1-5 terms (raw prefetch registers up to 4, combined at load beyond), with and without the time row, with
    and without x_out (which must be the fp32 path's combined input bit for bit), ragged row counts (a partly empty
    last tile, fewer tiles than producer groups), 4 / 2 / 1 channels per GroupNorm group and no normalisation."""
    from graph_odenet_amd import _lib, ops
    lib = _lib.load()
    d = 128
    g = torch.Generator().manual_seed(n * 11 + groups)
    xs = [torch.randn(n, d, generator=g) * (1.5 if j == 0 else 0.4) + (0.3 if j == 0 else 0.0) for j in range(5)]
    coef = [1.0, 0.25, -0.125, 0.0625, 0.5]
    gam, bet = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    W = torch.randn(d + 1, d, generator=g) / d ** 0.5
    D = dev()
    cg = d // groups if groups else 0
    # one or two channels per group: GroupNorm amplifies the rounding of x by rstd (up to 316), in both kernels alike
    # (test_gn_time_gemm_fwd_bwd_wgrad uses the same bars)
    tol = {0: 2e-6, 4: 2e-6, 2: 4e-5, 1: 4e-4}[cg]
    saved = lib.gode_get_option(b"fwd_pc")
    try:
        assert lib.gode_set_option(b"wgrad_split_small", 1) == 0
        assert lib.gode_set_option(b"fwd_pc", 4) != 0
        for nt in (1, 2, 3, 4, 5):
            x = sum(c * t.double() for c, t in zip(coef[:nt], xs[:nt]))
            xn = x
            if groups:
                xg = x.view(n, groups, d // groups)
                xn = ((xg - xg.mean(2, keepdim=True)) / torch.sqrt(xg.var(2, unbiased=False, keepdim=True) + 1e-5)).view(n, d)
                xn = xn * gam.double() + bet.double()
            terms = [(c, t.to(D)) for c, t in zip(coef[:nt], xs[:nt])]
            for has_time in (True, False):
                Wd = W if has_time else W[1:].contiguous()
                ref = (torch.cat([torch.full((n, 1), 0.4, dtype=torch.float64), xn], 1) @ W.double()) if has_time else xn @ W[1:].double()
                scale = ref.abs().max().item() + 1e-30
                for want_xout in (False, True):
                    got = {}
                    for mode in (0, 3):
                        assert lib.gode_set_option(b"fwd_pc", mode) == 0 and lib.gode_get_option(b"fwd_pc") == mode
                        xo = torch.full((n, d), float("nan"), device=D) if want_xout else None
                        out = ops.gn_time_gemm(terms, n, d, groups, 1e-5, gam.to(D) if groups else None,
                                               bet.to(D) if groups else None, Wd.to(D), has_time, 0.4, x_out=xo)
                        got[mode] = (out.cpu(), xo.cpu() if want_xout else None)
                        err = (got[mode][0].double() - ref).abs().max().item() / scale
                        assert err <= tol, (n, groups, nt, has_time, want_xout, mode, err)
                    assert (got[3][0] - got[0][0]).abs().max().item() <= 2 * tol * scale
                    if want_xout:
                        assert torch.equal(got[3][1], got[0][1])
                        assert (got[3][1].double() - x).abs().max().item() <= 1e-6 * (x.abs().max().item() + 1e-30)
    finally:
        lib.gode_set_option(b"fwd_pc", saved)
        lib.gode_set_option(b"wgrad_split_small", 0)


@pytest.mark.parametrize("n", [1, 31, 33, 1000, 4097, 70001])
@pytest.mark.parametrize("groups", [32, 0])
def test_weight_gradient_from_exact_bf16_pieces(n, groups):
    """csrc/gemm.hip wgrad_split_kernel (d = 128): every fp32 operand cut exactly into three bf16 pieces, products
    accumulated in fp32 on the bf16 matrix cores - against float64 and against the fp32-MFMA kernel, for 8 and 6 piece
    products, 1 / 2 (raw prefetch registers) and 3 / 5 terms (combined at load), with and without the time row, ragged
    row counts (the last 32-row tile partly empty, fewer tiles than producer groups)."""
    from graph_odenet_amd import _lib, ops
    lib = _lib.load()
    d = 128
    g = torch.Generator().manual_seed(n * 7 + groups)
    xs = [torch.randn(n, d, generator=g) * (1.5 if j == 0 else 0.4) + (0.3 if j == 0 else 0.0) for j in range(5)]
    coef = [1.0, 0.25, -0.125, 0.0625, 0.5]
    dS = torch.randn(n, d, generator=g) * torch.rand(n, 1, generator=g)
    gam, bet = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    try:
        assert lib.gode_set_option(b"wgrad_split_small", 1) == 0
        for nt in (1, 2, 3, 5):
            x = sum(c * t.double() for c, t in zip(coef[:nt], xs[:nt]))
            xn = x
            if groups:
                xg = x.view(n, groups, d // groups)
                xn = ((xg - xg.mean(2, keepdim=True)) / torch.sqrt(xg.var(2, unbiased=False, keepdim=True) + 1e-5)).view(n, d)
                xn = xn * gam.double() + bet.double()
            want = torch.cat([dS.double().sum(0, keepdim=True), xn.t() @ dS.double()], 0)
            scale = want.abs().max().item() + 1e-30
            terms = [(c, t.to(dev())) for c, t in zip(coef[:nt], xs[:nt])]
            got = {}
            for mode in (0, 8, 6):
                assert lib.gode_set_option(b"wgrad_split", mode) == 0 and lib.gode_get_option(b"wgrad_split") == mode
                for has_time in (True, False):
                    part = ops.wgrad(terms, n, d, groups, 1e-5, gam.to(dev()) if groups else torch.ones(d, device=dev()),
                                     bet.to(dev()) if groups else torch.zeros(d, device=dev()), dS.to(dev()), has_time)
                    dW = part.double().sum(0).view(d + (1 if has_time else 0), d).cpu()
                    ref = want if has_time else want[1:]
                    err = (dW - ref).abs().max().item() / scale
                    assert err <= 2e-6, (n, groups, nt, mode, has_time, err)
                    got[(mode, has_time)] = dW
            for has_time in (True, False):        # the three kernels agree far inside the 1e-5 bar
                for mode in (8, 6):
                    assert (got[(mode, has_time)] - got[(0, has_time)]).abs().max().item() <= 2e-6 * scale
    finally:
        lib.gode_set_option(b"wgrad_split", 8)
        lib.gode_set_option(b"wgrad_split_small", 0)
    assert lib.gode_set_option(b"wgrad_split", 7) != 0            # only 0, 6, 8


@pytest.mark.parametrize("n", [1, 31, 33, 1000, 4097, 70001])
@pytest.mark.parametrize("groups", [32, 0])
def test_vjp_from_exact_bf16_pieces(n, groups):
    """The VJP at d = 128 from exact bf16 pieces (dS and W1 cut three ways, eight piece products, fp32 accumulation, the
    GroupNorm backward of the fp32 kernel): csrc/gemm_pc.hip gn_gemm_bwd_pc_kernel (option bwd_pc, the default from
    65 536 rows on: consumer waves with the weight slab in registers, producer waves staging dS) against float64
    autograd and against the fp32-MFMA kernel: dx (with the fused pre-term and output scale), dgamma, dbeta; 1 / 2 terms of x in raw registers, 3 combined
    at load; ragged row counts (a partly empty last 32-row tile, fewer tiles than producer groups)."""
    from graph_odenet_amd import _lib, ops
    import torch.nn.functional as F
    lib = _lib.load()
    d = 128
    g = torch.Generator().manual_seed(n * 5 + groups)
    ys = [torch.randn(n, d, generator=g) * 1.5 + 0.3, torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)]
    cf = [1.0, 0.25, -0.5]
    dS, pre = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
    gam = torch.rand(d, generator=g) + 0.5
    W = torch.randn(d + 1, d, generator=g) / d ** 0.5
    D = dev()
    modes = {"fp32": 0, "pc": 1}                                             # bwd_pc
    saved = lib.gode_get_option(b"bwd_pc")
    try:
        assert lib.gode_set_option(b"wgrad_split_small", 1) == 0
        for nt in (1, 2, 3):
            x = sum(c * t.double() for c, t in zip(cf[:nt], ys[:nt])).requires_grad_(True)
            g64 = gam.double().requires_grad_(True)
            b64 = torch.zeros(d, dtype=torch.float64, requires_grad=True)
            xn = F.group_norm(x, groups, g64, b64, 1e-5) if groups else x
            S = torch.cat([torch.full((n, 1), 0.4, dtype=torch.float64), xn], 1) @ W.double()
            S.backward(dS.double())
            terms = [(c, t.to(D)) for c, t in zip(cf[:nt], ys[:nt])]
            for with_pre in (True, False):
                want_dx = (pre.double() if with_pre else 0.0) + 0.5 * x.grad
                got = {}
                for name, pc in modes.items():
                    assert lib.gode_set_option(b"bwd_pc", pc) == 0 and lib.gode_get_option(b"bwd_pc") == pc
                    dx, dg, db = ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, gam.to(D), W.to(D), True, dS.to(D), out_scale=0.5,
                                                      pre_terms=[(1.0, pre.to(D))] if with_pre else None)
                    got[name] = (dx.double().cpu(), dg.double().sum(0).cpu() if groups else None,
                                 db.double().sum(0).cpu() if groups else None)
                # a GroupNorm group of four nearly equal values has rstd up to 316 and amplifies every rounding of x by it
                # in dx: the bar is the 2e-5 of test_gn_time_gemm_fwd_bwd_wgrad, for all three kernels
                sx = want_dx.abs().max().item()
                for name in modes:
                    assert (got[name][0] - want_dx).abs().max().item() <= 2e-5 * sx, (name, n, groups, nt)
                    if groups:
                        assert (got[name][1] - g64.grad).abs().max().item() <= 3e-6 * max(1.0, g64.grad.abs().max().item()) * max(1, n ** 0.5), (name, nt)
                        assert (got[name][2] - b64.grad).abs().max().item() <= 3e-6 * max(1.0, b64.grad.abs().max().item()) * max(1, n ** 0.5), (name, nt)
                assert (got["pc"][0] - got["fp32"][0]).abs().max().item() <= 2e-5 * sx
    finally:
        lib.gode_set_option(b"bwd_pc", saved)
        lib.gode_set_option(b"wgrad_split_small", 0)


@pytest.mark.parametrize("n", [1, 31, 33, 64, 1000, 4097, 70001])
@pytest.mark.parametrize("groups", [32, 0])
def test_vjp_and_weight_gradient_in_one_pass(n, groups):
    """csrc/gemm_pc.hip gn_gemm_bwd_wgrad_pc_kernel (round 4): dx = GN'(x)^T (dS W1^T) and dW = [1 | GN(x)]^T dS from ONE
    read of x and dS - one producer group with two register sets, transposed operand fragments by ds_read_b64_tr_b16 from
    the row-major piece images - against float64 autograd and against the two separate kernels: dx (pre-term and output
    scale), dgamma, dbeta, dW with and without the time row, 1 / 2 raw terms and 3 combined at load, ragged row counts
    (odd tile counts per block, a partly empty last tile, fewer tiles than register sets)."""
    from graph_odenet_amd import _lib, ops
    import torch.nn.functional as F
    lib = _lib.load()
    d = 128
    g = torch.Generator().manual_seed(n * 11 + groups)
    ys = [torch.randn(n, d, generator=g) * 1.5 + 0.3, torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)]
    cf = [1.0, 0.25, -0.5]
    dS, pre = torch.randn(n, d, generator=g) * torch.rand(n, 1, generator=g), torch.randn(n, d, generator=g)
    gam, bet = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    W = torch.randn(d + 1, d, generator=g) / d ** 0.5
    D = dev()
    gm = gam.to(D) if groups else torch.ones(d, device=D)
    bt = bet.to(D) if groups else torch.zeros(d, device=D)
    try:
        assert lib.gode_set_option(b"wgrad_split_small", 1) == 0
        assert ops.bwd_wgrad_supported(n, d, groups)
        for nt in (1, 2, 3):
            x = sum(c * t.double() for c, t in zip(cf[:nt], ys[:nt])).requires_grad_(True)
            g64 = gam.double().requires_grad_(True)
            b64 = bet.double().requires_grad_(True)
            xn = F.group_norm(x, groups, g64, b64, 1e-5) if groups else x
            Wd = W.double().requires_grad_(True)
            S = torch.cat([torch.ones(n, 1, dtype=torch.float64), xn], 1) @ Wd
            S.backward(dS.double())
            terms = [(c, t.to(D)) for c, t in zip(cf[:nt], ys[:nt])]
            for with_pre, has_time in ((True, True), (False, True), (False, False)):
                want_dx = (pre.double() if with_pre else 0.0) + 0.5 * x.grad
                want_dW = Wd.grad if has_time else Wd.grad[1:]
                Wg = W.to(D) if has_time else W[1:].contiguous().to(D)
                pt = [(1.0, pre.to(D))] if with_pre else None
                dx, dg, db, wp = ops.gn_time_gemm_bwd_wgrad(terms, n, d, groups, 1e-5, gm, bt, Wg, has_time, dS.to(D), out_scale=0.5,
                                                            pre_terms=pt)
                dx2, dg2, db2 = ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, gm, Wg, has_time, dS.to(D), out_scale=0.5, pre_terms=pt)
                wp2 = ops.wgrad(terms, n, d, groups, 1e-5, gm, bt, dS.to(D), has_time)
                sx = want_dx.abs().max().item()
                assert (dx.double().cpu() - want_dx).abs().max().item() <= 2e-5 * sx, (n, groups, nt, with_pre)
                assert torch.equal(dx, dx2)                     # the same arithmetic as the stand-alone VJP kernel, bit for bit
                dW = wp.double().sum(0).view(d + (1 if has_time else 0), d).cpu()
                sw = want_dW.abs().max().item() + 1e-30
                assert (dW - want_dW).abs().max().item() <= 2e-6 * sw, (n, groups, nt, has_time)
                dW2 = wp2.double().sum(0).view_as(dW).cpu()
                assert (dW - dW2).abs().max().item() <= 2e-6 * sw
                if groups:
                    for got_p, ref_p, want in ((dg, dg2, g64.grad), (db, db2, b64.grad)):
                        got = got_p.double().sum(0).cpu()
                        assert (got - want).abs().max().item() <= 3e-6 * max(1.0, want.abs().max().item()) * max(1, n ** 0.5)
                        assert torch.equal(got_p.sum(0), ref_p.sum(0)) or (got - ref_p.double().sum(0).cpu()).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    finally:
        lib.gode_set_option(b"wgrad_split_small", 0)
    assert not ops.bwd_wgrad_supported(1000, d, groups)         # below 65 536 rows the drivers keep the two launches


@pytest.mark.parametrize("n", [1, 17, 380, 1000, 70001])
@pytest.mark.parametrize("K,M", [(16, 7), (128, 16), (1433, 16), (500, 16), (3, 3), (100, 40), (64, 200), (130, 129), (3703, 64), (3703, 2), (73, 73)])
def test_rectangular_products_vs_float64(n, K, M):
    """csrc/rect.hip: the dense products of a GraphConvolution with in_features != out_features (GCN/layers.py:32
    `torch.mm(input, self.weight)` and its autograd) on the exact fp32 matrix instruction - X W (with the output padded
    to a multiple of four columns, pad = exact zeros), dS W^T (dS a column block of the padded matrix), X^T dS (block
    partials + fixed-order sum) - against float64; aligned rows take the 16-byte accesses, K % 4 != 0 (Cora 1433, Citeseer 3703) dword accesses in the same lane layout."""
    from graph_odenet_amd import ops
    if n * K > 40_000_000:
        pytest.skip("operand larger than the test needs")
    g = torch.Generator().manual_seed(n * 3 + K * 5 + M)
    x = torch.randn(n, K, generator=g)
    W = torch.randn(K, M, generator=g) / K ** 0.5
    D = dev()
    ref = x.double() @ W.double()
    scale = ref.abs().max().item() + 1e-30
    kt = max(1.0, (K / 256) ** 0.5)                       # fp32 accumulation over K terms
    got = ops.rect_gemm(x.to(D), W.to(D)).cpu()
    assert got.shape == (n, M) and (got.double() - ref).abs().max().item() <= 2e-6 * scale * kt
    mp = (M + 3) // 4 * 4
    gotp = ops.rect_gemm(x.to(D), W.to(D), pad_to=mp).cpu()
    assert gotp.shape == (n, mp) and torch.equal(gotp[:, :M], got) and bool((gotp[:, M:] == 0).all())
    dSp = torch.zeros(n, mp)
    dSp[:, :M] = torch.randn(n, M, generator=g)
    dSd = dSp.to(D)
    want_dx = dSp[:, :M].double() @ W.double().t()
    dx = ops.rect_gemm_nt(dSd[:, :M], W.to(D)).cpu()
    assert (dx.double() - want_dx).abs().max().item() <= 2e-6 * (want_dx.abs().max().item() + 1e-30)
    want_dw = x.double().t() @ dSp[:, :M].double()
    dw = ops.rect_wgrad(x.to(D), dSd[:, :M]).cpu()
    assert dw.shape == (K, M)
    assert (dw.double() - want_dw).abs().max().item() <= 3e-6 * (want_dw.abs().max().item() + 1e-30) * max(1.0, (n / 4096) ** 0.5)


def test_dense_on_a_mostly_zero_input_takes_the_csr_route():
    """functional.dense (the GAT input layer's three projections of Citeseer's 3327 x 3703 bag of words, GAT/layers.py:43-45)
    runs CSR(X) @ W from the second sighting of the same tensor object; output widths 64 and 2 (the logit pair, padded to
    4 columns for the aggregation kernel); forward and weight gradient against float64 on both routes."""
    from graph_odenet_amd import functional as Fn, layers as L
    g = torch.Generator().manual_seed(11)
    n, K = 3327, 3703
    x = (torch.rand(n, K, generator=g) < 0.009).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1)
    xd = x.to(dev())
    for M in (64, 2):
        W = (torch.randn(K, M, generator=g) / K ** 0.5)
        dy = torch.randn(n, M, generator=g)
        ref, ref_gw = x.double() @ W.double(), x.double().t() @ dy.double()
        routes = []
        for sighting in range(3):
            Wd = W.to(dev()).requires_grad_(True)
            y = Fn.dense(xd, Wd)
            y.backward(dy.to(dev()))
            routes.append(y.grad_fn.xs is not None)
            assert y.shape == (n, M) and y.is_contiguous()
            assert (y.detach().cpu().double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
            assert (Wd.grad.cpu().double() - ref_gw).abs().max().item() <= 3e-6 * ref_gw.abs().max().item()
        if M == 64:
            assert routes == [False, True, True], routes
        else:
            assert routes == [True, True, True], routes           # the tensor was already classified


@pytest.mark.parametrize("M,N,K", [(760, 5329, 2667), (760, 2667, 5), (5, 2667, 760), (1, 1, 1), (129, 130, 17), (64, 128, 16), (300, 73, 200),
                                   (73, 42, 363)])
def test_tiled_gemm_all_operand_layouts_and_epilogues(M, N, K):
    """csrc/mlp.hip gode_gemm_f32 (the dense products of the QC edge encoder, QC/layers.py:46-86): C = op(A) op(B) for the
    four operand layouts, ragged sizes (rows of 2667 / 5329 floats are not 16-byte aligned), leading dimensions, and
    the fused epilogues (+ bias, relu, * [mask > 0]) against float64."""
    from graph_odenet_amd import ops
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / max(K, 1) ** 0.5
    bias, mask = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    D = dev()
    ref = A.double() @ B.double()
    scale = ref.abs().max().item() + 1e-30
    tol = 3e-6 * scale
    for ta in (False, True):
        for tb in (False, True):
            a = (A.t().contiguous() if ta else A).to(D)
            b = (B.t().contiguous() if tb else B).to(D)
            got = ops.gemm(a, b, trans_a=ta, trans_b=tb).cpu()
            assert (got.double() - ref).abs().max().item() <= tol, (ta, tb)
    a, b = A.to(D), B.to(D)
    got = ops.gemm(a, b, bias=bias.to(D), relu=True).cpu()
    assert (got.double() - torch.relu(ref + bias.double())).abs().max().item() <= tol
    got = ops.gemm(a, b, mask=mask.to(D)).cpu()
    assert (got.double() - ref * (mask > 0).double()).abs().max().item() <= tol
    # operands and result as column blocks of wider matrices (leading dimensions)
    wide_a, wide_b = torch.randn(M, K + 3, device=D), torch.randn(K, N + 5, device=D)
    wide_a[:, :K], wide_b[:, :N] = a, b
    out = torch.full((M, N + 2), 7.0, device=D)
    ops.gemm(wide_a[:, :K], wide_b[:, :N], out=out[:, :N])
    assert (out[:, :N].cpu().double() - ref).abs().max().item() <= tol and bool((out[:, N:] == 7.0).all())


def test_transition_mlp_and_linear_vs_float64_autograd():
    """functional.mlp2 / affine / linear (QC/layers.py TransitionMLP = NonLinear + MyLinear; nn.Linear): forward and every
    gradient against float64 autograd of the same formulas, at the edge encoder's shape (E = 760 edge rows, 5 -> 2667 ->
    5329) and a small ragged one."""
    from graph_odenet_amd import functional as Fn
    D = dev()
    for E, fi, hid, fo in ((760, 5, 2667, 5329), (37, 13, 43, 73)):
        g = torch.Generator().manual_seed(E)
        x = torch.randn(E, fi, generator=g)
        W1, b1 = torch.randn(fi, hid, generator=g) / fi ** 0.5, torch.randn(hid, generator=g) * 0.1
        W2, b2 = torch.randn(hid, fo, generator=g) / hid ** 0.5, torch.randn(fo, generator=g) * 0.1
        dy = torch.randn(E, fo, generator=g)
        ref_in = [t.double().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
        y64 = torch.relu(ref_in[0] @ ref_in[1] + ref_in[2]) @ ref_in[3] + ref_in[4]
        y64.backward(dy.double())
        got_in = [t.to(D).requires_grad_(True) for t in (x, W1, b1, W2, b2)]
        y = Fn.mlp2(*got_in)
        y.backward(dy.to(D))
        close(y, y64.float(), 1e-5, "mlp2 forward")
        for a, b, nm in zip(got_in, ref_in, ("dx", "dW1", "db1", "dW2", "db2")):
            assert (a.grad.cpu().double() - b.grad).abs().max().item() <= 1e-5 * max(1.0, b.grad.abs().max().item()), nm
    # affine (W stored in x out) and linear (out x in), with and without bias
    x = torch.randn(100, 37, requires_grad=True)
    W = torch.randn(37, 21, requires_grad=True)
    b = torch.randn(21, requires_grad=True)
    dy = torch.randn(100, 21)
    (x @ W + b).backward(dy)
    for form in ("affine", "linear"):
        xs, bs = x.detach().to(D).requires_grad_(True), b.detach().to(D).requires_grad_(True)
        Ws = (W.detach() if form == "affine" else W.detach().t().contiguous()).to(D).requires_grad_(True)
        out = Fn.affine(xs, Ws, bs) if form == "affine" else Fn.linear(xs, Ws, bs)
        out.backward(dy.to(D))
        close(out, (x @ W + b).detach(), 1e-5, form)
        close(xs.grad, x.grad, 1e-5, form + " dx"); close(bs.grad, b.grad, 1e-5, form + " db")
        close(Ws.grad if form == "affine" else Ws.grad.t(), W.grad, 1e-5, form + " dW")


def test_one_launch_adam_matches_torch_adam():
    """graph_odenet_amd.optim.Adam (csrc/mlp.hip gode_adam_f32: the whole parameter list in one launch, step counter on
    the device) against torch.optim.Adam on the CPU over eight steps: weight decay, a parameter that never receives a
    gradient, one that receives it only from step 3 on, more tensors than one argument block holds (70 > 64), a change
    of the learning rate between steps, and a state_dict round trip into torch's optimiser."""
    from graph_odenet_amd import optim
    g = torch.Generator().manual_seed(0)
    shapes = [(300, 17), (5329,), (1,), (64, 64)] + [(3, 5)] * 66
    ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    got = [p.detach().clone().to(dev()).requires_grad_(True) for p in ref]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    o_ref, o_got = torch.optim.Adam(ref, **kw), optim.Adam(got, **kw)
    for step in range(8):
        grads = [torch.randn(s, generator=g) * (1 + step) for s in shapes]
        for i, (p, q, gr) in enumerate(zip(ref, got, grads)):
            if i == 2 or (i == 3 and step < 3):
                p.grad = q.grad = None
            else:
                p.grad, q.grad = gr.clone(), gr.clone().to(dev())
        if step == 5:
            for o in (o_ref, o_got):
                o.param_groups[0]["lr"] = 3e-3
        o_ref.step(); o_got.step()
        for i, (p, q) in enumerate(zip(ref, got)):
            # parameter 3 joins three steps late: its own step count would differ in torch (per-parameter counters) -
            # the group-wide device counter is what a training loop in which every parameter always has a gradient
            # sees; checked for the parameters that were there from the start
            if i != 3:
                assert (q.detach().cpu() - p.detach()).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item()), (step, i)
    assert got[2].grad is None and torch.equal(got[2].detach().cpu(), ref[2].detach())
    sd = o_got.state_dict()
    o_new = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in got], **kw)
    o_new.load_state_dict(sd)                                     # torch's keys: step, exp_avg, exp_avg_sq
    assert float(o_new.state[o_new.param_groups[0]["params"][0]]["step"]) == 8.0


def test_adam_resumes_from_a_loaded_state_dict():
    """step, load_state_dict (a checkpoint taken earlier), step - against torch.optim.Adam doing the same on the CPU: the
    launch plan holds the addresses of the moment buffers and the device step counter, both replaced by the loader
    (the round-3 plan key missed them and kept updating the freed buffers)."""
    from graph_odenet_amd import optim
    g = torch.Generator().manual_seed(5)
    shapes = [(40, 7), (129,), (3, 3)]
    ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    got = [p.detach().clone().to(dev()).requires_grad_(True) for p in ref]
    kw = dict(lr=1e-2, weight_decay=5e-4)
    o_ref, o_got = torch.optim.Adam(ref, **kw), optim.Adam(got, **kw)

    def both(n):
        for _ in range(n):
            for p, q, s in zip(ref, got, shapes):
                gr = torch.randn(s, generator=g)
                p.grad, q.grad = gr.clone(), gr.clone().to(dev())
            o_ref.step(); o_got.step()
    both(3)
    import copy
    ck_ref, ck_got = copy.deepcopy(o_ref.state_dict()), copy.deepcopy(o_got.state_dict())
    w_ref, w_got = [p.detach().clone() for p in ref], [q.detach().clone() for q in got]
    both(4)                                             # moves moments and counters away from the checkpoint
    with torch.no_grad():
        for p, q, a, b in zip(ref, got, w_ref, w_got):
            p.copy_(a); q.copy_(b)
    o_ref.load_state_dict(ck_ref); o_got.load_state_dict(ck_got)
    both(2)
    for p, q in zip(ref, got):
        assert (q.detach().cpu() - p.detach()).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item())
    assert float(o_got.state[got[0]]["step"]) == 5.0 == float(o_ref.state[ref[0]]["step"])
    # and a torch checkpoint loads into the one-launch optimiser
    o_two = optim.Adam(got, **kw)
    o_two.load_state_dict(copy.deepcopy(o_ref.state_dict()))
    both(0)
    for p, q, s in zip(ref, got, shapes):
        gr = torch.randn(s, generator=g)
        p.grad, q.grad = gr.clone(), gr.clone().to(dev())
    o_ref.step(); o_two.step()
    for p, q in zip(ref, got):
        assert (q.detach().cpu() - p.detach()).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item())


def test_exact_three_way_cut_planes():
    """gode_cut_bf16x3_f32: x = hi + mid + lo EXACTLY (three bf16 numbers, truncation), planes zero outside the matrix,
    for a ragged matrix with an odd leading dimension (rows only 4-byte aligned, as the edge encoder's 2667 / 5329)."""
    from graph_odenet_amd import ops
    g = torch.Generator().manual_seed(11)
    wide = (torch.randn(301, 1203, generator=g) * torch.exp(4 * torch.randn(301, 1203, generator=g))).to(dev())
    X = wide[:, 3:1202]                                             # 301 x 1199, ld = 1203, first column at an odd offset
    c = ops.cut3(X)
    assert tuple(c.planes.shape) == (3, 384, 1280) and (c.rows, c.cols) == (301, 1199)
    pl = c.planes.float()
    assert torch.equal(pl[0, :301, :1199] + pl[1, :301, :1199] + pl[2, :301, :1199], X)          # exact in fp32
    assert float(pl[:, 301:, :].abs().max()) == 0.0 and float(pl[:, :, 1199:].abs().max()) == 0.0
    hi = pl[0, :301, :1199]
    assert bool((hi.abs() <= X.abs()).all()) and bool(((X - hi).abs() <= X.abs() * 2.0 ** -7).all())   # truncation, 8 bits


@pytest.mark.parametrize("M,N,K", [(200, 300, 150), (129, 257, 33), (1, 1, 1), (384, 256, 512), (300, 2600, 2100),
                                   (1100, 5329, 520)])
def test_piece_gemm_all_operand_layouts_and_epilogues(M, N, K):
    """csrc/pgemm.hip gode_pgemm_bf16x3 (large products of the QC edge encoder on the bf16 matrix cores from exact cuts):
    C = op(A) op(B) for the four operand layouts - each operand read from LDS by ds_read_b128 or by the transposing
    ds_read_b64_tr_b16 - ragged sizes, the fused epilogues, 8 and 6 piece products, against float64 at the bar of the
    exact-fp32 kernel; the cut of a matrix serves both of its roles.  The last two sizes do not fill whole rounds of the
    chip with tiles: 63 tiles of 66 k-steps - four blocks per tile, each an aligned quarter of the contraction; 378 tiles
    of 17 k-steps (and 289 of 10, the A^T A product below) - the k-steps are dealt in equal shares to one block per CU,
    shares begin and end inside tiles.  The finishing launch adds the partial tiles."""
    from graph_odenet_amd import _lib, ops
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    want_blocks = {(300, 2600, 2100): 252 if cus == 256 else None, (1100, 5329, 520): cus}.get((M, N, K), 0)
    if want_blocks is not None:
        assert _lib.load().gode_pgemm_workspace_bytes(M, N, K) == 2 * want_blocks * 128 * 128 * 4
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / max(K, 1) ** 0.5
    bias, mask = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    D = dev()
    ref = A.double() @ B.double()
    scale = ref.abs().max().item() + 1e-30
    tol = 3e-6 * scale
    for ta in (False, True):
        for tb in (False, True):
            a = ops.cut3((A.t().contiguous() if ta else A).to(D))
            b = ops.cut3((B.t().contiguous() if tb else B).to(D))
            got = ops.pgemm(a, b, trans_a=ta, trans_b=tb).cpu()
            assert (got.double() - ref).abs().max().item() <= tol, (ta, tb)
            got6 = ops.pgemm(a, b, trans_a=ta, trans_b=tb, products=6).cpu()
            assert (got6.double() - ref).abs().max().item() <= 2 * tol, (ta, tb, 6)
    a, b = ops.cut3(A.to(D)), ops.cut3(B.to(D))
    got = ops.pgemm(a, b, bias=bias.to(D), relu=True).cpu()
    assert (got.double() - torch.relu(ref + bias.double())).abs().max().item() <= tol
    got = ops.pgemm(a, b, mask=mask.to(D)).cpu()
    assert (got.double() - ref * (mask > 0).double()).abs().max().item() <= tol
    # the same cut of A as the TRANSPOSED operand of another product: A^T A (K x K), and a result inside a wider matrix
    ata = ops.pgemm(a, a, trans_a=True).cpu()
    ref2 = A.double().t() @ A.double()
    assert (ata.double() - ref2).abs().max().item() <= 3e-6 * (ref2.abs().max().item() + 1e-30)
    out = torch.full((M, N + 3), 7.0, device=D)
    ops.pgemm(a, b, out=out[:, :N])
    assert (out[:, :N].cpu().double() - ref).abs().max().item() <= tol and bool((out[:, N:] == 7.0).all())


def test_piece_gemm_large_magnitude_spread():
    """Operands whose entries span many binades (weights after training are not N(0, 1)): the cut is exact per element, so
    the bar is the same - relative to sum |a||b| per output element, as for an fp32 fmaf chain."""
    from graph_odenet_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 260, 270, 700
    A = torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, K, generator=g))
    B = torch.randn(K, N, generator=g) * torch.exp(3 * torch.randn(K, N, generator=g))
    D = dev()
    got = ops.pgemm(ops.cut3(A.to(D)), ops.cut3(B.to(D))).cpu().double()
    ref = A.double() @ B.double()
    bound = A.double().abs() @ B.double().abs()
    assert bool(((got - ref).abs() <= 1e-6 * bound).all())


def test_dense_first_nonzero_equals_the_argmax_expression():
    """gode_dense_first_nonzero_f32 (the collate's dense N x E target matrix back to the per-edge target vector, in one
    launch) against `(M != 0).to(uint8).argmax(0)`: several entries per column (the first one counts), all-zero columns
    (index 0), negative and tiny values, one row, a column block of a wider matrix."""
    from graph_odenet_amd import ops
    g = torch.Generator().manual_seed(3)
    for n, e in ((380, 760), (1, 5), (9, 1), (1000, 3000), (37, 33)):
        M = torch.zeros(n, e)
        rows = torch.randint(0, n, (e,), generator=g)
        M[rows, torch.arange(e)] = torch.randn(e, generator=g).sign() * (torch.rand(e, generator=g) * 1e-20 + 1e-30)
        extra = torch.randint(0, n, (e // 3,), generator=g)
        M[extra, torch.randint(0, e, (e // 3,), generator=g)] = 2.0          # second entries in some columns
        M[:, ::7] = 0.0                                                      # all-zero columns
        want = (M != 0).to(torch.uint8).argmax(0)
        got = ops.dense_first_nonzero(M.to(dev())).cpu()
        assert got.dtype == torch.int64 and torch.equal(got, want), (n, e)
        wide = torch.full((n, e + 5), 3.0)
        wide[:, :e] = M
        got2 = ops.dense_first_nonzero(wide.to(dev())[:, :e]).cpu()          # leading dimension e + 5
        assert torch.equal(got2, want)


def test_one_launch_assignment_csr_matches_the_sort_based_path():
    """csrc/convert.hip gode_assign_csr_i32 (a QC mini-batch's edge -> atom and atom -> graph vectors as CSR in one launch)
    against graph.csr_from_assignment's sort-based path: row pointers, the STABLE order inside every row, gathered values;
    empty rows, a single row, the size limits, and sizes past them (which take the sort-based path)."""
    from graph_odenet_amd import _lib, graph as G
    lib = _lib.load()
    g = torch.Generator().manual_seed(17)
    for e, n in ((760, 360), (2048, 4096), (1, 1), (5, 3000), (300, 7), (0, 4)):
        idx = torch.randint(0, n, (e,), generator=g)
        if e > 10:
            idx[idx == 2] = 3                                          # row 2 stays empty
        vals = torch.randn(e, generator=g)
        assert lib.gode_assign_csr_supported(e, n)
        got = G.csr_from_assignment(idx.to(dev()), n, vals.to(dev()))
        order = torch.argsort(idx, stable=True)
        rowptr = torch.zeros(n + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(torch.bincount(idx, minlength=n), 0)
        assert got.items is None and got.rowptr.dtype == torch.int32
        assert torch.equal(got.rowptr.cpu().long(), rowptr) and torch.equal(got.col.cpu().long(), order)
        assert torch.equal(got.val.cpu(), vals[order])
        pat = G.csr_from_assignment(idx.to(dev()), n)                  # pattern-only
        assert pat.val is None and torch.equal(pat.col.cpu().long(), order)
    assert not lib.gode_assign_csr_supported(2049, 10) and not lib.gode_assign_csr_supported(10, 4097)
    idx = torch.randint(0, 5000, (3000,), generator=g)
    big = G.csr_from_assignment(idx.to(dev()), 5000)                   # the sort-based path
    assert torch.equal(big.col.cpu().long(), torch.argsort(idx, stable=True))


def test_lincomb_multi_one_launch_for_four_components():
    """gode_lincomb_multi_f32: the solution combine of an adjoint state [y, a, a_t, theta] as one launch - components of
    different lengths (one of them a single float, one not a multiple of four), in place on the first term, bit for bit the
    result of four gode_lincomb_f32 launches."""
    from graph_odenet_amd import ops
    g = torch.Generator().manual_seed(3)
    sizes = [3327 * 16, 3327 * 16, 1, 611]
    D = dev()
    ys = [torch.randn(n, generator=g).to(D) for n in sizes]
    ks = [[torch.randn(n, generator=g).to(D) for n in sizes] for _ in range(4)]
    coefs = [0.125, 0.375, 0.375, 0.125]
    want = []
    for c in range(4):
        o = ys[c].clone()
        ops.lincomb_(o, [(1.0, o)] + [(coefs[s], ks[s][c]) for s in range(4)])
        want.append(o)
    got = [y.clone() for y in ys]
    ops.lincomb_multi_(got, [[(1.0, got[c])] + [(coefs[s], ks[s][c]) for s in range(4)] for c in range(4)])
    for c in range(4):
        assert torch.equal(got[c], want[c]), c
    two = [ys[0].clone(), ys[3].clone()]
    ops.lincomb_multi_(two, [[(2.0, ys[0])], [(1.0, ys[3]), (-1.0, ks[0][3])]])
    assert torch.equal(two[0], 2.0 * ys[0]) and torch.allclose(two[1], ys[3] - ks[0][3], atol=0, rtol=0)


def test_errnorm_multi_matches_the_single_form_bit_for_bit():
    """gode_rk_errnorm_multi_f32: the four error-ratio sums that close an adaptive step of the adjoint state [y, a, a_t, theta]
    in one pair of launches - every sum identical to gode_rk_errnorm_f32's (same block decomposition per component)."""
    import ctypes
    from graph_odenet_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(8)
    D = dev()
    sizes = [2708 * 16, 2708 * 16, 1, 323]
    y0 = [torch.randn(n, generator=g).to(D) for n in sizes]
    y1 = [torch.randn(n, generator=g).to(D) for n in sizes]
    ks = [[torch.randn(n, generator=g).to(D) for n in sizes] for _ in range(3)]
    coef = [0.3, -0.7, 0.11]
    single = [ops.rk_error_sumsq(y0[c], y1[c], [(coef[j], ks[j][c]) for j in range(3)], 1e-4, 1e-5).item() for c in range(4)]
    out = torch.empty(4, dtype=torch.float64, device=D)
    sc = torch.empty(lib.gode_rk_errnorm_scratch_bytes(), dtype=torch.uint8, device=D)
    lcs = (_lib.LinComb * 4)(*[_lib.lincomb([(coef[j], ks[j][c]) for j in range(3)]) for c in range(4)])
    p0 = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in y0])
    p1 = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in y1])
    ns = (ctypes.c_int64 * 4)(*sizes)
    _lib.check(lib.gode_rk_errnorm_multi_f32(_lib.ptr(out), p0, p1, lcs, ns, 4, 1e-4, 1e-5, _lib.ptr(sc), _lib.stream_ptr()),
               "gode_rk_errnorm_multi_f32")
    assert out.tolist() == single
