#!/usr/bin/env python3
"""Per-kernel timings at the C5 size (2^20 nodes, 10 M edges, d=128) with torch events.
Development aid; bench.py is the contract benchmark."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_odenet_amd import ops  # noqa: E402
from graph_odenet_amd.synth import rmat_graph  # noqa: E402


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--split", type=int, default=None)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = rmat_graph(a.scale, a.edges, seed=0, device=dev, split=a.split)
    gt = g.transpose()
    n, d = g.n_rows, a.d
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    print("nodes %d nnz %d items %d long_rows %d slots %d maxdeg %d | T: items %d long %d slots %d maxdeg %d" % (
        n, g.nnz, g.n_items, g.n_long, g.n_slots, int(deg.max()), gt.n_items, gt.n_long, gt.n_slots,
        int((gt.rowptr[1:] - gt.rowptr[:-1]).max())))
    bufs = [torch.randn(n, d, device=dev) for _ in range(6)]
    X, Y, K1, K2, K3, OUT = bufs
    bias = torch.randn(d, device=dev)
    W = torch.randn(d + 1, d, device=dev) / d ** 0.5
    gam, bet = torch.rand(d, device=dev) + 0.5, torch.rand(d, device=dev) - 0.5
    gb = g.algorithmic_bytes(d) / 1e9
    res = {}

    def rec(name, ms, gbytes=None, gflop=None):
        s = "%-34s %9.3f ms" % (name, ms)
        if gbytes:
            s += "  %8.1f GB/s" % (gbytes / ms * 1e3)
        if gflop:
            s += "  %8.1f TFLOP/s" % (gflop / ms)
        print(s, flush=True)
        res[name] = ms

    sel = a.only.split(",") if a.only else None

    def want(k):
        return sel is None or any(k.startswith(x) for x in sel)
    if want("spmm"):
        rec("spmm A (bias+relu)", timeit(lambda: ops.spmm(g, X, bias=bias, relu=True, out=OUT)), gb)
        rec("spmm A (+masked cotangent)", timeit(lambda: ops.spmm(g, X, bias=bias, relu=True, out=OUT,
                                                                   cot_terms=[(-1.0, Y), (0.1, K1)], out2=K3)), gb + 3 * n * d * 4 / 1e9)
        rec("spmm A^T", timeit(lambda: ops.spmm(gt, X, out=OUT)), gb)
    nd = n * d * 4 / 1e9
    fl = 2 * n * d * d / 1e9
    if want("gemm"):
        for nt, terms in ((1, [(1.0, X)]), (2, [(1.0, X), (0.1, K1)]), (4, [(1.0, X), (0.1, K1), (-0.1, K2), (0.1, K3)])):
            rec("gn_time_gemm fwd (%d terms)" % nt,
                timeit(lambda: ops.gn_time_gemm(terms, n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT)), (nt + 1) * nd, fl / 1e3)
        t4 = [(1.0, X), (0.1, K1), (-0.1, K2), (0.1, K3)]
        XO = torch.empty_like(X)
        rec("gn_time_gemm fwd (4 terms, +x_out)",
            timeit(lambda: ops.gn_time_gemm(t4, n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT, x_out=XO)), 6 * nd, fl / 1e3)
        rec("gn_time_gemm fwd (3 terms)",
            timeit(lambda: ops.gn_time_gemm(t4[:3], n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT)), 4 * nd, fl / 1e3)
        for nt, terms in ((1, [(1.0, X)]), (4, [(1.0, X), (0.1, K1), (-0.1, K2), (0.1, K3)])):
            rec("gn_time_gemm bwd (%d terms)" % nt,
                timeit(lambda: ops.gn_time_gemm_bwd(terms, n, d, 32, 1e-5, gam, W, True, Y, out=OUT)), (nt + 2) * nd, fl / 1e3)
            rec("wgrad (%d terms)" % nt, timeit(lambda: ops.wgrad(terms, n, d, 32, 1e-5, gam, bet, Y, True)), (nt + 1) * nd, fl / 1e3)
        t2 = [(1.0, X), (0.1, K1)]
        rec("gn_time_gemm bwd (2 terms)", timeit(lambda: ops.gn_time_gemm_bwd(t2, n, d, 32, 1e-5, gam, W, True, Y, out=OUT)), 4 * nd, fl / 1e3)
        rec("wgrad (2 terms)", timeit(lambda: ops.wgrad(t2, n, d, 32, 1e-5, gam, bet, Y, True)), 3 * nd, fl / 1e3)
        part = ops.wgrad(t2, n, d, 32, 1e-5, gam, bet, Y, True)
        o = torch.empty((d + 1) * d, device=dev)
        rec("reduce_parts dW (%d parts)" % part.shape[0], timeit(lambda: ops.reduce_parts_(o, part)), part.numel() * 4 / 1e9)
        _, dgp, dbp = ops.gn_time_gemm_bwd(t2, n, d, 32, 1e-5, gam, W, True, Y, out=OUT)
        o2 = torch.empty(d, device=dev)
        rec("reduce_parts dgamma (%d parts)" % dgp.shape[0], timeit(lambda: ops.reduce_parts_(o2, dgp)))
    if want("gat"):
        # GAT edge attention at the benchmark scale: the graph's nnz are the edges, o = d
        from graph_odenet_amd.gat_layers import EdgeGraph
        rp = g.rowptr.to(torch.int64)
        tgt = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
        src = g.col.to(torch.int64)
        E = src.numel()
        perm = torch.randperm(E, device=dev)                       # an edge list in no particular order
        src, tgt = src[perm], tgt[perm]
        Mtgt = torch.sparse_coo_tensor(torch.stack([tgt, torch.arange(E, device=dev)]), torch.ones(E, device=dev), (n, E))
        eg = EdgeGraph(src, tgt, Mtgt)
        print("edge graph: canonical %s, records %d, split rows %d" % (eg.canonical, eg.Mt.n_items, eg.Mt.n_long))
        Ps, Pt, A2 = torch.randn(n, d, device=dev), torch.randn(n, d, device=dev), torch.randn(n, 2, device=dev)
        proj = ops.gat_proj(Ps, Pt, A2)
        bw, bf = torch.zeros(1, device=dev), torch.randn(d, device=dev)
        a, amax = torch.empty(E, device=dev), torch.empty(1, device=dev)
        wgt, den, out = torch.empty(E, device=dev), torch.empty(n, device=dev), torch.empty(n, d, device=dev)
        dz, da = torch.empty(E, d, device=dev), torch.empty(E, device=dev)
        dPs, dPt, dA2 = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev), torch.empty(n, 2, device=dev)
        rec("gat logits + max", timeit(lambda: ops.gat_logits(proj, bw, eg.src, eg.tgt, a, amax)), E * 20 / 1e9)
        b_fwd = (E * (4 + 4 * d + 4 + 4 + 4) + 3 * n * d * 4) / 1e9     # src, gathered Ps row, logit, val, weight out; Pt, out
        rec("gat agg fwd", timeit(lambda: ops.gat_agg_fwd(eg, proj, d, bf, a, amax, 1e-6, out, wgt, den)), b_fwd)
        b_vjp = (E * (4 + 4 * d + 4 + 4 + 4 * d + 4) + 4 * n * d * 4 + E * (4 + 4 * d) + n * d * 4) / 1e9
        rec("gat vjp (agg bwd + max path + node sums)",
            timeit(lambda: ops.gat_vjp(eg, proj, d, bf, a, amax, wgt, den, out, dz, da, dPs, dPt, dA2, dout=Y)), b_vjp)
    if want("qc"):
        # QC edge-conditioned messages: 2000 molecules' worth of edges (the reference batches 20), h = 73
        from graph_odenet_amd.graph import incidence_from_index
        h, nq, Eq = 73, 36000, 76000
        gq = torch.Generator(device=dev).manual_seed(0)
        srcq = torch.randint(0, nq, (Eq,), generator=gq, device=dev).to(torch.int32)
        tgtq = torch.randint(0, nq, (Eq,), generator=gq, device=dev).to(torch.int32)
        Mtq = incidence_from_index(tgtq, nq)
        Aq = torch.randn(Eq, h, h, device=dev)
        Xq, dMq = torch.randn(nq, h, device=dev), torch.randn(nq, h, device=dev)
        bq = (Eq * h * h * 4 + Eq * h * 8 + nq * h * 8) / 1e9
        rec("qc edge matvec fwd (E=76k, h=73)", timeit(lambda: ops.edge_matvec_fwd(Mtq, srcq, Aq, Xq)), bq)
        erow = tgtq.clone(); evalq = torch.ones(Eq, device=dev)
        rec("qc edge matvec bwd (dA + dx)", timeit(lambda: ops.edge_matvec_bwd(erow, evalq, srcq, Aq, Xq, dMq)), 2 * bq)
    if want("ew"):
        o2 = torch.empty(d, device=dev)
        rec("colsum", timeit(lambda: ops.colsum_(o2, X)), nd)
        rec("lincomb 5 terms", timeit(lambda: ops.lincomb_(OUT, [(1.0, X), (.1, Y), (.3, K1), (.3, K2), (.1, K3)])), 6 * nd)
        rec("errnorm 6 terms", timeit(lambda: ops.rk_error_sumsq(X, Y, [(.1, K1), (.2, K2), (.3, K3), (.1, OUT), (.1, X), (.2, Y)], 1e-5, 1e-5)), 6 * nd)


if __name__ == "__main__":
    main()
