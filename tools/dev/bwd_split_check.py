#!/usr/bin/env python3
"""VJP at d = 128 from exact bf16 pieces (csrc/gemm.hip gn_gemm_bwd_split_kernel) against the fp32-MFMA kernel and
float64 autograd: error of dx / dgamma / dbeta, and time of each kernel."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
d = 128
lib.gode_set_option(b"wgrad_split_small", 1)
for n, groups in ((1000, 32), (4097, 32), (4097, 0), (1 << 17, 32), (1 << 20, 32)):
    x = torch.randn(n, d, device=dev) * 1.7 + 0.3
    k1 = torch.randn(n, d, device=dev)
    dS = torch.randn(n, d, device=dev)
    pre = torch.randn(n, d, device=dev)
    gam = torch.rand(d, device=dev) + 0.5
    W = torch.randn(d + 1, d, device=dev) / d ** 0.5
    terms = [(1.0, x), (0.25, k1)]
    want = None
    if n <= (1 << 17):
        xs = (x + 0.25 * k1).double().requires_grad_(True)
        g64 = gam.double().requires_grad_(True)
        b64 = torch.zeros(d, dtype=torch.float64, device=dev, requires_grad=True)
        xn = torch.nn.functional.group_norm(xs, groups, g64, b64, 1e-5) if groups else xs
        S = torch.cat([torch.full((n, 1), 0.4, dtype=torch.float64, device=dev), xn], 1) @ W.double()
        (S * dS.double()).sum().backward()
        want = (pre.double() + 0.5 * xs.grad, g64.grad if groups else None, b64.grad if groups else None)
    res = {}
    for mode in (0, 1):
        assert lib.gode_set_option(b"bwd_split", mode) == 0
        dx, dg, db = ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, gam, W, True, dS, out_scale=0.5, pre_terms=[(1.0, pre)])
        res[mode] = (dx.double(), dg.double().sum(0) if dg is not None else None, db.double().sum(0) if db is not None else None)
        out = torch.empty_like(dx)
        for _ in range(3):
            ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, gam, W, True, dS, out_scale=0.5, pre_terms=[(1.0, pre)], out=out, parts=(dg, db) if dg is not None else None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ops.gn_time_gemm_bwd(terms, n, d, groups, 1e-5, gam, W, True, dS, out_scale=0.5, pre_terms=[(1.0, pre)], out=out, parts=(dg, db) if dg is not None else None)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        msg = "n %8d groups %2d mode %d  %.3f ms" % (n, groups, mode, ms)
        if want is not None:
            e = [(a - b).abs().max().item() / (b.abs().max().item() + 1e-30) for a, b in zip(res[mode], want) if a is not None]
            msg += "  rel err dx/dgamma/dbeta vs float64: " + " ".join("%.2e" % v for v in e)
        print(msg, flush=True)
    e = [(a - b).abs().max().item() / (b.abs().max().item() + 1e-30) for a, b in zip(res[1], res[0]) if a is not None]
    print("          split vs fp32 kernel: " + " ".join("%.2e" % v for v in e), flush=True)
lib.gode_set_option(b"bwd_split", 0); lib.gode_set_option(b"wgrad_split_small", 0)
