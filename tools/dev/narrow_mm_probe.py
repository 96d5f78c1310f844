#!/usr/bin/env python3
"""gc3 of the benchmark model multiplies a 2^20 x 128 state by a 128 x 16 weight: which form of that product is fast?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops
dev = torch.device("cuda:0")
n, d, c = 1 << 20, 128, 16
x = torch.randn(n, d, device=dev); W = torch.randn(d, c, device=dev) / d ** 0.5
Wt = W.t().contiguous()


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


ref = x.double() @ W.double()
for name, fn in (("torch.mm(x, W)", lambda: torch.mm(x, W)),
                 ("F.linear(x, W^T contiguous)", lambda: torch.nn.functional.linear(x, Wt)),
                 ("(W^T @ x^T)^T", lambda: torch.mm(Wt, x.t()).t()),
                 ("gode gn_time_gemm generic (no GN, no time)", lambda: ops.gn_time_gemm([(1.0, x)], n, d, 0, 0.0, None, None, W, False, 0.0))):
    try:
        ms, r = t(fn)
        print("%-46s %.3f ms   max err %.2e" % (name, ms, (r.double() - ref).abs().max().item()))
    except Exception as e:
        print("%-46s failed: %s" % (name, e))
g = torch.randn(n, c, device=dev)
for name, fn in (("dX = torch.mm(g, W^T)", lambda: torch.mm(g, W.t())),
                 ("dW = torch.mm(x^T, g)", lambda: torch.mm(x.t(), g))):
    ms, r = t(fn)
    print("%-46s %.3f ms" % (name, ms))
refw = x.double().t() @ g.double()
for B in (64, 256, 1024, 4096):
    def f(B=B):
        return torch.bmm(x.view(B, n // B, d).transpose(1, 2), g.view(B, n // B, c)).sum(0)
    ms, r = t(f)
    print("dW = bmm over %4d row blocks + sum            %.3f ms   max err %.2e" % (B, ms, (r.double() - refw).abs().max().item() / refw.abs().max().item()))
try:
    ms, r = t(lambda: ops.wgrad([(1.0, x)], n, d, 0, 0.0, None, None, g, False))
    print("dW = ops.wgrad (generic kernel) partials         %.3f ms   max err %.2e" % (ms, (r.double().sum(0).view(d, c) - refw).abs().max().item() / refw.abs().max().item()))
except Exception as e:
    print("ops.wgrad failed:", e)
