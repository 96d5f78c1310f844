#!/usr/bin/env python3
"""Weight gradient at d = 128 from exact bf16 pieces (csrc/gemm.hip wgrad_split_kernel) against the fp32-MFMA kernel
and a float64 reference: error of each, and time of each."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_odenet_amd import ops
from graph_odenet_amd import _lib
lib = _lib.load()

dev = torch.device("cuda:0")
torch.manual_seed(0)


def ref64(x, gamma, beta, dS, groups, eps):
    x = x.double()
    n, d = x.shape
    xg = x.view(n, groups, d // groups)
    xn = (xg - xg.mean(2, keepdim=True)) / torch.sqrt(xg.var(2, unbiased=False, keepdim=True) + eps)
    xn = xn.view(n, d) * gamma.double() + beta.double()
    return torch.cat([dS.double().sum(0, keepdim=True), xn.t() @ dS.double()], 0)


for n in (1000, 4097, 1 << 17, 1 << 20):
    d = 128
    x = torch.randn(n, d, device=dev) * 1.7 + 0.3
    k1 = torch.randn(n, d, device=dev)
    dS = torch.randn(n, d, device=dev) * torch.rand(n, 1, device=dev)
    gam, bet = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev) * 0.1
    terms = [(1.0, x), (0.25, k1)]
    want = ref64(x + 0.25 * k1, gam, bet, dS, 32, 1e-5)
    scale = want.abs().max().item()
    for mode in (0, 6, 8):
        assert lib.gode_set_option(b"wgrad_split", mode) == 0
        part = ops.wgrad(terms, n, d, 32, 1e-5, gam, bet, dS, True)
        got = part.double().sum(0).view(d + 1, d)
        err = (got - want).abs().max().item() / scale
        for _ in range(3):
            ops.wgrad(terms, n, d, 32, 1e-5, gam, bet, dS, True, part=part)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ops.wgrad(terms, n, d, 32, 1e-5, gam, bet, dS, True, part=part)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        print("n %8d  mode %d  max err / max|dW| = %.3e   %.3f ms" % (n, mode, err, ms), flush=True)
    lib.gode_set_option(b"wgrad_split", 0)
